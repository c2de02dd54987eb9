#!/usr/bin/env python
"""Per-kernel timing of one encoder block's GEMM shapes at BASELINE config 2 (B = 256, JPEG-Ti, bf16), each measured
HOT (back-to-back: operands of the previous launch still in L2 / Infinity Cache) and COLD (a 1 GiB buffer is streamed
between launches, so every operand comes from HBM).  usage: python tools/nt_probe.py [tag]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L

DEV = "cuda"
M = 256 * 196


def ev():
    return torch.cuda.Event(enable_timing=True)


def time_hot(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def time_cold(fn, flush, n=12):
    tot = 0.0
    for _ in range(n):
        flush()
        a, b = ev(), ev()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3


def main(tag=""):
    lib = L.lib()
    dt = torch.bfloat16
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g).to(dt)  # noqa: E731
    x192, x576, x768 = rn(M, 192), rn(M, 576), rn(M, 768)
    r192, r768 = rn(M, 192), rn(M, 768)
    big = torch.empty(1 << 28, device=DEV, dtype=torch.float32)       # 1 GiB
    flush = lambda: big.add_(1.0)  # noqa: E731
    gamma, beta = torch.ones(192, device=DEV), torch.zeros(192, device=DEV)
    mean, rstd = torch.zeros(M, device=DEV), torch.ones(M, device=DEV)
    ws = torch.empty(64 << 20, device=DEV, dtype=torch.uint8)
    out = []

    def nt(name, A, N, K, epi, R=None, C2=False, bytes_=0):
        W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(dt)
        b = torch.randn(N, device=DEV, generator=g)
        Cc = torch.empty(M, N, device=DEV, dtype=dt)
        c2 = torch.empty(M, N, device=DEV, dtype=dt) if C2 else None
        f = lambda: L.check(lib.rgbnm_gemm_nt(1, epi, A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, b.data_ptr(),  # noqa: E731
                                              L.ptr(R), N, L.ptr(c2), N, None, 0, M, N, K, 0, L.stream()))
        out.append((name, time_hot(f), time_cold(f, flush), bytes_))

    MB = 1e6
    e = M * 2 / MB          # MB per feature column
    nt("qkv      wres<0>  K192 N576", x192, 576, 192, 0, bytes_=e * (192 + 576))
    nt("proj-dX  wres<0>  K192 N192", x192, 192, 192, 0, bytes_=e * (192 + 192))
    nt("fc1+gelu wres<2>  K192 N768", x192, 768, 192, 2, C2=True, bytes_=e * (192 + 768 * 2))
    nt("dgelu    wres<4>  K192 N768", x192, 768, 192, 4, R=r768, bytes_=e * (192 + 768 * 2))
    nt("fc2+res  kpipe<1> K768 N192", x768, 192, 768, 1, R=r192, bytes_=e * (768 + 192 * 2))
    nt("plain    kpipe<0> K768 N192", x768, 192, 768, 0, bytes_=e * (768 + 192))
    nt("plain    kpipe<0> K576 N192", x576, 192, 576, 0, bytes_=e * (576 + 192))

    # composite: one block forward / backward through the C ABI, per-kernel split comes from rocprof; here the totals
    import ctypes as C
    cfg = L.VitCfg(L.DT_BF16, 256, 196, 192, 3, 1e-5, 192 ** -0.5)
    for name, tm, tc, by in out:
        print(f"{tag:8s} {name:32s} hot {tm:7.1f} us ({by / tm / 1e6 * 1e0:5.2f} TB/s)   cold {tc:7.1f} us ({by / tc / 1e6:5.2f} TB/s)   {by:6.1f} MB")
    # stream reference on this box (16 B per lane grid-stride kernels of csrc/calib.hip)
    src = torch.empty(1 << 28, device=DEV, dtype=torch.uint8)
    dst = torch.empty(1 << 28, device=DEV, dtype=torch.uint8)
    sink = torch.zeros(4, device=DEV, dtype=torch.uint8)
    for mode, nm, mul in ((0, "copy", 2), (1, "read", 1), (2, "write", 1)):
        f = lambda: L.check(lib.rgbnm_calib_stream(src.data_ptr(), dst.data_ptr(), src.numel(), mode, 2048, sink.data_ptr(), L.stream()))  # noqa: E731
        t = time_hot(f, 20)
        print(f"{tag:8s} stream {nm:5s} 256 MiB: {t:7.1f} us = {mul * src.numel() / t / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")
