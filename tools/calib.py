#!/usr/bin/env python
"""Attainable peaks on this box: an MFMA-only bf16 loop and 16 B/lane stream kernels, timed with HIP events.
Prints one JSON object; bench.py quotes the nominal peaks (2.5 PFLOP/s dense bf16, 8 TB/s) and DESIGN.md §5 both."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgb_no_more_amd import lib as L


def _time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    out = {}
    for wgs, label in ((256, "1 workgroup (4 waves) per CU"), (1024, "4 workgroups (16 waves) per CU")):
        iters = 4096
        t = _time(lambda: L.check(lib.rgbnm_calib_mfma_bf16(wgs, iters, sink.data_ptr(), st)), 5)
        flops = wgs * 4 * iters * 4 * 32768.0
        out[f"mfma_bf16_tflops[{label}]"] = round(flops / t / 1e12, 1)
    nbytes = 1 << 30
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda").random_(0, 255)
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for mode, name, traffic in ((0, "copy", 2 * nbytes), (1, "read", nbytes), (2, "write", nbytes)):
        best = 0.0
        for wgs in (2048, 8192, 32768):
            t = _time(lambda: L.check(lib.rgbnm_calib_stream(a.data_ptr(), b.data_ptr(), nbytes, mode, wgs,
                                                             sink.data_ptr(), st)), 10)
            best = max(best, traffic / t / 1e9)
        out[f"hbm_{name}_GBps[1 GiB]"] = round(best, 0)
    out["device"] = torch.cuda.get_device_name(0)
    out["measured_at"] = {"csrc_sha16": L.source_hash()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
