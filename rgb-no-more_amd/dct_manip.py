"""Drop-in for the reference's `dct_manip` extension module on this path: `read_coefficients(path)` with the same
return contract (dct_manip/dct_manip.cpp:152-178), implemented by the host C library librgbnm_reader.so
(csrc/reader.c, include/rgbnm_reader.h) through ctypes -- no libtorch / pybind11 in the native part.

    dim, quant, Y, CbCr = dct_manip.read_coefficients("img.jpg")
    dim int32 (C,2); quant int16 (C,8,8); Y int16 (1,Hb,Wb,8,8); CbCr int16 (2,Hb/2,Wb/2,8,8) or None (grayscale)

`read_coefficients_batch` decodes many same-shaped files with a pthread pool into pinned batch tensors ready for one
H2D copy (replaces per-sample tensors + default collate, datasets.py:274-297,542-546).
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librgbnm_reader.so")
_lib = None
_ERRLEN = 512


class libjpeg_exception(Exception):
    """mirrors dct_manip.cpp:24-41 (libjpeg's formatted message)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"host reader missing: {LIB_PATH} (run __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.rgbnm_reader_abi_version.restype = C.c_int
        L.rgbnm_jpeg_info.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_int]
        L.rgbnm_jpeg_info_mem.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_char_p, C.c_int]
        L.rgbnm_read_coefficients.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        L.rgbnm_read_coefficients_mem.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_char_p, C.c_int]
        L.rgbnm_read_coefficients_batch.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rgbnm_read_coefficients_batch_crop.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_void_p, C.c_void_p]
        for f in ("rgbnm_jpeg_info", "rgbnm_jpeg_info_mem", "rgbnm_read_coefficients", "rgbnm_read_coefficients_mem",
                  "rgbnm_read_coefficients_batch", "rgbnm_read_coefficients_batch_crop"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def _raise(rc, err):
    msg = err.value.decode(errors="replace")
    if rc == -1:
        raise RuntimeError(msg)                      # "Unable to open file for reading: ..."
    if rc == -2:
        raise libjpeg_exception(msg)
    raise RuntimeError(f"read_coefficients failed ({rc}) {msg}")


def _read(path=None, data=None):
    L = lib()
    err = C.create_string_buffer(_ERRLEN)
    info = (C.c_int32 * 17)()
    if data is None:
        p = os.fsencode(path)
        rc = L.rgbnm_jpeg_info(p, info, err, _ERRLEN)
    else:
        rc = L.rgbnm_jpeg_info_mem(data, len(data), info, err, _ERRLEN)
    if rc:
        _raise(rc, err)
    nc = info[0]
    hb, wb = info[1], info[2]
    if nc not in (1, 3):
        raise libjpeg_exception(f"unsupported JPEG: {nc} components (the DCT path reads grayscale or YCbCr files)")
    dim = torch.empty((nc, 2), dtype=torch.int32)
    quant = torch.empty((nc, 8, 8), dtype=torch.int16)
    Y = torch.empty((1, hb, wb, 8, 8), dtype=torch.int16)
    CbCr = None
    if nc > 1:
        CbCr = torch.empty((2, info[5], info[6], 8, 8), dtype=torch.int16)
    cptr = CbCr.data_ptr() if CbCr is not None else None
    if data is None:
        rc = L.rgbnm_read_coefficients(p, dim.data_ptr(), quant.data_ptr(), Y.data_ptr(), cptr, err, _ERRLEN)
    else:
        rc = L.rgbnm_read_coefficients_mem(data, len(data), dim.data_ptr(), quant.data_ptr(), Y.data_ptr(), cptr, err, _ERRLEN)
    if rc:
        _raise(rc, err)
    return dim, quant, Y, CbCr


def read_coefficients(path: str):
    """(dim, quant, Y, CbCr|None) -- reference: dct_manip.read_coefficients (dct_manip.cpp:152-178)."""
    return _read(path=path)


def read_coefficients_bytes(data: bytes):
    """Same, from an in-memory JPEG."""
    return _read(data=bytes(data))


def default_threads():
    """Decoder threads worth starting: the CPUs this process may actually use (cgroup quota / affinity), not os.cpu_count()
    (the GPU box shows 256 hardware threads and grants 16: 256 decoder threads ran 35 % slower than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def alloc_batch(n, grid=(64, 64), pin_memory=True):
    """Reusable (pinned) staging buffers for read_coefficients_batch(out=...): allocating pinned memory per batch costs
    more than decoding it."""
    hb, wb = grid
    hbc, wbc = (hb + 1) // 2, (wb + 1) // 2
    kw = dict(dtype=torch.int16, pin_memory=pin_memory)
    return (torch.empty((n, 1, hb, wb, 8, 8), **kw), torch.empty((n, 2, hbc, wbc, 8, 8), **kw),
            torch.empty((n, 3, 8, 8), **kw))


def read_coefficients_batch(paths, threads=8, grid=(64, 64), pin_memory=False, out=None):
    """Decode len(paths) JPEGs of one coefficient grid (default 512x512 4:2:0 -> 64x64 luma blocks) in parallel.
    Returns Y (B,1,Hb,Wb,8,8), CbCr (B,2,Hb/2,Wb/2,8,8), quant (B,3,8,8) int16 CPU tensors (optionally pinned);
    out = buffers from alloc_batch() to decode into (ring of staging buffers)."""
    L = lib()
    n = len(paths)
    hb, wb = grid
    hbc, wbc = (hb + 1) // 2, (wb + 1) // 2
    if out is None:
        out = alloc_batch(n, grid, pin_memory)
    Y, CbCr, quant = out
    if tuple(Y.shape) != (n, 1, hb, wb, 8, 8) or tuple(CbCr.shape) != (n, 2, hbc, wbc, 8, 8) or \
            tuple(quant.shape) != (n, 3, 8, 8) or any(t.dtype != torch.int16 or not t.is_contiguous() for t in out):
        raise ValueError("out buffers do not match the batch (use alloc_batch)")
    status = torch.zeros(n, dtype=torch.int32)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    bad = L.rgbnm_read_coefficients_batch(arr, n, int(threads), hb, wb, hbc, wbc, Y.data_ptr(), CbCr.data_ptr(),
                                          quant.data_ptr(), status.data_ptr())
    if bad:
        i = int((status != 0).nonzero()[0])
        code = int(status[i])
        if code == -1:
            raise RuntimeError(f"Unable to open file for reading: {paths[i]}")
        raise libjpeg_exception(f"{paths[i]}: reader error {code} ({bad} of {n} files failed)")
    return Y, CbCr, quant


def _raise_batch(paths, status, bad):
    i = int((status != 0).nonzero()[0])
    code = int(status[i])
    if code == -1:
        raise RuntimeError(f"Unable to open file for reading: {paths[i]}")
    raise libjpeg_exception(f"{paths[i]}: reader error {code} ({bad} of {len(paths)} files failed)")


def alloc_packed(n, grid=(64, 64), pin_memory=True):
    """Staging buffers for read_coefficients_batch_crop(out=...): flat int16 Y / CbCr buffers sized for n WHOLE grids (a
    batch of crops never needs more), quant (n,3,8,8)."""
    hb, wb = grid
    hbc, wbc = (hb + 1) // 2, (wb + 1) // 2
    kw = dict(dtype=torch.int16, pin_memory=pin_memory)
    return (torch.empty(n * hb * wb * 64, **kw), torch.empty(n * 2 * hbc * wbc * 64, **kw), torch.empty((n, 3, 8, 8), **kw))


def packed_offsets(boxes):
    """boxes int (n,4) = (top, left, height, width) in luma blocks -> element offsets of every image's crop in the flat Y /
    CbCr buffers (running sums of the box sizes) and the totals: (y_off int64 (n,), c_off int64 (n,), n_y, n_c)."""
    b = torch.as_tensor(boxes, dtype=torch.int64)
    ysz = b[:, 2] * b[:, 3] * 64
    csz = 2 * (b[:, 2] // 2) * (b[:, 3] // 2) * 64
    y_off = torch.cumsum(ysz, 0) - ysz
    c_off = torch.cumsum(csz, 0) - csz
    return y_off.contiguous(), c_off.contiguous(), int(ysz.sum()), int(csz.sum())


def read_coefficients_batch_crop(paths, boxes, threads=8, grid=(64, 64), pin_memory=False, out=None):
    """Decode len(paths) same-grid JPEGs and keep only each file's crop box (luma blocks (top, left, height, width), all
    even; the chroma box is the luma box halved), packed back to back: returns (Ypacked, Cpacked, quant, y_off, c_off) --
    flat int16 buffers trimmed to the used length, image i's luma crop [h][w][8][8] at Ypacked[y_off[i]:], its chroma crop
    [2][h/2][w/2][8][8] at Cpacked[c_off[i]:].  What crosses PCIe afterwards is the crop, not the 64 x 64 grid."""
    L = lib()
    n = len(paths)
    hb, wb = grid
    hbc, wbc = (hb + 1) // 2, (wb + 1) // 2
    bx = torch.as_tensor(boxes, dtype=torch.int32).contiguous()
    if tuple(bx.shape) != (n, 4):
        raise ValueError("boxes must be (len(paths), 4)")
    y_off, c_off, ny, ncc = packed_offsets(bx)
    if out is None:
        out = alloc_packed(n, grid, pin_memory)
    Yp, Cp, quant = out
    if Yp.numel() < ny or Cp.numel() < ncc or tuple(quant.shape)[1:] != (3, 8, 8) or quant.shape[0] < n or \
            any(t.dtype != torch.int16 or not t.is_contiguous() for t in out):
        raise ValueError("out buffers too small for this batch (use alloc_packed)")
    status = torch.zeros(n, dtype=torch.int32)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    bad = L.rgbnm_read_coefficients_batch_crop(arr, n, int(threads), hb, wb, hbc, wbc, bx.data_ptr(), y_off.data_ptr(),
                                               c_off.data_ptr(), Yp.data_ptr(), Cp.data_ptr(), quant.data_ptr(),
                                               status.data_ptr())
    if bad:
        if int(status[(status != 0).nonzero()[0]]) == -3:
            raise ValueError("crop box outside the coefficient grid (or not even)")
        _raise_batch(paths, status, bad)
    return Yp[:ny], Cp[:ncc], quant[:n], y_off, c_off
