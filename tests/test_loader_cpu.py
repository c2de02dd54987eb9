"""DCTBatchLoader (rgb_no_more_amd/loader.py): the host side -- sharding identical to torch's
DistributedSampler (datasets.py:533-535), batching, the decoder thread + buffer ring, short last batch, error reporting --
runs without a GPU (device="cpu" yields the raw coefficient batches).  The H2D + HIP-transform half is in
tests/test_loader_gpu.py."""
import numpy as np
import pytest
import torch
from torch.utils.data.distributed import DistributedSampler

from rgb_no_more_amd import dct_manip as dm
from rgb_no_more_amd.loader import DCTBatchLoader


@pytest.fixture()
def files(golden, tmp_path):
    g = golden("g1_reader.npz")
    paths = []
    for i in range(11):
        p = tmp_path / f"img{i:02d}.jpg"
        p.write_bytes(g["c64x64_jpeg"].tobytes())
        paths.append(str(p))
    return paths, list(range(100, 111)), g


def test_sharding_matches_torch_distributed_sampler(files):
    paths, labels, _ = files
    for world in (1, 2, 3):
        for epoch in (0, 1, 5):
            got = []
            for rank in range(world):
                ld = DCTBatchLoader(paths, labels, batch_size=4, device="cpu", grid=(8, 8), seed=7, rank=rank, world_size=world)
                ld.set_epoch(epoch)
                ds = DistributedSampler(list(range(len(paths))), num_replicas=world, rank=rank, shuffle=True, seed=7, drop_last=False)
                ds.set_epoch(epoch)
                assert ld.indices() == list(iter(ds)), (world, epoch, rank)
                assert len(ld) == -(-len(ld.indices()) // 4)
                got += ld.indices()
            assert set(got) == set(range(len(paths)))            # padded by wrapping: every sample at least once
    ns = DCTBatchLoader(paths, labels, batch_size=4, device="cpu", grid=(8, 8), shuffle=False, rank=1, world_size=2)
    assert ns.indices() == list(iter(DistributedSampler(list(range(11)), num_replicas=2, rank=1, shuffle=False)))


def test_batches_content_order_and_short_tail(files):
    paths, labels, g = files
    ld = DCTBatchLoader(paths, labels, batch_size=4, device="cpu", grid=(8, 8), threads=3, prefetch=2, seed=3)
    ld.set_epoch(2)
    want = ld.indices()
    seen = []
    sizes = []
    for (Y, C, Q), lab in ld:
        assert Y.dtype == torch.int16 and Y.shape[1:] == (1, 8, 8, 8, 8) and C.shape[1:] == (2, 4, 4, 8, 8) and Q.shape[1:] == (3, 8, 8)
        assert np.array_equal(Y[0].numpy(), g["c64x64_Y"]) and np.array_equal(C[-1].numpy(), g["c64x64_CbCr"])
        assert np.array_equal(Q[0].numpy(), g["c64x64_quant"])
        sizes.append(Y.shape[0])
        seen += [int(v) - 100 for v in lab]
    assert sizes == [4, 4, 3] and seen == want
    # a second pass re-uses the ring and gives the same epoch again; drop_last drops the short batch
    assert [int(v) - 100 for _, lab in ld for v in lab] == want
    dl = DCTBatchLoader(paths, labels, batch_size=4, device="cpu", grid=(8, 8), drop_last=True)
    assert len(dl) == 2 and [b[0][0].shape[0] for b in dl] == [4, 4]
    # yielded host batches are copies: they survive the ring being refilled
    it = iter(DCTBatchLoader(paths * 3, labels * 3, batch_size=2, device="cpu", grid=(8, 8), prefetch=1, shuffle=False))
    first = next(it)
    keep = first[0][0].clone()
    for _ in it:
        pass
    assert torch.equal(first[0][0], keep)


def test_errors_name_the_file_and_stop_the_decoder(files, tmp_path):
    paths, labels, g = files
    bad = list(paths)
    bad[5] = str(tmp_path / "missing.jpg")
    ld = DCTBatchLoader(bad, labels, batch_size=4, device="cpu", grid=(8, 8), shuffle=False)
    with pytest.raises(RuntimeError, match="missing.jpg"):
        for _ in ld:
            pass
    gray = tmp_path / "gray.jpg"
    gray.write_bytes(g["g40x56_jpeg"].tobytes())                       # another grid: refused, names the file
    ld2 = DCTBatchLoader(paths[:3] + [str(gray)], labels[:4], batch_size=4, device="cpu", grid=(8, 8), shuffle=False)
    with pytest.raises(dm.libjpeg_exception, match="gray.jpg"):
        next(iter(ld2))
    # abandoning an iterator mid-epoch must not leave the decoder thread blocked
    import threading
    before = threading.active_count()
    it = iter(DCTBatchLoader(paths * 4, labels * 4, batch_size=2, device="cpu", grid=(8, 8), prefetch=1))
    next(it)
    it.close()
    assert threading.active_count() <= before + 0
    with pytest.raises(ValueError):
        DCTBatchLoader(paths, labels[:-1], batch_size=2)


def test_cropped_batch_read_is_a_slice_of_the_full_read(files):
    """rgbnm_read_coefficients_batch_crop: only each file's crop box leaves libjpeg's coefficient arrays, packed back to
    back -- and it is exactly the slice [top:top+h, left:left+w] of what the whole-grid reader returns (chroma: halved box)."""
    paths, _, g = files
    Y, C, Q = dm.read_coefficients_batch(paths[:5], threads=2, grid=(8, 8))
    boxes = [(0, 0, 8, 8), (2, 4, 4, 4), (6, 0, 2, 8), (0, 6, 8, 2), (4, 2, 2, 2)]
    Yp, Cp, Qp, yo, co = dm.read_coefficients_batch_crop(paths[:5], boxes, threads=3, grid=(8, 8))
    assert torch.equal(Q, Qp)
    assert Yp.numel() == sum(h * w * 64 for _, _, h, w in boxes) and Cp.numel() == sum(2 * (h // 2) * (w // 2) * 64 for _, _, h, w in boxes)
    for b, (i, j, h, w) in enumerate(boxes):
        y = Yp[int(yo[b]):int(yo[b]) + h * w * 64].view(h, w, 8, 8)
        c = Cp[int(co[b]):int(co[b]) + 2 * (h // 2) * (w // 2) * 64].view(2, h // 2, w // 2, 8, 8)
        assert torch.equal(y, Y[b, 0, i:i + h, j:j + w]), b
        assert torch.equal(c, C[b, :, i // 2:(i + h) // 2, j // 2:(j + w) // 2]), b
    for bad in ((1, 0, 2, 2), (0, 0, 10, 2), (0, 8, 2, 2), (0, 0, 0, 2)):       # odd / outside the 8 x 8 grid / empty
        with pytest.raises(ValueError):
            dm.read_coefficients_batch_crop(paths[:1], [bad], grid=(8, 8))
    # straight through the C ABI (no Python-side checks): a negative or oversized box is RD_EARG for that file -- nothing is
    # written through it (the chroma pre-fill used to run before the box was validated: a negative size wrapped to ~2^64 bytes)
    import ctypes as C
    lib = dm.lib()
    for bad in ((0, 0, -2, 4), (0, 0, 2, -4), (-2, 0, 2, 2), (0, 0, 4096, 4096), (6, 6, 4, 4)):
        arr = (C.c_char_p * 1)(paths[0].encode())
        box = torch.tensor([bad], dtype=torch.int32)
        off = torch.zeros(1, dtype=torch.int64)
        ybuf = torch.full((8 * 8 * 64,), 7, dtype=torch.int16)
        cbuf = torch.full((2 * 4 * 4 * 64,), 7, dtype=torch.int16)
        q = torch.zeros(192, dtype=torch.int16)
        st = torch.zeros(1, dtype=torch.int32)
        lib.rgbnm_read_coefficients_batch_crop(arr, 1, 1, 8, 8, 4, 4, C.c_void_p(box.data_ptr()), C.c_void_p(off.data_ptr()),
                                               C.c_void_p(off.data_ptr()), C.c_void_p(ybuf.data_ptr()), C.c_void_p(cbuf.data_ptr()),
                                               C.c_void_p(q.data_ptr()), C.c_void_p(st.data_ptr()))
        assert st.item() != 0, bad
        assert bool((ybuf == 7).all()) and bool((cbuf == 7).all()), bad
    with pytest.raises(ValueError):
        DCTBatchLoader(paths, list(range(11)), batch_size=4, device="cpu", grid=(8, 8), crop_on_host=True)
