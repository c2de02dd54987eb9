"""Batched host reader + H2D staging for the DCT datasets.

Replaces, for `--domain DCT`, the reference's per-sample path
    imagenet_dataset_indexing.__getitem__ (datasets.py:274-297: read_coefficients + dequantise per sample, CPU tensors)
    -> DistributedSampler / DataLoader workers / default collate (datasets.py:533-546)
    -> pipeline_utils.unpack_data (.to(device), pipeline_utils.py:70-73)
with one object:

    loader = DCTBatchLoader(paths, labels, batch_size=256, device="cuda:0", rank=rank, world_size=world,
                            transform=rg.datasets.get_transform("imagenet_dct", "train", fused=True, ...))
    for epoch in range(E):
        loader.set_epoch(epoch)                       # = trainloader.sampler.set_epoch(epoch), train.py:143
        for (Y, CbCr), labels in loader:              # device tensors, ToRange'd -- what unpack_data returns
            ...

How: a decoder thread calls `librgbnm_reader.so`'s pthread batch entry (entropy decode only, GIL released) straight into a
RING of re-used pinned int16 batch buffers, `prefetch` batches ahead of the consumer; the consumer issues ONE H2D copy per
tensor on a side stream, waits for it with an event (so the pinned buffer can go back to the ring), and runs the fused HIP
transform (de-quantise + crop + resize + flip + RandAugment + ToRange) on the raw coefficients.  No per-sample tensors, no
collate, no worker processes.  Sharding follows torch's DistributedSampler(shuffle, drop_last=False) index for index
(seed + epoch permutation, padded by wrapping, rank-strided), so a run is reproducible against the reference's sampler.

`crop_on_host=True` (train transform only): the crop boxes and the augmentation draws of a batch are sampled BEFORE its files
are decoded, the reader copies only each file's crop box out of libjpeg's coefficient arrays, back to back into flat pinned
buffers, and ONE H2D copy of the used prefix ships the batch: 787 KB -> ~380 KB per image on the sampler's mix of crop sides
("768 KB/img raw coefficients also make PCIe the next limit -- crop on host before H2D, or ship only the crop
box").  The kernels read the packed boxes in place (rgbnm_dct_augment_packed): same output bits as the whole-grid path.

Everything except the H2D copy and the transform is host logic and runs (and is tested) without a GPU: with device="cpu"
the loader yields the raw (Yq, CbCrq, quant) batches.  Files must share one coefficient grid (default 64 x 64 luma blocks =
512 x 512, the pre-resized layout of the reference's dataset, README "resize to 512"); a file with another grid or an
unreadable file raises at the batch that contains it, naming the file (the reference raises inside a worker)."""
import math
import queue
import threading

import torch

from . import dct_manip as dm


class DCTBatchLoader:
    def __init__(self, paths, labels, batch_size, device="cuda", grid=(64, 64), threads=None, prefetch=2, shuffle=True,
                 seed=0, rank=0, world_size=1, drop_last=False, transform=None, crop_on_host=False):
        if len(paths) != len(labels):
            raise ValueError("paths and labels differ in length")
        if batch_size <= 0 or prefetch < 1:
            raise ValueError("batch_size and prefetch must be positive")
        self.paths, self.labels = list(paths), torch.as_tensor(labels, dtype=torch.int64)
        self.batch_size, self.grid, self.prefetch = batch_size, tuple(grid), int(prefetch)
        self.threads = int(threads) if threads else dm.default_threads()
        self.shuffle, self.seed, self.rank, self.world_size, self.drop_last = shuffle, seed, rank, world_size, drop_last
        self.device = torch.device(device)
        self.transform = transform
        self.crop_on_host = bool(crop_on_host)
        if self.crop_on_host:
            from . import custom_transforms as CT
            if not isinstance(transform, CT.TrainTransform_DCT) or transform.eval_mode or self.device.type != "cuda":
                raise ValueError("crop_on_host needs the fused train transform (TrainTransform_DCT) and a HIP device: the crop "
                                 "boxes are drawn before the decode; the eval transform reads the whole grid")
        self.last_packed = None        # crop_on_host: the (packed parameters, nops) of the batch yielded last (tests)
        self.epoch = 0
        n = len(self.paths)
        # DistributedSampler(drop_last=False): every rank gets ceil(n / world) indices, the list is padded by wrapping
        self.num_samples = math.ceil(n / world_size) if n else 0
        self.total_size = self.num_samples * world_size
        self._ring = None

    # ---- sampler semantics (torch.utils.data.distributed.DistributedSampler.__iter__)
    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        n = len(self.paths)
        if n == 0:
            return []
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        pad = self.total_size - len(idx)
        if pad > 0:
            idx += (idx * math.ceil(pad / len(idx)))[:pad]
        return idx[self.rank:self.total_size:self.world_size]

    def __len__(self):
        if self.drop_last:
            return self.num_samples // self.batch_size
        return math.ceil(self.num_samples / self.batch_size)

    # ---- staging ring: prefetch queued + one being consumed + one being filled
    def _buffers(self):
        if self._ring is None:
            pin = self.device.type == "cuda"
            alloc = dm.alloc_packed if self.crop_on_host else dm.alloc_batch
            self._ring = [alloc(self.batch_size, self.grid, pin_memory=pin) for _ in range(self.prefetch + 2)]
        return self._ring

    def __iter__(self):
        idx = self.indices()
        nb = len(self)
        batches = [idx[b * self.batch_size:(b + 1) * self.batch_size] for b in range(nb)]
        ring = self._buffers()
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        sampler = None
        if self.crop_on_host:
            from . import custom_transforms as CT
            # one stream of draws per (seed, epoch, rank): the producer samples a batch's parameters before decoding it
            sampler = CT.FastParamSampler(self.transform, seed=[self.seed, self.epoch, self.rank])

        def producer():
            try:
                for k, bi in enumerate(batches):
                    if stop.is_set():
                        return
                    buf = ring[k % len(ring)]
                    if sampler is not None:
                        packed, nops = sampler.sample(len(bi), self.grid[0], self.grid[1])
                        Yp, Cp, Qp, yo, co = dm.read_coefficients_batch_crop([self.paths[i] for i in bi], packed["crop"],
                                                                             threads=self.threads, grid=self.grid, out=buf)
                        item = ((Yp, Cp, Qp, yo, co, packed, nops), self.labels[bi])
                    else:
                        view = tuple(t[:len(bi)] for t in buf)          # the last batch of an epoch may be short
                        out = dm.read_coefficients_batch([self.paths[i] for i in bi], threads=self.threads, grid=self.grid, out=view)
                        item = (out, self.labels[bi])
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                q.put(None)
            except BaseException as e:       # noqa: BLE001 -- delivered to the consumer, which re-raises it
                q.put(e)

        th = threading.Thread(target=producer, daemon=True, name="rgbnm-dct-decoder")
        th.start()
        copy_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                if sampler is not None:
                    from . import custom_transforms as CT
                    (Yp, Cp, Qp, yo, co, packed, nops), lab = item
                    with torch.cuda.stream(copy_stream):
                        dev = [t.to(self.device, non_blocking=True) for t in (Yp, Cp, Qp, yo.pin_memory(), co.pin_memory(), lab)]
                    copy_stream.synchronize()          # the pinned buffers go back to the decoder ring after this
                    for t in dev:
                        t.record_stream(torch.cuda.current_stream(self.device))
                    self.last_packed = (packed, nops)
                    self.h2d_bytes = 2 * (Yp.numel() + Cp.numel())
                    yield tuple(CT.apply_packed(self.transform, dev[0], dev[1], dev[2], packed, nops, y_off=dev[3], c_off=dev[4],
                                                grid=self.grid)), dev[5]
                    continue
                (Y, C, Q), lab = item
                if copy_stream is None:
                    yield self._finish(Y.clone(), C.clone(), Q.clone(), lab)     # host buffers are recycled: hand out copies
                    continue
                with torch.cuda.stream(copy_stream):
                    Yd = Y.to(self.device, non_blocking=True)
                    Cd = C.to(self.device, non_blocking=True)
                    Qd = Q.to(self.device, non_blocking=True)
                    ld = lab.to(self.device, non_blocking=True)
                copy_stream.synchronize()              # the pinned buffer goes back to the decoder ring after this
                for t in (Yd, Cd, Qd, ld):
                    t.record_stream(torch.cuda.current_stream(self.device))
                yield self._finish(Yd, Cd, Qd, ld)
        finally:
            stop.set()
            while th.is_alive():                        # drain so the producer can see `stop`
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)

    def _finish(self, Y, C, Q, lab):
        if self.transform is None:
            return (Y, C, Q), lab
        return tuple(self.transform(Y, C, Q)), lab
