"""Mirror of the DCT half of the reference's `utils/custom_transforms.py` (:378-1196) for BATCHED device tensors.

The reference composes per-sample CPU transforms inside DataLoader workers (datasets.py:354-366):
    RandomResizedCrop_DCT(28, scale=(0.05,1), ratio=(1,1)) -> RandomFlip_DCT(0.5) -> RandAugment_dct(2, 3, 11, ops)
    -> ToRange(-1, 1, -1024, 1016)
Here the same chain runs as two HIP kernels per batch (csrc/augment.hip) on the raw int16 coefficients that
`dct_manip.read_coefficients` returns.  Random parameters are drawn on the host with the reference's distributions
(`sample_params`, class names / argument meaning kept) or supplied explicitly (parity tests pin them).

    aug = TrainTransform_DCT(size=28, ops_list=cfg.TRAIN.AUGLIST, num_ops=2, magnitude=3)
    Y, CbCr = aug(Yq, CbCrq, quant)            # (B,1,28,28,8,8), (B,2,14,14,8,8) in [-1, 1]
"""
import ctypes as C
import itertools
import time
import math

import numpy as np
import torch

from . import dct_ops as dops
from . import lib as L

OPS = {"Identity": 0, "AutoContrast": 1, "Posterize": 2, "SolarizeAdd": 3, "Color": 4, "Contrast": 5, "Brightness": 6,
       "MidfreqAug": 7, "Cutout": 8, "TranslateX": 9, "TranslateY": 10, "Rotate90": 11, "AutoSaturation": 12,
       "Grayscale": 13, "ChromaDrop": 14, "Sharpness": 15, "Invert": 16, "Solarize": 17, "FreqEnhance": 18, "Equalize": 19}
CHROMA_OPS = {"Grayscale", "Color", "AutoSaturation", "ChromaDrop"}
# The reference's DFT-plane ops (utils/dct_ops.py:367-434, 957-1013): in no DCT op list of utils/configs.py; they run as device tensor
# ops between kernel passes (dct_ops.rotate_block / shear_block) and only through RandAugment_dct, not through the fused transform
DFT_OPS = {"Rotate", "ShearX", "ShearY"}
# default vitti list (utils/configs.py:93)
VITTI_OPS = ("AutoContrast,Posterize,SolarizeAdd,Color,Contrast,Brightness,MidfreqAug,Cutout,TranslateX,TranslateY,"
             "Rotate90,AutoSaturation,Grayscale,ChromaDrop").split(",")


class AugParams(C.Structure):
    _fields_ = [("crop_i", C.c_int), ("crop_j", C.c_int), ("crop_h", C.c_int), ("crop_w", C.c_int), ("flip", C.c_int),
                ("op", C.c_int * 2), ("fmag", C.c_float * 2), ("iarg0", C.c_int * 2), ("iarg1", C.c_int * 2),
                ("iarg2", C.c_int * 2)]


def _factors(n):
    return sorted(itertools.chain.from_iterable((i, n // i) for i in range(1, int(n ** 0.5) + 1) if n % i == 0))


def even_size_choices(size):
    return [c for c in _factors(size) if c % 2 == 0]


def _choose_closest(val, choices, maxval):
    """custom_transforms.py:571-578."""
    if val <= choices[-1]:
        return choices[int(np.argmin([abs(c - val) for c in choices]))]
    closest = float(np.rint(np.float32(val) / np.float32(choices[-1]))) * choices[-1]
    if closest > maxval:
        closest -= choices[-1]
    return closest


class RandomResizedCrop_DCT(torch.nn.Module):
    """Reference semantics (custom_transforms.py:527-663), ratio == (1, 1): `get_params` samples the crop box with the
    reference's distribution; `forward(coeff)` crops and resizes device int16 coefficients (tensor or (Y, CbCr)) on the HIP
    augment kernels and returns int16 coefficients on the `size` x `size` block grid."""

    def __init__(self, size, scale=(0.05, 1.0), ratio=(1, 1), chroma_scale=2, dtype_resize=torch.float32):
        super().__init__()
        if tuple(ratio) != (1, 1):
            raise NotImplementedError("the DCT pipelines use ratio=(1,1) (datasets.py:356,373)")
        if dtype_resize != torch.float32:
            raise NotImplementedError("resize runs in fp32 (the reference's default dtype_resize)")
        self.size, self.scale, self.chroma_scale = size, scale, chroma_scale
        self.even_size_choices = even_size_choices(size)

    def forward(self, coeff, box=None):
        """box: explicit (i, j, h, w) for every sample (parity tests); default: one get_params() draw per sample."""
        Y, C, single, batched = _unpack(coeff)
        B, H, W = Y.shape[0], Y.shape[2], Y.shape[3]
        boxes = [box] * B if box is not None else [self.get_params(H, W) for _ in range(B)]
        return _pack(*_run_chain(Y, C, self.size, boxes, None, None, 0, torch.int16), single, batched)

    def get_params(self, height, width, rng=torch):
        area = height * width
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()
            w = int(round(math.sqrt(target_area)))
            w = _choose_closest(w, self.even_size_choices, width)
            w = int(max(2, w))
            h = w
            if w <= width and h <= height:
                i = int(torch.randint(0, height - h + 1, size=(1,)).item() // self.chroma_scale * self.chroma_scale)
                j = int(torch.randint(0, width - w + 1, size=(1,)).item() // self.chroma_scale * self.chroma_scale)
                return i, j, h, w
        return self.fallback_box(height, width)

    def fallback_box(self, height, width):
        """central crop after 10 failed attempts (custom_transforms.py:612-629); ratio == (1, 1)"""
        in_ratio = float(width) / float(height)
        if in_ratio < 1:
            w = width
            h = int(round(w / 1))
        elif in_ratio > 1:
            h = height
            w = int(round(h * 1))
        else:
            w, h = width, height
        h = int(_choose_closest(h, self.even_size_choices, height))
        w = int(_choose_closest(w, self.even_size_choices, width))
        i = (height - h) // 2 // self.chroma_scale * self.chroma_scale
        j = (width - w) // 2 // self.chroma_scale * self.chroma_scale
        return i, j, max(1, h), max(1, w)


class ResizedCenterCrop_DCT(torch.nn.Module):
    """Eval transform (custom_transforms.py:819-911): crop size_crop/size_resize of the grid, centred, even offsets, then
    resize to size_crop (= resize to size_resize + centre crop, in the cheaper order)."""

    def __init__(self, size_resize, size_crop, chroma_scale=2, dtype_resize=torch.float32):
        super().__init__()
        self.size_resize, self.size_crop, self.chroma_scale = size_resize, size_crop, chroma_scale
        self.size = size_crop
        self.even_size_choices = even_size_choices(size_crop)

    def forward(self, coeff):
        Y, C, single, batched = _unpack(coeff)
        box = self.get_params(Y.shape[2], Y.shape[3])
        return _pack(*_run_chain(Y, C, self.size, [box] * Y.shape[0], None, None, 0, torch.int16), single, batched)

    def get_params(self, height, width):
        ratio = self.size_crop / self.size_resize
        w = _choose_closest(round(ratio * width), self.even_size_choices, width)
        h = _choose_closest(round(ratio * height), self.even_size_choices, height)
        i = (height - int(h)) // 2 // self.chroma_scale * self.chroma_scale
        j = (width - int(w)) // 2 // self.chroma_scale * self.chroma_scale
        return int(i), int(j), int(max(1, h)), int(max(1, w))


def magnitude_table(num_bins=11, image_size=(28, 28)):
    """RandAugment_dct._augmentation_space (custom_transforms.py:1066-1092): op -> (magnitudes, signed)."""
    ls = lambda a, b: torch.linspace(a, b, num_bins)  # noqa: E731
    z = torch.tensor(0.0)
    return {"Identity": (z, False), "AutoContrast": (z, False), "Posterize": (ls(0.0, 5.0).round().int(), False),
            "SolarizeAdd": (ls(0, 883), False), "Color": (ls(0.0, 0.9), True), "Contrast": (ls(0.0, 0.9), True),
            "Brightness": (ls(0.0, 0.9), True), "Sharpness": (ls(0.0, 0.9), True), "Cutout": (ls(0, 6), False),
            "TranslateX": (ls(0.0, 150.0 / 336.0 * image_size[1]), True),
            "TranslateY": (ls(0.0, 150.0 / 336.0 * image_size[0]), True), "Rotate90": (torch.tensor(1), True),
            "AutoSaturation": (z, False), "Grayscale": (z, False), "MidfreqAug": (ls(0.0, 0.9), True),
            "ChromaDrop": (z, False), "Invert": (z, False), "Equalize": (z, False), "Solarize": (ls(818, -818), False),
            "FreqEnhance": (ls(0.0, 0.9), True),
            "Rotate": (ls(0.0, 30.0), True), "ShearX": (ls(0.0, 17.0), True), "ShearY": (ls(0.0, 17.0), True)}   # :1073, :1081-1082


class _FilterBank:
    """8x8 fp32 multiplier tables of MidfreqAug / Sharpness, one per distinct magnitude, computed on the host exactly
    as the reference does (dct_ops.py:696-699, 725-737) and cached on the device."""

    def __init__(self):
        self.keys, self.tables, self.dev = {}, [], None

    def index(self, kind, mag):
        k = (kind, float(mag))
        if k not in self.keys:
            if kind == "MidfreqAug":
                F = dops.midfreq_filter(mag)
            else:
                fh = torch.linspace(1, 1 + 2 * mag, 8, dtype=torch.float32).unsqueeze(1).clamp(min=0)
                fw = torch.linspace(1, 1 + 2 * mag, 8, dtype=torch.float32).unsqueeze(0).clamp(min=0)
                F = fh.mm(fw)
            self.keys[k] = len(self.tables)
            self.tables.append(F.reshape(64).contiguous())
            self.dev = None
        return self.keys[k]

    def device_tensor(self, device):
        if not self.tables:
            return None
        if self.dev is None or self.dev.device != torch.device(device):
            self.dev = torch.stack(self.tables).to(device).contiguous()
        return self.dev


def encode_op(name, magnitude, aux, bank, grid=28):
    """_apply_op_dct (custom_transforms.py:944-1017) argument handling -> (op id, fmag, iarg0, iarg1, iarg2)."""
    if name not in OPS:
        raise ValueError(f"The provided operator {name} is not recognized.")
    op, f, a0, a1, a2 = OPS[name], 0.0, 0, 0, 0
    if name in ("TranslateX", "TranslateY"):
        a0 = int(magnitude - (magnitude % 2))                 # python modulo: -3.75 -> -4, +3.75 -> +2
        if abs(a0) >= grid:
            raise AssertionError("You cannot translate more than the image's size")
    elif name == "Brightness":
        f = float(np.float32((1.0 + magnitude) - 1))
    elif name in ("Color", "Contrast"):
        f = float(np.float32(1.0 + magnitude))
        assert 0 <= 1.0 + magnitude <= 3, "Contrast adjustment factor should be in range [0,3]"
    elif name == "Posterize":
        a0 = int(magnitude)
        a1 = round(2040 / (2 ** a0)) + 1
    elif name == "SolarizeAdd":
        a0 = int(magnitude)
    elif name == "Cutout":
        cs = round(magnitude)
        a0 = int(cs - (cs % 2))
        a1, a2 = int(aux[0]), int(aux[1])
    elif name == "Rotate90":
        a0 = int(magnitude)
        if a0 not in (1, -1):
            raise NotImplementedError("Rotate90 magnitude is +-1 in RandAugment_dct (custom_transforms.py:1086)")
    elif name == "ChromaDrop":
        a0 = int(bool(aux))
    elif name == "Solarize":
        a0 = math.floor(magnitude)                            # int DC > float threshold  <=>  DC > floor(threshold)
    elif name == "FreqEnhance":
        f = float(np.float32(1.0 + magnitude))
    elif name in ("MidfreqAug", "Sharpness"):
        assert -1 <= magnitude <= 1, "Intensity should be within the range of [-1, 1]"
        a0 = bank.index(name, magnitude)
    return op, f, a0, a1, a2


def sample_ops(ops_list, num_ops, magnitude, meta, grid):
    """RandAugment_dct.forward's draws for ONE sample (custom_transforms.py:1108-1123) -> [(name, magnitude, aux)].
    The reference's `list(set(...))` reordering (:1117-1119) makes its op stream irreproducible from a seed; the
    distribution is kept (uniform over the remaining list), the order of the list is not."""
    ops, ops_list = [], list(ops_list)
    for _k in range(num_ops if ops_list else 0):
        name = ops_list[int(torch.randint(len(ops_list), (1,)).item())]
        if name in CHROMA_OPS:
            if name == "Grayscale":
                ops_list = [o for o in ops_list if o not in CHROMA_OPS]
            else:
                ops_list = [o for o in ops_list if o != "Grayscale"]
        mags, signed = meta[name]
        mag = float(mags[magnitude].item()) if mags.ndim > 0 else float(mags.item())
        if signed and int(torch.randint(2, (1,)).item()):
            mag *= -1.0
        aux = None
        if name == "Cutout":
            aux = ((torch.randint(0, grid, (1,)).item()) // 2 * 2, (torch.randint(0, grid, (1,)).item()) // 2 * 2)
        elif name == "ChromaDrop":
            aux = torch.rand(1).item() > 0.5
        ops.append((name, mag, aux))
    return ops


# ----------------------------------------------------------------------------------------------------------------
# The reference's per-transform classes (custom_transforms.py:406-1138) on DEVICE tensors, so that
# `datasets.get_transform('imagenet_dct' | 'imagenet_dct_swin', ...)` composes them exactly like the reference does.
# Each forward takes what the reference's takes -- one int16 coefficient tensor (C,H,W,8,8) or a (Y, CbCr) tuple, optionally
# with a leading batch dimension -- and returns the same structure.  Every class runs the same two HIP kernels as the fused
# TrainTransform_DCT with the other stages switched off (identity crop, no flip, no ops, raw int16 output), so a Compose
# chain of them is bit-identical to the fused transform; the fused one does it in one launch pair per batch.
# ----------------------------------------------------------------------------------------------------------------
_RUNNERS = {}


def _unpack(coeff):
    single = not isinstance(coeff, (tuple, list))
    items = [coeff] if single else list(coeff)
    if len(items) > 2 or not all(torch.is_tensor(t) for t in items):
        raise TypeError("expected a coefficient tensor or a (Y, CbCr) tuple")
    Y = items[0]
    C = items[1] if len(items) == 2 else None
    batched = Y.dim() == 6
    if Y.dim() not in (5, 6):
        raise AssertionError("DCT coefficients should have 5 dimensions of (C, H, W, KH, KW) where KH, KW are typically 8")
    if not batched:
        Y, C = Y.unsqueeze(0), (C.unsqueeze(0) if C is not None else None)
    if Y.shape[1] != 1 or (C is not None and C.shape[1] != 2):
        raise NotImplementedError("the device transforms take a luma tensor (c = 1) optionally followed by CbCr (c = 2)")
    L.require_cuda(Y.contiguous(), None if C is None else C.contiguous())
    return Y.contiguous(), (None if C is None else C.contiguous()), single, batched


def _pack(oy, oc, single, batched):
    if not batched:
        oy, oc = oy[0], (None if oc is None else oc[0])
    return oy if single else (oy, oc)


def _run_chain(Y, C, size, boxes, flips, ops, entry_clamp, out_dtype):
    """One pass of the augment kernels with explicit per-sample parameters; stages that are None are switched off."""
    if Y.dtype != torch.int16 or (C is not None and C.dtype != torch.int16):
        raise TypeError("DCT transforms work on int16 coefficients (ToRange is the only transform that leaves int16)")
    if size not in (28, 32):
        raise NotImplementedError("HIP augment path covers 28 x 28 ('imagenet_dct') and 32 x 32 ('imagenet_dct_swin') grids")
    key = (size, out_dtype)
    if key not in _RUNNERS:
        _RUNNERS[key] = TrainTransform_DCT(size=size, out_dtype=out_dtype)
    t = _RUNNERS[key]
    B = Y.shape[0]
    params = [dict(box=boxes[b], flip=bool(flips[b]) if flips is not None else False, ops=list(ops[b]) if ops is not None else [])
              for b in range(B)]
    quant = t._unit_quant(B, Y.device)
    oy, oc = t(Y, C, quant, params=params, entry_flags=(1 if entry_clamp else 0) | 2)   # | 2: input is de-quantised already
    return oy, (oc if C is not None else None)


class Compose(torch.nn.Module):
    """torchvision.transforms.Compose for these modules (datasets.py:354-382 composes the reference's classes with it)."""

    def __init__(self, transforms):
        super().__init__()
        self.transforms = torch.nn.ModuleList(transforms)

    def forward(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def _whole(Y):
    return (0, 0, Y.shape[2], Y.shape[3])


class Resize_DCT(torch.nn.Module):
    """custom_transforms.py:468-525: resize the whole block grid to `size` (chroma: ceil(size / chroma_scale)).  Square grids of
    size / 2, size or 2 x size blocks on the 28 / 32 pipelines -- everything datasets.get_transform composes -- run on the HIP
    augment kernels; any other grid or size takes dct_ops.resize_dct (the reference's up-by-size/gcd, down-by-H/gcd, as device
    tensor ops; golden g23)."""

    def __init__(self, size, chroma_scale=2, dtype_resize=torch.float32, strict_even_size=False):
        super().__init__()
        if strict_even_size:
            assert size % 2 == 0, f"ERROR: Resize_dct should have even numbered 'size' parameter. Current size: {size}"
        if dtype_resize != torch.float32:
            raise NotImplementedError("resize runs in fp32 (the reference's default dtype_resize)")
        self.size, self.chroma_scale = size, chroma_scale

    def forward(self, coeff):
        Y, C, single, batched = _unpack(coeff)
        H, W = Y.shape[2], Y.shape[3]
        on_kernels = (self.size in (28, 32) and H == W and H in (self.size // 2, self.size, 2 * self.size) and Y.dtype == torch.int16
                      and (C is None or (C.dtype == torch.int16 and C.shape[2] * 2 == H and C.shape[3] * 2 == W and self.chroma_scale == 2)))
        if on_kernels:
            return _pack(*_run_chain(Y, C, self.size, [_whole(Y)] * Y.shape[0], None, None, 0, torch.int16), single, batched)
        oy = dops.resize_dct(Y, self.size)
        oc = None if C is None else dops.resize_dct(C, math.ceil(self.size / self.chroma_scale))
        return _pack(oy, oc, single, batched)


class RandomCrop_DCT(torch.nn.Module):
    """custom_transforms.py:671-745: crop a `size` x `size` block window at a random (even) offset, no resize."""

    def __init__(self, size, chroma_scale=2):
        super().__init__()
        self.size, self.chroma_scale = size, chroma_scale

    def get_params(self, height, width):
        h = w = self.size
        assert w <= width and h <= height, \
            (f"Crop window should be smaller than original image's height and width. Current window size: {h},{w}, "
             f"Original image size: {height},{width}")
        i = int(torch.randint(0, height - h + 1, size=(1,)).item()) // self.chroma_scale * self.chroma_scale
        j = int(torch.randint(0, width - w + 1, size=(1,)).item()) // self.chroma_scale * self.chroma_scale
        return i, j, h, w

    def forward(self, coeff, box=None):
        Y, C, single, batched = _unpack(coeff)
        B = Y.shape[0]
        boxes = [box] * B if box is not None else [self.get_params(Y.shape[2], Y.shape[3]) for _ in range(B)]
        return _pack(*_run_chain(Y, C, self.size, boxes, None, None, 0, torch.int16), single, batched)


class CenterCrop_DCT(torch.nn.Module):
    """custom_transforms.py:747-817: centre crop of `size` blocks (offsets floored to even), no resize."""

    def __init__(self, size, chroma_scale=2):
        super().__init__()
        self.size, self.chroma_scale = size, chroma_scale

    def get_params(self, height, width):
        cs = self.chroma_scale
        w = h = self.size
        assert w <= width and h <= height, \
            (f"Crop window should be smaller than original image's height and width. Current window size: {h},{w}, "
             f"Original image size: {height},{width}")
        i = int(height - self.size) // 2 // cs * cs
        j = int(width - self.size) // 2 // cs * cs
        return i, j, max(1, h // cs * cs), max(1, w // cs * cs)

    def forward(self, coeff):
        Y, C, single, batched = _unpack(coeff)
        box = self.get_params(Y.shape[2], Y.shape[3])
        return _pack(*_run_chain(Y, C, self.size, [box] * Y.shape[0], None, None, 0, torch.int16), single, batched)


class RandomFlip_DCT(torch.nn.Module):
    """custom_transforms.py:913-942: with probability p reverse the block order along W and negate the odd columns of
    every 8x8 block.  (One draw per CALL in the reference, i.e. per sample in its per-sample pipeline: one draw per sample.)"""

    def __init__(self, p=0.5, direction="horizontal"):
        super().__init__()
        if direction not in ("horizontal", "vertical"):
            raise ValueError("direction must be 'horizontal' or 'vertical' (custom_transforms.py:919)")
        self.p, self.direction = p, direction

    def forward(self, coeff, flip=None):
        Y, C, single, batched = _unpack(coeff)
        B = Y.shape[0]
        flips = [bool(flip)] * B if flip is not None else [not (torch.rand(1).item() > self.p) for _ in range(B)]
        if Y.shape[2] != Y.shape[3]:
            raise NotImplementedError("RandomFlip_DCT on the HIP path works on the square grids after the crop/resize stage")
        ops = None
        if self.direction == "vertical":
            # dct_ops.py:617-620 (reverse the block rows, negate the odd rows of every block) = the horizontal flip followed by a
            # half turn; both are exact index / sign work in the kernels (flip in kernel 1, two Rotate90 steps in kernel 2).
            # Unlike flip_dct, kernel 2 clamps to the coefficient range [-1024, 1016] (RandAugment's per-op clamp): a coefficient
            # in [-1024, -1017] that the flip negates comes out as 1016, the value RandAugment's entry clamp gives it one stage
            # later in every reference pipeline; bit exact on [-1016, 1016]
            ops = [[("Rotate90", 1.0, None), ("Rotate90", 1.0, None)] if f else [] for f in flips]
        return _pack(*_run_chain(Y, C, Y.shape[2], [_whole(Y)] * B, flips, ops, 0, torch.int16), single, batched)


# ops_list=None for the FUSED transform and datasets.get_transform(fused=...): the reference's own default
# (custom_transforms.py:1060-1062) minus the DFT-plane Rotate / ShearX / ShearY, which only the class below runs
DEFAULT_OPS = ("AutoContrast", "Equalize", "Invert", "Posterize", "Solarize", "SolarizeAdd", "Color", "Contrast",
               "Brightness", "Sharpness", "Cutout", "TranslateX", "TranslateY")
# RandAugment_dct(ops_list=None): the reference's default list itself, in its order
REFERENCE_DEFAULT_OPS = ("AutoContrast", "Equalize", "Invert", "Rotate", "Posterize", "Solarize", "SolarizeAdd", "Color", "Contrast",
                         "Brightness", "Sharpness", "ShearX", "ShearY", "Cutout", "TranslateX", "TranslateY")


class RandAugment_dct(torch.nn.Module):
    """custom_transforms.py:1024-1138: clamp, then num_ops operations drawn from ops_list at magnitude bin `magnitude`
    (random sign for the signed ones; chroma / grayscale mutual exclusion), clamp after every op.
    `Rotate` / `ShearX` / `ShearY` (the reference's default list has them, no DCT list of utils/configs.py does) resample the DFT plane
    of the padded image (`pad`, default sqrt(2) as in the reference) as device tensor ops between the kernel passes; their
    torchvision sampling is restated, not pinned (dct_ops.py, oracle/dft_np.py)."""

    def __init__(self, num_ops: int = 2, magnitude: int = 10, num_magnitude_bins: int = 11, pad=2 ** 0.5, ops_list=None):
        super().__init__()
        self.num_ops, self.magnitude, self.num_magnitude_bins, self.pad = num_ops, magnitude, num_magnitude_bins, pad
        if ops_list is None:
            ops_list = REFERENCE_DEFAULT_OPS
        bad = [o for o in ops_list if o not in OPS and o not in DFT_OPS]
        if bad:
            raise NotImplementedError(f"operations {bad} are not implemented on the HIP path")
        self.ops_list = list(ops_list)

    def forward(self, coeff, ops=None):
        """ops: explicit [(name, magnitude, aux), ...] applied to every sample (parity tests)."""
        if len(self.ops_list) == 0:
            return coeff
        Y, C, single, batched = _unpack(coeff)
        B, S = Y.shape[0], Y.shape[2]
        if Y.shape[2] != Y.shape[3]:
            raise NotImplementedError("RandAugment_dct on the HIP path works on the square grids after the crop/resize stage")
        meta = magnitude_table(self.num_magnitude_bins, (S, S))
        chosen = [list(ops)] * B if ops is not None else \
            [sample_ops(self.ops_list, self.num_ops, self.magnitude, meta, S) for _ in range(B)]
        # the kernel chains two operations per pass (cfg.TRAIN.NUMOPS default 2); longer chains run as further passes over the int16
        # result -- the entry clamp of a later pass is the identity on what the previous pass's per-op clamp left
        n = max([len(c) for c in chosen] + [1])
        if not any(o[0] in DFT_OPS for c in chosen for o in c):
            for k in range(0, n, 2):
                Y, C = _run_chain(Y, C, S, [_whole(Y)] * B, None, [c[k:k + 2] for c in chosen], 1, torch.int16)
            return _pack(Y, C, single, batched)
        # a DFT-plane op somewhere: one op per pass.  Pass k runs the kernel ops of position k (the entry clamp belongs to the first
        # pass only: the reference clamps once, custom_transforms.py:1106-1107); samples whose op k is Rotate / ShearX / ShearY sit
        # the pass out and go through dct_ops.rotate_block / shear_block (Y and CbCr with the same magnitude, :949-968)
        for k in range(n):
            kern = [[c[k]] if k < len(c) and c[k][0] not in DFT_OPS else [] for c in chosen]
            if k == 0 or any(kern):
                Y, C = _run_chain(Y, C, S, [_whole(Y)] * B, None, kern, 1 if k == 0 else 0, torch.int16)
            for b, c in enumerate(chosen):
                if k < len(c) and c[k][0] in DFT_OPS:
                    name, mag = c[k][0], c[k][1]
                    if name == "Rotate":
                        f = lambda t: dops.rotate_block(t, degrees=mag, pad=self.pad)          # noqa: E731
                    elif name == "ShearX":
                        f = lambda t: dops.shear_block(t, deg_x=mag, pad=self.pad)             # noqa: E731
                    else:
                        f = lambda t: dops.shear_block(t, deg_y=mag, pad=self.pad)             # noqa: E731
                    Y[b] = f(Y[b])
                    if C is not None:
                        C[b] = f(C[b])
        return _pack(Y, C, single, batched)


class ToRange(torch.nn.Module):
    """custom_transforms.py:406-454: x -> (x - orig_min) / (orig_max - orig_min) * (val_max - val_min) + val_min, cast to
    dtype.  The HIP kernel evaluates the pipelines' instance ToRange(-1, 1, -1024, 1016) (datasets.py:360,365,377,381)."""

    def __init__(self, val_min: float = -1., val_max: float = 1., orig_min: float = -1024, orig_max: float = 1024,
                 dtype=torch.float32):
        super().__init__()
        self.val_min, self.val_max, self.orig_min, self.orig_max, self.dtype = val_min, val_max, orig_min, orig_max, dtype
        self._fused = (val_min, val_max, orig_min, orig_max) == (-1, 1, -1024, 1016)
        if dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError("ToRange output dtype: float32 or bfloat16")

    def forward(self, coeff):
        Y, C, single, batched = _unpack(coeff)
        if not self._fused or Y.shape[2] != Y.shape[3] or Y.shape[2] not in (28, 32):
            # any other range (the class default is orig_max = 1024, which no pipeline uses) or grid: the reference's statements
            # (custom_transforms.py:447-451) as device tensor ops -- cast to self.dtype FIRST, then the two statements in that
            # dtype, same operations in the same order, same bits (for bf16 too)
            L.require_cuda(Y) if C is None else L.require_cuda(Y, C)

            def f(x):
                # a TENSOR divisor: dividing by a Python scalar is turned into a multiplication by its reciprocal on the device
                x = x.to(self.dtype)
                den = torch.full((), float(self.orig_max - self.orig_min), device=x.device, dtype=self.dtype)
                x = (x - self.orig_min) / den
                return self.val_min + x * (self.val_max - self.val_min)
            return _pack(f(Y), None if C is None else f(C), single, batched)
        return _pack(*_run_chain(Y, C, Y.shape[2], [_whole(Y)] * Y.shape[0], None, None, 0, self.dtype), single, batched)


class TrainTransform_DCT(torch.nn.Module):
    """Batched device version of get_transform('imagenet_dct', 'train') (datasets.py:354-361)."""

    def __init__(self, size=28, scale=(0.05, 1.0), flip_p=0.5, num_ops=2, magnitude=3, num_magnitude_bins=11,
                 ops_list=None, out_dtype=torch.float32, eval_mode=False, size_resize=32):
        super().__init__()
        if size not in (28, 32):
            raise NotImplementedError("HIP augment path covers 28x28-block (imagenet_dct) and 32x32-block "
                                      "(imagenet_dct_swin) outputs")
        if num_ops > 2:
            raise NotImplementedError("the fused transform chains two operations per sample (cfg.TRAIN.NUMOPS default 2); longer "
                                      "chains: the per-transform classes, RandAugment_dct(num_ops=N)")
        self.size, self.flip_p, self.num_ops, self.magnitude = size, flip_p, num_ops, magnitude
        self.num_magnitude_bins = num_magnitude_bins
        self.ops_list = list(VITTI_OPS if ops_list is None else ops_list)
        bad = [o for o in self.ops_list if o not in OPS]
        if bad:
            raise NotImplementedError(f"operations {bad} are not in the fused transform's kernels (the DFT-plane Rotate / ShearX / "
                                      "ShearY run through the per-transform class RandAugment_dct)")
        self.out_dtype = out_dtype
        self.eval_mode = eval_mode
        self.rrc = RandomResizedCrop_DCT(size, scale=scale, ratio=(1, 1))
        # eval box: ResizedCenterCrop_DCT(32, 28) for the ViT pipeline; Resize_DCT(32) of the whole grid for Swin
        # (datasets.py:362-366, 378-382)
        self.rcc = ResizedCenterCrop_DCT(size_resize, size) if size == 28 else None
        self.bank = _FilterBank()
        self._conv16 = None
        self._ws = None

    # ---- parameter sampling with the reference distributions -------------------------------------------
    def sample_params(self, B, height, width):
        """One dict per sample: box, flip, ops=[(name, magnitude, aux)] (custom_transforms.py:589-610, 934, 1109-1123).
        The reference's `list(set(...))` reordering (:1117-1119) makes its op stream irreproducible from a seed; the
        distribution is kept (uniform over the remaining list), the order of the list is not."""
        out = []
        meta = magnitude_table(self.num_magnitude_bins, (self.size, self.size))
        for _ in range(B):
            if self.eval_mode:
                box = self.rcc.get_params(height, width) if self.rcc is not None else (0, 0, height, width)
                out.append(dict(box=box, flip=False, ops=[]))
                continue
            box = self.rrc.get_params(height, width)
            flip = not (torch.rand(1).item() > self.flip_p)
            ops = sample_ops(self.ops_list, self.num_ops, self.magnitude, meta, self.size)
            out.append(dict(box=box, flip=flip, ops=ops))
        return out

    def _unit_quant(self, B, device):
        """all-ones tables: the coefficients handed to the per-transform classes are already de-quantised (datasets.py:288)"""
        q = getattr(self, "_uq", None)
        if q is None or q.shape[0] < B or q.device != torch.device(device):
            q = self._uq = torch.ones(max(B, 1), 3, 8, 8, device=device, dtype=torch.int16)
        return q[:B]

    def pack(self, params):
        B = len(params)
        arr = (AugParams * B)()
        nops = max([len(p["ops"]) for p in params] + [0])
        if nops > 2:
            raise ValueError("one pass of the augment kernels takes at most two operations per sample")
        for b, p in enumerate(params):
            i, j, h, w = p["box"]
            a = arr[b]
            a.crop_i, a.crop_j, a.crop_h, a.crop_w, a.flip = int(i), int(j), int(h), int(w), int(bool(p["flip"]))
            for s in range(2):
                if s < len(p["ops"]):
                    name, mag, aux = p["ops"][s]
                    a.op[s], a.fmag[s], a.iarg0[s], a.iarg1[s], a.iarg2[s] = encode_op(name, mag, aux, self.bank, self.size)
                else:
                    a.op[s] = 0
        return arr, nops

    def forward(self, Yq, CbCrq, quant, params=None, entry_flags=None):
        """Yq (B,1,Hy,Wy,8,8) int16, CbCrq (B,2,Hc,Wc,8,8) int16 or None, quant (B,3,8,8) int16 -- device tensors.
        entry_flags: rgbnm_dct_augment_ex's `entry_clamp` argument (default: clamp before the ops unless eval_mode)."""
        L.require_cuda(Yq, CbCrq, quant)
        if Yq.dtype != torch.int16 or quant.dtype != torch.int16 or (CbCrq is not None and CbCrq.dtype != torch.int16):
            raise TypeError("coefficients and quantisation tables must be int16 (dct_manip.read_coefficients layout)")
        B, _, Hy, Wy, _, _ = Yq.shape
        Hc, Wc = (CbCrq.shape[2], CbCrq.shape[3]) if CbCrq is not None else (Hy // 2, Wy // 2)
        if params is None:
            params = self.sample_params(B, Hy, Wy)
        arr, nops = self.pack(params)
        dev = Yq.device
        pdev = _params_to_device(self, np.frombuffer(bytes(arr), dtype=np.uint8), dev)
        if self._conv16 is None or self._conv16.device != dev:
            self._conv16 = dops.generate_conversion_matrix(8, 2).to(dev).contiguous()
        filt = self.bank.device_tensor(dev)
        S = self.size
        wsb = L.lib().rgbnm_dct_augment_workspace_ex(B, S)
        if self._ws is None or self._ws.numel() < wsb or self._ws.device != dev:
            self._ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        oy = torch.empty(B, 1, S, S, 8, 8, device=dev, dtype=self.out_dtype)
        oc = torch.empty(B, 2, S // 2, S // 2, 8, 8, device=dev, dtype=self.out_dtype)
        odt = 2 if self.out_dtype == torch.int16 else L.dt_of(self.out_dtype)      # 2: raw int16, no ToRange (rgbnm.h)
        rc = L.lib().rgbnm_dct_augment_ex(Yq.data_ptr(), L.ptr(CbCrq), quant.data_ptr(), pdev.data_ptr(),
                                          C.cast(arr, C.c_void_p), self._conv16.data_ptr(), L.ptr(filt), oy.data_ptr(),
                                          oc.data_ptr(), odt, S, B, Hy, Wy, Hc, Wc,
                                          (0 if self.eval_mode else 1) if entry_flags is None else entry_flags, nops,
                                          self._ws.data_ptr(), self._ws.numel(), L.stream())
        if rc == -1:
            raise L.RgbnmError("dct_augment: invalid parameters (crop side must be size/2, size or 2*size luma blocks "
                               "with even offsets inside the coefficient grid; see include/rgbnm.h)")
        L.check(rc, "dct_augment")
        return oy, oc


def EvalTransform_DCT(**kw):
    """get_transform('imagenet_dct', 'val'|'test') (datasets.py:362-366): ResizedCenterCrop_DCT(32, 28) + ToRange."""
    return TrainTransform_DCT(eval_mode=True, **kw)


# ----------------------------------------------------------------------------------------------------------------
# Vectorised parameter sampling (numpy) with the same distributions as `sample_params`: the per-sample Python loop
# costs ~50 us/sample, far more than the GPU spends on the sample; this one is ~0.1 ms per 256-sample batch.
# ----------------------------------------------------------------------------------------------------------------
AUG_DTYPE = np.dtype([("crop", "i4", 4), ("flip", "i4"), ("op", "i4", 2), ("fmag", "f4", 2), ("iarg0", "i4", 2),
                      ("iarg1", "i4", 2), ("iarg2", "i4", 2)])
assert AUG_DTYPE.itemsize == C.sizeof(AugParams)


class FastParamSampler:
    def __init__(self, transform: "TrainTransform_DCT", seed=0):
        t = self.t = transform
        self.rng = np.random.default_rng(seed)
        meta = magnitude_table(t.num_magnitude_bins, (t.size, t.size))
        names = list(t.ops_list)
        self.names = names
        enc = np.zeros((len(names), 2, 5), dtype=np.float64)        # [op][sign] -> (id, fmag, a0, a1, a2)
        self.signed = np.zeros(len(names), dtype=bool)
        for k, n in enumerate(names):
            mags, signed = meta[n]
            mag = float(mags[t.magnitude].item()) if mags.ndim > 0 else float(mags.item())
            self.signed[k] = signed
            for s, sg in enumerate((1.0, -1.0)):
                aux = (0, 0) if n == "Cutout" else (False if n == "ChromaDrop" else None)
                enc[k, s] = encode_op(n, mag * sg if signed else mag, aux, t.bank, t.size)
        self.enc = enc
        idx = np.arange(len(names))
        is_chroma = np.array([n in CHROMA_OPS for n in names])
        is_gray = np.array([n == "Grayscale" for n in names])
        self.cand_all = idx
        self.cand_after_gray = idx[~is_chroma]          # Grayscale first: no chroma op afterwards (:1116-1117)
        self.cand_after_chroma = idx[~is_gray]          # chroma op first: no Grayscale afterwards (:1118-1119)
        self.is_chroma, self.is_gray = is_chroma, is_gray
        self.k_cut = names.index("Cutout") if "Cutout" in names else -1
        self.k_drop = names.index("ChromaDrop") if "ChromaDrop" in names else -1
        self.choices = np.array(t.rrc.even_size_choices)

    def sides_from_draws(self, u, H, W):
        u = np.asarray(u, dtype=np.float32).astype(np.float64)
        w = np.rint(np.sqrt(H * W * u))          # int(round(math.sqrt(.))): half-even on both sides
        size = self.choices[-1]
        near = self.choices[np.abs(self.choices[None, :] - w[:, None]).argmin(1)]
        big = np.rint((w.astype(np.float32) / np.float32(size))).astype(np.int64) * size
        big = np.where(big > W, big - size, big)
        return np.maximum(2, np.where(w <= size, near, big)).astype(np.int64)

    def boxes_from_draws(self, u, ri, rj, H, W):
        """(i, j, h, w) of a FIRST attempt from explicit draws (u: fp32 uniform; ri / rj: the randint draws in
        [0, H - h] / [0, W - w]); rows whose box does not fit the grid get h = w = -1 (the reference re-draws)."""
        w = self.sides_from_draws(u, H, W)
        fit = (w <= W) & (w <= H)
        box = np.stack([np.asarray(ri) // 2 * 2, np.asarray(rj) // 2 * 2, w, w], axis=1).astype(np.int64)
        box[~fit] = -1
        return box

    def sample_boxes(self, B, H, W):
        """RandomResizedCrop_DCT.get_params for B samples: up to 10 attempts each, then the central-crop fallback."""
        t, r = self.t, self.rng
        box = np.full((B, 4), -1, dtype=np.int64)
        todo = np.arange(B)
        for _ in range(10):
            if todo.size == 0:
                break
            u = r.uniform(t.rrc.scale[0], t.rrc.scale[1], todo.size).astype(np.float32)
            w = self.sides_from_draws(u, H, W)
            fit = (w <= W) & (w <= H)
            wf = w[fit]
            ri = r.integers(0, H - wf + 1)
            rj = r.integers(0, W - wf + 1)
            box[todo[fit]] = np.stack([ri // 2 * 2, rj // 2 * 2, wf, wf], axis=1)
            todo = todo[~fit]
        if todo.size:
            box[todo] = np.asarray(t.rrc.fallback_box(H, W), dtype=np.int64)
        return box

    def sample(self, B, H, W):
        t, r = self.t, self.rng
        out = np.zeros(B, dtype=AUG_DTYPE)
        out["crop"] = self.sample_boxes(B, H, W)
        out["flip"] = r.random(B) <= t.flip_p
        nops = t.num_ops if self.names else 0
        prev = None
        for s in range(nops):
            if s == 0:
                k = self.cand_all[r.integers(0, len(self.cand_all), B)]
            else:
                ua = r.random(B)
                k = np.where(self.is_gray[prev], self.cand_after_gray[(ua * len(self.cand_after_gray)).astype(int)],
                             np.where(self.is_chroma[prev], self.cand_after_chroma[(ua * len(self.cand_after_chroma)).astype(int)],
                                      self.cand_all[(ua * len(self.cand_all)).astype(int)]))
            sg = np.where(self.signed[k], r.integers(0, 2, B), 0)
            e = self.enc[k, sg]
            out["op"][:, s] = e[:, 0]
            out["fmag"][:, s] = e[:, 1]
            out["iarg0"][:, s] = e[:, 2]
            out["iarg1"][:, s] = e[:, 3]
            out["iarg2"][:, s] = e[:, 4]
            cut = k == self.k_cut
            out["iarg1"][cut, s] = r.integers(0, t.size, cut.sum()) // 2 * 2
            out["iarg2"][cut, s] = r.integers(0, t.size, cut.sum()) // 2 * 2
            drop = k == self.k_drop
            out["iarg0"][drop, s] = r.random(drop.sum()) > 0.5
            prev = k
        return out, nops


_PARAM_SLOTS = 16


def _params_to_device(t, host, dev):
    """The packed per-image parameters of one batch -> device, WITHOUT stalling the host: a copy from pageable memory makes the
    host wait until the stream has drained (the whole previous step), so the bytes go through a small ring of pinned slots and one
    asynchronous copy each, like the mixup lambda (cls_transforms.py).  A slot
    is reused _PARAM_SLOTS batches later, after the event behind its copy has completed."""
    nbytes = host.nbytes
    ring = t.__dict__.get("_pring")
    if ring is None or ring["dev"] != dev or ring["host"].shape[1] < nbytes:
        ring = t.__dict__["_pring"] = {"dev": dev, "host": torch.empty(_PARAM_SLOTS, max(nbytes, 4096), dtype=torch.uint8).pin_memory(),
                                       "ev": [None] * _PARAM_SLOTS, "i": 0}
    i = ring["i"]
    ring["i"] = (i + 1) % _PARAM_SLOTS
    if ring["ev"][i] is not None:
        t0 = time.perf_counter()
        ring["ev"][i].synchronize()
        L.HOST_WAIT["sec"] += time.perf_counter() - t0
    slot = ring["host"][i, :nbytes]
    slot.numpy()[:] = host.view(np.uint8).reshape(-1)
    pdev = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    pdev.copy_(slot, non_blocking=True)
    if ring["ev"][i] is None:
        ring["ev"][i] = torch.cuda.Event()
    ring["ev"][i].record()
    return pdev


def apply_packed(transform, Yq, CbCrq, quant, packed, nops, y_off=None, c_off=None, grid=None, out=None):
    """Run the augment kernels with an AUG_DTYPE array produced by FastParamSampler.sample().
    y_off / c_off (device int64 (B,)) + grid=(Hy, Wy): Yq / CbCrq are the flat HOST-CROPPED buffers of
    dct_manip.read_coefficients_batch_crop (image b's crop box alone at Yq[y_off[b]:]); same output, bit for bit."""
    t = transform
    dev = Yq.device
    if y_off is None:
        B, _, Hy, Wy, _, _ = Yq.shape
        Hc, Wc = (CbCrq.shape[2], CbCrq.shape[3]) if CbCrq is not None else (Hy // 2, Wy // 2)
    else:
        B, (Hy, Wy) = len(packed), grid
        Hc, Wc = (Hy + 1) // 2, (Wy + 1) // 2
    host = np.ascontiguousarray(packed)
    pdev = _params_to_device(t, host, dev)
    if t._conv16 is None or t._conv16.device != dev:
        t._conv16 = dops.generate_conversion_matrix(8, 2).to(dev).contiguous()
    filt = t.bank.device_tensor(dev)
    S = t.size
    wsb = L.lib().rgbnm_dct_augment_workspace_ex(B, S)
    if t._ws is None or t._ws.numel() < wsb or t._ws.device != dev:
        t._ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    if out is None:
        oy = torch.empty(B, 1, S, S, 8, 8, device=dev, dtype=t.out_dtype)
        oc = torch.empty(B, 2, S // 2, S // 2, 8, 8, device=dev, dtype=t.out_dtype)
    else:                                   # caller-owned outputs (static buffers of a captured HIP graph)
        oy, oc = out
        if (tuple(oy.shape) != (B, 1, S, S, 8, 8) or tuple(oc.shape) != (B, 2, S // 2, S // 2, 8, 8) or oy.dtype != t.out_dtype
                or oc.dtype != t.out_dtype or not oy.is_contiguous() or not oc.is_contiguous()):
            raise ValueError("out tensors must be contiguous (B,1,S,S,8,8) / (B,2,S/2,S/2,8,8) in the transform's out_dtype")
    if y_off is None:
        L.check(L.lib().rgbnm_dct_augment_ex(Yq.data_ptr(), L.ptr(CbCrq), quant.data_ptr(), pdev.data_ptr(),
                                             host.ctypes.data, t._conv16.data_ptr(), L.ptr(filt), oy.data_ptr(),
                                             oc.data_ptr(), L.dt_of(t.out_dtype), S, B, Hy, Wy, Hc, Wc,
                                             0 if t.eval_mode else 1, nops, t._ws.data_ptr(), t._ws.numel(), L.stream()),
                "dct_augment")
    else:
        L.require_cuda(y_off, c_off)
        L.check(L.lib().rgbnm_dct_augment_packed(Yq.data_ptr(), L.ptr(CbCrq), y_off.data_ptr(), c_off.data_ptr(),
                                                 quant.data_ptr(), pdev.data_ptr(), host.ctypes.data, t._conv16.data_ptr(),
                                                 L.ptr(filt), oy.data_ptr(), oc.data_ptr(), L.dt_of(t.out_dtype), S, B, Hy, Wy,
                                                 Hc, Wc, 0 if t.eval_mode else 1, nops, t._ws.data_ptr(), t._ws.numel(),
                                                 L.stream()), "dct_augment_packed")
    return oy, oc
