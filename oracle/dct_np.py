"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement, in numpy, of the reference's DCT-domain data path.
Every function cites the reference file:line it follows (paths relative to /root/reference).
Pinned against golden vectors produced by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/g2..g8*.npz); see tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Layout everywhere: (C, H, W, KH, KW) int16 per sample, natural (row = vertical frequency) order.
"""
import math
import numpy as np

CMIN, CMAX = -1024, 1016  # custom_transforms.py:1020, dct_ops.py:34


# ----------------------------------------------------------------------------- a2
def dequantize(Y, CbCr, quant):
    """datasets.py:288-293: int16*int16 -> int16 WITH WRAP-AROUND, then clamp to [-1024, 1016];
    grayscale (CbCr is None) -> zero chroma of half the luma grid."""
    Y = np.asarray(Y, dtype=np.int16)
    q = np.asarray(quant, dtype=np.int16)
    oY = np.clip((Y.astype(np.int32) * q[0].astype(np.int32)).astype(np.int16), CMIN, CMAX)
    if CbCr is None:
        _, h, w, kh, kw = Y.shape
        oC = np.zeros((2, h // 2, w // 2, kh, kw), dtype=np.int16)
    else:
        C = np.asarray(CbCr, dtype=np.int16)
        prod = C.astype(np.int32) * q[1:3, None, None].astype(np.int32)
        oC = np.clip(prod.astype(np.int16), CMIN, CMAX)
    return oY.astype(np.int16), oC.astype(np.int16)


# ----------------------------------------------------------------------------- a13 / a5 matrices
def basis_matrix(length, dtype=np.float32):
    """dct_ops.py:150-169 generate_basis_matrix(scale=True), evaluated in `dtype` like torch does."""
    h = np.arange(length, dtype=dtype)[:, None]
    w = np.arange(length, dtype=dtype)[None, :] + dtype(0.5)
    b = (h @ w).astype(dtype)
    b = (b * dtype(np.pi) / dtype(length)).astype(dtype)
    b = np.cos(b).astype(dtype)
    b[0] *= dtype(1 / (2 ** 0.5))
    b *= dtype((2 / length) ** 0.5)
    return b.astype(dtype)


def conversion_matrix(length_small, mult, dtype=np.float32):
    """dct_ops.py:180-208 generate_conversion_matrix(scale=True): A = D_{big} . blockdiag(D_small)^T."""
    if mult == 1:
        return np.eye(length_small, dtype=dtype)
    big = basis_matrix(length_small * mult, dtype)
    small = basis_matrix(length_small, dtype)
    n = length_small
    blocks = np.zeros((n * mult, n * mult), dtype=dtype)
    for k in range(mult):
        blocks[k * n:(k + 1) * n, k * n:(k + 1) * n] = small
    return (big @ blocks.T).astype(dtype)


# ----------------------------------------------------------------------------- a4
def crop(coeff, top, left, height, width):
    """dct_ops.py:584-599 (in-bounds branch only; the OOB pad branch is never reached on this path)."""
    c, h, w, _, _ = coeff.shape
    assert top >= 0 and left >= 0 and top + height <= h and left + width <= w
    return np.ascontiguousarray(coeff[:, top:top + height, left:left + width])


# ----------------------------------------------------------------------------- a5
def upsample(coeff, L, M, dtype=np.float32):
    """dct_ops.py:436-482: zero-pad 8x8 -> 8Lx8M, scale sqrt(L*M), A^T . P . A, split into LxM blocks."""
    if L == 1 and M == 1:
        return coeff.astype(dtype)
    C, H, W, KH, KW = coeff.shape
    AL = conversion_matrix(KH, L, dtype)
    AM = AL if (L == M and KH == KW) else conversion_matrix(KW, M, dtype)
    P = np.zeros((C, H, W, L * KH, M * KW), dtype=dtype)
    P[..., :KH, :KW] = coeff.astype(dtype) * dtype((L * M) ** 0.5)
    t = np.einsum("lo,chwom->chwlm", AL.T, P).astype(dtype)
    t = np.einsum("chwlo,om->chwlm", t, AM).astype(dtype)
    t = t.reshape(C, H, W, L, KH, M, KW).transpose(0, 1, 3, 2, 5, 4, 6)  # c h l w m kh kw
    return np.ascontiguousarray(t.reshape(C, H * L, W * M, KH, KW))


def downsample(coeff, L, M, dtype=np.float32):
    """dct_ops.py:484-527: gather LxM blocks -> 8Lx8M, A . X . A^T, keep top-left 8x8, / sqrt(L*M)."""
    if L == 1 and M == 1:
        return coeff.astype(dtype)
    C, H, W, KH, KW = coeff.shape
    AL = conversion_matrix(KH, L, dtype)
    AM = AL if (L == M and KH == KW) else conversion_matrix(KW, M, dtype)
    x = coeff.astype(dtype).reshape(C, H // L, L, W // M, M, KH, KW).transpose(0, 1, 3, 2, 5, 4, 6)
    x = x.reshape(C, H // L, W // M, L * KH, M * KW)
    t = np.einsum("lo,chwom->chwlm", AL, x).astype(dtype)
    t = np.einsum("chwlo,om->chwlm", t, AM.T).astype(dtype)
    return (t[..., :KH, :KW] / dtype((L * M) ** 0.5)).astype(dtype)


def resize_raw(coeff, size, dtype=np.float32):
    """dct_ops.py:529-565 (before the final round): up by size/gcd then down by H/gcd."""
    C, H, W, KH, KW = coeff.shape
    hg, wg = math.gcd(H, size), math.gcd(W, size)
    up = upsample(coeff, size // hg, size // wg, dtype)
    return downsample(up, H // hg, W // wg, dtype)


def resize(coeff, size, dtype=np.float32):
    """dct_ops.py:577-578: torch.round (half-to-even) then cast back to the input integer dtype."""
    return np.rint(resize_raw(coeff, size, dtype)).astype(coeff.dtype)


# ----------------------------------------------------------------------------- a6
def flip(coeff, direction="horizontal", fixed_pos=False):
    """dct_ops.py:601-621."""
    out = coeff.copy()
    if direction == "horizontal":
        if not fixed_pos:
            out = out[:, :, ::-1].copy()
        out[..., 1::2] *= -1
    else:
        if not fixed_pos:
            out = out[:, ::-1].copy()
        out[..., 1::2, :] *= -1
    return out


# ----------------------------------------------------------------------------- a7
def rotate90(coeff, rotate):
    """dct_ops.py:99-130 (rotate counted counter-clockwise; torch.rot90 on dims (1,2))."""
    rotate = int(rotate)
    sign = (rotate / abs(rotate)) if rotate != 0 else 1
    r = abs(rotate) % 4
    if r == 0:
        return coeff.copy()
    if sign * r == 3 or sign * r == -1:          # clockwise
        out = np.rot90(coeff, k=-1, axes=(1, 2)).swapaxes(-2, -1)
        return flip(np.ascontiguousarray(out), "horizontal", fixed_pos=True)
    if r == 2:
        return flip(flip(coeff, "vertical"), "horizontal")
    out = np.rot90(coeff, k=1, axes=(1, 2)).swapaxes(-2, -1)   # counter-clockwise
    return flip(np.ascontiguousarray(out), "vertical", fixed_pos=True)


# ----------------------------------------------------------------------------- a8
def translate(coeff, magnitude, direction="H"):
    """dct_ops.py:748-774: torch.roll + zero fill (note: magnitude 0 leaves the tensor untouched
    because `[:0] = 0` is empty)."""
    magnitude = int(magnitude)
    ax = 1 if direction == "H" else 2
    out = np.roll(coeff, magnitude, axis=ax).copy()
    sl = [slice(None)] * 5
    if magnitude >= 0:
        sl[ax] = slice(0, magnitude)
    else:
        sl[ax] = slice(magnitude, None)
    out[tuple(sl)] = 0
    return out


def cutout(coeff, pad_size, center_h, center_w, replace=0):
    """dct_ops.py:776-815 with explicit (even) centre.  Columns [cw-pad, cw+pad) clipped to the grid.
    ROWS ARE MIRRORED: the reference passes padding_dims=(left,right,upper,lower) to F.pad
    (dct_ops.py:803-806), i.e. top pad = `upper_pad` = H-ch-pad, so the zeroed rows are
    [max(0,H-ch-pad), H-max(0,ch-pad)) -- reproduced literally."""
    C, H, W, _, _ = coeff.shape
    lower = max(0, center_h - pad_size)
    upper = max(0, H - center_h - pad_size)
    left = max(0, center_w - pad_size)
    right = max(0, W - center_w - pad_size)
    out = coeff.copy()
    out[:, upper:H - lower, left:W - right] = replace
    return out


# ----------------------------------------------------------------------------- a9
def _round_i16(x):
    return np.rint(x).astype(np.int16)


def brightness(coeff, factor):
    """dct_ops.py:817-837: DC += mean|DC| * (factor-1), fp32, round half-even."""
    out = coeff.copy()
    dc = out[:, :, :, 0, 0].astype(np.float32)
    m = np.float32(np.mean(np.abs(dc), dtype=np.float32))
    dc = dc + m * np.float32(factor - 1)
    out[:, :, :, 0, 0] = _round_i16(dc)
    return out


def contrast(coeff, factor):
    """dct_ops.py:839-860: DC *= factor (on CbCr this is `Color`)."""
    out = coeff.copy()
    dc = out[:, :, :, 0, 0].astype(np.float32) * np.float32(factor)
    out[:, :, :, 0, 0] = _round_i16(dc)
    return out


def autocontrast(coeff):
    """dct_ops.py:862-887: joint min/max over ALL channels of the tensor (Appendix A.7)."""
    out = coeff.copy()
    dc = out[:, :, :, 0, 0].astype(np.float32)
    mn, mx = dc.min(), dc.max()
    if mn == mx and mx == 0:
        return out
    dc = (dc - mn) / (mx - mn)
    dc = np.float32(CMIN) + dc * np.float32(CMAX - CMIN)
    out[:, :, :, 0, 0] = _round_i16(dc)
    return out


def posterize(coeff, bitoffset):
    """dct_ops.py:889-914: idx = round((DC+1024)/2^b); DC = linspace(-1024,1016,round(2040/2^b)+1)[idx]."""
    out = coeff.copy()
    dc = out[:, :, :, 0, 0].astype(np.float32) - np.float32(CMIN)
    dc = dc / np.float32(2 ** bitoffset)
    idx = np.rint(dc).astype(np.int64)
    steps = int(round((CMAX - CMIN) / (2 ** bitoffset))) + 1
    table = torch_linspace_f32(CMIN, CMAX, steps)
    out[:, :, :, 0, 0] = _round_i16(table[idx])
    return out


def torch_linspace_f32(start, end, steps):
    """torch.linspace(float32) CPU kernel: step=(end-start)/(steps-1); first half start+i*step,
    second half end-(steps-1-i)*step (ATen RangeFactories linspace), all in fp32."""
    start, end = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((end - start) / np.float32(steps - 1))
    i = np.arange(steps)
    half = steps // 2
    lo = (start + step * i.astype(np.float32)).astype(np.float32)
    hi = (end - step * (steps - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < half, lo, hi).astype(np.float32)


def solarize_add(coeff, addition, threshold=0):
    """dct_ops.py:653-679: DC < threshold -> DC += addition (int16), then clamp whole tensor."""
    out = coeff.copy()
    dc = out[:, :, :, 0, 0]
    mask = dc < threshold
    dc = dc.copy()
    dc[mask] += np.int16(addition)
    out[:, :, :, 0, 0] = dc
    return np.clip(out, CMIN, CMAX)


# ----------------------------------------------------------------------------- a10
def gaussian_window(n, std):
    """scipy.signal.windows.gaussian(n, std) closed form (dct_ops.py:732-733)."""
    k = np.arange(n, dtype=np.float64) - (n - 1.0) / 2.0
    return np.exp(-0.5 * (k / std) ** 2)


def midfreq_filter(intensity, KH=8, KW=8):
    """The 8x8 multiplier of dct_ops.py:725-741 expressed in UN-shifted coordinates:
    out[u,v] = coeff[u,v] * F[(u+KH/2)%KH, (v+KW/2)%KW]."""
    hi = KH // 2 - (KH // 8 * 2.2) * abs(intensity)
    wi = KW // 2 - (KW // 8 * 2.2) * abs(intensity)
    fh = gaussian_window(KH, hi).astype(np.float32)[:, None]
    fw = gaussian_window(KW, wi).astype(np.float32)[None, :]
    F = (fh @ fw).astype(np.float32)
    if intensity >= 0:
        F = (np.float32(1) / F).astype(np.float32)
    # blockshift rolls by KH//2: shifted[u] = coeff[(u - KH//2) % KH]; multiply; inverse roll.
    return np.roll(np.roll(F, -(KH // 2), 0), -(KW // 2), 1)


def midfreqaug(coeff, intensity):
    """dct_ops.py:710-746."""
    F = midfreq_filter(intensity)
    x = coeff.astype(np.float32) * F
    x = np.clip(x, np.float32(CMIN), np.float32(CMAX))
    return np.rint(x).astype(coeff.dtype)


def sharpblur(coeff, intensity):
    """dct_ops.py:681-708 (non-Ti op lists)."""
    KH, KW = coeff.shape[-2:]
    fh = np.clip(torch_linspace_f32(1, 1 + 2 * intensity, KH), 0, None)[:, None]
    fw = np.clip(torch_linspace_f32(1, 1 + 2 * intensity, KW), 0, None)[None, :]
    F = (fh @ fw).astype(np.float32)
    x = np.clip(coeff.astype(np.float32) * F, np.float32(CMIN), np.float32(CMAX))
    return np.rint(x).astype(coeff.dtype)


# ----------------------------------------------------------------------------- a11
def magnitude_table(num_bins=11, image_size=(28, 28)):
    """custom_transforms.py:1066-1092 (_augmentation_space): op -> (per-bin magnitudes, signed)."""
    ls = lambda a, b: torch_linspace_f32(a, b, num_bins)  # noqa: E731
    z = np.zeros((), np.float32)
    return {
        "Identity": (z, False), "AutoContrast": (z, False), "Equalize": (z, False), "Invert": (z, False),
        "Rotate": (ls(0, 30), True), "Posterize": (np.rint(ls(0, 5)).astype(np.int32), False),
        "Solarize": (ls(818, -818), False), "SolarizeAdd": (ls(0, 883), False),
        "Color": (ls(0, 0.9), True), "Contrast": (ls(0, 0.9), True), "Brightness": (ls(0, 0.9), True),
        "Sharpness": (ls(0, 0.9), True), "ShearX": (ls(0, 17.), True), "ShearY": (ls(0, 17.), True),
        "Cutout": (ls(0, 6), False),
        "TranslateX": (ls(0, 150.0 / 336.0 * image_size[1]), True),
        "TranslateY": (ls(0, 150.0 / 336.0 * image_size[0]), True),
        "Rotate90": (np.array(1), True), "AutoSaturation": (z, False), "Grayscale": (z, False),
        "MidfreqAug": (ls(0, 0.9), True), "FreqEnhance": (ls(0, 0.9), True), "ChromaDrop": (z, False),
    }


def equalize(coeff):
    """dct_ops.py:916-955 (CPU branch: bincount over the 2041 shifted DC values): per channel
    new = round((cdf[dc] - cdf_min) / (N - cdf_min) * 2039) + CMIN, division and product in fp32, round half to even.
    cdf_min = number of blocks holding the smallest DC.  All-equal DCs divide by zero in the reference (undefined
    int16 cast); here they are left unchanged."""
    out = coeff.copy()
    for c in range(coeff.shape[0]):
        dc = coeff[c, :, :, 0, 0].astype(np.int64) - CMIN
        hist = np.bincount(dc.reshape(-1), minlength=CMAX - CMIN + 1)
        nz = hist[hist != 0]
        denom = nz[1:].sum()
        if denom == 0:
            continue
        cdf = np.cumsum(hist)
        eq = np.rint((cdf - nz[0]).astype(np.float32) / np.float32(denom) * np.float32(CMAX - CMIN - 1))
        out[c, :, :, 0, 0] = (eq[dc] + CMIN).astype(coeff.dtype)
    return out


def apply_op(Y, C, op_name, magnitude, aux=None):
    """custom_transforms.py:944-1021 (_apply_op_dct) for the ops on the north-star path; random draws
    of the reference (Cutout centre, ChromaDrop coin) are passed explicitly via `aux`.
    Ends with the per-op clamp (custom_transforms.py:1019-1020)."""
    Y, C = Y.copy(), C.copy()
    if op_name == "TranslateX":
        tb = int(magnitude - (magnitude % 2))          # Python modulo: -3.75 -> -4, +3.75 -> +2
        Y = translate(Y, tb, "W")
        C = translate(C, tb // 2, "W")
    elif op_name == "TranslateY":
        tb = int(magnitude - (magnitude % 2))
        Y = translate(Y, tb, "H")
        C = translate(C, tb // 2, "H")
    elif op_name == "Brightness":
        Y = brightness(Y, 1.0 + magnitude)
    elif op_name == "Color":
        C = contrast(C, 1.0 + magnitude)
    elif op_name == "Contrast":
        Y = contrast(Y, 1.0 + magnitude)
    elif op_name == "Sharpness":
        Y = sharpblur(Y, magnitude)
    elif op_name == "Posterize":
        Y = posterize(Y, int(magnitude))
        C = posterize(C, int(magnitude))
    elif op_name == "AutoContrast":
        Y = autocontrast(Y)
    elif op_name == "Identity":
        pass
    elif op_name == "Cutout":
        cs = round(magnitude)
        cs = int(cs - (cs % 2))
        ch, cw = aux
        Y = cutout(Y, cs, ch, cw)
        C = cutout(C, cs // 2, ch // 2, cw // 2)
    elif op_name == "SolarizeAdd":
        Y = solarize_add(Y, int(magnitude), 0)
    elif op_name == "Rotate90":
        Y = rotate90(Y, magnitude)
        C = rotate90(C, magnitude)
    elif op_name == "AutoSaturation":
        C = autocontrast(C)
    elif op_name == "Grayscale":
        C = C * 0
    elif op_name == "MidfreqAug":
        Y = midfreqaug(Y, magnitude)
    elif op_name == "ChromaDrop":
        drop_cb = bool(aux)
        C[0 if drop_cb else 1] *= 0
    elif op_name == "Equalize":                        # dct_ops.py:916-955: histogram equalisation of the luma DCs
        Y = equalize(Y)
    elif op_name == "Invert":                          # dct_ops.py:623-629 (zero-centred coefficients: * -1)
        Y, C = Y * -1, C * -1
    elif op_name == "Solarize":                        # dct_ops.py:631-651, custom_transforms.py:981-983
        mask = Y[:, :, :, 0, 0] > magnitude            # blocks whose luma DC exceeds the threshold
        Y[mask] *= -1
        cm = np.tile(mask[:, ::2, ::2], (2, 1, 1))     # chroma block (r, c) follows luma block (2r, 2c)
        C[cm] *= -1
    elif op_name == "FreqEnhance":                     # dct_ops.py:1015-1035: AC * factor (fp32), round half to even
        f = np.float32(1.0 + magnitude)
        for T in (Y, C):
            dc = T[..., 0, 0].copy()
            T[...] = np.rint(T.astype(np.float32) * f).astype(T.dtype)
            T[..., 0, 0] = dc
    else:
        raise ValueError(f"The provided operator {op_name} is not recognized.")
    return (np.ascontiguousarray(np.clip(Y, CMIN, CMAX)), np.ascontiguousarray(np.clip(C, CMIN, CMAX)))


# ----------------------------------------------------------------------------- a12
def to_range(x, val_min=-1.0, val_max=1.0, orig_min=-1024, orig_max=1016):
    """custom_transforms.py:436-454: fp32 ((x - omin)/(omax-omin)) then vmin + (.)*(vmax-vmin)."""
    x = x.astype(np.float32)
    x = (x - np.float32(orig_min)) / np.float32(orig_max - orig_min)
    return (np.float32(val_min) + x * np.float32(val_max - val_min)).astype(np.float32)


# ----------------------------------------------------------------------------- a3
def _factors(n):
    out = []
    for i in range(1, int(n ** 0.5) + 1):
        if n % i == 0:
            out += [i, n // i]
    return sorted(out)


def even_size_choices(size):
    """custom_transforms.py:550-555."""
    return [c for c in _factors(size) if c % 2 == 0]


def _choose_closest(val, choices, maxval):
    """custom_transforms.py:571-578 (torch.argmin returns the FIRST minimum; torch.round half-even)."""
    if val <= choices[-1]:
        d = [abs(c - val) for c in choices]
        return choices[int(np.argmin(d))]
    closest = float(np.rint(np.float32(val) / np.float32(choices[-1]))) * choices[-1]
    if closest > maxval:
        closest -= choices[-1]
    return closest


def rrc_box_side(height, width, size, u):
    """custom_transforms.py:590,598-603 with ratio==(1,1): u ~ U(scale) -> side w=h (luma blocks).
    `u` is the python float that torch.empty(1).uniform_(a,b).item() returned (fp32 value)."""
    choices = even_size_choices(size)
    target_area = height * width * u
    w = int(round(math.sqrt(target_area)))
    w = _choose_closest(w, choices, width)
    w = int(max(2, w))
    return w


def rrc_params(height, width, size, u, ri, rj, chroma_scale=2):
    """custom_transforms.py:589-610 first attempt: returns (i, j, h, w) or None if the box does not fit
    (the reference then re-draws)."""
    w = rrc_box_side(height, width, size, u)
    h = w
    if w <= width and h <= height:
        i = int(ri // chroma_scale * chroma_scale)
        j = int(rj // chroma_scale * chroma_scale)
        return i, j, h, w
    return None


def rcc_params(height, width, size_resize=32, size_crop=28, chroma_scale=2):
    """custom_transforms.py:850-882 ResizedCenterCrop_DCT.get_params for the luma tensor (c==1)."""
    choices = even_size_choices(size_crop)
    ratio = size_crop / size_resize
    w = _choose_closest(round(ratio * width), choices, width)
    h = _choose_closest(round(ratio * height), choices, height)
    i = int((height - h) // 2)
    j = int((width - w) // 2)
    i = i // chroma_scale * chroma_scale
    j = j // chroma_scale * chroma_scale
    return int(i), int(j), int(max(1, h)), int(max(1, w))


# ----------------------------------------------------------------------------- whole train transform
def train_transform(Yq, Cq, quant, box, flip_h, ops, size=28):
    """datasets.py:286-293 + get_transform('imagenet_dct','train') (datasets.py:354-361) with all random
    draws made explicit: box=(i,j,h,w) luma blocks; flip_h bool; ops=[(name, magnitude, aux), ...]."""
    Y, C = dequantize(Yq, Cq, quant)
    i, j, h, w = box
    Y = resize(crop(Y, i, j, h, w), size)
    C = resize(crop(C, i // 2, j // 2, max(1, h // 2), max(1, w // 2)), math.ceil(size / 2))
    if flip_h:
        Y, C = flip(Y), flip(C)
    Y, C = np.clip(Y, CMIN, CMAX), np.clip(C, CMIN, CMAX)   # custom_transforms.py:1106-1108
    for name, mag, aux in ops:
        Y, C = apply_op(Y, C, name, mag, aux)
    return to_range(Y), to_range(C)


def eval_transform(Yq, Cq, quant, size_resize=32, size_crop=28):
    """datasets.py:362-366: ResizedCenterCrop_DCT(32, 28) + ToRange."""
    Y, C = dequantize(Yq, Cq, quant)
    i, j, h, w = rcc_params(Y.shape[1], Y.shape[2], size_resize, size_crop)
    Y = resize(crop(Y, i, j, h, w), size_crop)
    C = resize(crop(C, i // 2, j // 2, max(1, h // 2), max(1, w // 2)), math.ceil(size_crop / 2))
    return to_range(Y), to_range(C)
