"""TEST INFRASTRUCTURE ONLY (imported by tests/; never by the package, never timed).

numpy restatement of the reference's DFT-plane `Rotate` / `ShearX` / `ShearY` of RandAugment_dct:
  utils/dct_ops.py:135-146 (generate_fourier_basis), :210-232 (generate_conversion_matrix_dft), :303-364 (combine_blocks_dft /
  decompose_block_dft), :62-97 (blockshift / iblockshift), :367-434 (rotate_block), :957-1013 (shear_block),
  utils/dct_torch_utils.py:232-321 (rotate_dft_2d_spatial / shear_dft_2d_spatial), utils/custom_transforms.py:949-968 (dispatch).

**PARITY UNPINNED.**  The reference resamples the shifted DFT plane with `torchvision.transforms.functional.rotate / affine`
(nearest neighbour, expand=False, fill 0).  torchvision is a third-party dependency that is NOT installed in the survey container
(the reference pins no version: no requirements file; its torch is 2.x, i.e. torchvision >= 0.15), so no golden vector of these
three ops could be generated from the reference.  `tv_inverse_affine_matrix`, `tv_affine_grid` and `tv_grid_sample_nearest` below
restate torchvision's PUBLISHED tensor path (torchvision/transforms/functional.py `rotate`, `affine`,
`_get_inverse_affine_matrix`; _functional_tensor.py `_gen_affine_grid`, `_apply_grid_transform`; torch grid_sample with
mode='nearest', padding_mode='zeros', align_corners=False) -- anchored on the reference's call sites only.  Everything around the
resampling (conversion matrices, block shifts, 90-degree pre-rotation, rounding) restates the reference's own code and is checked by
structural properties in tests/test_oracle_dft_cpu.py (combine / decompose are inverses and equal numpy's FFT of the block-IDCT image,
angle 0 / shear 0 are the identity, multiples of 90 degrees equal the exact Rotate90).
"""
import math

import numpy as np

from . import dct_np as O


def fourier_basis(n):
    """dct_ops.py:135-146: exp(-2 pi i t k / n) / sqrt(n), complex64 arithmetic from an fp32 outer product."""
    t = np.arange(n, dtype=np.float32)
    prod = np.outer(t, t).astype(np.float32)
    return (np.exp(prod.astype(np.complex64) * np.complex64(-2j * np.pi / n)) / np.float32(n ** 0.5)).astype(np.complex64)


def conversion_matrix_dft(length_small, mult):
    """dct_ops.py:210-232: (DFT basis of the whole axis) @ (block-diagonal DCT basis)^T -- small DCT blocks -> one large DFT."""
    b = O.basis_matrix(length_small)                       # orthonormal DCT-II basis (dct_ops.py:150-169)
    n = length_small * mult
    blocks = np.zeros((n, n), dtype=np.float32)
    for i in range(mult):
        blocks[i * length_small:(i + 1) * length_small, i * length_small:(i + 1) * length_small] = b
    return (fourier_basis(n) @ blocks.T.astype(np.complex64)).astype(np.complex64)


def combine_blocks_dft(coeff):
    """dct_ops.py:303-332: (C,H,W,KH,KW) -> (C, H KH, W KW) complex: the DFT of the image the blocks decode to."""
    C, H, W, KH, KW = coeff.shape
    L = conversion_matrix_dft(KH, H)
    M = L if (H == W and KH == KW) else conversion_matrix_dft(KW, W)
    x = coeff.astype(np.complex64).transpose(0, 1, 3, 2, 4).reshape(C, H * KH, W * KW)
    y = np.einsum("ho,cow->chw", L, x) * np.float32((KH * H) ** 0.5)
    y = np.einsum("cho,ow->chw", y, np.conj(M.T)) / np.float32((KW * W) ** 0.5)
    return y.astype(np.complex64), L, M


def decompose_block_dft(coeff, H, W, KH, KW, L=None, M=None):
    """dct_ops.py:334-364."""
    L = conversion_matrix_dft(KH, H) if L is None else L
    M = (L if (H == W and KH == KW) else conversion_matrix_dft(KW, W)) if M is None else M
    y = np.einsum("ho,cow->chw", np.conj(L.T), coeff.astype(np.complex64)) / np.float32((KH * H) ** 0.5)
    y = np.einsum("cho,ow->chw", y, M) * np.float32((KW * W) ** 0.5)
    C = y.shape[0]
    return y.reshape(C, H, KH, W, KW).transpose(0, 1, 3, 2, 4).real.astype(np.float32)


def blockshift(x, dims=(1, 2)):
    """dct_ops.py:62-77: roll by H // 2, W // 2."""
    return np.roll(np.roll(x, x.shape[dims[0]] // 2, dims[0]), x.shape[dims[1]] // 2, dims[1])


def iblockshift(x, dims=(1, 2)):
    """dct_ops.py:79-97: roll by H - H // 2, W - W // 2."""
    h, w = x.shape[dims[0]], x.shape[dims[1]]
    return np.roll(np.roll(x, h - h // 2, dims[0]), w - w // 2, dims[1])


# ------------------------------------------------------------------ torchvision's tensor path, restated from its published source
def tv_inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix(..., inverted=True): python floats."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def tv_affine_grid(matrix, w, h):
    """_functional_tensor._gen_affine_grid with ow = w, oh = h (expand=False), fp32: normalised sampling positions (h, w, 2)."""
    theta = np.asarray(matrix, dtype=np.float32).reshape(2, 3)
    xg = O.torch_linspace_f32(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, w)
    yg = O.torch_linspace_f32(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, h)
    base = np.empty((h, w, 3), dtype=np.float32)
    base[..., 0] = xg[None, :]
    base[..., 1] = yg[:, None]
    base[..., 2] = 1.0
    rescaled = (theta.T / np.array([0.5 * w, 0.5 * h], dtype=np.float32)).astype(np.float32)       # (3, 2)
    return (base.reshape(-1, 3) @ rescaled).astype(np.float32).reshape(h, w, 2)


def tv_grid_sample_nearest(img, grid):
    """torch.nn.functional.grid_sample(mode='nearest', padding_mode='zeros', align_corners=False) on (C,H,W) fp32: source index =
    nearbyint(((g + 1) * size - 1) / 2) (round half to even), zeros outside."""
    C, H, W = img.shape
    ix = ((grid[..., 0] + np.float32(1)) * np.float32(W) - np.float32(1)) / np.float32(2)
    iy = ((grid[..., 1] + np.float32(1)) * np.float32(H) - np.float32(1)) / np.float32(2)
    jx, jy = np.rint(ix).astype(np.int64), np.rint(iy).astype(np.int64)
    ok = (jx >= 0) & (jx < W) & (jy >= 0) & (jy < H)
    out = np.zeros_like(img)
    out[:, ok] = img[:, jy[ok], jx[ok]]
    return out


def tv_rotate(img, angle):
    """functional.rotate(img, angle, NEAREST, expand=False, center=None, fill=None) for a float tensor (C,H,W)."""
    m = tv_inverse_affine_matrix([0.0, 0.0], -angle, [0.0, 0.0], 1.0, [0.0, 0.0])
    return tv_grid_sample_nearest(img, tv_affine_grid(m, img.shape[-1], img.shape[-2]))


def tv_affine_shear(img, deg_x, deg_y):
    """functional.affine(img, angle=0, translate=[0,0], scale=1, shear=[deg_x, deg_y], NEAREST, fill=0) for a float tensor."""
    m = tv_inverse_affine_matrix([0.0, 0.0], 0.0, [0.0, 0.0], 1.0, [deg_x, deg_y])
    return tv_grid_sample_nearest(img, tv_affine_grid(m, img.shape[-1], img.shape[-2]))


# ------------------------------------------------------------------ the reference's own flow around the resampling
def _pad(coeff, pad):
    C, H, W, KH, KW = coeff.shape
    if not pad:
        return coeff, 0, 0
    assert pad >= 1
    Hp, Wp = int(H * pad // 1), int(W * pad // 1)
    out = np.zeros((C, Hp, Wp, KH, KW), dtype=coeff.dtype)
    hm, wm = (Hp - H) // 2, (Wp - W) // 2
    out[:, hm:hm + H, wm:wm + W] = coeff
    return out, hm, wm


def _finish(x, coeff, hm, wm, pad):
    C, H, W, KH, KW = coeff.shape
    if pad:
        x = x[:, hm:hm + H, wm:wm + W]
    if np.issubdtype(coeff.dtype, np.integer):
        x = np.rint(x)                      # torch.round: half to even
    return x.astype(coeff.dtype)


def rotate_block(coeff, degrees, pad=False):
    """dct_ops.py:367-434 (window=False): multiples of 90 degrees by the exact Rotate90, the rest (-45..45) on the DFT plane."""
    C, H, W, KH, KW = coeff.shape
    x, hm, wm = _pad(coeff, pad)
    Hp, Wp = x.shape[1], x.shape[2]
    sign = degrees / abs(degrees) if degrees != 0 else 1
    rem = sign * (abs(degrees) % 360)
    shifted = (rem + 360 + 45) % 360
    rot90s = shifted // 90
    left = -((rot90s * 90) - (shifted - 45))
    x = O.rotate90(x, rot90s)
    x = blockshift(x)
    comp, L, M = combine_blocks_dft(x)
    sh = np.fft.fftshift(comp, axes=(-2, -1))
    deg = -left                                         # dct_torch_utils.py:247 "degrees *= -1"
    rot = tv_rotate(sh.real.astype(np.float32), deg) + 1j * tv_rotate(sh.imag.astype(np.float32), deg)
    un = np.fft.ifftshift(rot.astype(np.complex64), axes=(-2, -1))
    dec = decompose_block_dft(un, Hp, Wp, KH, KW, L, M)
    dec = iblockshift(dec)
    return _finish(dec, coeff, hm, wm, pad)


def shear_block(coeff, deg_x=0.0, deg_y=0.0, pad=False):
    """dct_ops.py:957-1013 (window=False) with dct_torch_utils.py:268-321 (real=False)."""
    C, H, W, KH, KW = coeff.shape
    x, hm, wm = _pad(coeff, pad)
    Hp, Wp = x.shape[1], x.shape[2]
    x = blockshift(x)
    comp, L, M = combine_blocks_dft(x)
    sh = np.fft.fftshift(comp, axes=(-2, -1))
    s = tv_affine_shear(sh.real.astype(np.float32), deg_x, deg_y) + 1j * tv_affine_shear(sh.imag.astype(np.float32), deg_x, deg_y)
    un = np.fft.ifftshift(s.astype(np.complex64), axes=(-2, -1))
    dec = decompose_block_dft(un, Hp, Wp, KH, KW, L, M)
    dec = iblockshift(dec)
    return _finish(dec, coeff, hm, wm, pad)


def apply_dft_op(Y, C, name, magnitude, pad=2 ** 0.5):
    """custom_transforms.py:949-968: the op runs on Y and on CbCr with the same magnitude."""
    if name == "Rotate":
        f = lambda t: rotate_block(t, magnitude, pad)          # noqa: E731
    elif name == "ShearX":
        f = lambda t: shear_block(t, deg_x=magnitude, pad=pad)  # noqa: E731
    elif name == "ShearY":
        f = lambda t: shear_block(t, deg_y=magnitude, pad=pad)  # noqa: E731
    else:
        raise ValueError(name)
    return f(Y), (None if C is None else f(C))
