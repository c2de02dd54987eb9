"""Host-side mirror of the reference's `utils/dct_ops.py` surface for the north-star path.

Constant-table generators (DCT basis / conversion matrices / Gaussian windows) are tiny host
computations done once and handed to the HIP kernels; the per-coefficient work is in csrc/*.hip
(see custom_transforms.py for the batched device transforms).
"""
import math

import torch


def generate_basis_matrix(length=8, scale=True, dtype=torch.float32, device="cpu"):
    """Orthonormal DCT-II basis D[u, x] = c(u) sqrt(2/L) cos(pi/L * u * (x + 1/2)).
    Reference: utils/dct_ops.py:150-169 (same evaluation order, in `dtype`)."""
    u = torch.arange(length, dtype=dtype, device=device).unsqueeze(1)
    x = torch.arange(length, dtype=dtype, device=device).unsqueeze(0) + 0.5
    basis = (u.mm(x) * torch.pi / length).cos()
    if scale:
        basis[0] *= 1 / (2 ** 0.5)
        basis *= (2 / length) ** 0.5
    return basis


def generate_conversion_matrix(length_small=2, mult=2, scale=True, dtype=torch.float32, device="cpu"):
    """A = D_{small*mult} . blockdiag(D_small, ..., D_small)^T  (orthonormal).
    Reference: utils/dct_ops.py:180-208.  A.X.A^T merges mult x mult small DCT blocks into one large one."""
    if mult == 1:
        return torch.eye(length_small, dtype=dtype, device=device)
    big = generate_basis_matrix(length_small * mult, scale=scale, dtype=dtype, device=device)
    small = generate_basis_matrix(length_small, scale=scale, dtype=dtype, device=device)
    blocks = torch.block_diag(*([small] * mult))
    inv = blocks.T if scale else torch.linalg.inv(blocks)
    return big.mm(inv)


def gaussian_window(n, std, dtype=torch.float64):
    """scipy.signal.windows.gaussian(n, std) in closed form (used by midfreqaug_dct, dct_ops.py:732-733)."""
    k = torch.arange(n, dtype=dtype) - (n - 1.0) / 2.0
    return torch.exp(-0.5 * (k / std) ** 2)


def midfreq_filter(intensity, KH=8, KW=8):
    """8x8 multiplier of midfreqaug_dct (dct_ops.py:710-746) in un-shifted coordinates, fp32."""
    hi = KH // 2 - (KH // 8 * 2.2) * abs(intensity)
    wi = KW // 2 - (KW // 8 * 2.2) * abs(intensity)
    fh = gaussian_window(KH, hi).to(torch.float32).unsqueeze(1)
    fw = gaussian_window(KW, wi).to(torch.float32).unsqueeze(0)
    F = fh.mm(fw)
    if intensity >= 0:
        F = 1 / F
    return torch.roll(F, shifts=(-(KH // 2), -(KW // 2)), dims=(0, 1)).contiguous()
