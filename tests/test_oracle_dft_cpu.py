"""oracle/dft_np.py (DFT-plane Rotate / ShearX / ShearY) -- PARITY UNPINNED: torchvision, whose rotate / affine the
reference calls (utils/dct_torch_utils.py:232-321), is not installed where the reference could be run, so there is no golden vector.
What CAN be pinned is pinned here: the reference's own arithmetic around the resampling against numpy's FFT, exact identities, and that
the restated torchvision sampling really is a rotation / shear of the decoded image."""
import numpy as np
import pytest
import scipy.ndimage as ndi

from oracle import dct_np as O
from oracle import dft_np as D


def _coeffs(C, H, W, seed, smooth=False):
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(-300, 300, size=(C, H, W, 8, 8)).astype(np.int16)
    yy, xx = np.mgrid[0:H * 8, 0:W * 8].astype(np.float32)
    img = np.zeros((C, H * 8, W * 8), np.float32)
    for c in range(C):          # two off-centre elongated blobs (one along x, one along y): rotations, their direction and both shears are visible
        img[c] = 200 * np.exp(-(((xx - W * 5.2) / (W * 1.6)) ** 2 + ((yy - H * 2.5) / (H * 0.4)) ** 2)) \
            + 150 * np.exp(-(((xx - W * 2.2) / (W * 0.4)) ** 2 + ((yy - H * 4.8) / (H * 1.4)) ** 2))
    return _encode(img, H, W)


def _decode(coeff):
    C, H, W, _, _ = coeff.shape
    b = O.basis_matrix(8).astype(np.float64)
    x = np.einsum("ku,chwkl,lv->chwuv", b, coeff.astype(np.float64), b)
    return x.transpose(0, 1, 3, 2, 4).reshape(C, H * 8, W * 8)


def _encode(img, H, W):
    C = img.shape[0]
    b = O.basis_matrix(8).astype(np.float64)
    x = img.reshape(C, H, 8, W, 8).transpose(0, 1, 3, 2, 4).astype(np.float64)
    return np.rint(np.einsum("ku,chwuv,lv->chwkl", b, x, b)).astype(np.int16)


def test_combine_is_the_fft_of_the_decoded_image_and_decompose_inverts_it():
    c = _coeffs(2, 6, 6, 1)
    comp, L, M = D.combine_blocks_dft(c)
    img = _decode(c)
    # dct_ops.py:329-330: rows get the forward DFT (unnormalised), columns the conjugate one divided by their length
    ref = np.fft.ifft(np.fft.fft(img, axis=-2), axis=-1)
    assert np.abs(comp - ref).max() < 2e-3 * np.abs(ref).max()
    back = D.decompose_block_dft(comp, 6, 6, 8, 8, L, M)
    assert np.abs(back - c).max() < 2e-2
    odd = _coeffs(1, 5, 7, 2)              # non-square grid: two conversion matrices
    comp, L, M = D.combine_blocks_dft(odd)
    assert np.abs(D.decompose_block_dft(comp, 5, 7, 8, 8, L, M) - odd).max() < 2e-2


def test_blockshifts_are_inverses():
    x = np.arange(2 * 5 * 7).reshape(2, 5, 7)
    assert np.array_equal(D.iblockshift(D.blockshift(x)), x)


@pytest.mark.parametrize("pad", [False, 2 ** 0.5])
def test_identities(pad):
    c = _coeffs(2, 6, 6, 3)
    assert np.array_equal(D.rotate_block(c, 0.0, pad), c)
    assert np.array_equal(D.shear_block(c, 0.0, 0.0, pad), c)
    for k, deg in ((1, 90.0), (2, 180.0), (3, 270.0), (-1, -90.0), (1, 450.0)):
        assert np.array_equal(D.rotate_block(c, deg, pad), O.rotate90(c, k)), deg


def test_torchvision_restatement_is_identity_at_zero_and_exact_at_right_angles():
    rng = np.random.default_rng(0)
    img = rng.standard_normal((3, 9, 9)).astype(np.float32)
    assert np.array_equal(D.tv_rotate(img, 0.0), img)
    assert np.array_equal(D.tv_affine_shear(img, 0.0, 0.0), img)
    # odd size: the centre is a pixel, a quarter turn maps the grid onto itself (counter-clockwise for positive angles, as PIL)
    assert np.array_equal(D.tv_rotate(img, 90.0), np.rot90(img, 1, axes=(-2, -1)))
    assert np.array_equal(D.tv_rotate(img, -90.0), np.rot90(img, -1, axes=(-2, -1)))


def _corr(a, b):
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def test_rotate_rotates_the_decoded_image():
    c = _coeffs(1, 12, 12, 0, smooth=True)
    img = _decode(c)[0]
    out = _decode(D.rotate_block(c, 20.0, pad=2 ** 0.5))[0]
    ccw = ndi.rotate(img, 20.0, reshape=False, order=1)
    cw = ndi.rotate(img, -20.0, reshape=False, order=1)
    assert _corr(out, img) < 0.8                               # it moved
    assert max(_corr(out, ccw), _corr(out, cw)) > 0.9          # ... by a rotation of 20 degrees
    # the reference's own comment (dct_torch_utils.py:247): counter-clockwise
    assert _corr(out, ccw) > _corr(out, cw)


def test_shear_shears_the_decoded_image():
    c = _coeffs(1, 12, 12, 0, smooth=True)
    img = _decode(c)[0]
    for kw in (dict(deg_x=15.0), dict(deg_y=15.0)):
        out = _decode(D.shear_block(c, pad=2 ** 0.5, **kw))[0]
        t = np.tan(np.radians(15.0))
        best = 0.0
        for s in (t, -t):
            m = np.array([[1.0, s], [0.0, 1.0]]) if "deg_x" in kw else np.array([[1.0, 0.0], [s, 1.0]])
            ctr = (np.array(img.shape) - 1) / 2
            for mm in (m, m.T):
                warped = ndi.affine_transform(img, mm, offset=ctr - mm @ ctr, order=1)
                best = max(best, _corr(out, warped))
        assert _corr(out, img) < 0.95 and best > 0.93, (kw, _corr(out, img), best)
