"""Host side of the one-launch encoder forward (csrc/vit_chain.hip, rgbnm.h rgbnm_vit_chain_fwd).

The kernel streams a block's four weight matrices (reference: eb_mha.qkv, eb_mha.projection, eb_ffb.0, eb_ffb.3 of
models/plainvit.py:412-529) from a "chain image": the matrices cut into the 24 KB chunks the kernel consumes, in consumption
order, each laid out exactly as it lies in LDS, so that the kernel's DMA wave copies linear 1 KB pieces.  The image is written
every step by rgbnm_chain_gather (dst[i] = src[idx[i]] over the bf16 operand shadows); this module builds the constant int32
table idx.  Layout of one block's image (elements):

  12 attention chunks of 12288:  q0 k0 v0 q1 k1 v1 q2 k2 v2  p0 p1 p2
     q/k/v chunk of head h : LDS row r (0..63, 384 B) holds output feature swap23(r) of that head -- so that registers
                             8 hs .. 8 hs + 7 of the swapped MFMA's accumulator are 8 consecutive features --, its 24 16-byte
                             chunks at positions pchunk(c, r) (bank-conflict-free ds_read_b128)
     p chunk of head h     : LDS row r (0..191, 128 B) holds output feature swap23(r); the 64 reduction indices (head dims of
                             head h) in the order the attention output leaves the PV accumulator: position k' holds dim
                             swap23(k'); 16-byte chunk c at position c ^ fswz(r)
  12 hidden chunks of 2 x 12288: W1 rows of hidden units 64 c + swap23(r) (as the q/k/v chunks) | W2 [192 rows swap23][64 k]
where swap23 exchanges index bits 2 and 3, fswz(r) = bit1(r) << 2 | bit2(r) | bit3(r) << 1 and
pchunk(c, r) = (c & ~7) | ((c & 7) ^ fswz(r)).
"""
import numpy as np

SLOT = 12288            # elements of one chunk (64 x 192 or 192 x 64 bf16)
BLOCK_ELEMS = 12 * SLOT + 12 * 2 * SLOT


def _fswz(r):
    return (((r >> 1) & 1) << 2) | ((r >> 2) & 1) | (((r >> 3) & 1) << 1)


def _swap23(i):
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def _rows384(row_of_r, base, ld):
    """[64 LDS rows][192] image of a K = 192 matrix slice: LDS row r <- source row row_of_r(r); chunks swizzled by pchunk."""
    p = np.arange(SLOT)
    r, e = p // 192, p % 192
    cp, j = e // 8, e % 8
    lc = (cp & ~7) | ((cp & 7) ^ _fswz(r))
    return base + row_of_r(r) * ld + lc * 8 + j


def _rows128(row_of_r, col_of_k, base, ld):
    """[192 LDS rows][64] image of a 64-wide reduction slice: chunk c of row r at c ^ fswz(r); position k' <- column col_of_k(k')."""
    p = np.arange(SLOT)
    r, e = p // 64, p % 64
    cp, j = e // 8, e % 8
    kk = (cp ^ _fswz(r)) * 8 + j
    return base + row_of_r(r) * ld + col_of_k(kk)


def block_index(ws_qkv, ws_proj, ws_fc1, ws_fc2, heads=3, E=192):
    """int64 index table of one block: element offsets into the flat operand-shadow buffer.  ws_*: offsets of the [N, K] shadows
    (qkv rows de-interleaved: q | k | v blocks of heads x 64)."""
    assert heads == 3 and E == 192
    inner, hid = heads * 64, 4 * E
    out = []
    for h in range(heads):
        for m in range(3):
            out.append(_rows384(lambda r, m=m, h=h: m * inner + h * 64 + _swap23(r), ws_qkv, E))
    for h in range(heads):
        out.append(_rows128(_swap23, lambda k, h=h: h * 64 + _swap23(k), ws_proj, inner))
    for c in range(hid // 64):
        out.append(_rows384(lambda r, c=c: c * 64 + _swap23(r), ws_fc1, E))
        out.append(_rows128(_swap23, lambda k, c=c: c * 64 + k, ws_fc2, hid))
    idx = np.concatenate(out)
    assert idx.size == BLOCK_ELEMS
    return idx


def block_index_bwd(wst_qkv, wst_proj, wst_fc1, wst_fc2, heads=3, E=192):
    """Index table of one block's BACKWARD chain image (csrc/vit_chain_bwd.hip), over the TRANSPOSED operand shadows
    (wst_*: offsets of the [K, N] shadows; the qkv one has its 576 columns de-interleaved).  Consumption order:
      12 x ( W2^T chunk: LDS row r (384 B) <- row 64 c + swap23(r) of fc2's [768, 192] shadow, pchunk swizzle
           | W1^T chunk: LDS row r (128 B) <- row r of fc1's [192, 768] shadow, columns 64 c .., chunk q at q ^ fswz(r) )
      3 projection chunks (head h): LDS row r (384 B) <- row 64 h + swap23(r) of the projection's [192, 192] transposed shadow
      9 qkv chunks j: LDS row r (128 B) <- row r of the qkv [192, 576] transposed shadow, columns 64 j .."""
    assert heads == 3 and E == 192
    inner, hid = heads * 64, 4 * E
    out = []
    for c in range(hid // 64):
        out.append(_rows384(lambda r, c=c: c * 64 + _swap23(r), wst_fc2, E))
        out.append(_rows128(lambda r: r, lambda k, c=c: c * 64 + k, wst_fc1, hid))
    for h in range(heads):
        out.append(_rows384(lambda r, h=h: h * 64 + _swap23(r), wst_proj, E))
    for j in range(3 * inner // 64):
        out.append(_rows128(lambda r: r, lambda k, j=j: j * 64 + k, wst_qkv, 3 * inner))
    idx = np.concatenate(out)
    assert idx.size == BLOCK_ELEMS
    return idx
