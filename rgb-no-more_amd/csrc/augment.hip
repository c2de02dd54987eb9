// DCT-domain data path on device, batched: quant-table de-normalise + clamp, block crop,
// resize {x2, identity, /2} by DCT conversion matrices, horizontal flip, two RandAugment ops, ToRange.
// Reference (per sample, CPU, inside DataLoader workers): datasets.py:286-293, utils/custom_transforms.py
// (RandomResizedCrop_DCT :631-663, RandomFlip_DCT :926-942, RandAugment_dct/_apply_op_dct :944-1127, ToRange
// :436-454), utils/dct_ops.py (resize :436-580, flip :601-621, rotate90 :99-130, translate :748-774, cutout
// :776-815, DC photometric ops :817-914, solarize_add :653-679, midfreqaug :710-746, sharpblur :681-708).
//
// Integer / index work is bit exact; resize is fp32 (one LSB on exact .5 ties).  All random draws
// are made on the host and arrive as explicit per-sample parameters (rgbnm_aug_params).
//
//   kernel 1  dct_resize      one wave per 16x16 super-block: raw int16 coefficients -> dequant/clamp -> A.X.A^T
//                             (or A^T.P.A) -> round -> flipped int16 image [B][75264]           (HBM: ~0.6 MB/img)
//   kernel 2  dct_randaug     one 1024-thread workgroup per image, the whole 28x28(+2x14x14) block image lives in
//                             LDS (147 KB): ops are applied in order (geometric ones as register-staged gathers,
//                             photometric ones with an exact integer DC reduction), then ToRange -> fp32/bf16.
#include "common.h"
#include "../../include/rgbnm.h"

namespace {

constexpr int CMIN = -1024, CMAX = 1016;
constexpr int S_Y = 28, S_C = 14;
constexpr int NY = S_Y * S_Y * 64;           // 50176
constexpr int NC1 = S_C * S_C * 64;          // 12544 per chroma plane
constexpr int NIMG = NY + 2 * NC1;           // 75264

__device__ __forceinline__ short dequant(short c, short q) {
  const short w = (short)((int)c * (int)q);   // int16 multiply WITH wrap-around (datasets.py:288)
  return w < CMIN ? (short)CMIN : (w > CMAX ? (short)CMAX : w);
}
__device__ __forceinline__ short clamp_s(int v) { return (short)(v < CMIN ? CMIN : (v > CMAX ? CMAX : v)); }

// ------------------------------------------------------------------------------------------------ kernel 1
struct ResizeArgs {
  const short* Yq; const short* Cq; const short* quant; const rgbnm_aug_params* prm; const float* A16; short* img;
  int B, Hy, Wy, Hc, Wc, has_chroma;
};

__global__ __launch_bounds__(256) void dct_resize_kernel(ResizeArgs a) {
  __shared__ float As[16][17];
  __shared__ float Xs[4][16][17];
  __shared__ float Ts[4][16][17];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  As[tid >> 4][tid & 15] = a.A16[tid];
  __syncthreads();
  const rgbnm_aug_params p = a.prm[b];
  const int mode = p.crop_w == 2 * S_Y ? 0 : (p.crop_w == S_Y ? 1 : 2);   // 0: /2   1: identity   2: x2
  const short* q = a.quant + (size_t)b * 192;
  short* img = a.img + (size_t)b * NIMG;
  const int nwave = gridDim.y * 4, wave = blockIdx.y * 4 + w;

  if (mode != 2) {
    // one unit = one OUTPUT block of plane pl (0 = Y, 1 = Cb, 2 = Cr)
    for (int unit = wave; unit < S_Y * S_Y + 2 * S_C * S_C; unit += nwave) {
      int pl, oy, ox, S;
      if (unit < S_Y * S_Y) { pl = 0; S = S_Y; oy = unit / S_Y; ox = unit % S_Y; }
      else { const int u2 = unit - S_Y * S_Y; pl = 1 + u2 / (S_C * S_C); S = S_C; oy = (u2 % (S_C * S_C)) / S_C; ox = u2 % S_C; }
      const int top = pl ? p.crop_i / 2 : p.crop_i, left = pl ? p.crop_j / 2 : p.crop_j;
      const int H = pl ? a.Hc : a.Hy, W = pl ? a.Wc : a.Wy;
      const short* src = pl ? a.Cq + ((size_t)b * 2 + (pl - 1)) * H * W * 64 : a.Yq + (size_t)b * H * W * 64;
      const bool zero = pl && !a.has_chroma;      // grayscale JPEG: zero chroma (datasets.py:291-293)
      const short* qt = q + pl * 64;
      const int oxf = p.flip ? S - 1 - ox : ox;
      short* dst = img + (pl == 0 ? 0 : NY + (pl - 1) * NC1) + (oy * S + oxf) * 64;
      const int u = lane >> 3, v = lane & 7;
      if (mode == 1) {
        short val = 0;
        if (!zero) val = dequant(src[((size_t)(top + oy) * W + left + ox) * 64 + lane], qt[lane]);
        if (p.flip && (v & 1)) val = (short)-val;
        dst[lane] = val;
      } else {
        // gather 2x2 blocks block-major into X (dct_ops.py:519), dequantised
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = lane + 64 * k, r = idx >> 4, c = idx & 15;
          short val = 0;
          if (!zero) {
            const size_t blk = (size_t)(top + 2 * oy + (r >> 3)) * W + left + 2 * ox + (c >> 3);
            val = dequant(src[blk * 64 + (r & 7) * 8 + (c & 7)], qt[(r & 7) * 8 + (c & 7)]);
          }
          Xs[w][r][c] = (float)val;
        }
        __builtin_amdgcn_wave_barrier();
        // T = A[0:8,:] . X   (8 x 16)
        {
          const int j0 = v * 2;
          float t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float av = As[u][i];
            t0 = fmaf(av, Xs[w][i][j0], t0);
            t1 = fmaf(av, Xs[w][i][j0 + 1], t1);
          }
          Ts[w][u][j0] = t0;
          Ts[w][u][j0 + 1] = t1;
        }
        __builtin_amdgcn_wave_barrier();
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) z = fmaf(Ts[w][u][j], As[v][j], z);
        z = z / 2.0f;                                   // / sqrt(L*M)  (dct_ops.py:526)
        short val = (short)__float2int_rn(z);           // torch.round = half-to-even, then int16 (dct_ops.py:578)
        if (p.flip && (v & 1)) val = (short)-val;
        dst[lane] = val;
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
    // x2: one unit = one INPUT block -> 2x2 output blocks (dct_ops.py:474-482)
    const int SYI = S_Y / 2, SCI = S_C / 2;
    for (int unit = wave; unit < SYI * SYI + 2 * SCI * SCI; unit += nwave) {
      int pl, iy, ix, S;
      if (unit < SYI * SYI) { pl = 0; S = S_Y; iy = unit / SYI; ix = unit % SYI; }
      else { const int u2 = unit - SYI * SYI; pl = 1 + u2 / (SCI * SCI); S = S_C; iy = (u2 % (SCI * SCI)) / SCI; ix = u2 % SCI; }
      const int top = pl ? p.crop_i / 2 : p.crop_i, left = pl ? p.crop_j / 2 : p.crop_j;
      const int H = pl ? a.Hc : a.Hy, W = pl ? a.Wc : a.Wy;
      const short* src = pl ? a.Cq + ((size_t)b * 2 + (pl - 1)) * H * W * 64 : a.Yq + (size_t)b * H * W * 64;
      const bool zero = pl && !a.has_chroma;
      const short* qt = q + pl * 64;
      short* base = img + (pl == 0 ? 0 : NY + (pl - 1) * NC1);
      {
        short val = 0;
        if (!zero) val = dequant(src[((size_t)(top + iy) * W + left + ix) * 64 + lane], qt[lane]);
        Xs[w][lane >> 3][lane & 7] = (float)val * 2.0f;            // * sqrt(L*M)
      }
      __builtin_amdgcn_wave_barrier();
      // T = A^T[:, 0:8] . P   (16 x 8):  T[r][j] = sum_{i<8} A[i][r] * P[i][j]
      {
        const int r = lane >> 2, j0 = (lane & 3) * 2;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float av = As[i][r];
          t0 = fmaf(av, Xs[w][i][j0], t0);
          t1 = fmaf(av, Xs[w][i][j0 + 1], t1);
        }
        Ts[w][r][j0] = t0;
        Ts[w][r][j0 + 1] = t1;
      }
      __builtin_amdgcn_wave_barrier();
      {
        const int r = lane >> 2, s0 = (lane & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = s0 + e;
          float y = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) y = fmaf(Ts[w][r][j], As[j][s], y);
          short val = (short)__float2int_rn(y);
          const int oy = 2 * iy + (r >> 3), ox = 2 * ix + (s >> 3), kv = s & 7;
          const int oxf = p.flip ? S - 1 - ox : ox;
          if (p.flip && (kv & 1)) val = (short)-val;
          base[(oy * S + oxf) * 64 + (r & 7) * 8 + kv] = val;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------ kernel 2
constexpr int AUG_THREADS = 1024;

struct Decoded { int pl, r, c, u, v, S, base; };
__device__ __forceinline__ Decoded decode(int e) {
  Decoded d;
  if (e < NY) { d.pl = 0; d.S = S_Y; d.base = 0; }
  else if (e < NY + NC1) { d.pl = 1; d.S = S_C; d.base = NY; }
  else { d.pl = 2; d.S = S_C; d.base = NY + NC1; }
  const int o = e - d.base, blk = o >> 6;
  d.r = blk / d.S; d.c = blk % d.S; d.u = (o >> 3) & 7; d.v = o & 7;
  return d;
}

__device__ __forceinline__ float linspace_f32(float start, float end, int steps, int i) {
  // torch.linspace fp32 CPU kernel: symmetric evaluation around the midpoint
  const float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

template <typename TO>
__global__ __launch_bounds__(AUG_THREADS) void dct_randaug_kernel(short* inter,
                                                                  const rgbnm_aug_params* __restrict__ prm,
                                                                  const float* __restrict__ filt, TO* __restrict__ outY,
                                                                  TO* __restrict__ outC, int entry_clamp, int nops) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  short* img = reinterpret_cast<short*>(smem_raw);                    // [NIMG]
  int* red = reinterpret_cast<int*>(smem_raw + NIMG * sizeof(short));  // [48]
  unsigned char* smask = smem_raw + NIMG * sizeof(short) + 256;        // [S_Y * S_Y] Solarize: luma blocks to invert
  const int b = blockIdx.x, tid = threadIdx.x;
  const rgbnm_aug_params* pp = prm + b;   // indexed per slot straight from memory (a local copy would go to scratch)
  {
    const uint4* src = reinterpret_cast<const uint4*>(inter + (size_t)b * NIMG);
    uint4* dst = reinterpret_cast<uint4*>(img);
    for (int i = tid; i < NIMG / 8; i += AUG_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  if (entry_clamp) {                                   // custom_transforms.py:1106-1108
    for (int e = tid; e < NIMG; e += AUG_THREADS) img[e] = clamp_s(img[e]);
    __syncthreads();
  }

  for (int slot = 0; slot < nops; ++slot) {
    const int op = pp->op[slot];
    const float fm = pp->fmag[slot];
    const int i0 = pp->iarg0[slot], i1 = pp->iarg1[slot], i2 = pp->iarg2[slot];
    if (op == RGBNM_OP_ROTATE90 || op == RGBNM_OP_TRANSLATEX || op == RGBNM_OP_TRANSLATEY) {
      // geometric permutations (dct_ops.py:99-130, 748-774): spill the current image to this image's slot of the
      // intermediate buffer (150 KB, stays in L2), then gather it back permuted -- no register staging, no LDS copy.
      short* gimg = inter + (size_t)b * NIMG;
      {
        uint4* dst = reinterpret_cast<uint4*>(gimg);
        const uint4* src = reinterpret_cast<const uint4*>(img);
        for (int i = tid; i < NIMG / 8; i += AUG_THREADS) dst[i] = src[i];
      }
      __syncthreads();      // workgroup-scope release/acquire: the same CU re-reads its own stores
      for (int e = 2 * tid; e < NIMG; e += 2 * AUG_THREADS) {   // elements e, e+1: same block, row u, columns v, v+1
        const Decoded d = decode(e);
        short v0 = 0, v1 = 0;
        if (op == RGBNM_OP_ROTATE90) {
          int sr, sc;
          if (i0 > 0) { sr = d.c; sc = d.S - 1 - d.r; }        // counter-clockwise
          else { sr = d.S - 1 - d.c; sc = d.r; }               // clockwise
          const short* sp = gimg + d.base + ((sr * d.S + sc) << 6) + d.v * 8 + d.u;   // per-block transpose
          v0 = sp[0];
          v1 = sp[8];
          if (i0 > 0) { if (d.u & 1) { v0 = (short)-v0; v1 = (short)-v1; } }         // odd rows negated
          else v1 = (short)-v1;                                                       // odd columns (v+1 is odd)
        } else {
          const int sh = d.pl ? (i0 >= 0 ? i0 / 2 : -((-i0 + 1) / 2)) : i0;           // python floor division //2
          int sr = d.r, sc = d.c;
          if (op == RGBNM_OP_TRANSLATEX) sc -= sh;
          else sr -= sh;
          if (sr >= 0 && sr < d.S && sc >= 0 && sc < d.S) {
            const unsigned pr = *reinterpret_cast<const unsigned*>(gimg + d.base + ((sr * d.S + sc) << 6) + d.u * 8 + d.v);
            v0 = (short)(pr & 0xffff);
            v1 = (short)(pr >> 16);
          }
        }
        *reinterpret_cast<unsigned*>(img + e) =
            (unsigned)(unsigned short)clamp_s(v0) | ((unsigned)(unsigned short)clamp_s(v1) << 16);
      }
    } else if (op == RGBNM_OP_CUTOUT || op == RGBNM_OP_GRAYSCALE || op == RGBNM_OP_CHROMADROP) {
      for (int e = tid; e < NIMG; e += AUG_THREADS) {
        const Decoded d = decode(e);
        bool z = false;
        if (op == RGBNM_OP_GRAYSCALE) z = d.pl != 0;
        else if (op == RGBNM_OP_CHROMADROP) z = d.pl == (i0 ? 1 : 2);
        else {
          // rows mirrored exactly as dct_ops.py:796-806 does; i0 = pad (luma), i1/i2 = centre (h, w) in luma blocks
          const int pad = d.pl ? i0 / 2 : i0, ch = d.pl ? i1 / 2 : i1, cw = d.pl ? i2 / 2 : i2;
          const int lower = max(0, ch - pad), upper = max(0, d.S - ch - pad);
          const int left = max(0, cw - pad), right = max(0, d.S - cw - pad);
          z = d.r >= upper && d.r < d.S - lower && d.c >= left && d.c < d.S - right;
        }
        img[e] = z ? (short)0 : clamp_s(img[e]);
      }
    } else if (op == RGBNM_OP_MIDFREQAUG || op == RGBNM_OP_SHARPNESS) {
      // Y only: x * F[u][v] -> clamp (fp32) -> round -> int16   (dct_ops.py:739-746 / 702-708)
      const float* F = filt + (size_t)i0 * 64;
      for (int e = tid; e < NY; e += AUG_THREADS) {
        float x = (float)img[e] * F[e & 63];
        x = fminf(fmaxf(x, (float)CMIN), (float)CMAX);
        img[e] = clamp_s(__float2int_rn(x));
      }
    } else if (op == RGBNM_OP_INVERT || op == RGBNM_OP_SOLARIZE || op == RGBNM_OP_FREQENHANCE) {
      // whole-block ops outside the default lists: invert_dct (dct_ops.py:623-629), solarize_dct (:631-651:
      // blocks whose ORIGINAL luma DC exceeds the threshold are negated; chroma block (r,c) follows luma block (2r,2c),
      // custom_transforms.py:981-983), freq_enhance_dct (:1015-1035: AC * factor, round half to even, DC untouched)
      if (op == RGBNM_OP_SOLARIZE) {
        for (int i = tid; i < S_Y * S_Y; i += AUG_THREADS) smask[i] = img[i * 64] > i0 ? 1 : 0;
        __syncthreads();
      }
      for (int e = tid; e < NIMG; e += AUG_THREADS) {
        const Decoded d = decode(e);
        int v = img[e];
        if (op == RGBNM_OP_INVERT) v = -v;
        else if (op == RGBNM_OP_SOLARIZE) {
          const int m = d.pl == 0 ? smask[d.r * S_Y + d.c] : smask[2 * d.r * S_Y + 2 * d.c];
          v = m ? -v : v;
        } else if ((e & 63) != 0) v = __float2int_rn((float)v * fm);
        img[e] = clamp_s(v);
      }
    } else if (op != RGBNM_OP_IDENTITY) {
      // ---- DC photometric ops: exact integer reduction over the DC set, fp32 update, round, clamp ----
      const bool onC = (op == RGBNM_OP_COLOR || op == RGBNM_OP_AUTOSATURATION);
      const bool both = (op == RGBNM_OP_POSTERIZE);
      const int ndc = onC ? 2 * S_C * S_C : S_Y * S_Y;
      const int dcbase = onC ? NY : 0;
      float gsum = 0.f, gmin = 0.f, gmax = 0.f;
      if (op == RGBNM_OP_BRIGHTNESS || op == RGBNM_OP_AUTOCONTRAST || op == RGBNM_OP_AUTOSATURATION) {
        int s = 0, mn = 32767, mx = -32768;
        for (int i = tid; i < ndc; i += AUG_THREADS) {
          const int dc = img[dcbase + i * 64];
          s += dc < 0 ? -dc : dc;
          mn = min(mn, dc);
          mx = max(mx, dc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          s += __shfl_xor(s, o, 64);
          mn = min(mn, __shfl_xor(mn, o, 64));
          mx = max(mx, __shfl_xor(mx, o, 64));
        }
        if ((tid & 63) == 0) { red[(tid >> 6) * 3] = s; red[(tid >> 6) * 3 + 1] = mn; red[(tid >> 6) * 3 + 2] = mx; }
        __syncthreads();
        s = 0; mn = 32767; mx = -32768;
        for (int wv = 0; wv < AUG_THREADS / 64; ++wv) { s += red[wv * 3]; mn = min(mn, red[wv * 3 + 1]); mx = max(mx, red[wv * 3 + 2]); }
        gsum = (float)s;               // |DC| <= 1024, <= 784 terms: exact in fp32 in any summation order
        gmin = (float)mn;
        gmax = (float)mx;
        __syncthreads();
      }
      const int total = both ? S_Y * S_Y + 2 * S_C * S_C : ndc;
      for (int i = tid; i < total; i += AUG_THREADS) {
        const int idx = both ? (i < S_Y * S_Y ? i * 64 : NY + (i - S_Y * S_Y) * 64) : dcbase + i * 64;
        const short dc16 = img[idx];
        float dc = (float)dc16;
        int res;
        if (op == RGBNM_OP_BRIGHTNESS) {
          dc = dc + (gsum / (float)ndc) * fm;                      // fm = factor - 1  (dct_ops.py:832)
          res = __float2int_rn(dc);
        } else if (op == RGBNM_OP_CONTRAST || op == RGBNM_OP_COLOR) {
          res = __float2int_rn(dc * fm);                           // fm = factor      (dct_ops.py:856)
        } else if (op == RGBNM_OP_AUTOCONTRAST || op == RGBNM_OP_AUTOSATURATION) {
          if (gmin == gmax && gmax == 0.f) res = dc16;             // dct_ops.py:879
          else {
            dc = (dc - gmin) / (gmax - gmin);
            dc = (float)CMIN + dc * (float)(CMAX - CMIN);
            res = __float2int_rn(dc);
          }
        } else if (op == RGBNM_OP_POSTERIZE) {
          dc = (dc - (float)CMIN) / (float)(1 << i0);              // i0 = bit offset
          const int k = __float2int_rn(dc);
          res = __float2int_rn(linspace_f32((float)CMIN, (float)CMAX, i1, k));   // i1 = table length
        } else {   // RGBNM_OP_SOLARIZEADD: int16 add below threshold 0 (dct_ops.py:672-677)
          res = dc16 < 0 ? (int)(short)(dc16 + (short)i0) : (int)dc16;
        }
        img[idx] = clamp_s((int)(short)res);
      }
    }
    __syncthreads();
    if (op != RGBNM_OP_ROTATE90 && op != RGBNM_OP_TRANSLATEX && op != RGBNM_OP_TRANSLATEY && op != RGBNM_OP_CUTOUT &&
        op != RGBNM_OP_GRAYSCALE && op != RGBNM_OP_CHROMADROP) {
      // per-op clamp of the untouched coefficients too (custom_transforms.py:1019-1020)
      for (int e = tid; e < NIMG; e += AUG_THREADS) img[e] = clamp_s(img[e]);
      __syncthreads();
    }
  }

  // ---- ToRange (custom_transforms.py:451-452): ((x + 1024) / 2040) * 2 - 1, fp32, then cast ----
  TO* oy = outY + (size_t)b * NY;
  TO* oc = outC + (size_t)b * 2 * NC1;
  for (int e4 = tid * 4; e4 < NIMG; e4 += AUG_THREADS * 4) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = (float)img[e4 + k];
      x = (x - (float)CMIN) / (float)(CMAX - CMIN);
      v[k] = -1.0f + x * 2.0f;
    }
    if (e4 < NY) store4<TO>(oy + e4, v);
    else store4<TO>(oc + (e4 - NY), v);
  }
}

}  // namespace

extern "C" {

size_t rgbnm_dct_augment_workspace(int B) { return (size_t)B * NIMG * sizeof(short); }

int rgbnm_dct_augment(const int16_t* Yq, const int16_t* CbCrq, const int16_t* quant, const rgbnm_aug_params* params_dev,
                      const rgbnm_aug_params* params_host, const float* conv16, const float* filters, void* outY,
                      void* outC, int out_dtype, int B, int Hy, int Wy, int Hc, int Wc, int entry_clamp, int nops,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!Yq || !quant || !params_dev || !params_host || !conv16 || !outY || !outC || !workspace || B <= 0) return RGBNM_EINVAL;
  if (nops < 0 || nops > 2) return RGBNM_EINVAL;
  if (workspace_bytes < rgbnm_dct_augment_workspace(B)) return RGBNM_EWORKSPACE;
  // host-side validation of every crop box (the HIP path implements the x2 / identity / /2 resize cases)
  for (int b = 0; b < B; ++b) {
    const rgbnm_aug_params& p = params_host[b];
    if (p.crop_h != p.crop_w) return RGBNM_EINVAL;
    if (p.crop_w != 2 * S_Y && p.crop_w != S_Y && 2 * p.crop_w != S_Y) return RGBNM_EINVAL;
    if ((p.crop_i & 1) || (p.crop_j & 1) || p.crop_i < 0 || p.crop_j < 0) return RGBNM_EINVAL;
    if (p.crop_i + p.crop_h > Hy || p.crop_j + p.crop_w > Wy) return RGBNM_EINVAL;
    if (CbCrq && (p.crop_i / 2 + p.crop_h / 2 > Hc || p.crop_j / 2 + p.crop_w / 2 > Wc)) return RGBNM_EINVAL;
    for (int s = 0; s < nops; ++s)
      if (p.op[s] < 0 || p.op[s] > RGBNM_OP_FREQENHANCE) return RGBNM_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  ResizeArgs a;
  a.Yq = Yq; a.Cq = CbCrq; a.quant = quant; a.prm = params_dev; a.A16 = conv16; a.img = (short*)workspace;
  a.B = B; a.Hy = Hy; a.Wy = Wy; a.Hc = Hc; a.Wc = Wc; a.has_chroma = CbCrq != nullptr;
  hipLaunchKernelGGL(dct_resize_kernel, dim3(B, 8), dim3(256), 0, st, a);
  LAUNCH_CHECK();
  const size_t smem = NIMG * sizeof(short) + 256 + 1024;    // image + reduction scratch + Solarize block mask
  if (out_dtype == DT_F32) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)dct_randaug_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return RGBNM_ELAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL((dct_randaug_kernel<float>), dim3(B), dim3(AUG_THREADS), smem, st, (short*)workspace,
                       params_dev, filters, (float*)outY, (float*)outC, entry_clamp, nops);
  } else if (out_dtype == DT_BF16) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)dct_randaug_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return RGBNM_ELAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL((dct_randaug_kernel<bf16>), dim3(B), dim3(AUG_THREADS), smem, st, (short*)workspace,
                       params_dev, filters, (bf16*)outY, (bf16*)outC, entry_clamp, nops);
  } else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // extern "C"
