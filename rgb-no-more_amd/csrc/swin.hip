// SwinV2 DCT specifics (BASELINE config 5; reference models/swinv2.py): everything of the model
// that is not a plain Linear - the Linears run on the GEMM family of gemm.hip.
//   swin_embed      8x8 DCT blocks DEcomposed into 4x4 (Y) / 2x2 (CbCr) sub-block tokens: X' = A^T X A per block,
//                   einops split '(p1 pdh) (p2 pdw)' (coefficient-major!), 16 + 2*4 = 24 features per token
//                   (swinv2.py:556-576, plainvit.py:50-88)
//   ln_generic      LayerNorm for any width (96 / 192 / 384 / 768 here), optional residual and per-sample scale:
//                   y = res + s_b * LN(x)  (res-post-norm + DropPath, swinv2.py:302-307), backward with partial
//                   gamma / beta sums for the batched reduction (reduce.hip)
//   (window attention lives in swin_attn.hip)
//   merge_gather    PatchMerging's 2x2 neighbourhood concat as one gather (scatter in backward) (swinv2.py:357-362)
//   token_mean      AdaptiveAvgPool1d(1) over tokens (swinv2.py:703-705)
#include "common.h"
#include "../../include/rgbnm.h"
#include "internal.h"

#define TRYRC(x) do { const int rc__ = (x); if (rc__ != RGBNM_OK) return rc__; } while (0)

namespace {
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));


// ------------------------------------------------------------------------------------------------ embed
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void swin_embed_kernel(const TI* __restrict__ y, const TI* __restrict__ cbcr,
                                                         const float* __restrict__ Ay, const float* __restrict__ Ac,
                                                         TO* __restrict__ feat, int B, int H, int W) {
  __shared__ float Xs[4][64], Ts[4][64], As[2][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 64) As[0][threadIdx.x] = Ay[threadIdx.x];
  else if (threadIdx.x < 128) As[1][threadIdx.x - 64] = Ac[threadIdx.x - 64];
  const int Hc = H / 2, Wc = W / 2;
  const long long nY = (long long)B * H * W, nC = (long long)B * 2 * Hc * Wc;
  const long long blk = (long long)blockIdx.x * 4 + w;
  const bool valid = blk < nY + nC;
  const bool luma = blk < nY;
  const int r = lane >> 3, c = lane & 7;
  if (valid) Xs[w][lane] = to_f32(luma ? y[blk * 64 + lane] : cbcr[(blk - nY) * 64 + lane]);
  __syncthreads();
  if (!valid) return;
  const float* A = As[luma ? 0 : 1];
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) t += A[k * 8 + r] * Xs[w][k * 8 + c];       // (A^T X)[r][c]
  Ts[w][lane] = t;
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += Ts[w][r * 8 + k] * A[k * 8 + c];       // (A^T X A)[r][c]
  const int TW = 2 * W;                                                    // token grid is 2H x 2W
  if (luma) {
    const int b = (int)(blk / (H * W)), hw = (int)(blk % (H * W)), h = hw / W, ww = hw % W;
    const int p1 = r >> 1, pdh = r & 1, p2 = c >> 1, pdw = c & 1;
    const long long tok = ((long long)b * 2 * H + 2 * h + pdh) * TW + 2 * ww + pdw;
    feat[tok * 24 + p1 * 4 + p2] = from_f32<TO>(v);
  } else {
    const long long cb = blk - nY;
    const int b = (int)(cb / (2 * Hc * Wc)), rem = (int)(cb % (2 * Hc * Wc));
    const int ch = rem / (Hc * Wc), hw = rem % (Hc * Wc), h = hw / Wc, ww = hw % Wc;
    const int p1 = r >> 2, pdh = r & 3, p2 = c >> 2, pdw = c & 3;
    const long long tok = ((long long)b * 2 * H + 4 * h + pdh) * TW + 4 * ww + pdw;
    feat[tok * 24 + 16 + ch * 4 + p1 * 2 + p2] = from_f32<TO>(v);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm, any E
constexpr int LN_MAXV = 3;      // E <= 768: up to 3 float4 per lane
template <typename T>
__global__ __launch_bounds__(256) void ln_generic_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const T* __restrict__ res,
                                                             const float* __restrict__ sscale, int rows_per_sample,
                                                             T* __restrict__ y, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int M, int E, float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = E / 4;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    f32x4 xv[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
        xv[i] = load4<T>(x + (size_t)row * E + v * 4);
        s += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
      }
    }
    const float mu = wave_sum(s) / E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + 64 * i < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xv[i][e] - mu;
          q += d * d;
        }
      }
    const float rs = rsqrtf(wave_sum(q) / E + eps);
    const float sc = sscale ? sscale[row / rows_per_sample] : 1.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + v * 4), bt = *reinterpret_cast<const f32x4*>(beta + v * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((xv[i][e] - mu) * rs * g[e] + bt[e]) * sc;
        if (res) {
          const f32x4 rv = load4<T>(res + (size_t)row * E + v * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
        store4<T>(y + (size_t)row * E + v * 4, o);
      }
    }
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

// dx = LN'(s_b * dy);  partial dgamma / dbeta per workgroup: part[blk][2][E]
template <typename T>
__global__ __launch_bounds__(256) void ln_generic_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ sscale, int rows_per_sample,
                                                             T* __restrict__ dx, float* __restrict__ part, int M, int E) {
  __shared__ float red[4][2 * 768];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nv = E / 4;
  f32x4 dg[LN_MAXV], db[LN_MAXV], gm[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (lane + 64 * i < nv) gm[i] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * i) * 4);
  }
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    const float sc = sscale ? sscale[row / rows_per_sample] : 1.f;
    f32x4 xh[LN_MAXV], gv[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
        const f32x4 xv = load4<T>(x + (size_t)row * E + v * 4);
        const f32x4 dv = load4<T>(dy + (size_t)row * E + v * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = dv[e] * sc;
          xh[i][e] = (xv[e] - mu) * rs;
          gv[i][e] = d * gm[i][e];
          s1 += gv[i][e];
          s2 += gv[i][e] * xh[i][e];
          dg[i][e] += d * xh[i][e];
          db[i][e] += d;
        }
      }
    }
    const float c1 = wave_sum(s1) / E, c2 = wave_sum(s2) / E;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gv[i][e] - c1 - xh[i][e] * c2);
        store4<T>(dx + (size_t)row * E + v * 4, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[w][v * 4 + e] = dg[i][e];
        red[w][E + v * 4 + e] = db[i][e];
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * E; e += 256)
    part[(size_t)blockIdx.x * 2 * E + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// ------------------------------------------------------------------------------------------------ LayerNorm, Swin widths
// The four stage widths of SwinV2-T (models/swinv2.py: embed_dim 96 doubling per stage: 96 / 192 / 384 / 768) with the row split
// over LPR lanes (8 / 16 / 32 / 64) so that every lane of a wave carries data -- the any-width kernel above puts one wave on a
// row, 24 of 64 lanes busy at width 96 and one 192-byte request in flight per wave, and took 15 % of the SwinV2-T step.
// E = NV * LPR * 4; element index e = (v * LPR + l) * 4 + i; G = 256 / LPR rows per workgroup pass.  Same formulas as above.
template <int LPR>
__device__ __forceinline__ float lpr_sum(float v) {
  if constexpr (LPR == 8) return group8_sum(v);
  else if constexpr (LPR == 16) return group16_sum(v);
  else if constexpr (LPR == 32) return group16_sum(v + __shfl_xor(v, 16, 64));
  else return wave_sum(v);
}

template <typename T, int NV, int LPR>
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const T* __restrict__ res,
                                                          const float* __restrict__ sscale, int rows_per_sample,
                                                          T* __restrict__ y, float* __restrict__ mean,
                                                          float* __restrict__ rstd, int M, float eps) {
  constexpr int E = NV * LPR * 4, G = 256 / LPR;
  const int l = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  f32x4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + (v * LPR + l) * 4);
    bt[v] = *reinterpret_cast<const f32x4*>(beta + (v * LPR + l) * 4);
  }
  for (int row = blockIdx.x * G + grp; row < M; row += gridDim.x * G) {
    f32x4 xv[NV], rv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) xv[v] = load4<T>(x + (size_t)row * E + (v * LPR + l) * 4);
    if (res) {
#pragma unroll
      for (int v = 0; v < NV; ++v) rv[v] = load4<T>(res + (size_t)row * E + (v * LPR + l) * 4);
    }
    const float sc = sscale ? sscale[row / rows_per_sample] : 1.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) s += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
    const float mu = lpr_sum<LPR>(s) / E;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = xv[v][i] - mu;
        q += d * d;
      }
    const float rs = rsqrtf(lpr_sum<LPR>(q) / E + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = ((xv[v][i] - mu) * rs * gm[v][i] + bt[v][i]) * sc;
      if (res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += rv[v][i];
      }
      store4<T>(y + (size_t)row * E + (v * LPR + l) * 4, o);
    }
    if (l == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

template <typename T, int NV, int LPR>
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ sscale,
                                                          int rows_per_sample, T* __restrict__ dx, float* __restrict__ part,
                                                          int M) {
  constexpr int E = NV * LPR * 4, G = 256 / LPR;
  __shared__ float red[G][E + 4];
  const int l = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  f32x4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + (v * LPR + l) * 4);
    dg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * G + grp; row < M; row += gridDim.x * G) {
    f32x4 xh[NV], gv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xh[v] = load4<T>(x + (size_t)row * E + (v * LPR + l) * 4);
      gv[v] = load4<T>(dy + (size_t)row * E + (v * LPR + l) * 4);
    }
    const float mu = mean[row], rs = rstd[row];
    const float sc = sscale ? sscale[row / rows_per_sample] : 1.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = gv[v][i] * sc;
        xh[v][i] = (xh[v][i] - mu) * rs;
        gv[v][i] = d * gm[v][i];
        s1 += gv[v][i];
        s2 += gv[v][i] * xh[v][i];
        dg[v][i] += d * xh[v][i];
        db[v][i] += d;
      }
    const float c1 = lpr_sum<LPR>(s1) / E, c2 = lpr_sum<LPR>(s2) / E;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = rs * (gv[v][i] - c1 - xh[v][i] * c2);
      store4<T>(dx + (size_t)row * E + (v * LPR + l) * 4, o);
    }
  }
  // column sums over the G row groups of the workgroup, fixed order: part[blk][2][E]
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[grp][(v * LPR + l) * 4 + i] = pass == 0 ? dg[v][i] : db[v][i];
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < G; ++r) a += red[r][e];
      part[((size_t)blockIdx.x * 2 + pass) * E + e] = a;
    }
  }
}

constexpr int LN_ROWS_BWD_BLOCKS = 2048;
static inline int ln_rows_lpr(int E) { return E == 96 ? 8 : E == 192 ? 16 : E == 384 ? 32 : E == 768 ? 64 : 0; }   // three vectors per lane at every width
// workgroups of the backward: every workgroup walks the same number of row passes (a ragged last round would idle most of the chip)
static inline int ln_rows_bwd_grid(int M, int E) {
  const int G = 256 / ln_rows_lpr(E), P = (M + G - 1) / G, it = (P + LN_ROWS_BWD_BLOCKS - 1) / LN_ROWS_BWD_BLOCKS;
  return (P + it - 1) / it;
}

template <typename T>
bool ln_rows_fwd(const void* x, const float* g, const float* b, const void* res, const float* ss, int rps, void* y, float* mean,
                 float* rstd, int M, int E, float eps, hipStream_t st) {
  const int lpr = ln_rows_lpr(E);
  if (!lpr) return false;
  const int G = 256 / lpr, grid = min((M + G - 1) / G, 8192);
#define LNF(NV, LPR) hipLaunchKernelGGL((ln_rows_fwd_kernel<T, NV, LPR>), dim3(grid), dim3(256), 0, st, (const T*)x, g, b, (const T*)res, ss, rps, (T*)y, mean, rstd, M, eps)
  if (E == 96) LNF(3, 8);
  else if (E == 192) LNF(3, 16);
  else if (E == 384) LNF(3, 32);
  else LNF(3, 64);
#undef LNF
  return true;
}

template <typename T>
int ln_rows_bwd(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, const float* ss, int rps,
                void* dx, float* part, int M, int E, hipStream_t st) {
  const int grid = ln_rows_bwd_grid(M, E);
#define LNB(NV, LPR) hipLaunchKernelGGL((ln_rows_bwd_kernel<T, NV, LPR>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, g, mean, rstd, ss, rps, (T*)dx, part, M)
  if (E == 96) LNB(3, 8);
  else if (E == 192) LNB(3, 16);
  else if (E == 384) LNB(3, 32);
  else LNB(3, 64);
#undef LNB
  return grid;
}

// ------------------------------------------------------------------------------------------------ merge / mean
// fwd: out[b, (y/2)(res/2) + x/2, (dy + 2 dx) C + c] = in[b, y res + x, c];  bwd: the inverse copy
template <typename T>
__global__ void merge_gather_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int res, int C, int inverse) {
  const long long n4 = (long long)B * res * res * C / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int c = (int)(e % C);
    const long long t = e / C;
    const int x = (int)(t % res), yy = (int)((t / res) % res), b = (int)(t / ((long long)res * res));
    const long long o = (((long long)b * (res / 2) + yy / 2) * (res / 2) + x / 2) * 4 * C + ((yy & 1) + 2 * (x & 1)) * C + c;
    typedef typename Vec4<T>::type V;
    if (inverse) *reinterpret_cast<V*>(out + e) = *reinterpret_cast<const V*>(in + o);
    else *reinterpret_cast<V*>(out + o) = *reinterpret_cast<const V*>(in + e);
  }
}

// y[b, c] = mean_n x[b, n, c];  dx[b, n, c] = dy[b, c] / N.  One workgroup per image, 16-byte column vectors, the rows split over
// 256 / (C / EPV) thread groups and summed in a fixed order through LDS (the first version walked the rows with one 2-byte load per
// thread: 0.5 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int C) {
  constexpr int EPV = 16 / (int)sizeof(T);
  __shared__ float red[256 * EPV];
  const int b = blockIdx.x, nvec = C / EPV;
  const int groups = nvec >= 256 ? 1 : 256 / nvec;
  const int grp = threadIdx.x / nvec;
  for (int v0 = 0; v0 < nvec; v0 += 256) {            // one pass unless C / EPV > 256
    const int v = v0 + (nvec >= 256 ? threadIdx.x : threadIdx.x % nvec);
    float a[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) a[e] = 0.f;
    if (v < nvec && grp < groups) {
      for (int n = grp; n < N; n += groups) {
        const T* p = x + ((size_t)b * N + n) * C + v * EPV;
        if constexpr (sizeof(T) == 2) {
          const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += (float)t[e];
        } else {
          const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] += t[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) red[threadIdx.x * EPV + e] = a[e];
    __syncthreads();
    if (grp == 0 && v < nvec) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float t = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) t += red[((nvec >= 256 ? 0 : g2 * nvec) + (nvec >= 256 ? threadIdx.x : threadIdx.x % nvec)) * EPV + e];
        y[(size_t)b * C + v * EPV + e] = from_f32<T>(t / N);
      }
    }
    __syncthreads();
  }
}
template <typename T>
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int C) {
  constexpr int EPV = 16 / (int)sizeof(T);
  const int b = blockIdx.x, nvec = C / EPV;
  const float inv = 1.f / N;
  for (int i = threadIdx.x; i < N * nvec; i += 256) {
    const int n = i / nvec, v = i % nvec;
    T o[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) o[e] = from_f32<T>(to_f32(dy[(size_t)b * C + v * EPV + e]) * inv);
    *reinterpret_cast<u32x4s*>(dx + ((size_t)b * N + n) * C + v * EPV) = *reinterpret_cast<const u32x4s*>(o);
  }
}
// any C (not a multiple of the 16-byte vector): scalar walk
template <typename T>
__global__ void token_mean_fwd_scalar_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += to_f32(x[((size_t)b * N + n) * C + c]);
    y[(size_t)b * C + c] = from_f32<T>(a / N);
  }
}
template <typename T>
__global__ void token_mean_bwd_scalar_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int C) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < N * C; i += blockDim.x)
    dx[(size_t)b * N * C + i] = from_f32<T>(to_f32(dy[(size_t)b * C + i % C]) / N);
}


// ------------------------------------------------------------------------------------------------ continuous position bias
// WindowAttention's parameter-only part (models/swinv2.py:158-168) for EVERY block of the model in two launches per direction:
//   table[h][e] = cpb_mlp(relative_coords_table[e])[h]        (Linear(2, 512) -> ReLU -> Linear(512, heads, no bias); e < 225)
//   bias[h][i][k] = 16 sigmoid(table[h][relative_position_index[i][k]])     scale[h] = exp(min(logit_scale[h], ln 100))
// The reference runs this as ~20 tiny torch kernels forward and as many backward per block; batched per stage it still was 150
// launches and 1.3 ms of a SwinV2-T step.  Sums run in a fixed order (no atomics): run-to-run identical bits.
constexpr int CPB_E = 225, CPB_HID = 512, CPB_POS = 4096, CPB_MAXH = 24, CPB_MAXB = 16, CPB_EPS = 8, CPB_SPLIT = (CPB_E + CPB_EPS - 1) / CPB_EPS;
struct CpbArgs { rgbnm_cpb_block blk[CPB_MAXB]; const float* coords; const int* index; const int* inv; float* table; float* dtable; };

// grid (nblocks, CPB_SPLIT): CPB_EPS table entries per workgroup (29 x 12 workgroups for SwinV2-T); their hidden activations in LDS
__global__ __launch_bounds__(256) void cpb_table_kernel(CpbArgs a) {
  __shared__ float hid[CPB_EPS][CPB_HID];
  const rgbnm_cpb_block& b = a.blk[blockIdx.x];
  const int e0 = blockIdx.y * CPB_EPS, ne = min(CPB_EPS, CPB_E - e0), tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (ne <= 0) return;
  for (int i = tid; i < ne * CPB_HID; i += 256) {
    const int e = i / CPB_HID, j = i % CPB_HID;
    const float v = fmaf(b.w1[2 * j + 1], a.coords[2 * (e0 + e) + 1], fmaf(b.w1[2 * j], a.coords[2 * (e0 + e)], b.b1[j]));
    hid[e][j] = v > 0.f ? v : 0.f;
  }
  __syncthreads();
  float* t = a.table + (size_t)blockIdx.x * CPB_MAXH * CPB_E;
  for (int pr = w; pr < b.heads * ne; pr += 4) {           // one (head, entry) dot product of length 512 per wave and turn
    const int h = pr / ne, e = pr % ne;
    const float* w2 = b.w2 + (size_t)h * CPB_HID;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < CPB_HID / 64; ++k) acc = fmaf(w2[lane + 64 * k], hid[e][lane + 64 * k], acc);
    acc = wave_sum(acc);
    if (lane == 0) t[h * CPB_E + e0 + e] = acc;
  }
}

// grid (nblocks, CPB_MAXH): bias of one head (4096 positions) + the head's logit scale
__global__ __launch_bounds__(256) void cpb_bias_kernel(CpbArgs a) {
  const rgbnm_cpb_block& b = a.blk[blockIdx.x];
  const int h = blockIdx.y;
  if (h >= b.heads) return;
  const float* t = a.table + ((size_t)blockIdx.x * CPB_MAXH + h) * CPB_E;
  for (int p = threadIdx.x; p < CPB_POS; p += 256)
    b.bias[(size_t)h * CPB_POS + p] = 16.f / (1.f + expf(-t[a.index[p]]));
  if (threadIdx.x == 0) b.scale[h] = expf(fminf(b.ls[h], 4.605170185988092f));       // ln(1 / 0.01)
}

// grid (nblocks, CPB_MAXH): d table[h][e] = 16 s (1 - s) * sum of d bias over the <= 64 positions that read entry e (inv: [225][64],
// padded with 4096), ascending position order; d logit_scale
__global__ __launch_bounds__(256) void cpb_dtable_kernel(CpbArgs a) {
  const rgbnm_cpb_block& b = a.blk[blockIdx.x];
  const int h = blockIdx.y, e = threadIdx.x;
  if (h >= b.heads) return;
  if (e < CPB_E) {
    const float* db = b.dbias + (size_t)h * CPB_POS;
    int pos[64];
    float val[64];
#pragma unroll
    for (int q = 0; q < 16; ++q) {                       // the entry's position list first, then 64 INDEPENDENT gathers, then the sum in order
      const int4 v = *reinterpret_cast<const int4*>(a.inv + e * 64 + 4 * q);
      pos[4 * q] = v.x; pos[4 * q + 1] = v.y; pos[4 * q + 2] = v.z; pos[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 64; ++q) val[q] = pos[q] < CPB_POS ? db[pos[q]] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) acc += val[q];
    const float s = 1.f / (1.f + expf(-a.table[((size_t)blockIdx.x * CPB_MAXH + h) * CPB_E + e]));
    a.dtable[((size_t)blockIdx.x * CPB_MAXH + h) * CPB_E + e] = acc * 16.f * s * (1.f - s);
  }
  if (e == 255) b.dls[h] = b.ls[h] <= 4.605170185988092f ? b.dscale[h] * expf(b.ls[h]) : 0.f;    // clamp passes its gradient inside the range
}

// grid (nblocks, 16): 32 hidden units per workgroup, EIGHT lanes per unit, each walking an eighth of the 225 entries; the eight partial
// sums are combined by xor-shuffles (fixed tree): d W2[h][j], d W1[j][0..1], d b1[j]
__global__ __launch_bounds__(256) void cpb_mlp_bwd_kernel(CpbArgs a) {
  __shared__ float dts[CPB_MAXH][CPB_E + 3];
  __shared__ float cs[CPB_E][2];
  const rgbnm_cpb_block& b = a.blk[blockIdx.x];
  const int H = b.heads, j = blockIdx.y * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
  for (int i = threadIdx.x; i < H * CPB_E; i += 256) dts[i / CPB_E][i % CPB_E] = a.dtable[(size_t)blockIdx.x * CPB_MAXH * CPB_E + i];
  for (int i = threadIdx.x; i < CPB_E * 2; i += 256) cs[i >> 1][i & 1] = a.coords[i];
  __syncthreads();
  const float wa = b.w1[2 * j], wb = b.w1[2 * j + 1], bj = b.b1[j];
  float w2[CPB_MAXH], dw2[CPB_MAXH];
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h) { w2[h] = h < H ? b.w2[(size_t)h * CPB_HID + j] : 0.f; dw2[h] = 0.f; }
  float da = 0.f, dbb = 0.f, dbias = 0.f;
  constexpr int PER = (CPB_E + 7) / 8;                  // 29
  const int e1 = min(CPB_E, (part + 1) * PER);
  for (int e = part * PER; e < e1; ++e) {
    const float v = fmaf(wb, cs[e][1], fmaf(wa, cs[e][0], bj));
    const float hv = v > 0.f ? v : 0.f;
    float dh = 0.f;
#pragma unroll
    for (int h = 0; h < CPB_MAXH; ++h) {
      const float d = h < H ? dts[h][e] : 0.f;
      dw2[h] = fmaf(d, hv, dw2[h]);
      dh = fmaf(d, w2[h], dh);
    }
    if (v > 0.f) { da = fmaf(dh, cs[e][0], da); dbb = fmaf(dh, cs[e][1], dbb); dbias += dh; }
  }
  auto red8 = [](float x) { x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); return x; };
#pragma unroll
  for (int h = 0; h < CPB_MAXH; ++h) {
    const float r = red8(dw2[h]);
    if (h < H && part == 0) b.dw2[(size_t)h * CPB_HID + j] = r;
  }
  da = red8(da); dbb = red8(dbb); dbias = red8(dbias);
  if (part == 0) {
    b.dw1[2 * j] = da;
    b.dw1[2 * j + 1] = dbb;
    b.db1[j] = dbias;
  }
}

}  // namespace

extern "C" {

int rgbnm_swin_embed(int in_dtype, int out_dtype, const void* y, const void* cbcr, const float* convY, const float* convC,
                     void* feat, int B, int Hb, int Wb, void* stream) {
  if (!y || !cbcr || !convY || !convC || !feat || B <= 0 || (Hb & 1) || (Wb & 1)) return RGBNM_EINVAL;
  const long long nblk = (long long)B * Hb * Wb + (long long)B * 2 * (Hb / 2) * (Wb / 2);
  const int grid = (int)((nblk + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define EMB(TI, TO) hipLaunchKernelGGL((swin_embed_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)y, (const TI*)cbcr, convY, convC, (TO*)feat, B, Hb, Wb)
  if (in_dtype == DT_F32 && out_dtype == DT_F32) EMB(float, float);
  else if (in_dtype == DT_F32 && out_dtype == DT_BF16) EMB(float, bf16);
  else if (in_dtype == DT_BF16 && out_dtype == DT_BF16) EMB(bf16, bf16);
  else if (in_dtype == DT_BF16 && out_dtype == DT_F32) EMB(bf16, float);
  else return RGBNM_EINVAL;
#undef EMB
  LAUNCH_CHECK();
  return RGBNM_OK;
}

size_t rgbnm_ln_generic_bwd_workspace(int M, int E) {
  int blocks = (M + 3) / 4;
  if (blocks > LN_ROWS_BWD_BLOCKS) blocks = LN_ROWS_BWD_BLOCKS;   // covers the any-width kernel (<= 1024) and the stage-width one
  return (size_t)blocks * 2 * E * sizeof(float);
}

int rgbnm_ln_generic_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* res,
                         const float* sample_scale, int rows_per_sample, void* y, float* mean, float* rstd, int M, int E,
                         float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || E % 4 || E > 768 || (sample_scale && rows_per_sample <= 0))
    return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (rgbnm_get_option("ln_rows") && ln_rows_lpr(E)) {
    if (dtype == DT_BF16) ln_rows_fwd<bf16>(x, gamma, beta, res, sample_scale, rows_per_sample, y, mean, rstd, M, E, eps, st);
    else if (dtype == DT_F32) ln_rows_fwd<float>(x, gamma, beta, res, sample_scale, rows_per_sample, y, mean, rstd, M, E, eps, st);
    else return RGBNM_EINVAL;
    LAUNCH_CHECK();
    return RGBNM_OK;
  }
  int grid = (M + 3) / 4;
  if (grid > 4096) grid = 4096;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(ln_generic_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, gamma, beta, (const bf16*)res, sample_scale, rows_per_sample, (bf16*)y, mean, rstd, M, E, eps);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(ln_generic_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, (const float*)res, sample_scale, rows_per_sample, (float*)y, mean, rstd, M, E, eps);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_ln_generic_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                         const float* rstd, const float* sample_scale, int rows_per_sample, void* dx, float* dgamma,
                         float* dbeta, int M, int E, int accumulate, void* workspace, size_t workspace_bytes,
                         void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || M <= 0 || E % 4 || E > 768)
    return RGBNM_EINVAL;
  if (workspace_bytes < rgbnm_ln_generic_bwd_workspace(M, E)) return RGBNM_EWORKSPACE;
  int grid = (M + 3) / 4;
  if (grid > 1024) grid = 1024;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (rgbnm_get_option("ln_rows") && ln_rows_lpr(E)) {
    if (dtype == DT_BF16) grid = ln_rows_bwd<bf16>(dy, x, gamma, mean, rstd, sample_scale, rows_per_sample, dx, part, M, E, st);
    else if (dtype == DT_F32) grid = ln_rows_bwd<float>(dy, x, gamma, mean, rstd, sample_scale, rows_per_sample, dx, part, M, E, st);
    else return RGBNM_EINVAL;
  } else if (dtype == DT_BF16)
    hipLaunchKernelGGL(ln_generic_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, sample_scale, rows_per_sample, (bf16*)dx, part, M, E);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(ln_generic_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)x, gamma, mean, rstd, sample_scale, rows_per_sample, (float*)dx, part, M, E);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  RgbnmReduceJob j;
  j.part = part; j.stride = 2LL * E; j.out = dgamma; j.n = E; j.S = grid; j.cols = 1; j.perm_heads = 0;
  j.accumulate = accumulate; j.epw = 8;
  const bool own = !rgbnm_reduce_defer_active();        // dgamma and dbeta in ONE reduction launch
  if (own) rgbnm_reduce_defer_begin();
  int rc = rgbnm_reduce_submit(j, st);
  if (rc == RGBNM_OK) {
    j.part = part + E; j.out = dbeta;
    rc = rgbnm_reduce_submit(j, st);
  }
  if (own) {
    const int rf = rgbnm_reduce_defer_flush(st);
    if (rc == RGBNM_OK) rc = rf;
  }
  return rc;
}

int rgbnm_merge_gather(int dtype, const void* in, void* out, int B, int res, int C, int inverse, void* stream) {
  if (!in || !out || B <= 0 || (res & 1) || C % 4) return RGBNM_EINVAL;
  const long long n4 = (long long)B * res * res * C / 4;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) hipLaunchKernelGGL(merge_gather_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)in, (bf16*)out, B, res, C, inverse);
  else if (dtype == DT_F32) hipLaunchKernelGGL(merge_gather_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)in, (float*)out, B, res, C, inverse);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_token_mean(int dtype, const void* in, void* out, int B, int N, int C, int backward, void* stream) {
  if (!in || !out || B <= 0 || N <= 0 || C <= 0) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define TM(T) do { const bool vec = C % (16 / (int)sizeof(T)) == 0;                                                              \
    if (backward) { if (vec) hipLaunchKernelGGL(token_mean_bwd_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)in, (T*)out, N, C);  \
                    else hipLaunchKernelGGL(token_mean_bwd_scalar_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)in, (T*)out, N, C); } \
    else { if (vec) hipLaunchKernelGGL(token_mean_fwd_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)in, (T*)out, N, C);           \
           else hipLaunchKernelGGL(token_mean_fwd_scalar_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)in, (T*)out, N, C); } } while (0)
  if (dtype == DT_BF16) TM(bf16);
  else if (dtype == DT_F32) TM(float);
  else return RGBNM_EINVAL;
#undef TM
  LAUNCH_CHECK();
  return RGBNM_OK;
}


size_t rgbnm_swin_cpb_table_elems(int nblocks) { return (size_t)(nblocks > 0 ? nblocks : 0) * CPB_MAXH * CPB_E; }

static int cpb_fill(CpbArgs& a, const rgbnm_cpb_block* blocks, int n, const float* coords, const int* index, const int* inv,
                    float* table, float* dtable, bool bwd) {
  for (int i = 0; i < n; ++i) {
    const rgbnm_cpb_block& b = blocks[i];
    if (b.heads < 1 || b.heads > CPB_MAXH || !b.w1 || !b.b1 || !b.w2 || !b.ls) return RGBNM_EINVAL;
    if (!bwd && (!b.bias || !b.scale)) return RGBNM_EINVAL;
    if (bwd && (!b.dbias || !b.dscale || !b.dw1 || !b.db1 || !b.dw2 || !b.dls)) return RGBNM_EINVAL;
    a.blk[i] = b;
  }
  for (int i = n; i < CPB_MAXB; ++i) a.blk[i] = blocks[0];
  a.coords = coords; a.index = index; a.inv = inv; a.table = table; a.dtable = dtable;
  return RGBNM_OK;
}

int rgbnm_swin_cpb_fwd(const rgbnm_cpb_block* blocks, int nblocks, const float* coords, const int* index, float* table, void* stream) {
  if (!blocks || nblocks < 1 || !coords || !index || !table) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int b0 = 0; b0 < nblocks; b0 += CPB_MAXB) {
    const int n = nblocks - b0 < CPB_MAXB ? nblocks - b0 : CPB_MAXB;
    CpbArgs a;
    TRYRC(cpb_fill(a, blocks + b0, n, coords, index, nullptr, table + (size_t)b0 * CPB_MAXH * CPB_E, nullptr, false));
    hipLaunchKernelGGL(cpb_table_kernel, dim3(n, CPB_SPLIT), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(cpb_bias_kernel, dim3(n, CPB_MAXH), dim3(256), 0, st, a);
    LAUNCH_CHECK();
  }
  return RGBNM_OK;
}

int rgbnm_swin_cpb_bwd(const rgbnm_cpb_block* blocks, int nblocks, const float* coords, const int* inv_index, const float* table,
                       float* dtable, void* stream) {
  if (!blocks || nblocks < 1 || !coords || !inv_index || !table || !dtable) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int b0 = 0; b0 < nblocks; b0 += CPB_MAXB) {
    const int n = nblocks - b0 < CPB_MAXB ? nblocks - b0 : CPB_MAXB;
    CpbArgs a;
    TRYRC(cpb_fill(a, blocks + b0, n, coords, nullptr, inv_index, const_cast<float*>(table) + (size_t)b0 * CPB_MAXH * CPB_E,
                   dtable + (size_t)b0 * CPB_MAXH * CPB_E, true));
    hipLaunchKernelGGL(cpb_dtable_kernel, dim3(n, CPB_MAXH), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(cpb_mlp_bwd_kernel, dim3(n, CPB_HID / 32), dim3(256), 0, st, a);
    LAUNCH_CHECK();
  }
  return RGBNM_OK;
}

}  // extern "C"
