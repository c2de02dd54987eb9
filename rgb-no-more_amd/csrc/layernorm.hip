// LayerNorm forward / backward over the embedding axis (E = 192 or 384; eps 1e-5) for the pre-LN
// residual blocks (reference: nn.LayerNorm in models/plainvit.py:513,522,551) plus the fused head pooling
// (LN -> token mean, plainvit.py:551-552).  HBM-bound: 16 lanes own one row, 8/16-byte vector accesses,
// statistics in fp32 with a two-pass (mean, then centred variance) reduction held in registers.
#include "common.h"
#include "../../include/rgbnm.h"
#include "internal.h"

namespace {


// E = NV * LPR * 4 : each of the LPR lanes of a row group holds NV vectors of 4 consecutive elements,
// element index e = (v*LPR + l16)*4 + i.  LPR = 16 (E = 192 / 384: JPEG-Ti / JPEG-S) or 64 (one wave per row; E = 512 / 768 /
// 1024: utils/configs.py:104-122 vitb / vitl) -- the wide rows would not fit one lane's registers at 16 lanes per row.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
  if constexpr (LPR == 16) return group16_sum(v);
  else if constexpr (LPR == 32) return group16_sum(v + __shfl_xor(v, 16, 64));
  else return wave_sum(v);
}

template <typename T, int NV, int LPR = 16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                     float eps) {
  constexpr int E = NV * LPR * 4, G = 256 / LPR;   // LPR lanes own one row, G rows per workgroup pass
  const int l16 = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  f32x4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + (v * LPR + l16) * 4);
    bt[v] = *reinterpret_cast<const f32x4*>(beta + (v * LPR + l16) * 4);
  }
  for (int row = blockIdx.x * G + grp; row < M; row += gridDim.x * G) {
    // every fused multiply-add is spelled out and contraction is off: the residual + LayerNorm epilogue of the row-panel
    // GEMM (gemm_nt_kpipe.hip, EPI_RES_LN) repeats this arithmetic in another lane layout and must produce the same bits
#pragma clang fp contract(off)
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = load4<T>(x + (size_t)row * E + (v * LPR + l16) * 4);
      s += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
    }
    const float mu = row_sum<LPR>(s) * (1.f / E);
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = xv[v][i] - mu;
        q = __builtin_fmaf(d, d, q);
      }
    const float rs = rsqrtf(__builtin_fmaf(row_sum<LPR>(q), 1.f / E, eps));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = __builtin_fmaf((xv[v][i] - mu) * rs, gm[v][i], bt[v][i]);
      store4<T>(y + (size_t)row * E + (v * LPR + l16) * 4, o);
    }
    if (l16 == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

// dx = [dres +] rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  partial dgamma/dbeta per block.
template <typename T, int NV, int LPR = 16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ part, int M) {
  constexpr int E = NV * LPR * 4, G = 256 / LPR;   // LPR lanes own one row, G rows per workgroup pass
  __shared__ float red[G][E + 4];
  const int l16 = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  f32x4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + (v * LPR + l16) * 4);
    dg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * G + grp; row < M; row += gridDim.x * G) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NV], gv[NV], rv[NV];
    float s1 = 0.f, s2 = 0.f;
    // the residual gradient is requested with the other two operands: asked for after the row sums it was a second, fully
    // exposed round trip per row (44 -> 30 us at M = 50176, E = 384)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      rv[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (dres) rv[v] = load4<T>(dres + (size_t)row * E + (v * LPR + l16) * 4);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const size_t off = (size_t)row * E + (v * LPR + l16) * 4;
      const f32x4 xv = load4<T>(x + off);
      const f32x4 dv = load4<T>(dy + off);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (E == 192) {      // rounds like the fused epilogues that repeat this arithmetic at E = 192 (ln_bwd_rows.h)
          float xh_, gv_, dg_ = dg[v][i], db_ = db[v][i];
          ln_bwd_acc(dv[i], xv[i], mu, rs, gm[v][i], xh_, gv_, s1, s2, dg_, db_);
          xh[v][i] = xh_; gv[v][i] = gv_; dg[v][i] = dg_; db[v][i] = db_;
        } else {                       // no fused twin at other widths: leave the compiler free to schedule across rows
          xh[v][i] = (xv[i] - mu) * rs;
          gv[v][i] = dv[i] * gm[v][i];
          s1 += gv[v][i];
          s2 += gv[v][i] * xh[v][i];
          dg[v][i] += dv[i] * xh[v][i];
          db[v][i] += dv[i];
        }
      }
    }
    float c1 = row_sum<LPR>(s1) * (1.f / E), c2 = row_sum<LPR>(s2) * (1.f / E);
    asm volatile("" : "+v"(c1), "+v"(c2));       // as ln_bwd_rows.h: products, never contracted into ln_bwd_dx's subtraction
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const size_t off = (size_t)row * E + (v * LPR + l16) * 4;
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (E == 192) o[i] = ln_bwd_dx(rs, gv[v][i], c1, xh[v][i], c2, rv[v][i]);
        else o[i] = rs * (gv[v][i] - c1 - xh[v][i] * c2) + rv[v][i];
      }
      store4<T>(dx + off, o);
    }
  }
  // block-level column reduction of dgamma / dbeta (fixed order => deterministic)
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[grp][(v * LPR + l16) * 4 + i] = pass == 0 ? dg[v][i] : db[v][i];
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < G; ++r) a += red[r][e];
      part[((size_t)blockIdx.x * 2 + pass) * E + e] = a;
    }
  }
}

// partial blocks [nblk][2][E] (gamma | beta) -> two jobs of the batched reduction (reduce.hip)
int submit_ln_reduce(const float* part, float* dgamma, float* dbeta, int nblk, int E, int accumulate, hipStream_t st) {
  RgbnmReduceJob j;
  j.part = part; j.stride = 2LL * E; j.out = dgamma; j.n = E; j.S = nblk; j.cols = 1; j.perm_heads = 0;
  j.accumulate = accumulate; j.epw = 8;
  const bool own = !rgbnm_reduce_defer_active();        // stand-alone call: gamma and beta in ONE reduction launch
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

// ---- head pooling: pooled[b] = mean_t LN(x[b,t,:])  (one workgroup per image) ------------------------
template <typename T, int NV, int LPR = 16>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ pooled,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int N,
                                                       float eps) {
  constexpr int E = NV * LPR * 4, G = 256 / LPR;   // LPR lanes own one row, G rows per workgroup pass
  __shared__ float red[G][E + 4];
  const int b = blockIdx.x;
  const int l16 = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  f32x4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // eight rows per turn, their loads issued together: one row per turn was a chain of 13 dependent memory latencies (12.6 us for
  // 75 KB per workgroup); the rows are still folded into acc in ascending order: the same bits
  constexpr int UNR = 8;
  for (int t0 = grp; t0 < N; t0 += UNR * G) {
    f32x4 xa[UNR][NV];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int tt = min(t0 + u * G, N - 1);
#pragma unroll
      for (int v = 0; v < NV; ++v) xa[u][v] = load4<T>(x + ((size_t)b * N + tt) * E + (v * LPR + l16) * 4);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
    const int t = t0 + u * G;
    if (t >= N) break;
    const size_t row = (size_t)b * N + t;
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = xa[u][v];
      s += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
    }
    const float mu = row_sum<LPR>(s) * (1.f / E);
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = xv[v][i] - mu;
        q += d * d;
      }
    const float rs = rsqrtf(row_sum<LPR>(q) * (1.f / E) + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[v][i] += (xv[v][i] - mu) * rs;
    if (l16 == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[grp][(v * LPR + l16) * 4 + i] = acc[v][i];
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < G; ++r) a += red[r][e];
    // mean_t (xhat*gamma + beta) = gamma * mean_t(xhat) + beta
    pooled[(size_t)b * E + e] = from_f32<T>(a * (1.f / N) * gamma[e] + beta[e]);
  }
}

// dx[b,t,:] = LN-backward of dy = dpooled[b,:]/N (same for every token); partial dgamma/dbeta per image.
template <typename T, int NV, int LPR = 16>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ dpooled, const T* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, T* __restrict__ dx,
                                                       float* __restrict__ part, int N) {
  constexpr int E = NV * LPR * 4, G = 256 / LPR;   // LPR lanes own one row, G rows per workgroup pass
  __shared__ float red[G][E + 4];
  const int b = blockIdx.x;
  const int l16 = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  const float invN = 1.f / N;
  f32x4 gm[NV], dv[NV], dg[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + (v * LPR + l16) * 4);
    dv[v] = load4<T>(dpooled + (size_t)b * E + (v * LPR + l16) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) dv[v][i] *= invN;
    dg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  constexpr int UNR = 4;          // four rows per turn with their loads issued together (see pool_fwd_kernel; eight measured the same here); dg folds rows in order
  for (int t0 = grp; t0 < N; t0 += UNR * G) {
    f32x4 xa[UNR][NV];
    float mua[UNR], rsa[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int tt = min(t0 + u * G, N - 1);
      mua[u] = mean[(size_t)b * N + tt];
      rsa[u] = rstd[(size_t)b * N + tt];
#pragma unroll
      for (int v = 0; v < NV; ++v) xa[u][v] = load4<T>(x + ((size_t)b * N + tt) * E + (v * LPR + l16) * 4);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
    const int t = t0 + u * G;
    if (t >= N) break;
    const size_t row = (size_t)b * N + t;
    const float mu = mua[u], rs = rsa[u];
    f32x4 xh[NV], gv[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const f32x4 xv = xa[u][v];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[v][i] = (xv[i] - mu) * rs;
        gv[v][i] = dv[v][i] * gm[v][i];
        s1 += gv[v][i];
        s2 += gv[v][i] * xh[v][i];
        dg[v][i] += dv[v][i] * xh[v][i];
      }
    }
    const float c1 = row_sum<LPR>(s1) * (1.f / E), c2 = row_sum<LPR>(s2) * (1.f / E);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = rs * (gv[v][i] - c1 - xh[v][i] * c2);
      store4<T>(dx + row * E + (v * LPR + l16) * 4, o);
    }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[grp][(v * LPR + l16) * 4 + i] = dg[v][i];
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < G; ++r) a += red[r][e];
    part[((size_t)b * 2 + 0) * E + e] = a;
    // dbeta contribution of this image = sum_t dy = dpooled[b,e]
    part[((size_t)b * 2 + 1) * E + e] = to_f32(dpooled[(size_t)b * E + e]);
  }
}

constexpr int LN_BWD_BLOCKS = 2048;  // 8 workgroups per CU: the row loop has no prefetch, so the loads in flight come from the number of resident workgroups

template <typename T>
int ln_fwd_t(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int M, int E, float eps,
             hipStream_t st) {
  const int grid = min(cdiv(M, E <= 384 ? 16 : 4), 4096);
  if (E == 192) hipLaunchKernelGGL((ln_fwd_kernel<T, 3>), dim3(grid), dim3(256), 0, st, (const T*)x, g, b, (T*)y, mean, rstd, M, eps);
  else if (E == 384) hipLaunchKernelGGL((ln_fwd_kernel<T, 6>), dim3(grid), dim3(256), 0, st, (const T*)x, g, b, (T*)y, mean, rstd, M, eps);
  else if (E == 512) hipLaunchKernelGGL((ln_fwd_kernel<T, 2, 64>), dim3(grid), dim3(256), 0, st, (const T*)x, g, b, (T*)y, mean, rstd, M, eps);
  else if (E == 768) hipLaunchKernelGGL((ln_fwd_kernel<T, 3, 64>), dim3(grid), dim3(256), 0, st, (const T*)x, g, b, (T*)y, mean, rstd, M, eps);
  else if (E == 1024) hipLaunchKernelGGL((ln_fwd_kernel<T, 4, 64>), dim3(grid), dim3(256), 0, st, (const T*)x, g, b, (T*)y, mean, rstd, M, eps);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

template <typename T>
int ln_bwd_t(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, const void* dres,
             void* dx, float* dgamma, float* dbeta, int M, int E, int accumulate, float* ws, hipStream_t st) {
  // every workgroup walks the same number of row passes: 3136 passes on 2048 workgroups left a third of them a second pass
  // E = 384 runs 32 lanes per row: at 16 its six vectors per operand cost 162 VGPRs (3 waves per SIMD)
  const int P = cdiv(M, E == 192 ? 16 : E == 384 ? 8 : 4), grid = cdiv(P, cdiv(P, LN_BWD_BLOCKS));
  if (E == 192) hipLaunchKernelGGL((ln_bwd_kernel<T, 3>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx, ws, M);
  else if (E == 384) hipLaunchKernelGGL((ln_bwd_kernel<T, 3, 32>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx, ws, M);
  else if (E == 512) hipLaunchKernelGGL((ln_bwd_kernel<T, 2, 64>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx, ws, M);
  else if (E == 768) hipLaunchKernelGGL((ln_bwd_kernel<T, 3, 64>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx, ws, M);
  else if (E == 1024) hipLaunchKernelGGL((ln_bwd_kernel<T, 4, 64>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx, ws, M);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return submit_ln_reduce(ws, dgamma, dbeta, grid, E, accumulate, st);
}

}  // namespace

extern "C" {

size_t rgbnm_layernorm_bwd_workspace(int M, int E) {
  int blocks = cdiv(M, 16);
  if (blocks < LN_BWD_BLOCKS) blocks = LN_BWD_BLOCKS;   // also covers pool_bwd (one block per image) up to 512
  return (size_t)blocks * 2 * E * sizeof(float);
}

int rgbnm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int M, int E, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0) return RGBNM_EINVAL;
  if (dtype == DT_BF16) return ln_fwd_t<bf16>(x, gamma, beta, y, mean, rstd, M, E, eps, (hipStream_t)stream);
  if (dtype == DT_F32) return ln_fwd_t<float>(x, gamma, beta, y, mean, rstd, M, E, eps, (hipStream_t)stream);
  return RGBNM_EINVAL;
}

int rgbnm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, int M, int E,
                        int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || M <= 0) return RGBNM_EINVAL;
  if (workspace_bytes < (size_t)LN_BWD_BLOCKS * 2 * E * sizeof(float)) return RGBNM_EWORKSPACE;
  if (dtype == DT_BF16) return ln_bwd_t<bf16>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, M, E, accumulate, (float*)workspace, (hipStream_t)stream);
  if (dtype == DT_F32) return ln_bwd_t<float>(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, M, E, accumulate, (float*)workspace, (hipStream_t)stream);
  return RGBNM_EINVAL;
}

int rgbnm_head_pool_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* pooled, float* mean,
                        float* rstd, int B, int N, int E, float eps, void* stream) {
  if (!x || !gamma || !beta || !pooled || !mean || !rstd || B <= 0 || N <= 0) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define POOL(T, ...) hipLaunchKernelGGL((pool_fwd_kernel<T, __VA_ARGS__>), dim3(B), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)pooled, mean, rstd, N, eps)
  if (dtype == DT_BF16 && E == 192) POOL(bf16, 3);
  else if (dtype == DT_BF16 && E == 384) POOL(bf16, 6);
  else if (dtype == DT_F32 && E == 192) POOL(float, 3);
  else if (dtype == DT_F32 && E == 384) POOL(float, 6);
  else if (dtype == DT_BF16 && E == 512) POOL(bf16, 2, 64);
  else if (dtype == DT_BF16 && E == 768) POOL(bf16, 3, 64);
  else if (dtype == DT_BF16 && E == 1024) POOL(bf16, 4, 64);
  else if (dtype == DT_F32 && E == 512) POOL(float, 2, 64);
  else if (dtype == DT_F32 && E == 768) POOL(float, 3, 64);
  else if (dtype == DT_F32 && E == 1024) POOL(float, 4, 64);
  else return RGBNM_EINVAL;
#undef POOL
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_head_pool_bwd(int dtype, const void* dpooled, const void* x, const float* gamma, const float* mean,
                        const float* rstd, void* dx, float* dgamma, float* dbeta, int B, int N, int E, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!dpooled || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return RGBNM_EINVAL;
  if (workspace_bytes < (size_t)B * 2 * E * sizeof(float)) return RGBNM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
#define POOLB(T, ...) hipLaunchKernelGGL((pool_bwd_kernel<T, __VA_ARGS__>), dim3(B), dim3(256), 0, st, (const T*)dpooled, (const T*)x, gamma, mean, rstd, (T*)dx, ws, N)
  if (dtype == DT_BF16 && E == 192) POOLB(bf16, 3);
  else if (dtype == DT_BF16 && E == 384) POOLB(bf16, 6);
  else if (dtype == DT_F32 && E == 192) POOLB(float, 3);
  else if (dtype == DT_F32 && E == 384) POOLB(float, 6);
  else if (dtype == DT_BF16 && E == 512) POOLB(bf16, 2, 64);
  else if (dtype == DT_BF16 && E == 768) POOLB(bf16, 3, 64);
  else if (dtype == DT_BF16 && E == 1024) POOLB(bf16, 4, 64);
  else if (dtype == DT_F32 && E == 512) POOLB(float, 2, 64);
  else if (dtype == DT_F32 && E == 768) POOLB(float, 3, 64);
  else if (dtype == DT_F32 && E == 1024) POOLB(float, 4, 64);
  else return RGBNM_EINVAL;
#undef POOLB
  LAUNCH_CHECK();
  return submit_ln_reduce(ws, dgamma, dbeta, B, E, accumulate, (hipStream_t)st);
}

}  // extern "C"
