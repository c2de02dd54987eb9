// NT GEMM for FEW ROWS (M <= 512, bf16): the classification head of the ViT (models/plainvit.py:531-547 -- LayerNorm, mean
// pool, Linear + Tanh, Linear -- and its backward), whose GEMMs have M = batch = 256 rows:
//     h1 = tanh(pooled . W1^T + b1)        256 x 192,  K = 192        logits = h1 . W2^T + b2      256 x 1000, K = 192
//     da = (dlogits . W2) * (1 - h1^2)     256 x 192,  K = 1000       dpooled = da . W1            256 x 192,  K = 192
// The tile-per-workgroup kernel (gemm.hip, 128 x 192 tiles) runs these on 2 - 12 workgroups, one k-tile after the other:
// 10 - 36 us each, 82 us per step for 0.2 GFLOP.  Here an output tile is 32 x 32 and a workgroup's four waves split the
// reduction axis between them, so 48 - 256 workgroups are in flight and a wave runs K / 64 MFMA steps:
//   * operands come straight from global memory as MFMA fragments (16 B per lane; the matrices are a few hundred KB and sit
//     in L2), four k-steps of loads in flight ahead of their MFMAs;
//   * swapped operands (D rows = features, D cols = tokens) as everywhere in this library: a lane owns one token and quads of
//     4 consecutive features;
//   * the four partial tiles meet in LDS (fixed order 0..3: deterministic), then bias / tanh / (1 - h^2) product and the store
//     (bf16 or fp32 rows) with the rounding sequence of gemm.hip's staged epilogue: bf16(acc + bias) first, then the function.
#include "common.h"
#include "internal.h"

namespace {

enum { EPI_NONE = 0, EPI_TANH = 5, EPI_DTANH = 6 };      // numbering of gemm.hip / rgbnm.h

struct SmallNT {
  const bf16* A; const bf16* W; void* C; const float* bias; const bf16* R;
  int lda, ldw, ldc, ldr;
  int M, N, K, c_f32, ntiles;
};

constexpr int RP = 36;      // pitch (floats) of a partial tile row: 16-byte aligned, conflict-free quads

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(SmallNT p) {
  __shared__ __attribute__((aligned(16))) float red[4][32 * RP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int m0 = (blockIdx.x / p.ntiles) * 32, n0 = (blockIdx.x % p.ntiles) * 32;
  // this lane's token row (B operand) and weight row (A operand); rows past the edge re-read the last valid row and are
  // never stored
  const bf16* arow = p.A + (size_t)min(m0 + l31, p.M - 1) * p.lda;
  const bf16* wrow = p.W + (size_t)min(n0 + l31, p.N - 1) * p.ldw;
  const int ksteps = (p.K + 15) / 16;                        // 16 reduction elements per MFMA
  const int per = (ksteps + 3) / 4;
  const int s0 = w * per, s1 = min(s0 + per, ksteps);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  auto frag = [&](const bf16* row, int s) -> bf16x8 {        // elements 16 s + 8 g .. + 7 (K % 8 == 0: whole or absent)
    const int k = 16 * s + 8 * g;
    return k < p.K ? *reinterpret_cast<const bf16x8*>(row + k) : zero;
  };
  for (int s = s0; s < s1; s += 4) {
    Frag<bf16> fa[4], fw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int su = s + u < s1 ? s + u : s1 - 1;
      fa[u].v = frag(arow, su);
      fw[u].v = frag(wrow, su);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s + u < s1) mma(acc, fw[u], fa[u]);                // D rows = features, D cols = tokens
  }
  // partial tile of this wave: token l31, features 8 q + 4 g + (0..3)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = {acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    *reinterpret_cast<f32x4*>(&red[w][l31 * RP + 8 * q + 4 * g]) = v;
  }
  __syncthreads();
  const int row = tid >> 3, f0 = (tid & 7) * 4;              // 32 rows x 8 quads
  const int gm = m0 + row, gn = n0 + f0;
  if (gm >= p.M || gn >= p.N) return;                         // N % 4 == 0: a quad is whole or absent
  f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][row * RP + f0]);
#pragma unroll
  for (int k = 1; k < 4; ++k) v += *reinterpret_cast<const f32x4*>(&red[k][row * RP + f0]);
  if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + gn);
  if (p.c_f32) {                                              // fp32 logits: no bf16 rounding anywhere (gemm.hip direct path)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (EPI == EPI_TANH) v[e] = tanhf(v[e]);
      if (EPI == EPI_DTANH) {
        const float h = (float)p.R[(size_t)gm * p.ldr + gn + e];
        v[e] *= (1.f - h * h);
      }
    }
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)gm * p.ldc + gn) = v;
    return;
  }
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = (float)(bf16)v[e];                              // the tile is rounded to bf16 before the function, as in gemm.hip
    if (EPI == EPI_TANH) x = tanhf(x);
    if (EPI == EPI_DTANH) {
      const float h = (float)p.R[(size_t)gm * p.ldr + gn + e];
      x *= (1.f - h * h);
    }
    o[e] = (bf16)x;
  }
  *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + (size_t)gm * p.ldc + gn) = o;
}

}  // namespace

// 1 = shape / epilogue not eligible (the caller uses the tile-per-workgroup kernel).
int rgbnm_launch_nt_small(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                          const void* R, int ldr, int c_f32, int M, int N, int K, hipStream_t st) {
  if (M > 512 || K % 8 || N % 4 || lda % 8 || ldw % 8 || ldc % 4 || (epi != EPI_NONE && epi != EPI_TANH && epi != EPI_DTANH))
    return 1;
  if (epi == EPI_DTANH && !R) return RGBNM_EINVAL;
  SmallNT p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = C; p.bias = bias; p.R = (const bf16*)R;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.M = M; p.N = N; p.K = K; p.c_f32 = c_f32;
  p.ntiles = cdiv(N, 32);
  const int grid = cdiv(M, 32) * p.ntiles;
  const double mn = (double)M * N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * K, ((double)M * K + (double)N * K) * 2.0 + mn * (c_f32 ? 4.0 : 2.0) +
                                                              (epi == EPI_DTANH ? mn * 2.0 : 0.0), st);
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL(gemm_nt_small_kernel<EPI_NONE>, dim3(grid), dim3(256), 0, st, p); break;
    case EPI_TANH: hipLaunchKernelGGL(gemm_nt_small_kernel<EPI_TANH>, dim3(grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(gemm_nt_small_kernel<EPI_DTANH>, dim3(grid), dim3(256), 0, st, p); break;
  }
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}
