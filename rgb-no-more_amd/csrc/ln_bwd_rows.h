// LayerNorm backward over the rows of a bf16 staging tile in LDS -- the shared epilogue of gemm_nt_kpipe<EPI_LNBWD> (qkv / fc1
// dX GEMMs) and mlp_bwd_kernel (fused FeedForwardBlock backward):
//     dx[row] = residual[row] + LayerNorm'(dy[row])  w.r.t. the LayerNorm input x[row] (saved mean / rstd, gamma),
//     part[panel][0 / 1][192] = this panel's column sums of dgamma / dbeta (reduced across panels by reduce.hip).
// Reference semantics: torch.nn.LayerNorm backward as autograd runs it under models/plainvit.py:452-479 (ResidualAdd(LN -> fn)).
//
// Layout: 8 lanes per row, a lane owns three 16-byte pieces (columns v*64 + l8*8 .. +7, v = 0..2), i.e. the elements of the
// "virtual lanes" 2 l8 and 2 l8 + 1 of the 16-lanes-per-row layout of ln_bwd_kernel (layernorm.hip).  The row sums keep one
// partial per virtual lane and are combined by group8_pair_sum, which reproduces the 16-lane butterfly's order, so dx has the
// bits of ln_bwd_kernel; dgamma / dbeta are summed in a different row grouping (fp32 rounding only).  Against the 16-lane
// form this kernel family used before: half the load / store instructions (16 instead of 8 bytes per lane) and half the row
// iterations.  It is not faster: stamps of mlp_bwd_kernel show the epilogue (27 k cycles = 12.7 us) waiting on memory -- 75 KB
// of x rows + 75 KB of residual rows in and 75 KB out per CU right behind the last du stores, i.e. 58 MB per launch at
// ~4.8 TB/s -- whichever layout issues the requests.  One implementation instead of two is the point of this file.
#pragma once
#include "common.h"

namespace rgbnm {

template <int NT, int BMROWS>
struct LnBwdRows {
  static constexpr int E = 192;
  static constexpr int GROUPS = NT / 8, ITERS = BMROWS / GROUPS;
  static_assert(BMROWS % GROUPS == 0, "row groups");
  static constexpr int RED_BYTES = GROUPS * (E + 4) * 4;

  bf16x8 lx[ITERS][3], lr[ITERS][3];
  float mu[ITERS], rs[ITERS];

  // operand rows of the LayerNorm input and their statistics (request these as early as registers allow)
  __device__ __forceinline__ void request_x(const bf16* __restrict__ X, int ldx, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, int m0, int rows, int tid) {
    const int l8 = tid & 7, grp = tid >> 3;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = it * GROUPS + grp, rr = m0 + (row < rows ? row : rows - 1);
      mu[it] = mean[rr];
      rs[it] = rstd[rr];
#pragma unroll
      for (int v = 0; v < 3; ++v) lx[it][v] = *reinterpret_cast<const bf16x8*>(X + (size_t)rr * ldx + v * 64 + l8 * 8);
    }
  }
  // residual-gradient rows (R may be null: no residual branch)
  __device__ __forceinline__ void request_res(const bf16* __restrict__ R, int ldr, int m0, int rows, int tid) {
    const int l8 = tid & 7, grp = tid >> 3;
    if (!R) return;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = it * GROUPS + grp, rr = m0 + (row < rows ? row : rows - 1);
#pragma unroll
      for (int v = 0; v < 3; ++v) lr[it][v] = *reinterpret_cast<const bf16x8*>(R + (size_t)rr * ldr + v * 64 + l8 * 8);
    }
  }

  // Cs: the staged dy rows (bf16, pitch cp elements, 8-byte aligned pieces); every thread of the NT calls this after the
  // barrier that completes the tile.  red: RED_BYTES of LDS scratch that does not overlap Cs.  Contains two __syncthreads.
  __device__ __forceinline__ void run(const bf16* Cs, int cp, bf16* __restrict__ C, int ldc, const float* __restrict__ gamma,
                                      bool has_res, float* __restrict__ part, int panel, float* red, int m0, int rows, int tid) {
    run<false>(Cs, cp, C, ldc, gamma, has_res, part, panel, red, m0, rows, tid, nullptr, [](int) { __syncthreads(); });
  }
  // The same with (i) wb != null: every dx row is also written back into the staging tile (pitch cp, same pieces), from which a
  // persistent caller takes its rows as operand fragments; (ii) the four workgroup barriers of the column sums supplied by the
  // caller (bar(0) .. bar(3)): a workgroup with a non-participating wave must let that wave mirror them (vit_chain_bwd.hip).
  // (iii) SPLIT2: the column sums of a pass are taken by 384 threads -- two per column, even / odd row groups, combined by
  // one lane exchange -- instead of 192 walking all 56 groups (fp32 sums in a different order than the other callers').
  template <bool SPLIT2 = false, typename BarF>
  __device__ __forceinline__ void run(const bf16* Cs, int cp, bf16* __restrict__ C, int ldc, const float* __restrict__ gamma,
                                      bool has_res, float* __restrict__ part, int panel, float* red, int m0, int rows, int tid,
                                      bf16* wb, BarF&& bar) {
    const int l8 = tid & 7, grp = tid >> 3;
    f32x4 gm[3][2];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) gm[v][hf] = *reinterpret_cast<const f32x4*>(gamma + v * 64 + l8 * 8 + hf * 4);
    float dg[3][8], db[3][8];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int i = 0; i < 8; ++i) { dg[v][i] = 0.f; db[v][i] = 0.f; }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = it * GROUPS + grp;
      if (row < rows) {
        float xh[3][8], gv[3][8];
        float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const bf16* cptr = Cs + row * cp + v * 64 + l8 * 8;
          const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(cptr), c1 = *reinterpret_cast<const bf16x4*>(cptr + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ln_bwd_acc((float)c0[i], (float)lx[it][v][i], mu[it], rs[it], gm[v][0][i], xh[v][i], gv[v][i], s1a, s2a, dg[v][i],
                       db[v][i]);
            ln_bwd_acc((float)c1[i], (float)lx[it][v][4 + i], mu[it], rs[it], gm[v][1][i], xh[v][4 + i], gv[v][4 + i], s1b, s2b,
                       dg[v][4 + i], db[v][4 + i]);
          }
        }
        float c1 = group8_pair_sum(s1a, s1b) * (1.f / E), c2 = group8_pair_sum(s2a, s2b) * (1.f / E);
        asm volatile("" : "+v"(c1), "+v"(c2));       // (products: never contracted into the subtractions of ln_bwd_dx, whatever the caller)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          bf16x8 ob;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            ob[i] = (bf16)ln_bwd_dx(rs[it], gv[v][i], c1, xh[v][i], c2, has_res ? (float)lr[it][v][i] : 0.f);
          *reinterpret_cast<bf16x8*>(C + (size_t)(m0 + row) * ldc + v * 64 + l8 * 8) = ob;
          if (wb) {
            bf16* wp = wb + row * cp + v * 64 + l8 * 8;                 // 8-byte aligned pieces, as they were read
            *reinterpret_cast<bf16x4*>(wp) = bf16x4{ob[0], ob[1], ob[2], ob[3]};
            *reinterpret_cast<bf16x4*>(wp + 4) = bf16x4{ob[4], ob[5], ob[6], ob[7]};
          }
        }
      }
    }
    // panel-level column sums of dgamma / dbeta (fixed order => deterministic)
    for (int pass = 0; pass < 2; ++pass) {
      bar(2 * pass);
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 q = pass == 0 ? (f32x4){dg[v][4 * hf], dg[v][4 * hf + 1], dg[v][4 * hf + 2], dg[v][4 * hf + 3]}
                                    : (f32x4){db[v][4 * hf], db[v][4 * hf + 1], db[v][4 * hf + 2], db[v][4 * hf + 3]};
          *reinterpret_cast<f32x4*>(red + grp * (E + 4) + v * 64 + l8 * 8 + hf * 4) = q;
        }
      bar(2 * pass + 1);
      if constexpr (SPLIT2) {
        static_assert(GROUPS % 2 == 0 && NT >= 2 * E, "two threads per column");
        if (tid < 2 * E) {
          const int e = tid >> 1, q = tid & 1;
          float a = 0.f;
#pragma unroll 7
          for (int r = 0; r < GROUPS / 2; ++r) a += red[(2 * r + q) * (E + 4) + e];
          a += lane_xor1(a);
          if (q == 0) part[((size_t)panel * 2 + pass) * E + e] = a;
        }
      } else {
        for (int e = tid; e < E; e += NT) {
          float a = 0.f;
#pragma unroll 8
          for (int r = 0; r < GROUPS; ++r) a += red[r * (E + 4) + e];
          part[((size_t)panel * 2 + pass) * E + e] = a;
        }
      }
    }
  }
};

}  // namespace rgbnm
