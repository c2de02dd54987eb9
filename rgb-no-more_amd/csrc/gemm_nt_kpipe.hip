// Row-panel NT GEMM for the N = 192 Linear layers with a long reduction (fc2: K = 768 + bias + residual; the dX
// GEMMs of fc1 and qkv: K = 768 / 576), bf16:   C[M,192] = epi(A[M,K] . W[192,K]^T).
// These read 58-77 MB of activations to produce 19 MB: HBM-read bound.  One 448-thread workgroup per CU owns ONE
// contiguous panel of rows (M / 256 = 196 tokens at B = 256: exactly one workgroup per CU, a single balanced round;
// 7 waves x 32 rows, the last 28 rows are padding) and all 192 output columns, and streams the reduction through a
// 3-stage LDS-DMA ring of 64-wide k-tiles (A panel 28 KB + W 24 KB per stage; two tiles = 104 KB in flight per CU,
// hand-counted vmcnt, one raw s_barrier per k-tile).  Rows are 128 bytes in LDS; 16-byte chunks are XOR-swizzled
// (chunk ^ rot3((row>>1)&7)) through the DMA source address so ds_read_b128 over 32 rows is conflict free.
// Epilogue: accumulators (MFMA issued with swapped operands: a lane owns one token) -> bf16 staging tile in the ring
// memory -> coalesced 16-byte row pieces, bias and the residual (prefetched into registers before the loop) fused.
#include "common.h"
#include "internal.h"
#include "../../include/rgbnm.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int BN = 192, TKB = 128;            // k-tile: 64 bf16 = 128 bytes per row
constexpr int NWAVES = 7, NTHREADS = 64 * NWAVES, BM = 32 * NWAVES;   // 224 rows of LDS per panel
constexpr int A_STAGE = BM * TKB;             // 28 KB
constexpr int W_STAGE = BN * TKB;             // 24 KB
constexpr int STAGE = A_STAGE + W_STAGE;      // 52 KB
constexpr int NSTAGE = 3;
constexpr int NDMA = 8;                       // DMA instructions per wave per k-tile: 56 slots for 52 KB (4 repeats)
constexpr int CP = BN + 4;                    // staging pitch (elements)
constexpr int BIAS_OFF = NSTAGE * STAGE;      // 192 floats behind the ring
constexpr int SMEM = NSTAGE * STAGE + BN * 4; // 160,512 B
constexpr int NVEC = (BM * (BN / 8) + NTHREADS - 1) / NTHREADS;      // 12 output vectors per thread
static_assert(BM * CP * 2 <= NSTAGE * STAGE, "staging tile lives in the ring");
static_assert(SMEM <= 160 * 1024, "LDS");

enum { EPI_NONE = 0, EPI_RES = 1, EPI_LNBWD = 2, EPI_RES_LN = 3, EPI_GELU = 4, EPI_DGELU = 5 };
// LayerNorm epilogues: 8 lanes per row, lane l8 holds elements v * 64 + l8 * 8 + (0..7), v = 0..2 -- three 16-byte vectors,
// so rows move as 16 B per lane (half the vmem instructions of the 16-lane / 8-byte layout of layernorm.hip; in these
// epilogues every wave of the CU stores at once and a store costs ~200 cycles to issue).  Each lane plays the two
// "virtual" lanes 2 l8 and 2 l8 + 1 of the 16-lane layout and the reductions follow that butterfly (xor 8, 4, 2 across
// lanes, xor 1 inside the lane), so statistics and outputs keep the bits of ln_fwd_kernel / ln_bwd_kernel.
constexpr int LN_E = 192, LN_GROUPS = NTHREADS / 8, LN_ITERS = BM / LN_GROUPS;   // 56 row groups of 8 lanes, 4 rounds
// (EPI_LNBWD keeps the 16-lane layout: with one output it issues half the stores of EPI_RES_LN, and the 8-lane form needs
// 48 more accumulator registers for dgamma / dbeta -- measured slower)
constexpr int LB_GROUPS = NTHREADS / 16, LB_ITERS = BM / LB_GROUPS;              // 28 row groups of 16 lanes, 8 rounds
constexpr int RED_OFF = 90112;                // column-reduction scratch behind the staging tile
static_assert(BM * CP * 2 <= RED_OFF && RED_OFF + LB_GROUPS * (LN_E + 4) * 4 <= NSTAGE * STAGE, "LN scratch");

struct KpArgs {
  const bf16* A; const bf16* W; bf16* C; const float* bias; const bf16* R;
  int lda, ldw, ldc, ldr;
  int M, K, rows_per_wg, npanels;
  int ntiles;                  // N / 192 column tiles (LayerNorm epilogues: 1); block -> (panel, tile) map is XCD aware
  bf16* C2; int ldc2;          // EPI_GELU: C = gelu(u), C2 = gelu'(u);  EPI_DGELU: C = (A.W^T) * R
  // EPI_LNBWD: C = [R +] LayerNorm'(A.W^T) w.r.t. its input X (saved mean / rstd), partial dgamma/dbeta per panel
  const bf16* X; const float* gamma; const float* mean; const float* rstd; float* part;
  int ldx;
  // EPI_RES_LN: C = A.W^T + bias + R, then Y2 = LayerNorm(C) (gamma, beta) with its row statistics saved
  const float* beta; bf16* Y2; float* mean_o; float* rstd_o; float eps; int ldy2;
};

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 16 virtual lanes of a row: s0 / s1 = partial sums of virtual lanes 2 l8 / 2 l8 + 1
__device__ __forceinline__ float group8_pair_sum(float s0, float s1) {
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  return s0 + s1;
}

__device__ __forceinline__ int fswz(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}

// DMAW: an eighth wave issues every LDS-DMA of the ring (52 per k-tile).  The seven compute waves are barrier-synchronised
// per k-tile, so their 8 DMA issues each used to fall into the same window -- no MFMA runs while every wave of the CU
// is stalled in vmem issue (the attention backward showed 700-1000 cycles per instruction in such bursts).
template <int EPI, bool DMAW>
__global__ __launch_bounds__(DMAW ? NTHREADS + 64 : NTHREADS) void gemm_nt_kpipe_kernel(KpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Bs = reinterpret_cast<float*>(smem + BIAS_OFF);
  // blocks that share an A row panel (its column tiles) are neighbours on one XCD (block b runs on XCD b % 8): the
  // panel's k-tiles come from HBM once and from that XCD's L2 for the other tiles
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int panel = (jj / p.ntiles) * 8 + xcd;
  if (panel >= p.npanels) return;
  const int n0 = (jj % p.ntiles) * BN;
  const int m0 = panel * p.rows_per_wg;
  const int rows = min(p.rows_per_wg, p.M - m0);          // valid rows of this panel (<= 224)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;

  if (DMAW && w == NWAVES) {
    const int rl = lane >> 3, pc = lane & 7;
    const int lc0 = (pc ^ fswz(rl)) * 8, lc1 = (pc ^ fswz(rl + 8)) * 8;    // slot i covers rows 8 i + rl: bit 3 = i & 1
    const int T = p.K / 64;
    auto issue_all = [&](int stage, int k0) {
      unsigned char* st = smem + stage * STAGE;
#pragma unroll 4
      for (int i = 0; i < BM / 8; ++i) {
        int r8 = i * 8 + rl;
        r8 = r8 < rows ? r8 : rows - 1;
        __builtin_amdgcn_global_load_lds((glb_ptr)(p.A + (size_t)(m0 + r8) * p.lda + ((i & 1) ? lc1 : lc0) + k0),
                                         (lds_ptr)(st + i * 1024), 16, 0, 0);
      }
#pragma unroll 4
      for (int i = 0; i < BN / 8; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr)(p.W + (size_t)(n0 + i * 8 + rl) * p.ldw + ((i & 1) ? lc1 : lc0) + k0),
                                         (lds_ptr)(st + A_STAGE + i * 1024), 16, 0, 0);
    };
    issue_all(0, 0);
    if (T > 1) issue_all(1, 64);
    int st_issue = 2;
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");   // k-tile t landed; t + 1 may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 < T) {
        issue_all(st_issue, (t + 2) * 64);
        st_issue = st_issue == NSTAGE - 1 ? 0 : st_issue + 1;
      }
    }
    return;                                  // ended waves drop out of the workgroup barrier
  }

  // ---- residual rows of this panel, straight into registers (oldest loads: they never delay a k-tile wait)
  bf16x8 rv[NVEC];
  if (EPI == EPI_RES || EPI == EPI_DGELU) {
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
      const int idx = tid + NTHREADS * i, row = idx / (BN / 8), vec = idx % (BN / 8);
      const int rr = row < rows ? row : rows - 1;
      rv[i] = *reinterpret_cast<const bf16x8*>(p.R + (size_t)(m0 + rr) * p.ldr + n0 + vec * 8);
    }
  }
  // EPI_RES_LN: residual rows in the 8-lanes-per-row layout of its LayerNorm pass (gamma / beta: after the loop)
  const int l8 = tid & 7, grp = tid >> 3;
  bf16x8 lr[LN_ITERS][3];
  if (EPI == EPI_RES_LN) {
#pragma unroll
    for (int it = 0; it < LN_ITERS; ++it) {
      const int row = it * LN_GROUPS + grp, rr = m0 + (row < rows ? row : rows - 1);
#pragma unroll
      for (int v = 0; v < 3; ++v)
        lr[it][v] = *reinterpret_cast<const bf16x8*>(p.R + (size_t)rr * p.ldr + v * 64 + l8 * 8);
    }
  }
  // EPI_LNBWD (16 lanes per row): the LayerNorm input rows, their statistics and gamma, also ahead of the loop
  const int l16 = tid & 15, grp16 = tid >> 4;
  bf16x4 lx[LB_ITERS][3], lrb[LB_ITERS][3];
  float lmu[LB_ITERS], lrs[LB_ITERS];
  f32x4 gmb[3];
  if (EPI == EPI_LNBWD) {
#pragma unroll
    for (int v = 0; v < 3; ++v) gmb[v] = *reinterpret_cast<const f32x4*>(p.gamma + (v * 16 + l16) * 4);
#pragma unroll
    for (int it = 0; it < LB_ITERS; ++it) {
      const int row = it * LB_GROUPS + grp16, rr = m0 + (row < rows ? row : rows - 1);
      lmu[it] = p.mean[rr];
      lrs[it] = p.rstd[rr];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        lx[it][v] = *reinterpret_cast<const bf16x4*>(p.X + (size_t)rr * p.ldx + (v * 16 + l16) * 4);
      }
    }
  }
  if (w < BN / 64) {
    if (p.bias) __builtin_amdgcn_global_load_lds((glb_ptr)(p.bias + n0 + 64 * w + lane), (lds_ptr)(Bs + 64 * w), 4, 0, 0);
    else Bs[64 * w + lane] = 0.f;
  }

  // ---- DMA slots: instruction i (0..55, i mod 52) covers 8 rows x 128 B; i < 28: A rows, else W rows
  const bf16* src[NDMA];
  int dst[NDMA];
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    int i = w + NWAVES * j;
    i = i < 52 ? i : i - 52;
    const int r8 = (i < 28 ? i : i - 28) * 8 + (lane >> 3), pc = lane & 7;
    const int lc = pc ^ fswz(r8);
    if (i < 28) {
      const int rr = r8 < rows ? r8 : rows - 1;
      src[j] = p.A + (size_t)(m0 + rr) * p.lda + lc * 8;
      dst[j] = i * 1024;
    } else {
      src[j] = p.W + (size_t)(n0 + r8) * p.ldw + lc * 8;
      dst[j] = A_STAGE + (i - 28) * 1024;
    }
  }
  int k0 = 0;
  auto issue = [&](int stage) {
    unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
#ifdef KP_NOW      // timing experiment only: weight k-tiles beyond the first keep re-reading k-tile 0 (an L1/L2-hot line set)
      const int i = (w + NWAVES * j) < 52 ? (w + NWAVES * j) : (w + NWAVES * j) - 52;
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[j] + (i < 28 ? k0 : 0)), (lds_ptr)(st + dst[j]), 16, 0, 0);
#elif defined(KP_NOA)    // timing experiment only: activation k-tiles keep re-reading k-tile 0
      const int i = (w + NWAVES * j) < 52 ? (w + NWAVES * j) : (w + NWAVES * j) - 52;
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[j] + (i < 28 ? 0 : k0)), (lds_ptr)(st + dst[j]), 16, 0, 0);
#else
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[j] + k0), (lds_ptr)(st + dst[j]), 16, 0, 0);
#endif
    }
    k0 += 64;
  };

  // ---- fragment addresses inside a stage
  const int fl = fswz(l31);
  int foff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) foff[c] = l31 * TKB + (((2 * c + g) ^ fl) << 4);
  const int a_row = 32 * w * TKB;

  f32x16 acc[6];
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  const int T = p.K / 64;
  int st_issue = 0, st_comp = 0;
  if (!DMAW) {
    issue(0);
    st_issue = 1;
    if (T > 1) {
      issue(1);
      st_issue = 2;
    }
  }
  for (int t = 0; t < T; ++t) {
    if (!DMAW) {
      if (t + 1 < T) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();        // k-tile t landed for every wave; every wave is done with k-tile t-1
    if (!DMAW && t + 2 < T) {
      issue(st_issue);
      st_issue = st_issue == NSTAGE - 1 ? 0 : st_issue + 1;
    }
    const unsigned char* sA = smem + st_comp * STAGE + a_row;
    const unsigned char* sW = smem + st_comp * STAGE + A_STAGE;
    st_comp = st_comp == NSTAGE - 1 ? 0 : st_comp + 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Frag<bf16> fa, fb[6];
      fa.v = *reinterpret_cast<const bf16x8*>(sA + foff[c]);
#pragma unroll
      for (int b = 0; b < 6; ++b) fb[b].v = *reinterpret_cast<const bf16x8*>(sW + 32 * b * TKB + foff[c]);
#pragma unroll
      for (int b = 0; b < 6; ++b) mma(acc[b], fb[b], fa);     // swapped: D rows <-> features, D cols <-> tokens
    }
  }
  if (DMAW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the bias DMA of waves 0..2 (nothing else waited for it)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave is done reading the ring: it becomes the staging tile

  // ---- pass 1: lane = token (32 w + l31), register quad = 4 consecutive features (+ bias) -> bf16 staging tile
  bf16* Cs = reinterpret_cast<bf16*>(smem);
  const int ml = 32 * w + l31;
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = 32 * b + 8 * q + 4 * g;
      f32x4 v = {acc[b][4 * q + 0], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]};
      v += *reinterpret_cast<const f32x4*>(Bs + nl);
      store4<bf16>(Cs + ml * CP + nl, v);
    }
  f32x4 gm[3][2], bt[3][2];        // EPI_RES_LN: gamma / beta of this lane's 24 elements, requested now (accumulators dead)
  if (EPI == EPI_RES_LN) {
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        gm[v][hf] = *reinterpret_cast<const f32x4*>(p.gamma + v * 64 + l8 * 8 + hf * 4);
        bt[v][hf] = *reinterpret_cast<const f32x4*>(p.beta + v * 64 + l8 * 8 + hf * 4);
      }
  }
  if (EPI == EPI_LNBWD && p.R) {   // residual gradient rows: requested now (the accumulators are dead), used below
#pragma unroll
    for (int it = 0; it < LB_ITERS; ++it) {
      const int row = it * LB_GROUPS + grp16, rr = m0 + (row < rows ? row : rows - 1);
#pragma unroll
      for (int v = 0; v < 3; ++v)
        lrb[it][v] = *reinterpret_cast<const bf16x4*>(p.R + (size_t)rr * p.ldr + (v * 16 + l16) * 4);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (EPI == EPI_RES_LN) {
    // ---- residual add, then LayerNorm forward of the finished rows: the arithmetic of ln_fwd_kernel (layernorm.hip) --
    // same operations, same summation order, contraction off and every FMA explicit -- so both produce the same bits
#pragma clang fp contract(off)
#pragma unroll
    for (int it = 0; it < LN_ITERS; ++it) {
      const int row = it * LN_GROUPS + grp;
      if (row < rows) {
        float xv[3][8];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const bf16* cp = Cs + row * CP + v * 64 + l8 * 8;            // 8-byte aligned (CP * 2 = 392)
          const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(cp), c1 = *reinterpret_cast<const bf16x4*>(cp + 4);
          bf16x8 xb;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            xb[i] = (bf16)((float)(i < 4 ? c0[i & 3] : c1[i & 3]) + (float)lr[it][v][i]);   // the residual stream is bf16 ...
            xv[v][i] = (float)xb[i];                                                       // ... and LayerNorm sees those values
          }
          *reinterpret_cast<bf16x8*>(p.C + (size_t)(m0 + row) * p.ldc + v * 64 + l8 * 8) = xb;
          s0 += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
          s1 += xv[v][4] + xv[v][5] + xv[v][6] + xv[v][7];
        }
        const float mu = group8_pair_sum(s0, s1) * (1.f / LN_E);
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d0 = xv[v][i] - mu, d1 = xv[v][4 + i] - mu;
            q0 = __builtin_fmaf(d0, d0, q0);
            q1 = __builtin_fmaf(d1, d1, q1);
          }
        const float rs = rsqrtf(__builtin_fmaf(group8_pair_sum(q0, q1), 1.f / LN_E, p.eps));
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          bf16x8 ob;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            ob[i] = (bf16)__builtin_fmaf((xv[v][i] - mu) * rs, gm[v][i >> 2][i & 3], bt[v][i >> 2][i & 3]);
          *reinterpret_cast<bf16x8*>(p.Y2 + (size_t)(m0 + row) * p.ldy2 + v * 64 + l8 * 8) = ob;
        }
        if (l8 == 0) {
          p.mean_o[m0 + row] = mu;
          p.rstd_o[m0 + row] = rs;
        }
      }
    }
    return;
  }
  if (EPI == EPI_LNBWD) {
    // ---- LayerNorm backward on the staged rows (same arithmetic as ln_bwd_kernel, layernorm.hip): 16 lanes per row
    f32x4 dg[3], db[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      dg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
      db[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < LB_ITERS; ++it) {
      const int row = it * LB_GROUPS + grp16;
      if (row < rows) {
        const float mu = lmu[it], rs = lrs[it];
        f32x4 xh[3], gv[3];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          const bf16x4 dvb = *reinterpret_cast<const bf16x4*>(Cs + row * CP + (v * 16 + l16) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float dv = (float)dvb[i];
            xh[v][i] = ((float)lx[it][v][i] - mu) * rs;
            gv[v][i] = dv * gmb[v][i];
            s1 += gv[v][i];
            s2 += gv[v][i] * xh[v][i];
            dg[v][i] += dv * xh[v][i];
            db[v][i] += dv;
          }
        }
        const float c1 = group16_sum(s1) * (1.f / LN_E), c2 = group16_sum(s2) * (1.f / LN_E);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = rs * (gv[v][i] - c1 - xh[v][i] * c2);
          if (p.R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += (float)lrb[it][v][i];
          }
          store4<bf16>(p.C + (size_t)(m0 + row) * p.ldc + (v * 16 + l16) * 4, o);
        }
      }
    }
    // panel-level column sums of dgamma / dbeta (fixed order => deterministic); reduced across panels by reduce.hip
    float* red = reinterpret_cast<float*>(smem + RED_OFF);
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[grp16 * (LN_E + 4) + (v * 16 + l16) * 4 + i] = pass == 0 ? dg[v][i] : db[v][i];
      __syncthreads();
      for (int e = tid; e < LN_E; e += NTHREADS) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < LB_GROUPS; ++r) a += red[r * (LN_E + 4) + e];
        p.part[((size_t)panel * 2 + pass) * LN_E + e] = a;
      }
    }
    return;
  }
  // ---- pass 2: valid rows x 24 vectors of 8 features, coalesced 384-byte rows
#pragma unroll
  for (int i = 0; i < NVEC; ++i) {
    const int idx = tid + NTHREADS * i, row = idx / (BN / 8), vec = idx % (BN / 8);
    if (row >= rows) continue;
    const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(Cs + row * CP + vec * 8);
    const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(Cs + row * CP + vec * 8 + 4);
    bf16x8 cv = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    if (EPI == EPI_RES) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cv[e] = (bf16)((float)cv[e] + (float)rv[i][e]);
    }
    if (EPI == EPI_DGELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cv[e] = (bf16)((float)cv[e] * (float)rv[i][e]);
    }
    if (EPI == EPI_GELU) {     // one erfc / exp2 evaluation yields gelu(u) (-> C) and gelu'(u) (-> C2), as in gemm.hip
      bf16x8 dv;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2 u = {(float)cv[e], (float)cv[e + 1]};
        f32x2 gv, dgv;
        gelu_pair_fast(u, gv, dgv);
        dv[e] = (bf16)dgv[0];
        dv[e + 1] = (bf16)dgv[1];
        cv[e] = (bf16)gv[0];
        cv[e + 1] = (bf16)gv[1];
      }
      *reinterpret_cast<bf16x8*>(p.C2 + (size_t)(m0 + row) * p.ldc2 + n0 + vec * 8) = dv;
    }
    *reinterpret_cast<bf16x8*>(p.C + (size_t)(m0 + row) * p.ldc + n0 + vec * 8) = cv;
  }
}

template <int EPI, bool DMAW>
int launch_v(const KpArgs& p, hipStream_t st) {
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)gemm_nt_kpipe_kernel<EPI, DMAW>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) !=
        hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  hipLaunchKernelGGL((gemm_nt_kpipe_kernel<EPI, DMAW>), dim3(cdiv(p.npanels, 8) * 8 * p.ntiles),
                     dim3(DMAW ? NTHREADS + 64 : NTHREADS), SMEM, st, p);
  LAUNCH_CHECK();
  return RGBNM_OK;
}
template <int EPI>
int launch(const KpArgs& p, hipStream_t st) {
  return rgbnm_get_option("nt_dmawave") ? launch_v<EPI, true>(p, st) : launch_v<EPI, false>(p, st);
}

}  // namespace

// x = A . W^T + bias + R ; y = LayerNorm(x): fc2 (+ the next block's LN1) and proj (+ LN2) in one launch each.
int rgbnm_launch_nt_kpipe_res_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* R,
                                 int ldr, void* x, int ldc, const float* gamma, const float* beta, void* y, int ldy,
                                 float* mean, float* rstd, float eps, int M, int N, int K, hipStream_t st) {
  if (N != BN || K % 64 || K < 128 || lda % 8 || ldw % 8 || ldc % 8 || ldr % 8 || ldy % 8 || M < 8192) return 1;
  KpArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = (bf16*)x; p.bias = bias; p.R = (const bf16*)R;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.M = M; p.K = K;
  p.X = nullptr; p.mean = p.rstd = nullptr; p.part = nullptr; p.ldx = 0;
  p.gamma = gamma; p.beta = beta; p.Y2 = (bf16*)y; p.mean_o = mean; p.rstd_o = rstd; p.eps = eps; p.ldy2 = ldy;
  p.ntiles = 1; p.C2 = nullptr; p.ldc2 = 0;
  int rows = cdiv(M, 256);
  if (rows > BM) rows = BM;
  p.rows_per_wg = rows;
  p.npanels = cdiv(M, rows);
  const double mn = (double)M * N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * K, ((double)M * K + (double)N * K) * 2.0 + mn * 2.0 * 3.0, st);
  const int rc = launch<EPI_RES_LN>(p, st);
  rgbnm_trace_end(slot, st);
  return rc;
}

// dx = [dres +] LayerNorm'(A . W^T): the dX GEMM of fc1 / qkv with the LayerNorm backward fused into its epilogue
// (saves writing and re-reading the [M,192] gradient and one launch).  part: [npanels][2][192] partial dgamma/dbeta;
// *npanels_out tells the caller how many slices to reduce.  Returns 1 when the shape is not eligible.
int rgbnm_launch_nt_kpipe_lnbwd(const void* A, int lda, const void* W, int ldw, const void* X, int ldx,
                                const float* gamma, const float* mean, const float* rstd, const void* dres, int ldr,
                                void* dx, int ldc, float* part, int* npanels_out, int M, int N, int K, hipStream_t st) {
  if (N != BN || K % 64 || K < 256 || lda % 8 || ldw % 8 || ldc % 4 || ldx % 4 || (dres && ldr % 4) || M < 8192) return 1;
  KpArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = (bf16*)dx; p.bias = nullptr; p.R = (const bf16*)dres;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.M = M; p.K = K;
  p.X = (const bf16*)X; p.gamma = gamma; p.mean = mean; p.rstd = rstd; p.part = part; p.ldx = ldx;
  p.beta = nullptr; p.Y2 = nullptr; p.mean_o = p.rstd_o = nullptr; p.eps = 0.f; p.ldy2 = 0;
  p.ntiles = 1; p.C2 = nullptr; p.ldc2 = 0;
  int rows = cdiv(M, 256);
  if (rows > BM) rows = BM;
  p.rows_per_wg = rows;
  p.npanels = cdiv(M, rows);
  *npanels_out = p.npanels;
  const double mn = (double)M * N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * K, ((double)M * K + (double)N * K) * 2.0 + mn * 2.0 * (dres ? 3.0 : 2.0), st);
  const int rc = launch<EPI_LNBWD>(p, st);
  rgbnm_trace_end(slot, st);
  return rc;
}

// returns 1 when the shape is not eligible (caller falls back to the tile-per-workgroup kernel)
int rgbnm_launch_nt_kpipe(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                          const void* R, int ldr, void* C2, int ldc2, int M, int N, int K, hipStream_t st) {
  // epi uses the numbering of gemm.hip: 0 none, 1 residual, 2 GELU (+ GELU' into C2), 4 dGELU product
  if (N % BN || K % 64 || K < 256 || lda % 8 || ldw % 8 || ldc % 8 || M < 8192) return 1;
  if (epi != 0 && epi != 1 && epi != 2 && epi != 4) return 1;
  if ((epi == 1 || epi == 4) && (!R || ldr % 8)) return 1;
  if (epi == 2 && (!C2 || ldc2 % 8)) return 1;
  KpArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = (bf16*)C; p.bias = bias; p.R = (const bf16*)R;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.M = M; p.K = K;
  p.X = nullptr; p.gamma = p.mean = p.rstd = nullptr; p.part = nullptr; p.ldx = 0;
  p.beta = nullptr; p.Y2 = nullptr; p.mean_o = p.rstd_o = nullptr; p.eps = 0.f; p.ldy2 = 0;
  p.C2 = (bf16*)C2; p.ldc2 = ldc2;
  p.ntiles = N / BN;
  // N = 192: one panel per CU when it fits (M / 256 rows, at most 224: one balanced round).  Several column tiles: full
  // 224-row panels (the workgroup's arithmetic intensity against the L2 -> CU fabric is what bounds these shapes)
  int rows = p.ntiles == 1 ? cdiv(M, 256) : BM;
  if (rows > BM) rows = BM;
  p.rows_per_wg = rows;
  p.npanels = cdiv(M, rows);
  const double mn = (double)M * N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * K, ((double)M * K + (double)N * K) * 2.0 + mn * 2.0 +
                                                              (epi != 0 ? mn * 2.0 : 0.0), st);
  const int rc = epi == 1 ? launch<EPI_RES>(p, st) : epi == 2 ? launch<EPI_GELU>(p, st)
               : epi == 4 ? launch<EPI_DGELU>(p, st) : launch<EPI_NONE>(p, st);
  rgbnm_trace_end(slot, st);
  return rc;
}
