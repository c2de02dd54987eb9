// GEMM family for the JPEG-ViT hot path (all Linear layers, forward and backward) on gfx950.
//
//   gemm_nt : C[M,N] = epi( A[M,K] . W[N,K]^T )            forward Linear (reference nn.Linear layout
//                                                          (out,in), plainvit.py:195,441,443,487,490,553,555)
//                                                          and dX = dY . W via the transposed weight shadow.
//   gemm_tn : dW[No,Ki] = dY[M,No]^T . X[M,Ki]  (+ db)     weight gradients; split over tokens, fp32
//                                                          partials + deterministic reduce.
//   prep_weights : fp32 master -> T shadow [N,K] (rows optionally de-interleaved for qkv) and [K,N].
//
// Tiles are 128 x (64*NB) with four waves in a 2x2 grid; each wave owns 64 x (32*NB) as 32x32 MFMA
// tiles (v_mfma_f32_32x32x16_bf16, or 4x v_mfma_f32_32x32x2_f32 in strict fp32 mode).  LDS rows are
// 128 B of the reduction axis + 16 B pad (144 B pitch): ds_read_b128 of 32 rows at one k-offset is
// bank-conflict free (9*i mod 16 distinct).
#include "common.h"
#include <mutex>
#include <vector>
#include "../../include/rgbnm.h"
#include "internal.h"

namespace {

enum { EPI_NONE = 0, EPI_RES = 1, EPI_GELU = 2, EPI_POS = 3, EPI_DGELU = 4, EPI_TANH = 5, EPI_DTANH = 6 };

struct GemmNT {
  const void* A; const void* W; void* C; const float* bias; const void* R; void* C2; const float* pos;
  int lda, ldw, ldc, ldr, ldc2, pos_period;
  int M, N, K;
  int c_f32;
  int mtiles, ntiles;
};

constexpr int BM = 128;
constexpr int PITCH_B = 144;  // bytes per LDS tile row

// STAGED (bf16 only): the MFMA is issued with operands swapped (D rows <-> weight rows, D cols <-> tokens) so a
// lane owns ONE token and 4 consecutive output features per register quad; the tile is rounded to bf16 into LDS
// (8-byte writes, pitch BN+4 elements: conflict free) and leaves the CU as coalesced 16-byte rows, where the
// residual / GELU / dGELU operands are also read as 16-byte vectors.
template <typename T, int NB, int EPI, bool STAGED>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p) {
  constexpr int BN = 64 * NB;
  constexpr int CP = BN + 4;                    // staging pitch (elements)
  constexpr int TILE_BYTES = (BM + BN) * PITCH_B;
  constexpr int STAGE_BYTES = STAGED ? BM * CP * 2 : 0;
  constexpr int SMEM_BYTES = TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES;
  constexpr int EPV = 16 / (int)sizeof(T);      // elements per 16-byte vector
  constexpr int BK = 128 / (int)sizeof(T);      // elements per k-tile row
  constexpr int PE = PITCH_B / (int)sizeof(T);  // LDS pitch in elements
  constexpr int CH = 32 / (int)sizeof(T);       // elements per 32-byte mma chunk
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = reinterpret_cast<T*>(smem + BM * PITCH_B);

  // XCD-aware block -> tile map: blocks that share an A row-panel run on one XCD (b % 8), back to back.
  const int id = blockIdx.x;
  const int xcd = id & 7, j = id >> 3;
  const int nt_i = j % p.ntiles;
  const int mt_i = (j / p.ntiles) * 8 + xcd;
  if (mt_i >= p.mtiles) return;
  const int m0 = mt_i * BM, n0 = nt_i * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, g = lane >> 5;
  const int vcol = tid & 7, vrow = tid >> 3;

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* W = reinterpret_cast<const T*>(p.W);

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  Frag<T> ra[4], rb[2 * NB];
  const int KT = (p.K + BK - 1) / BK;

  auto gload = [&](int kt) {
    const int k = kt * BK + vcol * EPV;
    const bool kin = (k + EPV) <= p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = m0 + vrow + 32 * i;
      row = row < p.M ? row : p.M - 1;
      if (kin) ra[i] = load_frag<T>(A + (size_t)row * p.lda + k);
      else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) ra[i].v[e] = (T)0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) {
      int row = n0 + vrow + 32 * i;
      row = row < p.N ? row : p.N - 1;
      if (kin) rb[i] = load_frag<T>(W + (size_t)row * p.ldw + k);
      else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) rb[i].v[e] = (T)0.f;
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<decltype(ra[i].v)*>(As + (vrow + 32 * i) * PE + vcol * EPV) = ra[i].v;
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) *reinterpret_cast<decltype(rb[i].v)*>(Bs + (vrow + 32 * i) * PE + vcol * EPV) = rb[i].v;
  };

  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) gload(kt + 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      Frag<T> fa[2], fb[NB];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = load_frag<T>(As + (wm * 64 + a * 32 + l31) * PE + c * CH + g * Frag<T>::EPL);
#pragma unroll
      for (int b = 0; b < NB; ++b) fb[b] = load_frag<T>(Bs + (wn * 32 * NB + b * 32 + l31) * PE + c * CH + g * Frag<T>::EPL);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (STAGED) mma(acc[a][b], fb[b], fa[a]);
          else mma(acc[a][b], fa[a], fb[b]);
        }
    }
    __syncthreads();
    if (kt + 1 < KT) {
      lstore();
      __syncthreads();
    }
  }

  if constexpr (STAGED) {
    // ---- pass 1: registers -> LDS (bf16(acc + bias [+ pos])) ; lane = token, quad = 4 features ----
    bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ml = wm * 64 + a * 32 + l31;
      int prow = m0 + ml;
      prow = prow < p.M ? prow : p.M - 1;
      const float* posr = (EPI == EPI_POS) ? p.pos + (size_t)(prow % p.pos_period) * p.N : nullptr;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = wn * 32 * NB + b * 32 + 8 * q + 4 * g;
          int nc = n0 + nl;
          nc = nc + 4 <= p.N ? nc : 0;
          f32x4 v = {acc[a][b][4 * q + 0], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
          if (p.bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + nc);
            v += bv;
          }
          if (EPI == EPI_POS) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(posr + nc);
            v += pv;
          }
          store4<bf16>(Cs + ml * CP + nl, v);
        }
    }
    __syncthreads();
    // ---- pass 2: LDS -> global, 8 features (16 B) per thread-iteration, rows fully coalesced ----
    constexpr int VPR = BN / 8;
    bf16* C = reinterpret_cast<bf16*>(p.C);
    const bf16* R = reinterpret_cast<const bf16*>(p.R);
    bf16* C2 = reinterpret_cast<bf16*>(p.C2);
    int row = tid / VPR, vec = tid % VPR;
#pragma unroll 4
    for (int i = 0; i < BM * VPR / 256; ++i, row += 256 / VPR, vec += 256 % VPR) {
      if (vec >= VPR) { vec -= VPR; row += 1; }
      const int gm = m0 + row, gn = n0 + vec * 8;
      if (gm >= p.M || gn >= p.N) continue;
      const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(Cs + row * CP + vec * 8);
      const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(Cs + row * CP + vec * 8 + 4);
      bf16x8 cv = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      if (EPI == EPI_GELU) {
        // one erf/exp evaluation yields gelu(u) (-> C) and gelu'(u) (-> C2, consumed by EPI_DGELU in backward)
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
        *reinterpret_cast<bf16x8*>(C2 + (size_t)gm * p.ldc2 + gn) = dv;
      }
      if (EPI != EPI_NONE && EPI != EPI_POS && EPI != EPI_GELU) {
        bf16x8 rv;
        if (EPI == EPI_RES || EPI == EPI_DGELU || EPI == EPI_DTANH)
          rv = *reinterpret_cast<const bf16x8*>(R + (size_t)gm * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = (float)cv[e];
          if (EPI == EPI_RES) v += (float)rv[e];
          if (EPI == EPI_DGELU) v *= (float)rv[e];
          if (EPI == EPI_TANH) v = tanhf(v);
          if (EPI == EPI_DTANH) {
            const float h = (float)rv[e];
            v *= (1.f - h * h);
          }
          cv[e] = (bf16)v;
        }
      }
      *reinterpret_cast<bf16x8*>(C + (size_t)gm * p.ldc + gn) = cv;
    }
    return;
  }

  // ---- direct epilogue (fp32 mode, fp32 outputs, ragged N) --------------------------------------
  T* C = reinterpret_cast<T*>(p.C);
  float* Cf = reinterpret_cast<float*>(p.C);
  const T* R = reinterpret_cast<const T*>(p.R);
  T* C2 = reinterpret_cast<T*>(p.C2);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = n0 + wn * 32 * NB + b * 32 + l31;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + a * 32 + acc_row(r, lane);
        if (row >= p.M) continue;
        float v = acc[a][b][r] + bv;
        if (EPI == EPI_RES) v += to_f32(R[(size_t)row * p.ldr + col]);
        if (EPI == EPI_GELU) {
          const float u = to_f32(from_f32<T>(v));   // pre-activation at the activation dtype's precision
          C2[(size_t)row * p.ldc2 + col] = from_f32<T>(dgelu_f(u));
          v = gelu_f(u);
        }
        if (EPI == EPI_POS) v += p.pos[(size_t)(row % p.pos_period) * p.N + col];
        if (EPI == EPI_DGELU) v *= to_f32(R[(size_t)row * p.ldr + col]);
        if (EPI == EPI_TANH) v = tanhf(v);
        if (EPI == EPI_DTANH) {
          const float h = to_f32(R[(size_t)row * p.ldr + col]);
          v *= (1.f - h * h);
        }
        if (p.c_f32) Cf[(size_t)row * p.ldc + col] = v;
        else C[(size_t)row * p.ldc + col] = from_f32<T>(v);
      }
    }
  }
}

template <typename T, int NB, bool STAGED>
int launch_nt_epi(const GemmNT& p, int epi, hipStream_t st) {
  const int grid = ((p.mtiles + 7) / 8) * 8 * p.ntiles;
  const double esz = sizeof(T), mn = (double)p.M * p.N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * p.K,
                                     ((double)p.M * p.K + (double)p.N * p.K) * esz + mn * (p.c_f32 ? 4.0 : esz) +
                                         ((epi == EPI_RES || epi == EPI_DGELU || epi == EPI_DTANH || epi == EPI_GELU) ? mn * esz : 0.0),
                                     st);
  switch (epi) {
#define CASE(E) case E: hipLaunchKernelGGL((gemm_nt_kernel<T, NB, E, STAGED>), dim3(grid), dim3(256), 0, st, p); break;
    CASE(EPI_NONE) CASE(EPI_RES) CASE(EPI_GELU) CASE(EPI_POS) CASE(EPI_DGELU) CASE(EPI_TANH) CASE(EPI_DTANH)
#undef CASE
    default: return RGBNM_EINVAL;
  }
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

template <typename T, int NB>
int launch_nt_sel(const GemmNT& p, int epi, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (p.M <= 512 && !p.pos && !p.C2 && rgbnm_get_option("nt_small")) {
      // few rows (the classification head: M = batch): 32 x 32 tiles, the reduction split over a workgroup's four waves
      // (gemm_nt_small.hip); 1 = shape / epilogue not eligible
      const int rc = rgbnm_launch_nt_small(epi, p.A, p.lda, p.W, p.ldw, p.C, p.ldc, p.bias, p.R, p.ldr, p.c_f32, p.M, p.N,
                                           p.K, st);
      if (rc != 1) return rc;
    }
    const bool ok = !p.c_f32 && (p.N % 8 == 0) && (p.ldc % 8 == 0) && (!p.R || p.ldr % 8 == 0) &&
                    (!p.C2 || p.ldc2 % 8 == 0) && rgbnm_get_option("nt_staged");
    if (ok && !p.pos && rgbnm_get_option("nt_kpipe")) {
      // N % 192 == 0 with a long reduction (K >= 256): 224-row panels x 192-column tiles, k-tiles through an LDS-DMA ring
      // (gemm_nt_kpipe.hip); N = 192: one row panel per CU
      const int rc = rgbnm_launch_nt_kpipe(epi, p.A, p.lda, p.W, p.ldw, p.C, p.ldc, p.bias, p.R, p.ldr, p.C2, p.ldc2, p.M,
                                           p.N, p.K, st);
      if (rc != 1) return rc;
    }
    if (ok && !p.pos && rgbnm_get_option("nt_wres")) {
      // K = 192 layers: persistent weight-resident kernel (gemm_nt_wres.hip); 1 = shape not eligible
      const int rc = rgbnm_launch_nt_wres(epi, p.A, p.lda, p.W, p.ldw, p.C, p.ldc, p.bias, p.R, p.ldr, p.C2, p.ldc2,
                                          p.M, p.N, p.K, st);
      if (rc != 1) return rc;
    }
    if (ok) return launch_nt_epi<T, NB, true>(p, epi, st);
  }
  return launch_nt_epi<T, NB, false>(p, epi, st);
}

template <typename T>
int launch_nt(GemmNT p, int epi, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return RGBNM_EINVAL;
  const int epv = 16 / (int)sizeof(T);
  if (p.K % epv || p.lda % epv || p.ldw % epv) return RGBNM_EINVAL;
  p.mtiles = cdiv(p.M, BM);
  if (p.N % 192 == 0) {
    p.ntiles = p.N / 192;
    return launch_nt_sel<T, 3>(p, epi, st);
  }
  p.ntiles = cdiv(p.N, 128);
  return launch_nt_sel<T, 2>(p, epi, st);
}

// ------------------------------------------------------------------------------------------------
// TN: dW[No,Ki] = sum_m dY[m,No] * X[m,Ki].  Tiles are staged in their natural [token][feature] layout
// (coalesced 16-byte rows); the MFMA fragments (reduction = tokens) are gathered with strided LDS reads.
// ------------------------------------------------------------------------------------------------
struct GemmTN {
  const void* dY; const void* X; float* part; float* bpart;
  int ldy, ldx;
  int M, No, Ki, S, tok_per_split;
  int rtiles, ctiles;
};

template <typename T> __device__ __forceinline__ Frag<T> gather_frag(const T* base, int pitch_e) {
  Frag<T> f;
#pragma unroll
  for (int j = 0; j < Frag<T>::EPL; ++j) f.v[j] = base[j * pitch_e];
  return f;
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the address of 4 contiguous bf16 (row i>>2 of a
// 4 x 16 block, columns 4*(i&3)..+3) and receives column i of that block (4 rows).  With tiles stored in their
// natural [token][feature] layout this hands every lane 4 tokens of ITS feature: two reads = one MFMA fragment
// (8 reduction slots), instead of 8 strided 2-byte reads.  A and B fragments use the same token<->slot mapping.
__device__ __forceinline__ bf16x8 tr_pack(u32x2 lo, u32x2 hi) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int NB, int PAB, int PBB>   // pitches in BYTES
__device__ __forceinline__ void tr_load_chunk(unsigned addrA, unsigned addrB, Frag<bf16> (&fa)[2], Frag<bf16> (&fb)[NB]) {
  u32x2 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h, b2l, b2h;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %10\n\t"
      "ds_read_b64_tr_b16 %1, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %2, %10 offset:64\n\t"
      "ds_read_b64_tr_b16 %3, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %11\n\t"
      "ds_read_b64_tr_b16 %5, %11 offset:%14\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:64\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%15\n\t"
      "ds_read_b64_tr_b16 %8, %11 offset:%16\n\t"
      "ds_read_b64_tr_b16 %9, %11 offset:%17\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h), "=&v"(b2l),
        "=&v"(b2h)
      : "v"(addrA), "v"(addrB), "i"(4 * PAB), "i"(64 + 4 * PAB), "i"(4 * PBB), "i"(64 + 4 * PBB),
        "i"(NB == 3 ? 128 : 0), "i"(NB == 3 ? 128 + 4 * PBB : 4 * PBB)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
  fa[0].v = tr_pack(a0l, a0h);
  fa[1].v = tr_pack(a1l, a1h);
  fb[0].v = tr_pack(b0l, b0h);
  fb[1].v = tr_pack(b1l, b1h);
  if constexpr (NB == 3) fb[2].v = tr_pack(b2l, b2h);
}

template <typename T, int NB, bool TR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
  constexpr int BN = 64 * NB;
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int TK = 128 / (int)sizeof(T);                 // tokens per k-tile (64 bf16 / 32 f32)
  constexpr int PA = (BM * (int)sizeof(T) + 64) / (int)sizeof(T);   // pitch (elements), +64 B pad
  constexpr int PB = (BN * (int)sizeof(T) + 64) / (int)sizeof(T);
  constexpr int VA = BM / EPV, VB = BN / EPV;              // 16-byte vectors per tile row
  constexpr int NLA = TK * VA / 256, NLB = TK * VB / 256;  // vectors per thread
  constexpr int EPL = Frag<T>::EPL;
  __shared__ __attribute__((aligned(16))) unsigned char smem[TK * (PA + PB) * sizeof(T)];
  T* Ys = reinterpret_cast<T*>(smem);
  T* Xs = Ys + TK * PA;

  // 1-D grid, XCD-aware: all output tiles of one token split run on the same XCD (they re-read the same X rows)
  const int ntile = p.rtiles * p.ctiles;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int tile = jj % ntile, s = (jj / ntile) * 8 + xcd;
  if (s >= p.S) return;
  const int rt = tile / p.ctiles, ct = tile % p.ctiles;
  const int r0 = rt * BM, c0 = ct * BN;
  const int tok0 = s * p.tok_per_split;
  const int tok1 = min(p.M, tok0 + p.tok_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, g = lane >> 5;
  const T* dY = reinterpret_cast<const T*>(p.dY);
  const T* X = reinterpret_cast<const T*>(p.X);

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = (p.bpart != nullptr) && (ct == 0) && (tid < BM);

  Frag<T> ra[NLA], rb[NLB];
  auto gload = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int idx = tid + 256 * i, row = idx / VA, v = idx % VA;
      const int tok = t0 + row, f = r0 + v * EPV;
      if (tok < tok1 && f + EPV <= p.No) ra[i] = load_frag<T>(dY + (size_t)tok * p.ldy + f);
      else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) ra[i].v[e] = (T)0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + 256 * i, row = idx / VB, v = idx % VB;
      const int tok = t0 + row, f = c0 + v * EPV;
      if (tok < tok1 && f + EPV <= p.Ki) rb[i] = load_frag<T>(X + (size_t)tok * p.ldx + f);
      else {
#pragma unroll
        for (int e = 0; e < EPV; ++e) rb[i].v[e] = (T)0.f;
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int idx = tid + 256 * i, row = idx / VA, v = idx % VA;
      *reinterpret_cast<decltype(ra[i].v)*>(Ys + row * PA + v * EPV) = ra[i].v;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + 256 * i, row = idx / VB, v = idx % VB;
      *reinterpret_cast<decltype(rb[i].v)*>(Xs + row * PB + v * EPV) = rb[i].v;
    }
  };

  if (tok0 < tok1) {
    gload(tok0);
    lstore();
    __syncthreads();
    for (int t0 = tok0; t0 < tok1; t0 += TK) {
      const bool more = (t0 + TK) < tok1;
      if (more) gload(t0 + TK);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Frag<T> fa[2], fb[NB];
        const int trow = c * 2 * EPL + g * EPL;
        if constexpr (TR) {
          const int tr = trow + ((lane & 15) >> 2);
          const int fc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
          const unsigned aA = (unsigned)(size_t)(Ys + tr * PA + wm * 64 + fc);
          const unsigned aB = (unsigned)(size_t)(Xs + tr * PB + wn * 32 * NB + fc);
          tr_load_chunk<NB, PA * 2, PB * 2>(aA, aB, fa, fb);
        } else {
#pragma unroll
          for (int a = 0; a < 2; ++a) fa[a] = gather_frag<T>(Ys + trow * PA + wm * 64 + a * 32 + l31, PA);
#pragma unroll
          for (int b = 0; b < NB; ++b) fb[b] = gather_frag<T>(Xs + trow * PB + wn * 32 * NB + b * 32 + l31, PB);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) mma(acc[a][b], fa[a], fb[b]);
      }
      if (do_bias) {
#pragma unroll 8
        for (int t = 0; t < TK; ++t) bsum += to_f32(Ys[t * PA + tid]);
      }
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }

  float* part = p.part + (size_t)s * p.No * p.Ki;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = c0 + wn * 32 * NB + b * 32 + l31;
    if (col >= p.Ki) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + wm * 64 + a * 32 + acc_row(r, lane);
        if (row < p.No) part[(size_t)row * p.Ki + col] = acc[a][b][r];
      }
  }
  if (do_bias && r0 + tid < p.No) p.bpart[(size_t)s * p.No + r0 + tid] = bsum;
}

// out[rowmap(n)][k] (+)= sum_s part[s][n][k];  rowmap de-interleaves qkv rows (n = s3*H*64 + h*64 + d ->
// reference row h*192 + d*3 + s3, plainvit.py:447 '(h d qkv)') when perm_heads > 0.
__device__ __forceinline__ int qkv_row(int n, int heads) {
  const int inner = heads * 64;
  const int s3 = n / inner, rem = n % inner;
  const int h = rem / 64, d = rem % 64;
  return h * 192 + d * 3 + s3;
}

// weight partials [S][No][Ki] and bias partials [S][No] -> two jobs of the batched reduction (reduce.hip)
int submit_tn_reduce(const GemmTN& p, float* dW, float* db, int S, int perm_heads, int accumulate, hipStream_t st) {
  RgbnmReduceJob j;
  j.part = p.part; j.stride = (long long)p.No * p.Ki; j.out = dW; j.n = p.No * p.Ki; j.S = S; j.cols = p.Ki;
  j.perm_heads = perm_heads; j.accumulate = accumulate; j.epw = 64;
  if (!db) return rgbnm_reduce_submit(j, st);
  const bool own = !rgbnm_reduce_defer_active();        // stand-alone call: weight and bias partials in ONE reduction launch
  if (own) rgbnm_reduce_defer_begin();
  int rc = rgbnm_reduce_submit(j, st);
  if (rc == RGBNM_OK) {
    j.part = p.bpart; j.stride = p.No; j.out = db; j.n = p.No; j.cols = 1;
    rc = rgbnm_reduce_submit(j, st);
  }
  if (own) {
    const int rf = rgbnm_reduce_defer_flush(st);
    if (rc == RGBNM_OK) rc = rf;
  }
  return rc;
}

// deferred (grouped) weight-gradient launches
struct TnPending { GemmTN p; float* dW; float* db; int perm_heads, accumulate; };
thread_local bool g_tn_defer = false;     // per host thread, like the reduction queue (reduce.hip)
constexpr int TN_QMAX = 52;      // = TN_MAXJOBS of gemm_tn_pipe.hip
thread_local TnPending g_tn_q[TN_QMAX];
thread_local int g_tn_n = 0, g_tn_cap = 4;
thread_local int g_tn_tiles = 0;          // output tiles of the queued jobs (gemm_tn_pipe.hip: 128 x 192 tiles, one workgroup each)
thread_local int g_tn_depth = 0;          // begin / end pairs opened INSIDE the open bracket: they join it (see rgbnm_tn_defer_begin_n)
// A bracket may be abandoned from ANOTHER host thread than the one whose queue holds its jobs (swinv2.py: a backward pass that
// never reached its last node is noticed by the next forward, on the caller's thread, while the queue is thread_local to the
// autograd worker): the abandoning thread names the bracket (rgbnm_gemm_tn_group_abort), the owning thread drops the queue -- its
// operands may be freed by then: they are never launched -- the next time it touches it.
thread_local unsigned long long g_tn_id = 0;
std::mutex g_tn_abort_mu;
std::vector<unsigned long long> g_tn_aborted;
void tn_check_abort() {
  if (!g_tn_id) return;
  std::lock_guard<std::mutex> lk(g_tn_abort_mu);
  for (size_t i = 0; i < g_tn_aborted.size(); ++i)
    if (g_tn_aborted[i] == g_tn_id) {
      g_tn_aborted.erase(g_tn_aborted.begin() + i);
      g_tn_n = g_tn_tiles = g_tn_depth = 0;
      g_tn_defer = false;
      g_tn_cap = 4;
      g_tn_id = 0;
      return;
    }
}

int tn_flush(hipStream_t st) {
  const int n = g_tn_n;
  g_tn_n = 0;
  g_tn_tiles = 0;
  if (!n) return RGBNM_OK;
  RgbnmTnJob jobs[TN_QMAX];
  for (int i = 0; i < n; ++i) {
    const GemmTN& p = g_tn_q[i].p;
    jobs[i].dY = p.dY; jobs[i].X = p.X; jobs[i].part = p.part; jobs[i].bpart = g_tn_q[i].db ? p.bpart : nullptr;
    jobs[i].ldy = p.ldy; jobs[i].ldx = p.ldx; jobs[i].M = p.M; jobs[i].No = p.No; jobs[i].Ki = p.Ki;
    jobs[i].dW = g_tn_q[i].dW; jobs[i].db = g_tn_q[i].db; jobs[i].perm_heads = g_tn_q[i].perm_heads;
    jobs[i].accumulate = g_tn_q[i].accumulate;
    jobs[i].smax = p.S;
  }
  int S = 0, direct = 0;
  const int rc = rgbnm_launch_tn_pipe_group(jobs, n, &S, st, rgbnm_get_option("tn_direct") ? &direct : nullptr);
  if (rc != RGBNM_OK) return rc < 0 ? rc : RGBNM_EINVAL;     // eligibility was checked when the jobs were queued
  if (direct) return RGBNM_OK;                               // no token split: the kernel wrote dW / db itself
  const bool own = !rgbnm_reduce_defer_active();             // the reductions of one grouped launch: one launch (also when the
  if (own) rgbnm_reduce_defer_begin();                       // queue runs by itself in the middle of a long bracket)
  int rr = RGBNM_OK;
  for (int i = 0; i < n && rr == RGBNM_OK; ++i)
    rr = submit_tn_reduce(g_tn_q[i].p, g_tn_q[i].dW, g_tn_q[i].db, S, g_tn_q[i].perm_heads, g_tn_q[i].accumulate, st);
  if (own) {
    const int rf = rgbnm_reduce_defer_flush(st);
    if (rr == RGBNM_OK) rr = rf;
  }
  return rr;
}

bool tn_groupable(const GemmTN& p) {
  return rgbnm_get_option("tn_pipe") && p.M % 64 == 0 && p.M >= 64 &&
         p.Ki % 192 == 0 && p.No % 8 == 0 && p.ldy % 8 == 0 && p.ldx % 8 == 0;
}

template <typename T>
int launch_tn(GemmTN p, float* dW, float* db, int perm_heads, int accumulate, hipStream_t st) {
  const int epv = 16 / (int)sizeof(T);
  if (p.M <= 0 || p.No % epv || p.Ki % epv || p.ldy % epv || p.ldx % epv) return RGBNM_EINVAL;
  const int TK = 128 / (int)sizeof(T);
  p.rtiles = cdiv(p.No, BM);
  const bool nb3 = (p.Ki % 192 == 0);
  p.ctiles = nb3 ? p.Ki / 192 : cdiv(p.Ki, 128);
  const int tiles = p.rtiles * p.ctiles;
  if constexpr (sizeof(T) == 2) {
    tn_check_abort();
    if (g_tn_defer && tn_groupable(p)) {
      // a grouped launch has one workgroup per output tile and at most 256 of them (rgbnm_launch_tn_pipe_group): what is queued
      // runs before a job that would take the queue past that, or that has another row count (the queue used to fail
      // with RGBNM_EINVAL for widths whose blocks have more than 256 / n tiles)
      const int jt = cdiv(p.No, 128) * (p.Ki / 192);
      if (g_tn_n && (g_tn_q[0].p.M != p.M || g_tn_tiles + jt > 256)) { const int rf = tn_flush(st); if (rf != RGBNM_OK) return rf; }
      g_tn_q[g_tn_n++] = TnPending{p, dW, db, perm_heads, accumulate};
      g_tn_tiles += jt;
      return g_tn_n == g_tn_cap ? tn_flush(st) : RGBNM_OK;
    }
    if (rgbnm_get_option("tn_pipe")) {
      int Sp = 0;
      const int rc = rgbnm_launch_tn_pipe(p.dY, p.ldy, p.X, p.ldx, p.part, db ? p.bpart : nullptr, p.M, p.No, p.Ki,
                                          &Sp, st, p.S);
      if (rc < 0) return rc;
      if (rc == 0) {
        return submit_tn_reduce(p, dW, db, Sp, perm_heads, accumulate, st);
      }
    }
  }
  // split the token axis so that ~512 workgroups exist; each split is a whole number of k-tiles
  int ktiles = cdiv(p.M, TK);
  int S = min(p.S, max(1, min(ktiles, cdiv(rgbnm_get_option("tn_wgs"), tiles))));
  int kt_per = cdiv(ktiles, S);
  S = cdiv(ktiles, kt_per);
  p.S = S;
  p.tok_per_split = kt_per * TK;
  bool tr = false;
  if constexpr (sizeof(T) == 2) tr = rgbnm_get_option("tn_tr") != 0;
  if constexpr (sizeof(T) == 2) {
    if (tr) {
      if (nb3) hipLaunchKernelGGL((gemm_tn_kernel<T, 3, true>), dim3(tiles * ((S + 7) / 8) * 8), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((gemm_tn_kernel<T, 2, true>), dim3(tiles * ((S + 7) / 8) * 8), dim3(256), 0, st, p);
    }
  }
  if (!tr) {
    if (nb3) hipLaunchKernelGGL((gemm_tn_kernel<T, 3, false>), dim3(tiles * ((S + 7) / 8) * 8), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<T, 2, false>), dim3(tiles * ((S + 7) / 8) * 8), dim3(256), 0, st, p);
  }
  LAUNCH_CHECK();
  return submit_tn_reduce(p, dW, db, S, perm_heads, accumulate, st);
}

// ------------------------------------------------------------------------------------------------
// prep_weights: per Linear, fp32 master W[N,K] -> shadow Ws[N,K] (T) and WsT[K,N] (T); rows of qkv are
// de-interleaved so that q|k|v come out as contiguous [heads*64] column blocks of the GEMM output.
// ------------------------------------------------------------------------------------------------
// 32 x 32 tiles through LDS so that BOTH shadows are written with coalesced rows (the transposed one was a stride-N
// scatter before: 57 us per step for JPEG-Ti, 460 us for SwinV2-T's 28 M parameters).
// One-launch encoder (vit_chain.hip / vit_chain_bwd.hip): where element (R, c) of a block Linear's [N, K] operand lies in the
// forward chain image, and element (kr, n) of its [K, N] transpose in the backward one -- the arithmetic inverse of the layout
// rgb-no-more_amd/chain.py builds as an index table (block_index / block_index_bwd; tests/test_chain_fwd.py compares the two).
// kind: 1 qkv (rows de-interleaved q | k | v), 2 projection, 3 fc1, 4 fc2;  E = 192, 3 heads.
__device__ __forceinline__ int ch_fswz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 1) | (((r >> 3) & 1) << 1); }
__device__ __forceinline__ int ch_swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
constexpr int CH_SLOT = 12288;
__device__ __forceinline__ int ch_pos384(int rl, int col) {        // [64 LDS rows][192]: 16-byte chunks swizzled inside groups of 8
  const int lc = col >> 3;
  return rl * 192 + (((lc & ~7) | ((lc & 7) ^ ch_fswz(rl))) << 3) + (col & 7);
}
__device__ __forceinline__ int ch_pos128(int rl, int kk) {         // [192 LDS rows][64]: chunk q at q ^ fswz(row)
  return rl * 64 + (((kk >> 3) ^ ch_fswz(rl)) << 3) + (kk & 7);
}
__device__ __forceinline__ int chain_fwd_pos(int kind, int R, int c) {
  switch (kind) {
    case 1: return (((R % 192) >> 6) * 3 + R / 192) * CH_SLOT + ch_pos384(ch_swap23(R & 63), c);
    case 2: return (9 + (c >> 6)) * CH_SLOT + ch_pos128(ch_swap23(R), ch_swap23(c & 63));
    case 3: return (12 + 2 * (R >> 6)) * CH_SLOT + ch_pos384(ch_swap23(R & 63), c);
    default: return (13 + 2 * (c >> 6)) * CH_SLOT + ch_pos128(ch_swap23(R), c & 63);
  }
}
__device__ __forceinline__ int chain_bwd_pos(int kind, int kr, int n) {
  switch (kind) {
    case 4: return (2 * (kr >> 6)) * CH_SLOT + ch_pos384(ch_swap23(kr & 63), n);
    case 3: return (2 * (n >> 6) + 1) * CH_SLOT + ch_pos128(kr, n & 63);
    case 2: return (24 + (kr >> 6)) * CH_SLOT + ch_pos384(ch_swap23(kr & 63), n);
    default: return (27 + (n >> 6)) * CH_SLOT + ch_pos128(kr, n & 63);
  }
}

// chain_fwd / chain_bwd (bf16 only, may be null): the chain images, written straight from the fp32 masters for every descriptor
// with chain_kind != 0 -- the per-block [N, K] / [K, N] shadows of those Linears are skipped when skip_chain_shadows is set (nothing
// reads them while the one-launch kernels run).  The de-interleaved qkv bias (gather_bias, a launch of its own before round 6) is
// written by the first three workgroups of a descriptor's row of the grid.
template <typename T>
__global__ __launch_bounds__(256) void prep_weights_kernel(const rgbnm_linear_desc* __restrict__ descs,
                                                           const float* __restrict__ master, T* __restrict__ shadow,
                                                           float* __restrict__ bias_out, T* __restrict__ chain_fwd,
                                                           T* __restrict__ chain_bwd, int skip_chain_shadows) {
  constexpr int TU = 2;        // tiles per turn, their loads in flight together (one tile per turn: four dependent load latencies
                               // and two barriers per tile -- 22.9 us for 45 MB)
  __shared__ float tile[TU][32][33];
  const rgbnm_linear_desc d = descs[blockIdx.y];
  if (bias_out && d.perm_heads > 0 && blockIdx.x < 3)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.N; i += 3 * 256)
      bias_out[d.bperm_off + i] = master[d.b_off + qkv_row(i, d.perm_heads)];
  if (bias_out && d.perm_heads <= 0 && d.bias_mode && blockIdx.x < 3) {     // prepared bias operands (SwinV2: see rgbnm_linear_desc)
    const int third = d.N / 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.N; i += 3 * 256) {
      float v;
      if (d.bias_mode == 1) v = master[d.b_off + i];
      else v = i < third ? master[d.b_off + i] : (i < 2 * third ? 0.f : master[d.b2_off + i - 2 * third]);
      bias_out[d.bperm_off + i] = v;
      if (d.pair) bias_out[d.bperm_off + d.N + i] = v;
    }
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
  const int tr = (d.N + 31) / 32, tc = (d.K + 31) / 32;
  const int kind = (chain_fwd || chain_bwd) ? d.chain_kind : 0;
  const bool shadows = !(kind && skip_chain_shadows);
  for (int t0 = blockIdx.x; t0 < tr * tc; t0 += TU * gridDim.x) {
    float vv[TU][4];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int t = t0 + u * gridDim.x;
      const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;             // r = shadow (GEMM) row
        float v = 0.f;
        if (t < tr * tc && r < d.N && c < d.K) {
          const int src = d.perm_heads > 0 ? qkv_row(r, d.perm_heads) : r;
          v = master[d.w_off + (size_t)src * d.K + c];
        }
        vv[u][k] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int t = t0 + u * gridDim.x;
      const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (t < tr * tc && r < d.N && c < d.K) {
          v = vv[u][k] + ((d.add_identity && r == c) ? 1.0f : 0.0f);
          if (!shadows) {
          } else if (d.pair) {                                   // diag(W, W): row pitch 2K, second copy at (N, K)
            shadow[d.ws_off + (size_t)r * (2 * d.K) + c] = from_f32<T>(v);
            shadow[d.ws_off + (size_t)(d.N + r) * (2 * d.K) + d.K + c] = from_f32<T>(v);
          } else {
            shadow[d.ws_off + (size_t)r * d.K + c] = from_f32<T>(v);
          }
        }
        tile[u][ty + 8 * k][tx] = v;
      }
    }
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
      // chain images (N, K multiples of 32): eight neighbours of the fast index lie together in both layouts (kind 2 forward:
      // four), so threads 0 - 127 store the tile's 128 row vectors, threads 128 - 255 its 128 column vectors -- one 16-byte store
      // per thread and image instead of four 2-byte ones (11 M scattered 2-byte stores per step were most of this kernel)
      if (kind) {
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const int t = t0 + u * gridDim.x;
          if (t >= tr * tc) break;
          const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
          const int id = threadIdx.x & 127, a = id >> 2, g8 = (id & 3) * 8;
          typedef T vec8 __attribute__((ext_vector_type(8)));
          typedef T vec4 __attribute__((ext_vector_type(4)));
          vec8 o;
          if (threadIdx.x < 128) {
            if (chain_fwd) {
#pragma unroll
              for (int i = 0; i < 8; ++i) o[i] = from_f32<T>(tile[u][a][g8 + i]);
              if (kind == 2) {
                *reinterpret_cast<vec4*>(chain_fwd + d.chain_off + chain_fwd_pos(kind, r0 + a, c0 + g8)) = vec4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<vec4*>(chain_fwd + d.chain_off + chain_fwd_pos(kind, r0 + a, c0 + g8 + 4)) = vec4{o[4], o[5], o[6], o[7]};
              } else {
                *reinterpret_cast<vec8*>(chain_fwd + d.chain_off + chain_fwd_pos(kind, r0 + a, c0 + g8)) = o;
              }
            }
          } else if (chain_bwd) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = from_f32<T>(tile[u][g8 + i][a]);
            *reinterpret_cast<vec8*>(chain_bwd + d.chain_off + chain_bwd_pos(kind, c0 + a, r0 + g8)) = o;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int t = t0 + u * gridDim.x;
      if (t >= tr * tc) break;
      const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < d.N && c < d.K) {
          if (!shadows) {
          } else if (d.pair) {
            shadow[d.wst_off + (size_t)c * (2 * d.N) + r] = from_f32<T>(tile[u][tx][ty + 8 * k]);
            shadow[d.wst_off + (size_t)(d.K + c) * (2 * d.N) + d.N + r] = from_f32<T>(tile[u][tx][ty + 8 * k]);
          } else {
            shadow[d.wst_off + (size_t)c * (d.ldn > 0 ? d.ldn : d.N) + r] = from_f32<T>(tile[u][tx][ty + 8 * k]);
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

// Brackets nest by JOINING: a begin inside an open bracket (rgbnm_head_bwd or rgbnm_vit_block_bwd called by somebody who has
// opened rgbnm_gemm_tn_group_begin_n) only counts, and its flush neither launches nor closes anything -- the jobs wait for the
// outer end (the inner flush used to close the caller's bracket silently).  A begin on a thread whose queue still holds
// jobs starts from an EMPTY queue: they are leftovers of a pass that died between begin and end, their operands may be gone.
void rgbnm_tn_defer_begin_n(int max_jobs) {
  tn_check_abort();
  if (g_tn_defer) { ++g_tn_depth; return; }
  g_tn_n = g_tn_tiles = g_tn_depth = 0;
  g_tn_id = 0;
  g_tn_defer = true;
  g_tn_cap = max_jobs < 1 ? 1 : (max_jobs > TN_QMAX ? TN_QMAX : max_jobs);
}
void rgbnm_tn_defer_begin() { rgbnm_tn_defer_begin_n(4); }
int rgbnm_tn_defer_flush(hipStream_t st) {
  tn_check_abort();
  if (g_tn_depth > 0) { --g_tn_depth; return RGBNM_OK; }
  g_tn_defer = false;
  g_tn_cap = 4;
  g_tn_id = 0;
  return tn_flush(st);
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int rgbnm_gemm_nt(int dtype, int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc,
                  const float* bias, const void* R, int ldr, void* C2, int ldc2, const float* pos, int pos_period,
                  int M, int N, int K, int c_f32, void* stream) {
  GemmNT p;
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.R = R; p.C2 = C2; p.pos = pos;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldc2 = ldc2; p.pos_period = pos_period > 0 ? pos_period : 1;
  p.M = M; p.N = N; p.K = K; p.c_f32 = c_f32; p.mtiles = p.ntiles = 0;
  if (!A || !W || !C) return RGBNM_EINVAL;
  if ((epi == EPI_RES || epi == EPI_DGELU || epi == EPI_DTANH) && !R) return RGBNM_EINVAL;
  if (epi == EPI_GELU && !C2) return RGBNM_EINVAL;
  if (epi == EPI_POS && !pos) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) return launch_nt<bf16>(p, epi, st);
  if (dtype == DT_F32) return launch_nt<float>(p, epi, st);
  return RGBNM_EINVAL;
}

size_t rgbnm_gemm_tn_workspace(int M, int No, int Ki) {
  // worst case split count
  return (size_t)RGBNM_TN_MAX_SPLIT * ((size_t)No * Ki + No) * sizeof(float);
}
size_t rgbnm_gemm_tn_workspace_splits(int No, int Ki, int splits) {
  if (splits < 1) splits = 1;
  if (splits > RGBNM_TN_MAX_SPLIT) splits = RGBNM_TN_MAX_SPLIT;
  return (size_t)splits * ((size_t)No * Ki + No) * sizeof(float);
}

int rgbnm_gemm_tn(int dtype, const void* dY, int ldy, const void* X, int ldx, float* dW, float* db, int M, int No,
                  int Ki, int perm_heads, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dY || !X || !dW || !workspace) return RGBNM_EINVAL;
  // the workspace holds smax split slices (part[s][No][Ki] then bpart[s][No]): the full rgbnm_gemm_tn_workspace gives 128, a caller
  // that knows better (rgbnm_gemm_tn_workspace_splits) brings fewer and the token axis is split no further than that
  if (No <= 0 || Ki <= 0) return RGBNM_EINVAL;
  size_t smax = workspace_bytes / (((size_t)No * Ki + No) * sizeof(float));
  if (smax < 1) return RGBNM_EWORKSPACE;
  if (smax > (size_t)RGBNM_TN_MAX_SPLIT) smax = RGBNM_TN_MAX_SPLIT;
  GemmTN p;
  p.dY = dY; p.X = X; p.ldy = ldy; p.ldx = ldx; p.M = M; p.No = No; p.Ki = Ki; p.S = (int)smax;
  p.part = reinterpret_cast<float*>(workspace);
  p.bpart = db ? p.part + smax * No * Ki : nullptr;
  p.tok_per_split = 0; p.rtiles = p.ctiles = 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) return launch_tn<bf16>(p, dW, db, perm_heads, accumulate, st);
  if (dtype == DT_F32) return launch_tn<float>(p, dW, db, perm_heads, accumulate, st);
  return RGBNM_EINVAL;
}

void rgbnm_gemm_tn_group_begin(void) { rgbnm_tn_defer_begin(); }
void rgbnm_gemm_tn_group_begin_n(int max_jobs) { rgbnm_tn_defer_begin_n(max_jobs); }
void rgbnm_gemm_tn_group_begin_id(int max_jobs, unsigned long long id) {
  const bool outer = !g_tn_defer;
  rgbnm_tn_defer_begin_n(max_jobs);
  if (outer) g_tn_id = id;
}
void rgbnm_gemm_tn_group_abort(unsigned long long id) {
  if (!id) return;
  if (id == g_tn_id) { tn_check_abort(); g_tn_n = g_tn_tiles = g_tn_depth = 0; g_tn_defer = false; g_tn_cap = 4; g_tn_id = 0; return; }   // own thread: now
  std::lock_guard<std::mutex> lk(g_tn_abort_mu);
  if (g_tn_aborted.size() > 1024) g_tn_aborted.erase(g_tn_aborted.begin());      // (ids nobody came back for)
  g_tn_aborted.push_back(id);
}

int rgbnm_gemm_tn_group_end(void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool own = !rgbnm_reduce_defer_active();        // the queued jobs' reductions: one launch
  if (own) rgbnm_reduce_defer_begin();
  int rc = rgbnm_tn_defer_flush(st);
  if (own) {
    const int rf = rgbnm_reduce_defer_flush(st);
    if (rc == RGBNM_OK) rc = rf;
  }
  return rc;
}

int rgbnm_prep_weights_chain(int dtype, const rgbnm_linear_desc* descs_dev, int ndesc, const float* master, void* shadow,
                             float* bias_perm, void* chain_fwd, void* chain_bwd, int skip_chain_shadows, void* stream) {
  if (!descs_dev || !master || !shadow || ndesc <= 0) return RGBNM_EINVAL;
  if ((chain_fwd || chain_bwd) && dtype != DT_BF16) return RGBNM_EINVAL;
  if (skip_chain_shadows && !chain_fwd && !chain_bwd) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((prep_weights_kernel<bf16>), dim3(64, ndesc), dim3(256), 0, st, descs_dev, master, (bf16*)shadow, bias_perm,
                       (bf16*)chain_fwd, (bf16*)chain_bwd, skip_chain_shadows);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL((prep_weights_kernel<float>), dim3(64, ndesc), dim3(256), 0, st, descs_dev, master, (float*)shadow, bias_perm,
                       (float*)nullptr, (float*)nullptr, 0);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_prep_weights(int dtype, const rgbnm_linear_desc* descs_dev, int ndesc, const float* master, void* shadow,
                       float* bias_perm, void* stream) {
  return rgbnm_prep_weights_chain(dtype, descs_dev, ndesc, master, shadow, bias_perm, nullptr, nullptr, 0, stream);
}

}  // extern "C"
