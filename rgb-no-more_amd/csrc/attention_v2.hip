// bf16 attention, second generation (reference: MultiHeadAttention.forward, models/plainvit.py:445-464).
// Same math and register orientation as attention.hip (S^T = K.Q^T, a lane owns one query / one key); what changes
// is the memory system:
//   * Q/K/V/dO tiles of one (image, head) arrive by LDS-DMA (global_load_lds, 16 B/lane) in their natural
//     [token][64] layout -- one bulk transfer per workgroup instead of per-wave fragment loads from L2;
//   * one XOR swizzle of the 16-byte chunk index, f(row) = rot3((row>>1)&7), makes BOTH access patterns bank-conflict
//     free: ds_read_b128 of 32 token rows (MFMA operands with the head dim as reduction axis) and
//     ds_read_b64_tr_b16 of 4 rows x 64 B (operands with tokens as reduction axis: V^T, K^T, Q^T, dO^T), so no
//     transposed copies are staged at all;
//   * backward is ONE kernel (phase A: wave = 32 queries -> dQ and D; phase B: wave = 32 keys -> dK, dV) sharing the
//     four LDS tiles.
#include "common.h"
#include <type_traits>
#include "internal.h"
#include "../../include/rgbnm.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;
constexpr int NTILE = 7;
constexpr int NPAD = NTILE * 32;          // 224 token rows per LDS array
constexpr int ROWB = 128;                 // bytes per token row
constexpr int ARR = NPAD * ROWB;          // 28 KB
constexpr int NTHREADS = NTILE * 64;
constexpr int NTHREADS3 = NTHREADS + 64;   // attn3_bwd_kernel: 7 compute waves + 1 DMA wave

__device__ __forceinline__ int fswz(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}

// LDS-DMA one [N][64] bf16 matrix (row stride ld elements): 28 instructions of 1 KB (8 rows); wave w issues
// w, w+7, w+14, w+21.  Rows >= N replicate row N-1 (finite data; their probabilities are masked to zero).
__device__ __forceinline__ void dma_matrix(const bf16* __restrict__ src, int ld, int N, unsigned char* dst, int w,
                                           int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = w + 7 * j;
    const int row = 8 * i + (lane >> 3), pc = lane & 7;
    const int lc = pc ^ fswz(row);
    const int srow = row < N ? row : N - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + (size_t)srow * ld + lc * 8), (lds_ptr)(dst + i * 1024), 16, 0, 0);
  }
}

// rowfrag with the per-lane part folded into one offset rb = l31 * ROWB + ((g ^ fswz(l31)) << 4): chunk c of the row is at
// rb ^ (c << 5) because (2c + g) ^ fl = 2c ^ (g ^ fl).  rb is re-derived (made opaque) at the start of each phase so the
// per-array, per-chunk addresses are not kept in registers -- or spilled -- across the whole persistent loop.
__device__ __forceinline__ Frag<bf16> rowfrag_x(const unsigned char* arr, unsigned rb, int t, int c) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(arr + (rb ^ (unsigned)(c << 5)) + t * 32 * ROWB);
  return f;
}

// MFMA groups of one 32-row tile with their row fragments in a PINNED order (sched_barrier after every step).  Left alone the
// compiler emits read - wait - MFMA with ONE fragment buffer: an exposed LDS latency per MFMA (36 of attn3_fwd's 80 MFMAs, 50 of
// attn3_bwd's 196).  row_mma4: one accumulator, the four fragments requested together; row_pair_mma: two accumulators (scores and
// their gradient), four fragments ahead, every MFMA followed by the read of the fragment two steps on.  Same MFMA order per
// accumulator: same bits.  -DAV2_ROWPIPE=0: the plain loops (experiments).
#ifndef AV2_ROWPIPE
#define AV2_ROWPIPE 1
#endif
__device__ __forceinline__ void row_mma4(f32x16& acc, const unsigned char* arr, unsigned rb, int t, const Frag<bf16> (&x)[4]) {
#if AV2_ROWPIPE
  Frag<bf16> f[4];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) f[c] = rowfrag_x(arr, rb, t, c);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) mma(acc, f[c], x[c]);
  __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
  for (int c = 0; c < 4; ++c) mma(acc, rowfrag_x(arr, rb, t, c), x[c]);
#endif
}
__device__ __forceinline__ void row_pair_mma(f32x16& sa, f32x16& da, const unsigned char* arrS, const unsigned char* arrD, unsigned rb,
                                             int t, const Frag<bf16> (&xs)[4], const Frag<bf16> (&xd)[4]) {
#if AV2_ROWPIPE
#define AV2_SB __builtin_amdgcn_sched_barrier(0)
  Frag<bf16> fs[4], fd[4];
  AV2_SB;
  fs[0] = rowfrag_x(arrS, rb, t, 0);
  fd[0] = rowfrag_x(arrD, rb, t, 0);
  fs[1] = rowfrag_x(arrS, rb, t, 1);
  fd[1] = rowfrag_x(arrD, rb, t, 1);
  AV2_SB;
  mma(sa, fs[0], xs[0]); AV2_SB;
  fs[2] = rowfrag_x(arrS, rb, t, 2); AV2_SB;
  mma(da, fd[0], xd[0]); AV2_SB;
  fd[2] = rowfrag_x(arrD, rb, t, 2); AV2_SB;
  mma(sa, fs[1], xs[1]); AV2_SB;
  fs[3] = rowfrag_x(arrS, rb, t, 3); AV2_SB;
  mma(da, fd[1], xd[1]); AV2_SB;
  fd[3] = rowfrag_x(arrD, rb, t, 3); AV2_SB;
  mma(sa, fs[2], xs[2]);
  mma(da, fd[2], xd[2]);
  mma(sa, fs[3], xs[3]);
  mma(da, fd[3], xd[3]);
  AV2_SB;
#undef AV2_SB
#else
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mma(sa, rowfrag_x(arrS, rb, t, c), xs[c]);
    mma(da, rowfrag_x(arrD, rb, t, c), xd[c]);
  }
#endif
}

// The same 28 instructions issued by ONE wave (the DMA wave of attn3_bwd_kernel).
__device__ __forceinline__ void dma_matrix_all(const bf16* __restrict__ src, int ld, int N, unsigned char* dst, int lane) {
#pragma unroll 4
  for (int i = 0; i < 28; ++i) {
    const int row = 8 * i + (lane >> 3), pc = lane & 7;
    const int lc = pc ^ fswz(row);
    const int srow = row < N ? row : N - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + (size_t)srow * ld + lc * 8), (lds_ptr)(dst + i * 1024), 16, 0, 0);
  }
}

// MFMA operand with the head dim as reduction axis: lane <-> token row (32t + l31), 32-byte chunk c, half g.
__device__ __forceinline__ Frag<bf16> rowfrag(const unsigned char* arr, int t, int l31, int c, int g, int fl) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(arr + (32 * t + l31) * ROWB + (((2 * c + g) ^ fl) << 4));
  return f;
}

__device__ __forceinline__ bf16x8 pack8(u32x2 lo, u32x2 hi) {
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

// Transposed operands (tokens as reduction axis) of tile T, fragment FI, for both 32-wide d tiles:
// a0 = per-lane LDS address for (dt=0, rd=0); dt=1 flips address bit 6, rd=1 flips bit 5 and adds 8 rows.
template <int T, int FI>
__device__ __forceinline__ void tfrag2(unsigned a0, Frag<bf16>& f0, Frag<bf16>& f1) {
  u32x2 x0, x1, y0, y1;
  const unsigned a00 = a0, a01 = (a0 ^ 32u) + 1024u, a10 = a0 ^ 64u, a11 = (a0 ^ 96u) + 1024u;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4 offset:%8\n\t"
      "ds_read_b64_tr_b16 %1, %5 offset:%8\n\t"
      "ds_read_b64_tr_b16 %2, %6 offset:%8\n\t"
      "ds_read_b64_tr_b16 %3, %7 offset:%8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1)
      : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "i"(T * 4096 + FI * 2048)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
  f0.v = pack8(x0, x1);
  f1.v = pack8(y0, y1);
}

// Both fragments (FI = 0, 1) x both d tiles of tile T from ONE array: 8 reads, one wait.
template <int T>
__device__ __forceinline__ void tfrag4(unsigned a0, Frag<bf16> (&f)[4]) {
  u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
  const unsigned a00 = a0, a01 = (a0 ^ 32u) + 1024u, a10 = a0 ^ 64u, a11 = (a0 ^ 96u) + 1024u;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %2, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "i"(T * 4096), "i"(T * 4096 + 2048)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
  f[0].v = pack8(r0, r1);   // fi=0, dt=0
  f[1].v = pack8(r2, r3);   // fi=0, dt=1
  f[2].v = pack8(r4, r5);   // fi=1, dt=0
  f[3].v = pack8(r6, r7);   // fi=1, dt=1
}

// The same for TWO arrays (phase B of backward: dO^T and Q^T): 16 reads, one wait.
template <int T>
__device__ __forceinline__ void tfrag8(unsigned a0, unsigned b0, Frag<bf16> (&fa)[4], Frag<bf16> (&fb)[4]) {
  u32x2 r0, r1, r2, r3, r4, r5, r6, r7, q0, q1, q2, q3, q4, q5, q6, q7;
  const unsigned a00 = a0, a01 = (a0 ^ 32u) + 1024u, a10 = a0 ^ 64u, a11 = (a0 ^ 96u) + 1024u;
  const unsigned b00 = b0, b01 = (b0 ^ 32u) + 1024u, b10 = b0 ^ 64u, b11 = (b0 ^ 96u) + 1024u;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%24\n\t"
      "ds_read_b64_tr_b16 %1, %17 offset:%24\n\t"
      "ds_read_b64_tr_b16 %2, %18 offset:%24\n\t"
      "ds_read_b64_tr_b16 %3, %19 offset:%24\n\t"
      "ds_read_b64_tr_b16 %4, %16 offset:%25\n\t"
      "ds_read_b64_tr_b16 %5, %17 offset:%25\n\t"
      "ds_read_b64_tr_b16 %6, %18 offset:%25\n\t"
      "ds_read_b64_tr_b16 %7, %19 offset:%25\n\t"
      "ds_read_b64_tr_b16 %8, %20 offset:%24\n\t"
      "ds_read_b64_tr_b16 %9, %21 offset:%24\n\t"
      "ds_read_b64_tr_b16 %10, %22 offset:%24\n\t"
      "ds_read_b64_tr_b16 %11, %23 offset:%24\n\t"
      "ds_read_b64_tr_b16 %12, %20 offset:%25\n\t"
      "ds_read_b64_tr_b16 %13, %21 offset:%25\n\t"
      "ds_read_b64_tr_b16 %14, %22 offset:%25\n\t"
      "ds_read_b64_tr_b16 %15, %23 offset:%25\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(q0), "=&v"(q1),
        "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
      : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(b00), "v"(b01), "v"(b10), "v"(b11), "i"(T * 4096),
        "i"(T * 4096 + 2048)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
  fa[0].v = pack8(r0, r1); fa[1].v = pack8(r2, r3); fa[2].v = pack8(r4, r5); fa[3].v = pack8(r6, r7);
  fb[0].v = pack8(q0, q1); fb[1].v = pack8(q2, q3); fb[2].v = pack8(q4, q5); fb[3].v = pack8(q6, q7);
}

__device__ __forceinline__ Frag<bf16> pfrag(const float (&p)[16], int fi) {
  Frag<bf16> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (bf16)p[fi * 8 + j];
  return f;
}

struct LaneGeo {
  int lane, l31, g, fl;
  unsigned tr0;    // byte offset inside an array of the (t=0, fi=0, dt=0, rd=0) transpose read
};
__device__ __forceinline__ LaneGeo lane_geo() {
  LaneGeo L;
  L.lane = threadIdx.x & 63;
  L.l31 = L.lane & 31;
  L.g = L.lane >> 5;
  L.fl = fswz(L.l31);
  const int k = (L.lane >> 2) & 3, G1 = (L.lane >> 4) & 1, l3 = L.lane & 3;
  const int pc = (2 * G1 + (l3 >> 1)) ^ (((k >> 1) << 2) | L.g);
  L.tr0 = (unsigned)((4 * L.g + k) * ROWB + pc * 16 + 8 * (l3 & 1));
  return L;
}

template <int T> struct TileLoop {
  template <typename F> static __device__ __forceinline__ void run(F&& f) {
    TileLoop<T - 1>::run(f);
    f(std::integral_constant<int, T - 1>{});
  }
};
template <> struct TileLoop<0> {
  template <typename F> static __device__ __forceinline__ void run(F&&) {}
};

// ------------------------------------------------------------------------------------------------ forward
template <int NTOK, int T> __device__ __forceinline__ bool tile_on(int N) {
  if constexpr (NTOK > 0) return T * 32 < NTOK;
  else return T * 32 < N;
}
template <int NTOK, int T> __device__ __forceinline__ bool tile_ragged(int N) {
  if constexpr (NTOK > 0) return T * 32 + 32 > NTOK;
  else return T * 32 + 32 > N;
}

// NTOK > 0: compile-time token count (196): only the ragged last key tile carries a mask.  The exponent is one FMA +
// exp2 (log2 e folded into the scale), and 1/sum multiplies the 32 output values instead of the 112 probabilities.
template <int NTOK>
__global__ __launch_bounds__(NTHREADS) void attn2_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                             float* __restrict__ lse, int N_rt, int heads, float scale) {
  const int N = NTOK > 0 ? NTOK : N_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + ARR;
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int inner = heads * HD, ld = 3 * inner;
  const bf16* Q = qkv + (size_t)b * N * ld + h * HD;
  const bf16* K = Q + inner;
  const bf16* V = Q + 2 * inner;
  const LaneGeo L = lane_geo();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  dma_matrix(K, ld, N, Ks, w, L.lane);
  dma_matrix(V, ld, N, Vs, w, L.lane);

  const int q = w * 32 + L.l31;
  const int qc = q < N ? q : N - 1;
  Frag<bf16> qf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qf[c] = load_frag<bf16>(Q + (size_t)qc * ld + c * 16 + L.g * 8);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (w * 32 >= N) return;

  // Two passes over the key tiles, S recomputed in the second one (4 extra MFMAs per tile, the MFMA pipe is mostly idle
  // here): keeping the 7 x 16 scores of a lane alive between the passes cost 112 registers and with them the second
  // workgroup per CU (155 VGPRs -> 3 waves per SIMD -> one 7-wave workgroup); now 2 workgroups overlap DMA and compute.
  float m = -INFINITY;
  TileLoop<NTILE>::run([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (tile_on<NTOK, t>(N)) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) mma(acc, rowfrag(Ks, t, L.l31, c, L.g, L.fl), qf[c]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (tile_ragged<NTOK, t>(N) && t * 32 + acc_row(r, L.lane) >= N) v = -INFINITY;
        m = fmaxf(m, v);
      }
    }
  });
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float c2 = scale * 1.4426950408889634f, mc2 = m * c2;
  float sum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  const unsigned vt = (unsigned)(size_t)Vs + L.tr0;
  TileLoop<NTILE>::run([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (tile_on<NTOK, t>(N)) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) mma(acc, rowfrag(Ks, t, L.l31, c, L.g, L.fl), qf[c]);
      float pr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -mc2));
        if (tile_ragged<NTOK, t>(N) && t * 32 + acc_row(r, L.lane) >= N) pr[r] = 0.f;
        sum += pr[r];
      }
      Frag<bf16> vv[4];
      tfrag4<t>(vt, vv);
      Frag<bf16> pf = pfrag(pr, 0);
      mma(o[0], vv[0], pf);
      mma(o[1], vv[1], pf);
      pf = pfrag(pr, 1);
      mma(o[0], vv[2], pf);
      mma(o[1], vv[3], pf);
    }
  });
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  if (L.g == 0 && q < N) lse[(size_t)bh * N + q] = m * scale + __logf(sum);
  if (q < N) {
    bf16* orow = out + ((size_t)b * N + q) * inner + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {o[dt][rq * 4 + 0], o[dt][rq * 4 + 1], o[dt][rq * 4 + 2], o[dt][rq * 4 + 3]};
        store4<bf16>(orow + dt * 32 + rq * 8 + L.g * 4, v * inv);
      }
  }
}

// ----------------------------------------------------------------------------------------------- backward
#ifdef ATTN_PROF
__device__ unsigned long long g_attn_prof[768 * 8 * 8];
#define APROF(i) do { if (L.lane == 0) g_attn_prof[(blockIdx.x * 8 + w) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define APROF(i) do {} while (0)
#endif
__global__ __launch_bounds__(NTHREADS) void attn2_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                             const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                             int N, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + ARR;
  unsigned char* Vs = smem + 2 * ARR;
  unsigned char* Gs = smem + 3 * ARR;                       // dO
  float* lse_s = reinterpret_cast<float*>(smem + 4 * ARR);  // [NPAD]
  float* D_s = lse_s + NPAD;                                // [NPAD]
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int inner = heads * HD, ld = 3 * inner;
  const bf16* Q = qkv + (size_t)b * N * ld + h * HD;
  const bf16* K = Q + inner;
  const bf16* V = Q + 2 * inner;
  const bf16* O = out + (size_t)b * N * inner + h * HD;
  const bf16* dO = dout + (size_t)b * N * inner + h * HD;
  const LaneGeo L = lane_geo();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  APROF(0);
  dma_matrix(Q, ld, N, Qs, w, L.lane);
  dma_matrix(K, ld, N, Ks, w, L.lane);
  dma_matrix(V, ld, N, Vs, w, L.lane);
  dma_matrix(dO, inner, N, Gs, w, L.lane);
  for (int i = threadIdx.x; i < NPAD; i += NTHREADS) lse_s[i] = i < N ? lse[(size_t)bh * N + i] : 0.f;

  const int row = w * 32 + L.l31;            // this lane's query (phase A) / key (phase B)
  const int rc = row < N ? row : N - 1;
  // O fragment of this lane's query for D = rowsum(dO * O): issued before the wait so it overlaps the DMA
  Frag<bf16> of[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) of[c] = load_frag<bf16>(O + (size_t)rc * inner + c * 16 + L.g * 8);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  APROF(1);
  const bool active = w * 32 < N;

  // ================= phase A: wave = 32 queries -> dQ, D =================
  if (active) {
    Frag<bf16> qf[4], gf[4];
    float Dq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qf[c] = rowfrag(Qs, w, L.l31, c, L.g, L.fl);
      gf[c] = rowfrag(Gs, w, L.l31, c, L.g, L.fl);
#pragma unroll
      for (int e = 0; e < 8; ++e) Dq += (float)gf[c].v[e] * (float)of[c].v[e];
    }
    Dq += __shfl_xor(Dq, 32, 64);
    if (L.g == 0) D_s[row] = Dq;
    const float lq = lse_s[row];

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const unsigned kt = (unsigned)(size_t)Ks + L.tr0;
    TileLoop<NTILE>::run([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (t * 32 < N) {
        f32x16 sa, da;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          mma(sa, rowfrag(Ks, t, L.l31, c, L.g, L.fl), qf[c]);
          mma(da, rowfrag(Vs, t, L.l31, c, L.g, L.fl), gf[c]);
        }
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = t * 32 + acc_row(r, L.lane);
          const float p = kk < N ? __expf(sa[r] * scale - lq) : 0.f;
          ds[r] = p * (da[r] - Dq) * scale;
        }
        Frag<bf16> kk[4];
        tfrag4<t>(kt, kk);
        Frag<bf16> sf = pfrag(ds, 0);
        mma(dq[0], kk[0], sf);
        mma(dq[1], kk[1], sf);
        sf = pfrag(ds, 1);
        mma(dq[0], kk[2], sf);
        mma(dq[1], kk[3], sf);
      }
    });
    APROF(2);
    if (row < N) {
      bf16* drow = dqkv + ((size_t)b * N + row) * ld + h * HD;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 v = {dq[dt][rq * 4 + 0], dq[dt][rq * 4 + 1], dq[dt][rq * 4 + 2], dq[dt][rq * 4 + 3]};
          store4<bf16>(drow + dt * 32 + rq * 8 + L.g * 4, v);
        }
    }
  }
  APROF(3);
  __syncthreads();   // D_s complete
  APROF(4);
  if (!active) return;

  // ================= phase B: wave = 32 keys -> dK, dV =================
  Frag<bf16> kf[4], vf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    kf[c] = rowfrag(Ks, w, L.l31, c, L.g, L.fl);
    vf[c] = rowfrag(Vs, w, L.l31, c, L.g, L.fl);
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const unsigned qt_ = (unsigned)(size_t)Qs + L.tr0, gt_ = (unsigned)(size_t)Gs + L.tr0;
  TileLoop<NTILE>::run([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (t * 32 < N) {
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        mma(sa, rowfrag(Qs, t, L.l31, c, L.g, L.fl), kf[c]);   // rows = queries, cols = keys
        mma(da, rowfrag(Gs, t, L.l31, c, L.g, L.fl), vf[c]);
      }
      float pp[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = t * 32 + acc_row(r, L.lane);
        const float p = qq < N ? __expf(sa[r] * scale - lse_s[qq]) : 0.f;
        pp[r] = p;
        ds[r] = p * (da[r] - D_s[qq]) * scale;
      }
      Frag<bf16> gg[4], qq4[4];
      tfrag8<t>(gt_, qt_, gg, qq4);
      Frag<bf16> f = pfrag(pp, 0);
      mma(dv[0], gg[0], f);
      mma(dv[1], gg[1], f);
      f = pfrag(pp, 1);
      mma(dv[0], gg[2], f);
      mma(dv[1], gg[3], f);
      f = pfrag(ds, 0);
      mma(dk[0], qq4[0], f);
      mma(dk[1], qq4[1], f);
      f = pfrag(ds, 1);
      mma(dk[0], qq4[2], f);
      mma(dk[1], qq4[3], f);
    }
  });
  APROF(5);
  if (row < N) {
    bf16* krow = dqkv + ((size_t)b * N + row) * ld + inner + h * HD;
    bf16* vrow = krow + inner;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 a = {dk[dt][rq * 4 + 0], dk[dt][rq * 4 + 1], dk[dt][rq * 4 + 2], dk[dt][rq * 4 + 3]};
        f32x4 c = {dv[dt][rq * 4 + 0], dv[dt][rq * 4 + 1], dv[dt][rq * 4 + 2], dv[dt][rq * 4 + 3]};
        store4<bf16>(krow + dt * 32 + rq * 8 + L.g * 4, a);
        store4<bf16>(vrow + dt * 32 + rq * 8 + L.g * 4, c);
      }
  }
#ifdef ATTN_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  APROF(6);
#endif
}

// ------------------------------------------------------------------------ backward, persistent (third generation)
// Same math and LDS layout as attn2_bwd_kernel; what changes is the schedule.  Measured on attn2_bwd (B = 256, cycle
// stamps): 23 % of a workgroup's life is the initial DMA wait and 8 % the store drain, with one workgroup per CU
// (116 KB of LDS) nothing hides either.  Here each of (at most) 256 workgroups walks several (image, head) pairs and
// the two LDS array pairs are refilled a full phase ahead:
//     phase A(h) reads K,V (all rows)  + own Q,dO,O rows (coalesced register loads issued after phase B(h-1), turned
//                into fragments through the wave's private LDS tile)
//     -- barrier -- K,V(h+1) DMA starts
//     phase B(h) reads Q,dO (all rows) + own K,V rows (registers, taken from LDS before the barrier)
//     -- barrier -- Q,dO(h+1) DMA starts
// 7 compute waves + 1 DMA wave.  Cycle stamps of the first persistent version showed 40 % of a pair's time going to the
// ISSUE of vmem instructions -- 24 scattered stores, 13 fragment-shaped loads and 16 LDS-DMAs per wave, 700-1000 cycles
// each, because every CU reaches its burst at the same moment and the fragment loads thrashed L1.  Now the DMA wave
// issues every LDS-DMA and every global store (it can stall for thousands of cycles without holding anybody up) and the
// compute waves touch global memory only through 13 coalesced loads per pair: 69 -> 57 us per launch at B = 256.
// The exponent is one FMA + exp2 (log2 e folded into scale and lse), `scale` multiplies dQ/dK once at the end, lse/D
// rows are read as float4, and the key/query mask is applied on the ragged last tile only.

// NTOK > 0: the token count is a compile-time constant (196 for JPEG-Ti/S): every tile test folds away and only the
// ragged last tile carries a mask.  (With a runtime N the compiler kept 14 tile predicates and 100+ lane masks alive in
// SGPRs and spilled them through v_writelane / v_readlane: 650 of the kernel's 2900 VALU instructions.)
// Gradient rows leave the persistent backward as full 128-byte lines, and not from the compute waves.  In the swapped
// orientation a lane owns one token and 4 consecutive head-dim values per accumulator quad, so direct stores are 8 bytes
// per lane at a 1152-byte stride; worse, every CU reaches its store / load burst at the same time and a vmem instruction
// then takes ~700-1000 cycles to ISSUE (cycle stamps: 40 % of a pair's time went to issuing 24 stores, 13 loads and 16
// LDS-DMAs per wave).  So the compute waves only park their 32 x 64 bf16 tiles in LDS (8-byte writes, conflict free):
//   dQ -> the wave's own rows of Ks (free after the mid barrier),  dK -> a private 4.5 KB tile,  dV -> own rows of Gs
// and the DMA wave reads them back as 16 B per lane, 8 lanes per row, and issues every global store (whole lines).
constexpr int STG_PITCH = 144;
constexpr int STG_WAVE = 32 * STG_PITCH;       // 4.5 KB per wave
__device__ __forceinline__ void tile_park_private(unsigned char* stg, const f32x16 (&acc)[2], float mul, const LaneGeo& L) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 v = {acc[dt][rq * 4 + 0], acc[dt][rq * 4 + 1], acc[dt][rq * 4 + 2], acc[dt][rq * 4 + 3]};
      store4<bf16>(reinterpret_cast<bf16*>(stg + L.l31 * STG_PITCH) + dt * 32 + rq * 8 + L.g * 4, v * mul);
    }
}
// rows w*32 .. w*32+31 of a 128-byte-pitch array; 16-byte chunk c of row r sits at chunk c ^ (r & 7)
__device__ __forceinline__ void tile_park_rows(unsigned char* arr, int w, const f32x16 (&acc)[2], float mul, const LaneGeo& L) {
  // one opaque base offset, chunk selected by XOR with a constant: the 8 swizzled addresses are loop invariants that the
  // compiler otherwise hoists to the kernel prologue and then SPILLS (each reload a serialised scratch round trip)
  const int ln = lane_id_here();   // (and off0 itself is recomputed here, not kept live across the pair)
  const unsigned off0 = (unsigned)((w * 32 + (ln & 31)) * ROWB + ((ln & 7) << 4) + (ln >> 5) * 8);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 v = {acc[dt][rq * 4 + 0], acc[dt][rq * 4 + 1], acc[dt][rq * 4 + 2], acc[dt][rq * 4 + 3]};
      store4<bf16>(reinterpret_cast<bf16*>(arr + (off0 ^ (unsigned)((dt * 4 + rq) << 4))), v * mul);
    }
}
// DMA wave: the parked tiles of all compute waves -> registers -> global rows (column block of 64 at g0, row stride ld)
template <bool PRIVATE>
__device__ __forceinline__ void tiles_read(const unsigned char* src, int N, int lane, u32x4 (&v)[NTILE][4]) {
  const int rl = lane >> 3, seg = lane & 7;
#pragma unroll
  for (int wv = 0; wv < NTILE; ++wv) {
    if (wv * 32 < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 8 + rl;
        v[wv][i] = PRIVATE ? *reinterpret_cast<const u32x4*>(src + wv * STG_WAVE + r * STG_PITCH + seg * 16)
                           : *reinterpret_cast<const u32x4*>(src + (wv * 32 + r) * ROWB + ((seg ^ (r & 7)) << 4));
      }
    }
  }
}
__device__ __forceinline__ void tiles_write(const u32x4 (&v)[NTILE][4], bf16* __restrict__ g0, size_t ld, int N, int lane) {
  const int rl = lane >> 3, seg = lane & 7;
#pragma unroll
  for (int wv = 0; wv < NTILE; ++wv)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wv * 32 + i * 8 + rl;
      if (row < N) *reinterpret_cast<u32x4*>(g0 + (size_t)row * ld + seg * 8) = v[wv][i];
    }
}
template <bool PRIVATE>
__device__ __forceinline__ void tiles_to_global(const unsigned char* src, bf16* __restrict__ g0, size_t ld, int N, int lane) {
  u32x4 v[NTILE][4];
  tiles_read<PRIVATE>(src, N, lane, v);
  tiles_write(v, g0, ld, N, lane);
}

template <int NTOK>
__global__ __launch_bounds__(NTHREADS3) void attn3_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                             const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                             int N_rt, int heads, int nbh, float scale) {
  const int N = NTOK > 0 ? NTOK : N_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + ARR;
  unsigned char* Vs = smem + 2 * ARR;
  unsigned char* Gs = smem + 3 * ARR;                       // dO
  float* L2_s = reinterpret_cast<float*>(smem + 4 * ARR);   // [NPAD] lse * log2(e)
  float* D_s = L2_s + NPAD;                                 // [NPAD] rowsum(dO * O)
  const int inner = heads * HD, ld = 3 * inner;
  const LaneGeo L = lane_geo();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* stg = smem + 4 * ARR + 2 * NPAD * (int)sizeof(float) + w * STG_WAVE;   // this wave's store tile
  const unsigned rb0 = (unsigned)(L.l31 * ROWB + ((L.g ^ L.fl) << 4));
  const int row = w * 32 + L.l31;            // this lane's query (phase A) / key (phase B)
  const int rc = row < N ? row : N - 1;
  const bool active = w * 32 < N;
  const float LOG2E = 1.4426950408889634f;
  const float c2 = scale * LOG2E;

  auto issue_kv = [&](int bh) {             // DMA wave only: 56 instructions
    const bf16* Q = qkv + (size_t)(bh / heads) * N * ld + (bh % heads) * HD;
    dma_matrix_all(Q + inner, ld, N, Ks, L.lane);
    dma_matrix_all(Q + 2 * inner, ld, N, Vs, L.lane);
  };
  auto issue_qg = [&](int bh) {
    const bf16* Q = qkv + (size_t)(bh / heads) * N * ld + (bh % heads) * HD;
    dma_matrix_all(Q, ld, N, Qs, L.lane);
    dma_matrix_all(dout + (size_t)(bh / heads) * N * inner + (bh % heads) * HD, inner, N, Gs, L.lane);
  };
  // Own rows (phase A operands Q, dO and O for D): 12 COALESCED loads per lane -- lane = (row i*8 + lane/8, 16-byte
  // segment lane%8), 8 whole lines per instruction -- parked in the private LDS tile at the top of the pair and read back
  // as MFMA fragments.  (Fragment-shaped loads touched 32 lines per instruction, 4 instructions per line: the 84 KB of own
  // rows per pair thrashed the 32 KB L1 and pulled up to 4x that from L2 while every store and DMA queued behind them.)
  u32x4 qraw[4], graw[4], oraw[4];
  float lq = 0.f;
  auto load_own_qg = [&](int bh) {           // 8 loads per lane: Q and dO rows
    const int b = bh / heads, h = bh % heads;
    const int lane_ = lane_id_here();        // the row addresses are recomputed here, not hoisted and spilled
    const int seg = lane_ & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = w * 32 + i * 8 + (lane_ >> 3);
      r = r < N ? r : N - 1;
      qraw[i] = *reinterpret_cast<const u32x4*>(qkv + ((size_t)b * N + r) * ld + h * HD + seg * 8);
      graw[i] = *reinterpret_cast<const u32x4*>(dout + ((size_t)b * N + r) * inner + h * HD + seg * 8);
    }
  };
  auto load_own_o = [&](int bh) {            // 5 loads per lane: O rows and the row's log-sum-exp
    const int b = bh / heads, h = bh % heads;
    const int lane_ = lane_id_here();
    const int seg = lane_ & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = w * 32 + i * 8 + (lane_ >> 3);
      r = r < N ? r : N - 1;
      oraw[i] = *reinterpret_cast<const u32x4*>(out + ((size_t)b * N + r) * inner + h * HD + seg * 8);
    }
    int rq_ = w * 32 + (lane_ & 31);
    rq_ = rq_ < N ? rq_ : N - 1;
    lq = lse[(size_t)bh * N + rq_];
  };
  auto load_own = [&](int bh) {
    load_own_qg(bh);
    load_own_o(bh);
  };

  int bh = blockIdx.x;
  if (bh >= nbh) return;
#ifdef ATTN_PROF
  int pit = 0;
#define APROF3(i) do { if (L.lane == 0 && pit == 1) g_attn_prof[(blockIdx.x * 8 + w) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define APROF3(i) do {} while (0)
#endif
  if (w == NTILE) {
    // ---- DMA wave: every LDS-DMA of the workgroup.  Issuing these instructions stalls for thousands of cycles when all
    // CUs hit their load / store bursts together (cycle stamps: 8 DMA instructions per compute wave took 6.3 k cycles to
    // issue); a wave with nothing else to do absorbs that stall while the compute waves go straight into the next phase.
    // Four barriers per pair; what each one tells the other side:
    //   mid : compute waves are done with K,V; dQ tiles are parked (private tiles) | Q,dO of this pair have landed
    //   mid2:                                                 | the dQ tiles are in registers, the private tiles are free
    //   end : compute waves are done with Q,dO, D_s, L2_s; dK tiles are parked | K,V of the next pair have landed
    //   end2: dV tiles are parked in the Gs rows              | the dK tiles are in registers, the private tiles are free
    const unsigned char* stg0 = smem + 4 * ARR + 2 * NPAD * (int)sizeof(float);
    issue_kv(bh);
    issue_qg(bh);
    asm volatile("s_waitcnt vmcnt(56)" ::: "memory");     // K,V landed: only the younger Q,dO group may be in flight
    __builtin_amdgcn_s_barrier();                          // start
    for (; bh < nbh; bh += gridDim.x) {
      const int nxt = bh + gridDim.x;
      bf16* g0 = dqkv + (size_t)(bh / heads) * N * ld + (bh % heads) * HD;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // Q,dO landed (and the previous pair's stores are done)
      __builtin_amdgcn_s_barrier();                        // mid
      u32x4 tq[NTILE][4];
      tiles_read<true>(stg0, N, L.lane, tq);               // dQ tiles -> registers
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // mid2
      tiles_write(tq, g0, ld, N, L.lane);
      if (nxt < nbh) issue_kv(nxt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // K,V of the next pair landed (during phase B)
      __builtin_amdgcn_s_barrier();                        // end
      tiles_read<true>(stg0, N, L.lane, tq);               // dK tiles -> registers (the private tiles are free again)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // end2
      tiles_write(tq, g0 + inner, ld, N, L.lane);
      tiles_read<false>(Gs, N, L.lane, tq);                // dV
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (nxt < nbh) issue_qg(nxt);                        // Q,dO of the next pair first: phase B needs them
      tiles_write(tq, g0 + 2 * inner, ld, N, L.lane);
    }
    return;
  }
  load_own(bh);
  __builtin_amdgcn_s_barrier();                            // start: K,V of the first pair are in LDS
  for (; bh < nbh; bh += gridDim.x) {
    const int b = bh / heads, h = bh % heads;
    const int nxt = bh + gridDim.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own rows (and the previous pair's stores)
    APROF3(0);
    APROF3(1);

    // ================= phase A: wave = 32 queries -> dQ, D =================
    Frag<bf16> kf[4], vf[4];
    if (active) {
      f32x16 dq[2];
      Frag<bf16> qf[4], gf[4];
      const int rl = L.lane >> 3, seg = L.lane & 7;
      // D = rowsum(dO * O) straight from the row-major pieces: 8 products per lane, 8 lanes per row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 gv = __builtin_bit_cast(bf16x8, graw[i]), ov = __builtin_bit_cast(bf16x8, oraw[i]);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)gv[e] * (float)ov[e];
        d += lane_xor1(d);
        d += lane_xor2(d);
        d += lane_xor4(d);
        if (seg == 0) D_s[w * 32 + i * 8 + rl] = d;
      }
      // Q, dO: row-major pieces -> private tile -> fragments (lane <-> row l31, 16-byte chunk 2c + g)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (i * 8 + rl) * STG_PITCH + seg * 16) = qraw[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c)
        qf[c].v = *reinterpret_cast<const bf16x8*>(stg + L.l31 * STG_PITCH + (2 * c + L.g) * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (i * 8 + rl) * STG_PITCH + seg * 16) = graw[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c)
        gf[c].v = *reinterpret_cast<const bf16x8*>(stg + L.l31 * STG_PITCH + (2 * c + L.g) * 16);
      const float Dq = D_s[row];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float lq2 = lq * LOG2E;
      if (L.g == 0) L2_s[row] = lq2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
      unsigned rb = rb0, tr = L.tr0;
      asm volatile("" : "+v"(rb), "+v"(tr));
      const unsigned kt = (unsigned)(size_t)Ks + tr;
      TileLoop<NTILE>::run([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (tile_on<NTOK, t>(N)) {
          f32x16 sa, da;
#pragma unroll
          for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
          row_pair_mma(sa, da, Ks, Vs, rb, t, qf, gf);
          float ds[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -lq2));
            ds[r] = pr * (da[r] - Dq);
          }
          if (tile_ragged<NTOK, t>(N)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (t * 32 + acc_row(r, L.lane) >= N) ds[r] = 0.f;
          }
          Frag<bf16> kk[4];
          tfrag4<t>(kt, kk);
          Frag<bf16> sf = pfrag(ds, 0);
          mma(dq[0], kk[0], sf);
          mma(dq[1], kk[1], sf);
          sf = pfrag(ds, 1);
          mma(dq[0], kk[2], sf);
          mma(dq[1], kk[3], sf);
        }
      });
      tile_park_private(stg, dq, scale, L);
#pragma unroll
      for (int c = 0; c < 4; ++c) {          // own K,V rows for phase B, before the arrays are refilled
        kf[c] = rowfrag_x(Ks, rb, w, c);
        vf[c] = rowfrag_x(Vs, rb, w, c);
      }
    }
    APROF3(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // mid: D_s / L2_s complete; every wave is done with K,V; dQ tiles parked
    __builtin_amdgcn_s_barrier();            // mid2: the DMA wave has taken the dQ tiles (the private tile is free for dK)
    APROF3(3);

    // ================= phase B: wave = 32 keys -> dK, dV =================
    if (active) {
      f32x16 dk[2], dv[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
      unsigned rb = rb0, tr = L.tr0;
      asm volatile("" : "+v"(rb), "+v"(tr));
      const unsigned qt_ = (unsigned)(size_t)Qs + tr, gt_ = (unsigned)(size_t)Gs + tr;
      TileLoop<NTILE>::run([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (tile_on<NTOK, t>(N)) {
          f32x16 sa, da;
#pragma unroll
          for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
          row_pair_mma(sa, da, Qs, Gs, rb, t, kf, vf);
          float pp[16], ds[16];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {   // accumulator rows 4 q4 .. 4 q4 + 3 = queries t*32 + 8 q4 + 4 g + 0..3
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(L2_s + t * 32 + 8 * q4 + 4 * L.g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(D_s + t * 32 + 8 * q4 + 4 * L.g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * q4 + e;
              const float pr = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -l4[e]));
              pp[r] = pr;
              ds[r] = pr * (da[r] - d4[e]);
            }
          }
          if (tile_ragged<NTOK, t>(N)) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (t * 32 + acc_row(r, L.lane) >= N) { pp[r] = 0.f; ds[r] = 0.f; }
          }
          Frag<bf16> gg[4];
          tfrag4<t>(gt_, gg);
          Frag<bf16> f = pfrag(pp, 0);
          mma(dv[0], gg[0], f);
          mma(dv[1], gg[1], f);
          f = pfrag(pp, 1);
          mma(dv[0], gg[2], f);
          mma(dv[1], gg[3], f);
          tfrag4<t>(qt_, gg);
          f = pfrag(ds, 0);
          mma(dk[0], gg[0], f);
          mma(dk[1], gg[1], f);
          f = pfrag(ds, 1);
          mma(dk[0], gg[2], f);
          mma(dk[1], gg[3], f);
        }
      });
      APROF3(4);
      // the next pair's own Q / dO rows: issued as soon as the tile loop's temporaries are dead (only dk / dv are live), so that
      // they fly during the two parks and the two barriers instead of after them; the O rows follow after the dV park (all 13
      // loads here, or any of them earlier, and the 256-register budget spills)
      if (nxt < nbh) load_own_qg(nxt);
      tile_park_private(stg, dk, scale, L);
      APROF3(5);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // end: every wave is done with Q,dO, D_s, L2_s; dK tiles parked
      APROF3(6);
      tile_park_rows(Gs, w, dv, 1.0f, L);
    } else {
      if (nxt < nbh) load_own_qg(nxt);
      __builtin_amdgcn_s_barrier();          // end
    }
    if (nxt < nbh) load_own_o(nxt);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // end2: dV tiles parked
    APROF3(7);
#ifdef ATTN_PROF
    ++pit;
#endif
  }
}

// ------------------------------------------------------------------------ forward, persistent
// The backward's recipe applied to the forward: <= 256 workgroups walk the (image, head) pairs; K,V are double buffered
// (4 x 28 KB) and a DMA wave fetches the next pair's while the 7 compute waves work on the current one -- ONE barrier
// per pair: "K,V of this pair have landed" / "every compute wave has left the previous pair (its buffer is free)".
// A compute wave's own Q rows come as 4 coalesced loads issued a pair ahead and are turned into fragments through its
// private LDS tile; the same tile takes the output rows, which leave as whole 128-byte lines.
template <int NTOK>
__global__ __launch_bounds__(NTHREADS3) void attn3_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                             float* __restrict__ lse, int N_rt, int heads, int nbh,
                                                             float scale) {
  const int N = NTOK > 0 ? NTOK : N_rt;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int inner = heads * HD, ld = 3 * inner;
  const LaneGeo L = lane_geo();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bh = blockIdx.x;
  if (bh >= nbh) return;

  if (w == NTILE) {                                     // ---- DMA wave
    auto issue_kv = [&](int p, int buf) {
      const bf16* Q = qkv + (size_t)(p / heads) * N * ld + (p % heads) * HD;
      dma_matrix_all(Q + inner, ld, N, smem + buf * 2 * ARR, L.lane);
      dma_matrix_all(Q + 2 * inner, ld, N, smem + buf * 2 * ARR + ARR, L.lane);
    };
    int buf = 0;
    issue_kv(bh, 0);
    for (; bh < nbh; bh += gridDim.x) {
      const int nxt = bh + gridDim.x;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      buf ^= 1;
      if (nxt < nbh) issue_kv(nxt, buf);
    }
    return;
  }

  unsigned char* stg = smem + 4 * ARR + w * STG_WAVE;
  const int q = w * 32 + L.l31;
  const bool active = w * 32 < N;
  const float c2 = scale * 1.4426950408889634f;
  u32x4 qraw[4];
  auto load_own = [&](int p) {
    const int lane_ = lane_id_here();
    const int b = p / heads, h = p % heads;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r = w * 32 + i * 8 + (lane_ >> 3);
      r = r < N ? r : N - 1;
      qraw[i] = *reinterpret_cast<const u32x4*>(qkv + ((size_t)b * N + r) * ld + h * HD + (lane_ & 7) * 8);
    }
  };
  if (active) load_own(bh);
  int buf = 0;
  for (; bh < nbh; bh += gridDim.x) {
    const int b = bh / heads, h = bh % heads;
    const int nxt = bh + gridDim.x;
    const unsigned char* Ks = smem + buf * 2 * ARR;
    const unsigned char* Vs = Ks + ARR;
    buf ^= 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own Q rows (and the previous pair's stores)
    __builtin_amdgcn_s_barrier();
    if (!active) continue;
    Frag<bf16> qf[4];
    {
      const int rl = L.lane >> 3, seg = L.lane & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (i * 8 + rl) * STG_PITCH + seg * 16) = qraw[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c)
        qf[c].v = *reinterpret_cast<const bf16x8*>(stg + L.l31 * STG_PITCH + (2 * c + L.g) * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (nxt < nbh) load_own(nxt);
    unsigned rb = (unsigned)(L.l31 * ROWB + ((L.g ^ L.fl) << 4)), tr = L.tr0;
    asm volatile("" : "+v"(rb), "+v"(tr));

    float m = -INFINITY;
    TileLoop<NTILE>::run([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (tile_on<NTOK, t>(N)) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        row_mma4(acc, Ks, rb, t, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[r];
          if (tile_ragged<NTOK, t>(N) && t * 32 + acc_row(r, L.lane) >= N) v = -INFINITY;
          m = fmaxf(m, v);
        }
      }
    });
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc2 = m * c2;
    float sum = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    const unsigned vt = (unsigned)(size_t)Vs + tr;
    TileLoop<NTILE>::run([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if (tile_on<NTOK, t>(N)) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        row_mma4(acc, Ks, rb, t, qf);
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pr[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -mc2));
          if (tile_ragged<NTOK, t>(N) && t * 32 + acc_row(r, L.lane) >= N) pr[r] = 0.f;
          sum += pr[r];
        }
        Frag<bf16> vv[4];
        tfrag4<t>(vt, vv);
        Frag<bf16> pf = pfrag(pr, 0);
        mma(o[0], vv[0], pf);
        mma(o[1], vv[1], pf);
        pf = pfrag(pr, 1);
        mma(o[0], vv[2], pf);
        mma(o[1], vv[3], pf);
      }
    });
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (L.g == 0 && q < N) lse[(size_t)bh * N + q] = m * scale + __logf(sum);
    // output rows: private tile -> 4 stores of whole lines
    tile_park_private(stg, o, inv, L);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const int lane_ = lane_id_here();
      const int rl = lane_ >> 3, seg = lane_ & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 8 + rl;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + r * STG_PITCH + seg * 16);
        if (w * 32 + r < N)
          *reinterpret_cast<u32x4*>(out + ((size_t)b * N + w * 32 + r) * inner + h * HD + seg * 8) = v;
      }
    }
  }
}

// (Round 6 pruned attn3_proj_fwd_kernel, option attn_proj: attention + output projection + LayerNorm of an image in one launch,
// bit-equal but 47.7 us against 24.5 + 19.2 us for the two launches; the one-launch encoder kernels superseded it.)

constexpr int SMEM_FWD = 2 * ARR;
constexpr int SMEM_FWD3 = 4 * ARR + NTILE * STG_WAVE;
#define ATTN_PROF_END 1
constexpr int SMEM_BWD = 4 * ARR + 2 * NPAD * (int)sizeof(float);
constexpr int SMEM_BWD3 = SMEM_BWD + NTILE * STG_WAVE;      // + the per-wave store tiles of attn3_bwd_kernel

}  // namespace

int rgbnm_launch_attn2_fwd(const void* qkv, void* out, float* lse, int B, int N, int heads, float scale,
                           hipStream_t st) {
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)attn2_fwd_kernel<196>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_FWD) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn2_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_FWD) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  const double bhn = (double)B * heads * N;
  const int slot = rgbnm_trace_begin(TR_ATTN_FWD, 4.0 * bhn * N * HD, bhn * HD * 2.0 * 4.0, st);
  if (rgbnm_get_option("attn_persist") && B * heads >= 256) {
    static DevOnce attr3;
    if (attr3.need()) {
      if (hipFuncSetAttribute((const void*)attn3_fwd_kernel<196>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_FWD3) != hipSuccess ||
          hipFuncSetAttribute((const void*)attn3_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_FWD3) != hipSuccess)
        return RGBNM_ELAUNCH;
      attr3.done();
    }
    const int nbh = B * heads;
    if (N == 196)
      hipLaunchKernelGGL(attn3_fwd_kernel<196>, dim3(256), dim3(NTHREADS3), SMEM_FWD3, st, (const bf16*)qkv, (bf16*)out, lse, N,
                         heads, nbh, scale);
    else
      hipLaunchKernelGGL(attn3_fwd_kernel<0>, dim3(256), dim3(NTHREADS3), SMEM_FWD3, st, (const bf16*)qkv, (bf16*)out, lse, N,
                         heads, nbh, scale);
  } else if (N == 196)
    hipLaunchKernelGGL(attn2_fwd_kernel<196>, dim3(B * heads), dim3(NTHREADS), SMEM_FWD, st, (const bf16*)qkv, (bf16*)out,
                       lse, N, heads, scale);
  else
    hipLaunchKernelGGL(attn2_fwd_kernel<0>, dim3(B * heads), dim3(NTHREADS), SMEM_FWD, st, (const bf16*)qkv, (bf16*)out, lse,
                       N, heads, scale);
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_launch_attn2_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                           int N, int heads, float scale, hipStream_t st) {
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)attn2_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BWD) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  const double bhn = (double)B * heads * N;
  const int slot = rgbnm_trace_begin(TR_ATTN_BWD, 10.0 * bhn * N * HD, bhn * HD * 2.0 * 8.0, st);
  if (rgbnm_get_option("attn_persist")) {
    static DevOnce attr3;
    if (attr3.need()) {
      if (hipFuncSetAttribute((const void*)attn3_bwd_kernel<196>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BWD3) != hipSuccess ||
          hipFuncSetAttribute((const void*)attn3_bwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BWD3) != hipSuccess)
        return RGBNM_ELAUNCH;
      attr3.done();
    }
    const int nbh = B * heads;
    if (N == 196)
      hipLaunchKernelGGL(attn3_bwd_kernel<196>, dim3(nbh < 256 ? nbh : 256), dim3(NTHREADS3), SMEM_BWD3, st, (const bf16*)qkv,
                         (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, N, heads, nbh, scale);
    else
      hipLaunchKernelGGL(attn3_bwd_kernel<0>, dim3(nbh < 256 ? nbh : 256), dim3(NTHREADS3), SMEM_BWD3, st, (const bf16*)qkv,
                         (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, N, heads, nbh, scale);
  } else {
    hipLaunchKernelGGL(attn2_bwd_kernel, dim3(B * heads), dim3(NTHREADS), SMEM_BWD, st, (const bf16*)qkv, (const bf16*)out,
                       (const bf16*)dout, lse, (bf16*)dqkv, N, heads, scale);
  }
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

#ifdef ATTN_PROF
extern "C" int rgbnm_debug_attn_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_prof), sizeof(unsigned long long) * 768 * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif
