// The whole encoder of JPEG-Ti as ONE persistent launch, one workgroup per image (reference: the twelve
// TransformerEncoderBlocks of models/plainvit.py:493-529 applied in sequence, :601-611).
//
// Why: nothing in the encoder forward couples two images -- LayerNorm, the Linears, attention and the residual adds are all
// row- or image-local -- so an image can stay on one CU for all twelve blocks.  The per-operation launches (qkv GEMM, attention,
// projection + LN2, fused MLP: 48 per forward) each pay a start (DMA ring fill), a memory-bound epilogue and a tail in which
// the slowest workgroup holds the next launch back, and every activation makes a round trip through L2 / HBM between them
// (463 MB per block read + written).  Here the residual stream, the LayerNorm outputs, q and the attention output never leave
// the registers of the wave that owns the 32 tokens; K and V of one head live in LDS; what goes to HBM is exactly what the
// backward needs (307 MB per block), written by the wave that produced it while the next phase already runs.  Workgroups drift
// apart over the twelve blocks, so the store bursts of different CUs no longer coincide.
//
// Structure: 512 threads = 7 compute waves (32 tokens each, 196 = 6 x 32 + 4) + 1 DMA wave that streams the weights of the
// whole forward, in consumption order, from a per-step "chain image" (written by rgbnm_chain_gather: rows and swizzles exactly
// as they lie in LDS, so every transfer is a linear 1 KB LDS-DMA) into three 24 KB slots (attention part: q, k, v and output
// projection chunks of one head each) / two 48 KB stages (MLP part: hidden chunks of 64, as mlp_fused.hip).  One workgroup
// barrier per step, 29 per block; the DMA wave runs a static schedule two short steps (one long step) ahead.
//
// Register convention ("D layout"): every [tokens, features] activation is held as the accumulator of a swapped MFMA
// (D rows = features, D cols = tokens): lane (l31, g) owns token 32 w + l31 and, of every 16 features, the 8 with bit 3 == g.
// All weight matrices are stored with their OUTPUT rows permuted by swapping index bits 2 and 3, so that accumulator registers
// 8 hs .. 8 hs + 7 of tile b are the 8 CONSECUTIVE features 32 b + 16 hs + 8 g + (0..7): one bf16x8 that is at the same time
// (i) a B-operand fragment of the next GEMM (reduction index in natural order), (ii) a 16-byte piece of the row in memory.
// The attention output leaves the PV MFMAs in plain accumulator order; the projection weights' reduction index is permuted to
// match (rgbnm_chain_gather).  LayerNorm runs on these registers: a token's 192 features sit in two lanes (l31, l31 + 32).
//
// Arithmetic vs the per-operation path: same operand rounding points (q, k, v, attention output, LayerNorm outputs, gelu, the
// residual stream are rounded to bf16 where that path stores them), fp32 accumulation; the two residual adds are done in fp32
// on the accumulator (one rounding instead of two) and the LayerNorm sums run in a different order -- results agree to bf16
// rounding, not bit for bit (tests/test_chain_fwd.py compares with the per-operation kernels and with the reference golden).
#include "common.h"
#include <string.h>
#include <type_traits>
#include "internal.h"
#include "../../include/rgbnm.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2v __attribute__((ext_vector_type(2)));

constexpr int E = 192, HID = 768, HD = 64, HEADS = 3, NTOK = 196, NTILE = 7, NPAD = 224, INNER = HEADS * HD;
constexpr int NCW = 7, NTHREADS = 64 * (NCW + 1);
constexpr int ROWB = 128;
constexpr int SLOT = 24576;                       // one 64 x 192 (or 192 x 64) bf16 weight chunk
// ---- LDS, attention part of a block
constexpr int A_SLOT0 = 0;                        // three chunk slots
constexpr int K_OFF = 3 * SLOT;                   // 73728: K[224][64] (rows 196.. hold the LN1 parameters: masked keys)
constexpr int ARR = NPAD * ROWB;                  // 28672
constexpr int V_OFF = K_OFF + ARR;                // 102400
constexpr int STGA_OFF = V_OFF + ARR;             // 131072: private 32 x 128 B tiles of the compute waves
constexpr int STG_TILE = 32 * ROWB;               // 4096
// the seventh wave's tile needs 4 rows: the parameter vectors start behind them (fp32): LN2 gamma | beta | projection bias | qkv bias
constexpr int MISC_OFF = STGA_OFF + 6 * STG_TILE + 4 * ROWB;      // 156160
constexpr int BPROJ_OFF = MISC_OFF + 2 * E * 4, BQKV_OFF = BPROJ_OFF + E * 4;
constexpr int LN1P_OFF = K_OFF + NTOK * ROWB;     // 98816: LN1 gamma | beta (fp32), inside the K pad rows
// ---- LDS, MLP part
constexpr int TAB_LIMIT = 13328;                  // the GELU table image (mlp_fused.hip F_TAB_BYTES) at offset 0
constexpr int ST1_OFF = 13440;                    // odd hidden chunks: W1 chunk | W2 chunk (128-byte aligned)
constexpr int STAGE = 2 * SLOT;
constexpr int TA_OFF = ST1_OFF + STAGE;           // 62592: gelu tiles of wave 5, then wave 6's
constexpr int B1R_OFF = TA_OFF + 2 * STG_TILE + 2 * 4 * ROWB;   // 71808: fc1-bias ring, 2 x 64 floats (inside slot 2 of the attention part:
                                                                //        the first piece is fetched after the last projection step)
constexpr int B2_OFF = B1R_OFF + 512;             // 72320: fc2 bias (fetched with the first fc1-bias piece)
constexpr int FLAG_OFF = B2_OFF + E * 4;           // 73088: the last hidden chunk whose tiles of waves 4-6 the DMA wave has taken
constexpr int ST0_OFF = K_OFF;                    // even hidden chunks (over K, V: dead by then)
constexpr int TB_OFF = ST0_OFF + STAGE;           // 122880: gelu | gelu' tiles of waves 0..4
constexpr int SMEM = 163840;
#ifndef PIPE_QKV
#define PIPE_QKV 6
#endif
#ifndef PIPE_P
#define PIPE_P 6
#endif
#ifndef PIPE_M1
#define PIPE_M1 5
#endif
#ifndef PIPE_M2
#define PIPE_M2 6
#endif
static_assert(BQKV_OFF + 3 * INNER * 4 <= SMEM, "LDS");
static_assert(TB_OFF + 10 * STG_TILE <= SMEM, "LDS");
static_assert(FLAG_OFF + 16 <= ST0_OFF, "LDS");
static_assert(LN1P_OFF + 2 * E * 4 <= V_OFF, "LDS");
static_assert(TAB_LIMIT <= ST1_OFF && ST1_OFF % 128 == 0 && TA_OFF % 128 == 0, "LDS");

struct ChainBlk {            // = rgbnm_chain_block (rgbnm.h) with typed pointers
  const bf16* wimg;          // 12 attention chunks (q0 k0 v0 q1 k1 v1 q2 k2 v2 p0 p1 p2) | 12 x (W1 chunk | W2 chunk)
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *bqkv, *bproj, *b1, *b2;
  bf16* xn1; float *mean1, *rstd1;
  bf16* qkv; float* lse; bf16* attn; bf16* x_mid; bf16* xn2; float *mean2, *rstd2;
  bf16* gp; bf16* gl; bf16* x_out;       // gp = gelu'(u) (rgbnm_block_acts.u), gl = gelu(u)
};
static_assert(sizeof(ChainBlk) == sizeof(rgbnm_chain_block), "rgbnm_chain_block layout");
constexpr int MAX_DEPTH = 12;
struct ChainArgs {            // passed BY VALUE (kernel argument segment): nothing to upload, safe inside a stream capture
  ChainBlk blk[MAX_DEPTH]; const bf16* x0;
  int depth, nimg;
  float eps, scale;
  const unsigned* tab_img; int tab_pieces;
  unsigned kneg, kpos, klo, koff, ksgn;
};

__device__ __forceinline__ int fswz(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}

struct Geo {
  int lane, l31, g, fl;
  unsigned tr0;    // byte offset inside a [token][64] array of the (t=0, fi=0, dt=0, rd=0) transpose read (attention_v2.hip)
};
__device__ __forceinline__ Geo make_geo() {
  Geo L;
  L.lane = threadIdx.x & 63;
  L.l31 = L.lane & 31;
  L.g = L.lane >> 5;
  L.fl = fswz(L.l31);
  const int k = (L.lane >> 2) & 3, G1 = (L.lane >> 4) & 1, l3 = L.lane & 3;
  const int pc = (2 * G1 + (l3 >> 1)) ^ (((k >> 1) << 2) | L.g);
  L.tr0 = (unsigned)((4 * L.g + k) * ROWB + pc * 16 + 8 * (l3 & 1));
  return L;
}

// The same from a lane id the optimiser cannot trace (common.h lane_id_here): every phase derives its per-lane constants anew, so
// none of them is live -- or spilled -- across the register-heavy phases in between
__device__ __forceinline__ Geo fresh_geo() {
  Geo L;
  L.lane = lane_id_here();
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

__device__ __forceinline__ bf16x8 pack8(u32x2 lo, u32x2 hi) {
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}
// Both fragments (FI = 0, 1) x both d tiles of key tile T of the V array (tokens as reduction axis): 8 transpose reads, one wait
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
// The same through the compiler's builtin: the reads are ordinary DS loads to the scheduler (they can be requested ahead and
// waited for where they are used; the asm form above waits on the spot)
typedef bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4v* lds_b64_ptr;
__device__ __forceinline__ u32x2 tr_read(const unsigned char* smem, unsigned off) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_ptr)(smem + off)));
}
template <int T>
__device__ __forceinline__ void tfrag4_b(const unsigned char* smem, unsigned a0, Frag<bf16> (&f)[4]) {
  const unsigned a00 = a0, a01 = (a0 ^ 32u) + 1024u, a10 = a0 ^ 64u, a11 = (a0 ^ 96u) + 1024u;
  f[0].v = pack8(tr_read(smem, a00 + T * 4096), tr_read(smem, a01 + T * 4096));
  f[1].v = pack8(tr_read(smem, a10 + T * 4096), tr_read(smem, a11 + T * 4096));
  f[2].v = pack8(tr_read(smem, a00 + T * 4096 + 2048), tr_read(smem, a01 + T * 4096 + 2048));
  f[3].v = pack8(tr_read(smem, a10 + T * 4096 + 2048), tr_read(smem, a11 + T * 4096 + 2048));
}
__device__ __forceinline__ Frag<bf16> pfrag(const float (&p)[16], int fi) {
  Frag<bf16> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (bf16)p[fi * 8 + j];
  return f;
}

// A per-lane value the optimiser must treat as new: addresses derived from it are re-derived where they are used (one XOR / add
// each) instead of being hoisted out of the block loop as invariants and spilled (common.h, lane_id_here)
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
// workgroup barrier that the compiler may not move LDS / global accesses across (the builtin alone is "no memory")
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Global stores of the saved tensors: NON-TEMPORAL (`global_store_dwordx4 ... nt`).  Nobody reads them for a millisecond (the
// backward), so they should not push the weight images and the next phases' lines out of L2: -3 % on this launch and -1.3 % on
// the backward launch behind it (interleaved A/B; the same hint did nothing for the per-operation kernels of
// round 2, whose outputs the next launch reads at once).  -DX_STORE=0 plain, =2 write-through (sc1), =3 sc0 sc1 (experiments).
#ifndef X_STORE
#define X_STORE 1
#endif
template <typename V, typename P>
__device__ __forceinline__ void gstore(P* ptr, const V& v) {
#if X_STORE == 1
  __builtin_nontemporal_store(v, reinterpret_cast<V*>(ptr));
#elif X_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(reinterpret_cast<V*>(ptr)), "v"(v) : "memory");   // (store-data hazard: the compiler cannot see this store)
#elif X_STORE == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(reinterpret_cast<V*>(ptr)), "v"(v) : "memory");
#else
  *reinterpret_cast<V*>(ptr) = v;
#endif
}

// Between a wave's accesses to ITS OWN LDS tile (write the fragment layout, read row pieces back, overwrite with the next tile) no
// wait is needed: the LDS executes one wave's DS instructions in order, and the compiler counts lgkmcnt for the registers that are
// used.  What must not happen is the compiler reordering the accesses (differently typed pointers): a compiler-only fence.  The
// drains that stood here cost two LDS round trips per stored tile (~50 tiles per block and wave).  -DX_LDSWAIT restores them.
#ifdef X_LDSWAIT
__device__ __forceinline__ void own_tile_fence() { wait_lds(); }
#else
__device__ __forceinline__ void own_tile_fence() { asm volatile("" ::: "memory"); }
#endif

// ---- DMA wave helpers: linear 1 KB pieces
template <int NKB>
__device__ __forceinline__ void dma_linear(const unsigned char* src, unsigned char* dst, int lane) {
#pragma unroll
  for (int i = 0; i < NKB; ++i)
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + i * 1024 + lane * 16), (lds_ptr)(dst + i * 1024), 16, 0, 0);
}
__device__ __forceinline__ void dma_f32x192(const float* src, unsigned char* dst, int lane) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + 64 * i + lane), (lds_ptr)(dst + 256 * i), 4, 0, 0);
}

// ---- D-layout rows: 12 pieces (index c = 2 b + hs) of 8 bf16 = features 16 c + 8 g + (0..7) of the lane's token, held as PACKED
// dwords (element j of a piece = half j & 1 of dword j >> 1): as bf16 vectors built element by element the compiler kept the
// 96 values of a row set in 96 registers and spilled them
struct Rows { u32x4 v[12]; };
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {       // two fp32 -> bf16 pair (round to nearest even), a in the low half
  const bf16x2v v = {(bf16)a, (bf16)b};
  unsigned r = __builtin_bit_cast(unsigned, v);
  asm volatile("" : "+v"(r));     // packed HERE: the optimiser otherwise sinks the conversion to the (conditional) use and keeps the fp32 pair
  return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ void ln_stats(const Rows& x, float& mu, float& rs, float eps) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned u = opaque(x.v[c][d]);
      s0 += bf_lo(u);
      s1 += bf_hi(u);
    }
  float s = s0 + s1;
  s += __shfl_xor(s, 32, 64);
  mu = s * (1.f / E);
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned u = opaque(x.v[c][d]);
      const float d0 = bf_lo(u) - mu, d1 = bf_hi(u) - mu;
      q0 = __builtin_fmaf(d0, d0, q0);
      q1 = __builtin_fmaf(d1, d1, q1);
    }
  float q = q0 + q1;
  q += __shfl_xor(q, 32, 64);
  rs = rsqrtf(__builtin_fmaf(q, 1.f / E, eps));
}
// y = (x - mu) * rs * gamma + beta, gamma | beta = 2 x 192 floats in LDS
__device__ __forceinline__ void ln_apply(const Rows& x, float mu, float rs, const float* gb, int g, Rows& y) {
#ifdef X_NOLN        // (experiments only, wrong numbers: what the five LayerNorm applications per block cost -- y = x)
#pragma unroll
  for (int c = 0; c < 12; ++c) y.v[c] = x.v[c];
  return;
#endif
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const float* gp = gb + 16 * c + 8 * g;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(gp + E), b1 = *reinterpret_cast<const f32x4*>(gp + E + 4);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float gm0 = d < 2 ? g0[2 * d] : g1[2 * d - 4], gm1 = d < 2 ? g0[2 * d + 1] : g1[2 * d - 3];
      const float bt0 = d < 2 ? b0[2 * d] : b1[2 * d - 4], bt1 = d < 2 ? b0[2 * d + 1] : b1[2 * d - 3];
      // (the packed dword is read through an opaque copy: otherwise the unpacked fp32 values of the residual stream are computed
      // once per block, shared by every use -- three LN1 recomputations and the residual add -- and held / spilled in between)
      const unsigned u = opaque(x.v[c][d]);
      y.v[c][d] = pack2(__builtin_fmaf((bf_lo(u) - mu) * rs, gm0, bt0), __builtin_fmaf((bf_hi(u) - mu) * rs, gm1, bt1));
    }
    if (c & 1) __builtin_amdgcn_sched_barrier(0);     // (else all 48 parameter reads are issued up front: 192 registers)
  }
}

// One 64-feature third (pieces c0 .. c0 + 3) of the wave's 32 rows -> private tile -> whole 128-byte row pieces -> global
// (row r of the wave at dst + r * ld).  The tile is swizzled by 16-byte chunk: chunk q of row r at q ^ (r & 7).
__device__ __forceinline__ void tile_out(unsigned char* smem, unsigned stg, const u32x4& p0, const u32x4& p1, const u32x4& p2,
                                         const u32x4& p3, bf16* dst, int ld, int live) {
  const int ln = lane_id_here();
  // own row ln & 31, chunk 2 q + g at (2 q + g) ^ (row & 7): one base offset, q selected by XOR (the tile is 128-byte aligned)
  const unsigned wo = stg + (unsigned)((ln & 31) * ROWB + (((ln >> 5) ^ (ln & 7)) << 4));
  if ((ln & 31) < live) {       // (the seventh wave's tile has 4 rows; the parameter vectors lie behind them)
    *reinterpret_cast<u32x4*>(smem + wo) = p0;
    *reinterpret_cast<u32x4*>(smem + (wo ^ 32u)) = p1;
    *reinterpret_cast<u32x4*>(smem + (wo ^ 64u)) = p2;
    *reinterpret_cast<u32x4*>(smem + (wo ^ 96u)) = p3;
  }
  own_tile_fence();
  const int rl = ln >> 3, seg = ln & 7;
  const unsigned ro = stg + (unsigned)(rl * ROWB + ((seg ^ rl) << 4));      // row i * 8 + rl: (row & 7) == rl
  bf16* gp = dst + (size_t)rl * ld + seg * 8;
  // all four row pieces are requested before the first store (one LDS round trip, not four), and a full tile -- six of the seven
  // waves -- stores behind ONE uniform branch instead of four divergent ones
  u32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(smem + ro + i * 8 * ROWB);
#ifndef X_NOSAVE      // (experiments only: the kernel without its global stores -- what the saved tensors cost)
  if (live == 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) gstore(gp + (size_t)i * 8 * ld, v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i * 8 + rl < live) gstore(gp + (size_t)i * 8 * ld, v[i]);
  }
#endif
  own_tile_fence();
}
__device__ __forceinline__ void rows_out(unsigned char* smem, unsigned stg, const Rows& x, bf16* dst, int live) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    tile_out(smem, stg, x.v[4 * t], x.v[4 * t + 1], x.v[4 * t + 2], x.v[4 * t + 3], dst + 64 * t, E, live);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc (32 output features x 32 tokens, swapped) += W chunk rows [32 ht .. +31] . x : 12 k-steps; chunk rows are 384 B, 16-byte
// chunks swizzled by pchunk (as the W1 stage of mlp_fused.hip)
__device__ __forceinline__ void gemm_k192(f32x16& acc, const unsigned char* sW, int ht, const Rows& x, const Geo& L) {
  int wbase = L.l31 * (E * 2) + ((L.g ^ L.fl) << 4) + ht * 32 * (E * 2);
  asm volatile("" : "+v"(wbase));
  Frag<bf16> fb;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    Frag<bf16> fx;
#ifdef X_HALFW      // (experiments only, wrong numbers: every weight fragment read from LDS serves two MFMAs -- what a wave that owned
                    // 64 rows would save on the LDS port)
    if ((c & 1) == 0)
#endif
    fb.v = *reinterpret_cast<const bf16x8*>(sW + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
    fx.v = as_bf16x8(x.v[c]);
    mma(acc, fb, fx);
  }
}

// Both 32-row halves (ht = 0, 1) of a 64 x 192 chunk times x as ONE stream of 24 MFMAs with the weight fragments requested
// DEPTH MFMAs ahead.  Left to itself the scheduler (256 registers: "minimum pressure" everywhere) emits read -> wait -> MFMA with a
// single fragment buffer, i.e. one exposed LDS latency (100+ cycles under load) per 32-cycle MFMA; two waves per SIMD hide half
// of it at best.  The order is pinned with sched_group_barrier, the compiler counts the lgkmcnt values.
template <int DEPTH>
__device__ __forceinline__ void gemm_k192x2(f32x16& a0, f32x16& a1, const unsigned char* sW, const Rows& x, const Geo& L) {
  int wb0 = L.l31 * (E * 2) + ((L.g ^ L.fl) << 4);
  asm volatile("" : "+v"(wb0));
  const int wb1 = wb0 + 32 * (E * 2);
  Frag<bf16> fb[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    const int c = i >> 1;
    fb[i].v = *reinterpret_cast<const bf16x8*>(sW + ((((i & 1) ? wb1 : wb0) ^ ((c % 4) << 5)) + 128 * (c / 4)));
  }
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    Frag<bf16> fx;
    fx.v = as_bf16x8(x.v[i >> 1]);
    mma((i & 1) ? a1 : a0, fb[i], fx);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
  for (int i = 0; i < 24 - DEPTH; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
}

#ifdef CHAIN_PROF
// experiments only (RGBNM_HIPCC_FLAGS=-DCHAIN_PROF): cycle stamps of block CHAIN_PROF_BLK of the first 8 workgroups, slot 2 s = in
// front of the barrier of step s, 2 s + 1 = behind it, [126] / [127] = s_memrealtime at kernel start / end; tools/chain_prof.py
#ifndef CHAIN_PROF_BLK
#define CHAIN_PROF_BLK 1
#endif
__device__ unsigned long long g_chain_prof[8 * 8 * 128];
#define CP(i)                                                                                          \
  do {                                                                                                 \
    if (ib == CHAIN_PROF_BLK && blockIdx.x < 8) {                                                      \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
      if ((threadIdx.x & 63) == 0) g_chain_prof[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 128 + (i)] = t_; \
    }                                                                                                  \
  } while (0)
#define CP_RT(i)                                                                                       \
  do {                                                                                                 \
    if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)                                                     \
      g_chain_prof[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 128 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CP(i) do {} while (0)
#define CP_RT(i) do {} while (0)
#endif
#define BAR(s) do { CP(2 * (s)); wg_barrier(); CP(2 * (s) + 1); } while (0)

__global__ __launch_bounds__(NTHREADS) void vit_chain_fwd_kernel(ChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int img = blockIdx.x;
  if (img >= p.nimg) return;
  const Geo L = make_geo();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int depth = p.depth;
  CP_RT(126);

  if (w == NCW) {
#ifdef X_DMAPRIO
    __builtin_amdgcn_s_setprio(X_DMAPRIO);
#endif
    // ================================================================ DMA wave: the static weight schedule
    const int lane = L.lane;
    auto chunkA = [&](const ChainBlk& b, int idx, int slot) {
      dma_linear<24>(reinterpret_cast<const unsigned char*>(b.wimg) + (size_t)idx * SLOT, smem + A_SLOT0 + slot * SLOT, lane);
    };
    auto chunkM = [&](const ChainBlk& b, int c) {
      unsigned char* st = smem + ((c & 1) ? ST1_OFF : ST0_OFF);
      dma_linear<48>(reinterpret_cast<const unsigned char*>(b.wimg) + (size_t)12 * SLOT + (size_t)c * STAGE, st, lane);
    };
    auto bias1 = [&](const ChainBlk& b, int c) {
      __builtin_amdgcn_global_load_lds((glb_ptr)(b.b1 + c * 64 + lane), (lds_ptr)(smem + B1R_OFF + (c & 1) * 256), 4, 0, 0);
    };
    auto ln1p = [&](const ChainBlk& b) {
      dma_f32x192(b.ln1_g, smem + LN1P_OFF, lane);
      dma_f32x192(b.ln1_b, smem + LN1P_OFF + E * 4, lane);
    };
    auto misc = [&](const ChainBlk& b) {
      dma_f32x192(b.ln2_g, smem + MISC_OFF, lane);
      dma_f32x192(b.ln2_b, smem + MISC_OFF + E * 4, lane);
      dma_f32x192(b.bproj, smem + BPROJ_OFF, lane);
#pragma unroll
      for (int i = 0; i < 3; ++i) dma_f32x192(b.bqkv + i * INNER, smem + BQKV_OFF + i * INNER * 4, lane);
    };
    {
      const ChainBlk& b0 = p.blk[0];
      ln1p(b0);
      misc(b0);
      chunkA(b0, 0, 0);
      chunkA(b0, 1, 1);
      wait_vm<24>();                        // the parameter vectors (older than the 48 chunk pieces) and q0 have landed
      wg_barrier();         // init
    }
    for (int ib = 0; ib < depth; ++ib) {
      const ChainBlk& b = p.blk[ib];
      wait_vm<24>(); BAR(0);                             // 0: q0
      // v0 goes into slot 2 only now: the slot's tail holds the fc2 bias and the hand-off flag of the PREVIOUS block's MLP part,
      // which the compute waves read in that block's epilogue -- behind barrier 0 every wave has left it (fetched there from the
      // tail of the previous block, the last pieces of this chunk landed on the bias while slow waves were still adding it:
      // features 160..191 of x_out, tools/chain_determinism.py)
      chunkA(b, 2, 2); wait_vm<24>(); BAR(1);            // 1: k0
      chunkA(b, 3, 0); wait_vm<24>(); BAR(2);            // 2: v0
      chunkA(b, 4, 1); BAR(3);                           // 3: attention 0
      chunkA(b, 5, 2); wait_vm<48>(); BAR(4);            // 4: q1
      wait_vm<24>(); BAR(5);                             // 5: k1
      chunkA(b, 6, 0); wait_vm<24>(); BAR(6);            // 6: v1
      chunkA(b, 7, 1); BAR(7);                           // 7: attention 1
      chunkA(b, 8, 2); wait_vm<48>(); BAR(8);            // 8: q2
      wait_vm<24>(); BAR(9);                             // 9: k2
      chunkA(b, 9, 0); wait_vm<24>(); BAR(10);           // 10: v2
      chunkA(b, 10, 1); BAR(11);                         // 11: attention 2
      chunkA(b, 11, 2); wait_vm<48>(); BAR(12);          // 12: p0
      chunkM(b, 0); wait_vm<63>(); BAR(13);              // 13: p1 (older than p2 and the 48 pieces behind it)
      {                                                                   // GELU table image + first fc1-bias piece
        for (int i = 0; i * 64 < p.tab_pieces; ++i)
          if (i * 64 + lane < p.tab_pieces)
            __builtin_amdgcn_global_load_lds((glb_ptr)(p.tab_img + (i * 64 + lane) * 4), (lds_ptr)(smem + i * 1024), 16, 0, 0);
      }
      wait_vm<48>(); BAR(14);                            // 14: p2 (at least the 48 pieces of chunk 0 are younger)
      BAR(15);                                           // 14b: the projection MFMAs are done: slot 2 is free
      bias1(b, 0);
      dma_f32x192(b.b2, smem + B2_OFF, lane);
      volatile int* flag = reinterpret_cast<volatile int*>(smem + FLAG_OFF);
      if (lane == 0) *flag = 0;
      // The gelu / gelu' tiles of waves 4-6 -- the younger, slower wave of each SIMD pair (mlp_fused.hip) -- leave through this
      // wave: behind the barrier that starts chunk c it copies the tiles of chunk c - 1 into registers, raises the flag their
      // owners poll before overwriting them, and stores them (whole 128-byte row pieces) before fetching the next weight chunk.
      u32x4 tv[3][2][4];
      const int trow = lane >> 3, tvec = lane & 7;
      auto take_tiles = [&]() {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const unsigned tb = (unsigned)(q == 0 ? TB_OFF + 8 * STG_TILE : (q == 1 ? TA_OFF : TA_OFF + 2 * STG_TILE));
          const int lr = q == 2 ? NTOK - 6 * 32 : 32;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = trow + 8 * i;
              if (row < lr) tv[q][t][i] = *reinterpret_cast<const u32x4*>(smem + tb + t * lr * ROWB + row * ROWB + ((tvec ^ (row & 7)) << 4));
            }
        }
        wait_lds();
      };
      auto store_tiles = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int lr = q == 2 ? NTOK - 6 * 32 : 32;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = trow + 8 * i;
            if (row < lr) {
              const size_t go = ((size_t)img * NTOK + 32 * (4 + q) + row) * HID + chunk * 64 + tvec * 8;
#ifndef X_NOSAVE
              gstore(b.gl + go, tv[q][0][i]);
              gstore(b.gp + go, tv[q][1][i]);
#endif
            }
          }
        }
      };
      wait_vm<0>(); BAR(16);                             // 15: hidden chunk 0, table, bias
      for (int c = 0; c < 11; ++c) {
        if (c >= 1) {
          take_tiles();
          if (lane == 0) *flag = c;
          store_tiles(c - 1);
        }
        chunkM(b, c + 1);
        bias1(b, c + 1);
        wait_vm<0>(); BAR(17 + c);                       // 16 + c
      }
      take_tiles();                                      // (chunk 10's, behind barrier 27 = the start of chunk 11)
      if (lane == 0) *flag = 11;
      store_tiles(10);
      if (ib + 1 < depth) ln1p(p.blk[ib + 1]);                           // stage 0 (over the K pad rows) is dead since barrier 26
      wait_vm<0>(); BAR(28);                             // 27: every wave has left the last hidden chunk
      take_tiles();                                      // chunk 11's: the compute waves poll the flag before they touch their row tiles
      if (lane == 0) *flag = 12;
      store_tiles(11);
      if (ib + 1 < depth) {
        const ChainBlk& nb = p.blk[ib + 1];
        misc(nb);
        chunkA(nb, 0, 0);
        chunkA(nb, 1, 1);                                // (chunk 2 behind the next barrier 0, see there)
      }
    }
    return;
  }

  // ==================================================================== compute waves
#if defined(X_PRIO) && X_PRIO == 1      // experiments: static issue priority for the younger wave of every SIMD pair
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
#elif defined(X_PRIO) && X_PRIO == 2    // ... for the older one
  if (w < 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int row0 = 32 * w;
  const int live = NTOK - row0 < 32 ? NTOK - row0 : 32;          // 32, or 4 for the seventh wave
  const int tok = row0 + L.l31 < NTOK ? row0 + L.l31 : NTOK - 1; // this lane's token (clamped: finite data in the pad rows)
  const size_t grow0 = (size_t)img * NTOK + row0;                // first global row of the wave
  const unsigned stg = (unsigned)(STGA_OFF + w * STG_TILE);      // LDS offset of the private tile
  const float c2 = p.scale * 1.4426950408889634f;
  const float* ln1p = reinterpret_cast<const float*>(smem + LN1P_OFF);
  const float* ln2p = reinterpret_cast<const float*>(smem + MISC_OFF);
  const float* bprj = reinterpret_cast<const float*>(smem + BPROJ_OFF);

  Rows xr;                                                        // the residual stream of this lane's token
  {
    const bf16* xrow = p.x0 + ((size_t)img * NTOK + tok) * E + 8 * L.g;
#pragma unroll
    for (int c = 0; c < 12; ++c) xr.v[c] = *reinterpret_cast<const u32x4*>(xrow + 16 * c);
  }
  float mu1, rs1;
  ln_stats(xr, mu1, rs1, p.eps);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();                                   // init: LN parameters of block 0 are in LDS
  {
    const ChainBlk& b0 = p.blk[0];
    Rows xn;
    ln_apply(xr, mu1, rs1, ln1p, L.g, xn);
    rows_out(smem, stg, xn, b0.xn1 + grow0 * E, live);
    if (L.g == 0 && L.l31 < live) {
      b0.mean1[grow0 + L.l31] = mu1;
      b0.rstd1[grow0 + L.l31] = rs1;
    }
  }

  for (int ib = 0; ib < depth; ++ib) {
    const ChainBlk& b = p.blk[ib];
    Frag<bf16> of[HEADS][4];                                      // attention output of the three heads as projection operands
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      const Geo L = fresh_geo();
      Rows xn;
      __builtin_amdgcn_sched_barrier(0);
      ln_apply(xr, mu1, rs1, ln1p, L.g, xn);                      // LN1 output, recomputed per head (48 registers not held)
      __builtin_amdgcn_sched_barrier(0);
      Frag<bf16> qf[4];
      // ---------------- q, k, v of head h: three steps of 24 MFMAs
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const float* bias = reinterpret_cast<const float*>(smem + BQKV_OFF) + m * INNER + h * HD + 8 * L.g;
        BAR(4 * h + m);                           // step 4 h + m: the chunk has landed
        const unsigned char* sW = smem + A_SLOT0 + m * SLOT;
        f32x16 acc[2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ht][r] = 0.f;
#ifdef X_NOPIPE
        gemm_k192(acc[0], sW, 0, xn, L);
        gemm_k192(acc[1], sW, 1, xn, L);
#else
        gemm_k192x2<PIPE_QKV>(acc[0], acc[1], sW, xn, L);
#endif
        u32x4 pc[4];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
          for (int hs = 0; hs < 2; ++hs) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 32 * ht + 16 * hs);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 32 * ht + 16 * hs + 4);
#pragma unroll
            for (int d = 0; d < 4; ++d)
              pc[2 * ht + hs][d] = pack2(acc[ht][8 * hs + 2 * d] + (d < 2 ? b0[2 * d] : b1[2 * d - 4]),
                                         acc[ht][8 * hs + 2 * d + 1] + (d < 2 ? b0[2 * d + 1] : b1[2 * d - 3]));
          }
        if (m == 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) qf[c].v = as_bf16x8(pc[c]);
        } else {
          // K / V array row of this lane's token: 16-byte chunk q at q ^ fswz(row).  The seventh wave leaves the K pad rows
          // alone (LN1 parameters) and fills the V pad rows with its clamped (finite) rows.
          const unsigned ao = opaque((unsigned)((m == 1 ? K_OFF : V_OFF) + (row0 + L.l31) * ROWB + ((L.g ^ L.fl) << 4)));
          if (m == 2 || L.l31 < live) {
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(smem + (ao ^ (unsigned)(c << 5))) = pc[c];
          }
        }
        tile_out(smem, stg, pc[0], pc[1], pc[2], pc[3], b.qkv + grow0 * (3 * INNER) + m * INNER + h * HD, 3 * INNER, live);
      }
      // ---------------- attention of head h (the forward of attention_v2.hip: scores recomputed in the second pass)
      BAR(4 * h + 3);                             // step 4 h + 3: every K, V row is written
      {
        const Geo L = fresh_geo();
        const unsigned char* Ks = smem + K_OFF;
        unsigned rb = (unsigned)(L.l31 * ROWB + ((L.g ^ L.fl) << 4)), tr = L.tr0;
        asm volatile("" : "+v"(rb), "+v"(tr));
        float mx = -INFINITY;
#if defined(X_NOPIPE) || defined(X_NOATTN)
#ifdef X_NOATTN
        TileLoop<0>::run([&](auto tc) {
#else
        TileLoop<NTILE>::run([&](auto tc) {
#endif
          constexpr int t = decltype(tc)::value;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            Frag<bf16> kf;
            kf.v = *reinterpret_cast<const bf16x8*>(Ks + (rb ^ (unsigned)(c << 5)) + t * 32 * ROWB);
            mma(acc, kf, qf[c]);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if (t * 32 + 32 > NTOK && t * 32 + acc_row(r, L.lane) >= NTOK) v = -INFINITY;
            mx = fmaxf(mx, v);
          }
        });
#else
        // software-pipelined by hand, one fence per key tile: the K fragments of tile t + 1 are requested, then the 4 MFMAs of tile t
        // are issued and the maxima of tile t - 1 run under them (two score tiles live)
        {
          Frag<bf16> kc[4], kn[4];
          f32x16 prev;
#pragma unroll
          for (int c = 0; c < 4; ++c) kc[c].v = *reinterpret_cast<const bf16x8*>(Ks + (rb ^ (unsigned)(c << 5)));
          TileLoop<NTILE + 1>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            f32x16 acc;
            if (t + 1 < NTILE) {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                kn[c].v = *reinterpret_cast<const bf16x8*>(Ks + (rb ^ (unsigned)(c << 5)) + (t + 1) * 32 * ROWB);
            }
            if (t < NTILE) {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
              for (int c = 0; c < 4; ++c) mma(acc, kc[c], qf[c]);
            }
            if (t >= 1) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float v = prev[r];
                if (t * 32 > NTOK && (t - 1) * 32 + acc_row(r, L.lane) >= NTOK) v = -INFINITY;
                mx = fmaxf(mx, v);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            prev = acc;
#pragma unroll
            for (int c = 0; c < 4; ++c) kc[c] = kn[c];
          });
        }
#endif
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc2 = mx * c2;
        float sum = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        const unsigned vt = (unsigned)V_OFF + tr;
#if defined(X_NOPIPE) || defined(X_NOATTN)
#ifdef X_NOATTN
        TileLoop<0>::run([&](auto tc) {
#else
        TileLoop<NTILE>::run([&](auto tc) {
#endif
          constexpr int t = decltype(tc)::value;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            Frag<bf16> kf;
            kf.v = *reinterpret_cast<const bf16x8*>(Ks + (rb ^ (unsigned)(c << 5)) + t * 32 * ROWB);
            mma(acc, kf, qf[c]);
          }
          float pr[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            pr[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -mc2));
            if (t * 32 + 32 > NTOK && t * 32 + acc_row(r, L.lane) >= NTOK) pr[r] = 0.f;
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
        });
#else
        // one fenced region per key tile: S_t (K fragments requested in the previous region), then the V fragments of this tile and
        // the K fragments of the next one are requested and travel under the exponentials, P.V_t
        {
          Frag<bf16> kc[4], kn[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) kc[c].v = *reinterpret_cast<const bf16x8*>(Ks + (rb ^ (unsigned)(c << 5)));
          __builtin_amdgcn_sched_barrier(0);
          TileLoop<NTILE>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) mma(acc, kc[c], qf[c]);
            Frag<bf16> vv[4];
            tfrag4_b<t>(smem, vt, vv);
            if (t + 1 < NTILE) {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                kn[c].v = *reinterpret_cast<const bf16x8*>(Ks + (rb ^ (unsigned)(c << 5)) + (t + 1) * 32 * ROWB);
            }
            float pr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#ifdef X_NOEXP      // (experiments only, wrong numbers: what the exponentials cost)
              pr[r] = fmaf(acc[r], c2, -mc2);
#else
              pr[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -mc2));
#endif
              if (t * 32 + 32 > NTOK && t * 32 + acc_row(r, L.lane) >= NTOK) pr[r] = 0.f;
              sum += pr[r];
            }
            Frag<bf16> pf = pfrag(pr, 0);
            mma(o[0], vv[0], pf);
            mma(o[1], vv[1], pf);
            pf = pfrag(pr, 1);
            mma(o[0], vv[2], pf);
            mma(o[1], vv[3], pf);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (t + 1 < NTILE) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) kc[c] = kn[c];
          });
        }
#endif
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (L.g == 0 && L.l31 < live) b.lse[((size_t)img * HEADS + h) * NTOK + row0 + L.l31] = mx * p.scale + __logf(sum);
        // o: register r of tile dt = head dim 32 dt + 8 (r >> 2) + 4 g + (r & 3).  As projection operand (k order permuted in the
        // chain image): registers 8 hs .. 8 hs + 7.  As memory rows: 8-byte pieces through the private tile.
        u32x4 op[4];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int hs = 0; hs < 2; ++hs) {
#pragma unroll
            for (int d = 0; d < 4; ++d) op[2 * dt + hs][d] = pack2(o[dt][8 * hs + 2 * d] * inv, o[dt][8 * hs + 2 * d + 1] * inv);
            of[h][2 * dt + hs].v = as_bf16x8(op[2 * dt + hs]);
          }
        {
          const int ln = lane_id_here();
          // 8-byte pieces: head dims 8 q + 4 g + (0..3) of the own row, chunk q at q ^ (row & 7)
          const unsigned wo = stg + (unsigned)((ln & 31) * ROWB + ((ln & 7) << 4) + (ln >> 5) * 8);
          if ((ln & 31) < live)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              const u32x4 src = op[2 * dt + (rq >> 1)];
              const u32x2 q2 = {src[2 * (rq & 1)], src[2 * (rq & 1) + 1]};
              *reinterpret_cast<u32x2*>(smem + (wo ^ (unsigned)((dt * 4 + rq) << 4))) = q2;
            }
          own_tile_fence();
          const int rl = ln >> 3, seg = ln & 7;
          const unsigned ro = stg + (unsigned)(rl * ROWB + ((seg ^ rl) << 4));
          bf16* dst = b.attn + (grow0 + rl) * INNER + h * HD + seg * 8;
          u32x4 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(smem + ro + i * 8 * ROWB);
#ifndef X_NOSAVE
          if (live == 32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gstore(dst + (size_t)i * 8 * INNER, v[i]);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (i * 8 + rl < live) gstore(dst + (size_t)i * 8 * INNER, v[i]);
          }
#endif
          own_tile_fence();
        }
      }
    }
    // ---------------- output projection: 3 steps of 24 MFMAs (k = the 64 dims of head h), then residual + LN2
    const Geo L = fresh_geo();
    f32x16 accp[6];
#pragma unroll
    for (int bt = 0; bt < 6; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[bt][r] = 0.f;
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      BAR(12 + h);                                // steps 12, 13, 14
      // chunk image: [192 rows][128 B], chunk q at q ^ fswz(row)
      const unsigned wo = opaque((unsigned)(A_SLOT0 + h * SLOT + L.l31 * ROWB + ((L.g ^ L.fl) << 4)));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<bf16> fw[6];
#pragma unroll
        for (int bt = 0; bt < 6; ++bt)
#ifdef X_HALFW
          if (bt & 1) fw[bt].v = fw[bt - 1].v; else
#endif
          fw[bt].v = *reinterpret_cast<const bf16x8*>(smem + (wo ^ (unsigned)(s << 5)) + bt * 32 * ROWB);
#pragma unroll
        for (int bt = 0; bt < 6; ++bt) mma(accp[bt], fw[bt], of[h][s]);
      }
#ifndef X_NOPIPE
      // the 24 weight fragments of the step PIPE_P MFMAs ahead of their use
      __builtin_amdgcn_sched_group_barrier(0x100, PIPE_P, 0);
#pragma unroll
      for (int i = 0; i < 24 - PIPE_P; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < PIPE_P; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    BAR(15);                                                      // step 14b: the chunk slots are free (fc1-bias ring lives there)
    float mu2, rs2;
    {
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        const float* bp = bprj + 16 * c + 8 * L.g;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned u = opaque(xr.v[c][d]);
          xr.v[c][d] = pack2(accp[c >> 1][8 * (c & 1) + 2 * d] + (d < 2 ? b0[2 * d] : b1[2 * d - 4]) + bf_lo(u),
                             accp[c >> 1][8 * (c & 1) + 2 * d + 1] + (d < 2 ? b0[2 * d + 1] : b1[2 * d - 3]) + bf_hi(u));
        }
        if (c & 1) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      ln_stats(xr, mu2, rs2, p.eps);
      __builtin_amdgcn_sched_barrier(0);
      rows_out(smem, stg, xr, b.x_mid + grow0 * E, live);
      __builtin_amdgcn_sched_barrier(0);
      if (L.g == 0 && L.l31 < live) {
        b.mean2[grow0 + L.l31] = mu2;
        b.rstd2[grow0 + L.l31] = rs2;
      }
    }
    Rows fa;
    __builtin_amdgcn_sched_barrier(0);
    ln_apply(xr, mu2, rs2, ln2p, L.g, fa);
    __builtin_amdgcn_sched_barrier(0);
    rows_out(smem, stg, fa, b.xn2 + grow0 * E, live);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- MLP: 12 hidden chunks of 64 (mlp_fused.hip), accumulator initialised with the residual
    f32x16 acc2[6];
#pragma unroll
    for (int c = 0; c < 12; ++c)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned u = opaque(xr.v[c][d]);
        acc2[c >> 1][8 * (c & 1) + 2 * d] = bf_lo(u);
        acc2[c >> 1][8 * (c & 1) + 2 * d + 1] = bf_hi(u);
      }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned tg = (unsigned)(w < 5 ? TB_OFF + 2 * w * STG_TILE
                                         : (w == 5 ? TA_OFF : TA_OFF + 2 * STG_TILE));       // gelu tile (LDS offset); gelu' follows
    const unsigned tpoff = (unsigned)(live * ROWB);                // (32 or 4 rows per tile)
    const bool lane_live = L.l31 < live;
    const bool handed = w >= 4;                                    // this wave's gelu / gelu' tiles leave through the DMA wave
    const volatile int* flag = reinterpret_cast<const volatile int*>(smem + FLAG_OFF);
    const unsigned kneg = p.kneg, kpos = p.kpos, klo = p.klo, koff = p.koff, ksgn = p.ksgn;
    unsigned k4v = 0x00040004u;
    asm volatile("" : "+v"(k4v));
    const int woff0 = L.l31 * (E * 2) + ((L.g ^ L.fl) << 4);
#ifdef X_NOMLP
    for (int chunk = 0; chunk < HID / 64; ++chunk) BAR(16 + chunk);
    for (int chunk = 0; chunk < 0; ++chunk) {
#else
    for (int chunk = 0; chunk < HID / 64; ++chunk) {
#endif
      BAR(16 + chunk);                            // steps 15 .. 26
      const unsigned st_off = (unsigned)((chunk & 1) ? ST1_OFF : ST0_OFF);
      const unsigned char* sW1 = smem + st_off;
      const float* bch = reinterpret_cast<const float*>(smem + B1R_OFF) + (chunk & 1) * 64;
      // per-lane bases of this chunk, made new here so that nothing derived from them is a block-loop invariant
      const unsigned w2o = opaque(st_off + (unsigned)(SLOT + L.l31 * ROWB + ((L.g ^ L.fl) << 4)));       // W2 chunk, fragment s at ^ (s << 5)
      const unsigned two = opaque(tg + (unsigned)(L.l31 * ROWB + ((L.g ^ (L.l31 & 7)) << 4)));          // own tile row, piece q at ^ (q << 5)
#pragma unroll
      for (int ht = 0; ht < 2; ++ht) {
        f32x16 a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[r] = 0.f;
        int wbase = woff0 + ht * 32 * (E * 2);
        asm volatile("" : "+v"(wbase));
#ifdef X_NOPIPE
        Frag<bf16> fb;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
          Frag<bf16> fx;
#ifdef X_HALFW
          if ((c & 1) == 0)
#endif
          fb.v = *reinterpret_cast<const bf16x8*>(sW1 + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
          fx.v = as_bf16x8(fa.v[c]);
          mma(a1, fb, fx);
        }
#else
        {
          Frag<bf16> fb[12];
#pragma unroll
          for (int c = 0; c < 12; ++c)
            fb[c].v = *reinterpret_cast<const bf16x8*>(sW1 + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
#pragma unroll
          for (int c = 0; c < 12; ++c) {
            Frag<bf16> fx;
            fx.v = as_bf16x8(fa.v[c]);
            mma(a1, fb[c], fx);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, PIPE_M1, 0);
#pragma unroll
          for (int i = 0; i < 12 - PIPE_M1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#pragma unroll
          for (int i = 0; i < PIPE_M1; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the six W2 fragments of the first 16 hidden columns travel under the GELU arithmetic
        Frag<bf16> fw2[12];
#pragma unroll
        for (int bt = 0; bt < 6; ++bt)
          fw2[bt].v = *reinterpret_cast<const bf16x8*>(smem + (w2o ^ (unsigned)((2 * ht) << 5)) + bt * 32 * ROWB);
        __builtin_amdgcn_sched_barrier(0);
#endif
        Frag<bf16> pg[2];
        if (handed && ht == 0 && chunk > 0) {                      // the DMA wave has taken the previous chunk's tiles (normally long ago)
          while (__builtin_amdgcn_readfirstlane(*flag) < chunk) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
          const float* bp = bch + 32 * ht + 16 * hs + 8 * L.g;
          const f32x4 bl = *reinterpret_cast<const f32x4*>(bp), bh = *reinterpret_cast<const f32x4*>(bp + 4);
          bf16x8 gv, dv;
          // table GELU of eight elements (see mlp_fused.hip: gelu_full_kernel / mlp_fwd_kernel<true>)
          unsigned pb[4], agv[4], alo[4], ahi[4], e0[4], e1[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = 2 * jj;
            const f32x2 uu = f32x2{a1[8 * hs + j], a1[8 * hs + j + 1]} + (j < 4 ? f32x2{bl[j], bl[j + 1]} : f32x2{bh[j - 4], bh[j - 3]});
            const bf16x2v pbv = {(bf16)uu[0], (bf16)uu[1]};
            pb[jj] = __builtin_bit_cast(unsigned, pbv);
#ifdef X_NOGELU      // (experiments only, wrong numbers: what the table GELU's index arithmetic and lookups cost -- gelu = gelu' = u)
            agv[jj] = alo[jj] = ahi[jj] = 0;
            continue;
#endif
            unsigned p1, p2, ak, i4, sg;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(p1) : "v"(pb[jj]), "s"(kneg));
            asm("v_pk_min_i16 %0, %1, %2" : "=v"(p2) : "v"(p1), "s"(kpos));
            p1 &= 0x7FFF7FFFu;
            p2 &= 0x7FFF7FFFu;
            asm("v_pk_max_u16 %0, %1, %2" : "=v"(agv[jj]) : "v"(p1), "s"(0x00800080u));
            asm("v_pk_max_u16 %0, %1, %2" : "=v"(ak) : "v"(p2), "s"(klo));
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(i4) : "v"(ak), "v"(k4v), "s"(koff));
            asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(sg) : "s"(0x000F000Fu), "v"(pb[jj]));
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(i4) : "v"(sg), "s"(ksgn), "v"(i4));
            alo[jj] = i4 & 0xffffu;
            ahi[jj] = i4 >> 16;
          }
#ifndef X_NOGELU
          asm volatile(
              "ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
              "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(e0[0]), "=&v"(e1[0]), "=&v"(e0[1]), "=&v"(e1[1]), "=&v"(e0[2]), "=&v"(e1[2]), "=&v"(e0[3]), "=&v"(e1[3])
              : "v"(alo[0]), "v"(ahi[0]), "v"(alo[1]), "v"(ahi[1]), "v"(alo[2]), "v"(ahi[2]), "v"(alo[3]), "v"(ahi[3])
              : "memory");
#endif
          u32x4 gq, dq;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
#ifdef X_NOGELU
            gq[jj] = dq[jj] = pb[jj];
#else
            const unsigned dpair = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x05040100u);
            unsigned gm;
            asm("v_pk_sub_u16 %0, %1, %2" : "=v"(gm) : "v"(agv[jj]), "v"(dpair));
            gq[jj] = (pb[jj] & 0x80008000u) | gm;
            dq[jj] = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x07060302u);
#endif
          }
          gv = __builtin_bit_cast(bf16x8, gq);
          dv = __builtin_bit_cast(bf16x8, dq);
          pg[hs].v = gv;
          if (lane_live) {
            const unsigned to = two ^ (unsigned)((2 * ht + hs) << 5);
            *reinterpret_cast<bf16x8*>(smem + to) = gv;
            *reinterpret_cast<bf16x8*>(smem + to + tpoff) = dv;
          }
        }
#ifdef X_NOPIPE
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
          const int s = 2 * ht + hs;
          Frag<bf16> fw[6];
#pragma unroll
          for (int bt = 0; bt < 6; ++bt)
#ifdef X_HALFW
            if (bt & 1) fw[bt].v = fw[bt - 1].v; else
#endif
            fw[bt].v = *reinterpret_cast<const bf16x8*>(smem + (w2o ^ (unsigned)(s << 5)) + bt * 32 * ROWB);
#pragma unroll
          for (int bt = 0; bt < 6; ++bt) mma(acc2[bt], fw[bt], pg[hs]);
        }
#else
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bt = 0; bt < 6; ++bt)
          fw2[6 + bt].v = *reinterpret_cast<const bf16x8*>(smem + (w2o ^ (unsigned)((2 * ht + 1) << 5)) + bt * 32 * ROWB);
#pragma unroll
        for (int bt = 0; bt < 6; ++bt) mma(acc2[bt], fw2[bt], pg[0]);
#pragma unroll
        for (int bt = 0; bt < 6; ++bt) mma(acc2[bt], fw2[6 + bt], pg[1]);
        // (hs = 0's fragments landed under the GELU; hs = 1's are requested one per MFMA)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      // the chunk's gelu / gelu' tiles out as whole 128-byte row pieces (handed tiles: written out before the barrier behind which
      // the DMA wave takes them)
      if (handed) wait_lds();
      else own_tile_fence();
      if (!handed) {
        const int ln = lane_id_here();
        if (live == 32) {                 // (waves 0 - 3: always) all eight row pieces requested before the first store
          bf16x8 v0[4], v1[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
            const unsigned so = tg + (unsigned)(row * ROWB + ((vec ^ (row & 7)) << 4));
            v0[i] = *reinterpret_cast<const bf16x8*>(smem + so);
            v1[i] = *reinterpret_cast<const bf16x8*>(smem + so + tpoff);
          }
#ifndef X_NOSAVE
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
            const size_t go = (grow0 + row) * HID + chunk * 64 + vec * 8;
            gstore(b.gl + go, v0[i]);
            gstore(b.gp + go, v1[i]);
          }
#endif
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
            if (row < live) {
              const unsigned so = tg + (unsigned)(row * ROWB + ((vec ^ (row & 7)) << 4));
              const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(smem + so);
              const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(smem + so + tpoff);
              const size_t go = (grow0 + row) * HID + chunk * 64 + vec * 8;
#ifndef X_NOSAVE
              gstore(b.gl + go, v0);
              gstore(b.gp + go, v1);
#endif
            }
          }
        }
        own_tile_fence();
      }
    }
    BAR(28);                                      // step 27: the MLP stages are dead; the last tiles of waves 4-6 are being taken
    while (__builtin_amdgcn_readfirstlane(*flag) < HID / 64) __builtin_amdgcn_s_sleep(1);
    // ---------------- x_out = x_mid + fc2(...) + b2 ; next block's LN1
    {
      const float* b2p = reinterpret_cast<const float*>(smem + B2_OFF) + 8 * L.g;
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b2p + 16 * c), b1 = *reinterpret_cast<const f32x4*>(b2p + 16 * c + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d)
          xr.v[c][d] = pack2(acc2[c >> 1][8 * (c & 1) + 2 * d] + (d < 2 ? b0[2 * d] : b1[2 * d - 4]),
                             acc2[c >> 1][8 * (c & 1) + 2 * d + 1] + (d < 2 ? b0[2 * d + 1] : b1[2 * d - 3]));
        if (c & 1) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      rows_out(smem, stg, xr, b.x_out + grow0 * E, live);
      __builtin_amdgcn_sched_barrier(0);
      if (ib + 1 < depth) {
        const ChainBlk& nb = p.blk[ib + 1];
        ln_stats(xr, mu1, rs1, p.eps);
        __builtin_amdgcn_sched_barrier(0);
        Rows xn;
        ln_apply(xr, mu1, rs1, ln1p, L.g, xn);
        __builtin_amdgcn_sched_barrier(0);
        rows_out(smem, stg, xn, nb.xn1 + grow0 * E, live);
        __builtin_amdgcn_sched_barrier(0);
        if (L.g == 0 && L.l31 < live) {
          nb.mean1[grow0 + L.l31] = mu1;
          nb.rstd1[grow0 + L.l31] = rs1;
        }
      }
    }
    CP(58);
  }
  CP_RT(127);
}

// dst[i] = src[idx[i]]: the chain image from the [N,K] operand shadows (the index table is built once on the host)
__global__ __launch_bounds__(256) void chain_gather_kernel(const bf16* __restrict__ src, const int* __restrict__ idx,
                                                           bf16* __restrict__ dst, long long n8) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const int4 a = *reinterpret_cast<const int4*>(idx + 8 * i), c = *reinterpret_cast<const int4*>(idx + 8 * i + 4);
    bf16x8 v;
    v[0] = src[a.x]; v[1] = src[a.y]; v[2] = src[a.z]; v[3] = src[a.w];
    v[4] = src[c.x]; v[5] = src[c.y]; v[6] = src[c.z]; v[7] = src[c.w];
    *reinterpret_cast<bf16x8*>(dst + 8 * i) = v;
  }
}

}  // namespace

int rgbnm_gelu_table_query(const unsigned** img, int* A0, int* P1, int* N1, int* ndw);   // mlp_fused.hip

#ifdef CHAIN_PROF
extern "C" int rgbnm_chain_prof_read(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_chain_prof), sizeof(unsigned long long) * 8 * 8 * 128) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

size_t rgbnm_chain_block_bytes(void) { return sizeof(ChainBlk); }
long long rgbnm_chain_image_elems(void) { return 12ll * (SLOT / 2) + 12ll * (STAGE / 2); }

int rgbnm_chain_gather(const void* src, const int* idx, void* dst, long long n, void* stream) {
  if (!src || !idx || !dst || n <= 0 || n % 8) return RGBNM_EINVAL;
  const long long n8 = n / 8;
  int grid = (int)((n8 + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(chain_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, idx, (bf16*)dst, n8);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

// 1 = not eligible (the caller runs the blocks one by one)
int rgbnm_vit_chain_fwd(const rgbnm_vit_cfg* c, const rgbnm_chain_block* blocks, int depth, const void* x0, void* stream) {
  if (!c || !blocks || !x0 || depth <= 0) return RGBNM_EINVAL;
  if (c->dtype != RGBNM_DT_BF16 || c->E != E || c->heads != HEADS || c->N != NTOK || c->B < 1 || depth > MAX_DEPTH) return 1;
  const unsigned* img = nullptr;
  int A0 = 0, P1 = 0, N1 = 0, ndw = 0;
  if (rgbnm_gelu_table_query(&img, &A0, &P1, &N1, &ndw) != 1 || ndw * 4 > TAB_LIMIT) return 1;
  ChainArgs p;
  memcpy(p.blk, blocks, sizeof(ChainBlk) * depth);
  p.x0 = (const bf16*)x0; p.depth = depth; p.nimg = c->B;
  p.eps = c->ln_eps; p.scale = c->attn_scale;
  p.tab_img = img; p.tab_pieces = (ndw * 4 + 15) / 16;
  p.kneg = 0x00010001u * (unsigned)(0x8000 | N1);
  p.kpos = 0x00010001u * (unsigned)P1;
  p.klo = 0x00010001u * (unsigned)(A0 - 1);
  p.koff = 0x00010001u * (unsigned)((0x10000 - 4 * (A0 - 1)) & 0xffff);
  p.ksgn = 0x00010001u * (unsigned)(4 * (P1 - A0 + 2));
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)vit_chain_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  // algorithmic work per block and image: 0.20292 GFLOP; bytes that must cross HBM: the twelve saved tensors
  // (xn1, qkv, attn, x_mid, xn2, gelu, gelu', x_out: (5 * 192 + 576 + 2 * 768) bf16 per token + statistics) and the weights once
  const double tok = (double)c->B * NTOK;
  const double flops = depth * tok * 2.0 * (E * 3.0 * INNER + 2.0 * NTOK * INNER + INNER * E + 2.0 * E * HID);
  const double bytes = depth * (tok * ((5.0 * E + 3.0 * INNER + 2.0 * HID) * 2.0 + 4 * 4 + HEADS * 4) + 12.0 * SLOT + 12.0 * STAGE) + tok * E * 2.0;
  const int slot = rgbnm_trace_begin(TR_CHAIN_FWD, flops, bytes, (hipStream_t)stream);
  hipLaunchKernelGGL(vit_chain_fwd_kernel, dim3(c->B), dim3(NTHREADS), SMEM, (hipStream_t)stream, p);
  rgbnm_trace_end(slot, (hipStream_t)stream);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // extern "C"
