// The data path of the whole encoder BACKWARD of JPEG-Ti as ONE persistent launch, one workgroup per image (reference: the
// twelve TransformerEncoderBlocks of models/plainvit.py:493-529 run backwards by autograd).
//
// What couples images in the backward are only the weight / bias / LayerNorm-parameter gradients (sums over all tokens); the
// chain dy -> d(x_mid) -> d(attention output) -> d(q, k, v) -> dx of a block is image-local exactly like the forward.  This
// kernel runs that chain for all blocks of one image on one CU and leaves behind, per block, the operands of the four
// weight-gradient GEMMs (du, d(x_mid), d(qkv), dy: gemm_tn_pipe.hip runs them afterwards, one grouped launch per block as
// before) and the image's partial sums of the LayerNorm parameter gradients.  It replaces 4 launches per block -- fused MLP
// backward, the attention-output dX GEMM, attention backward, the qkv dX GEMM with the LayerNorm backward -- and their starts,
// memory-bound epilogues and tails; the hand-offs between them stay inside the CU's L2 (d(attention output), d(qkv)) or LDS
// (d(x_mid), the next block's dy).
//
// The phases are the bodies of those kernels, same arithmetic and summation order, so the results are the SAME BITS as the
// per-operation path (tests/test_chain_bwd.py):
//   M   mlp_fused.hip mlp_bwd_kernel: du = (dy . W2) * gelu'(u) chunk by chunk, dxn2 += du . W1; LayerNorm backward through a
//       row-major staging tile (ln_bwd_rows.h), which also hands d(x_mid) back to the waves as operand fragments
//   P   d(attention output) = d(x_mid) . Wproj: three steps of 24 MFMAs per wave (as the q / k / v steps of vit_chain.hip)
//   A   attention_v2.hip attn3_bwd_kernel for the three heads of the image (K,V / Q,dO arrays by LDS-DMA, gradient tiles out
//       through the DMA wave)
//   X   dxn1 = d(qkv) . Wqkv: nine steps of 24 MFMAs, the wave's own d(qkv) rows coming back from L2 by LDS-DMA; LayerNorm
//       backward as in M, whose staging tile hands the next block its dy
// Weights come from a second chain image (transposed shadows, consumption order, LDS layout: rgb-no-more_amd/chain.py).
#include "common.h"
#include <string.h>
#include <type_traits>
#include "internal.h"
#include "ln_bwd_rows.h"
#include "../../include/rgbnm.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2v __attribute__((ext_vector_type(2)));

constexpr int E = 192, HID = 768, HD = 64, HEADS = 3, NTOK = 196, NTILE = 7, NPAD = 224, INNER = HEADS * HD, LDQ = 3 * INNER;
constexpr int NCW = 7, NTHREADS = 64 * (NCW + 1), CTHREADS = 64 * NCW, BM = 32 * NCW;
constexpr int ROWB = 128, ARR = NPAD * ROWB, SLOT = 24576, STAGE = 2 * SLOT, CH = 64, NCHUNK = HID / CH;
constexpr int CP = E + 4;                           // pitch (elements) of the LayerNorm-backward staging tile
constexpr int SMEM = 163840;
#ifndef PIPE_P
#define PIPE_P 6
#endif
#ifndef PIPE_M1
#define PIPE_M1 5
#endif
#ifndef PIPE_X
#define PIPE_X 7
#endif
// ---- M: two weight stages | per wave: gelu' tile, du tile
constexpr int M_STG = 2 * STAGE;                    // 98304
constexpr int TILE = 32 * ROWB;                     // 4096
// ---- LayerNorm-backward epilogues: staging tile at 0, column-sum scratch behind it
typedef rgbnm::LnBwdRows<CTHREADS, BM> LnBwd;
constexpr int RED_OFF = (BM * CP * 2 + 1023) / 1024 * 1024;      // 88064
static_assert(RED_OFF + LnBwd::RED_BYTES <= 132096, "LDS");
// ---- P: slot 0 behind the epilogue's scratch (fetched during the epilogue), slots 1, 2 over the dead staging tile; out tiles
constexpr int P_SLOT0 = 132096, P_SLOT1 = 0, P_TILES = 98304;     // chunks 0, 2 in slot 0, chunk 1 in slot 1; K, V of head 0 arrive meanwhile
static_assert(P_SLOT0 + SLOT <= SMEM && P_TILES + NCW * TILE <= P_SLOT0, "LDS");
// ---- A: the layout of attn3_bwd_kernel
constexpr int A_Q = 0, A_K = ARR, A_V = 2 * ARR, A_G = 3 * ARR, A_L2 = 4 * ARR, A_D = A_L2 + NPAD * 4, A_STG = A_D + NPAD * 4;
constexpr int STG_PITCH = 144, STG_WAVE = 32 * STG_PITCH;
constexpr int A_SIDE = A_STG + NCW * STG_WAVE;      // 1 KB per wave: rows 0..7 of the wave's own next Q tile (see load_own_q)
static_assert(A_SIDE + NCW * 1024 <= SMEM, "LDS");
// ---- X: three weight slots | three sets of d(qkv) row tiles
// chunks 0, 1 over the K, V arrays (fetched during the last head's second phase), chunk 2 over Q; the tile sets fill the rest
constexpr int X_SLOT0 = A_K, X_SLOT1 = A_K + SLOT, X_SLOT2 = 0, X_TILES = A_K + 2 * SLOT, X_TSET = NCW * TILE;
static_assert(X_SLOT1 + SLOT <= X_TILES && X_TILES + 3 * X_TSET <= SMEM && SLOT <= A_K, "LDS");

struct BwdBlk {              // = rgbnm_chain_bwd_block (rgbnm.h) with typed pointers
  const bf16* wimg;          // 12 x (W2^T chunk | W1^T chunk) | 3 projection chunks | 9 qkv chunks
  const float *ln1_g, *ln2_g;
  const bf16* x_in; const float *mean1, *rstd1;
  const bf16* qkv; const float* lse; const bf16* attn; const bf16* x_mid; const float *mean2, *rstd2; const bf16* gp;
  const bf16* dy;            // gradient w.r.t. the block's output
  bf16* du; bf16* dx_mid; bf16* dqkv; bf16* dx;     // dx = gradient w.r.t. the block's input (the next block's dy)
  float *part2, *part1;      // [nimg][2][192] partial sums of d(gamma), d(beta) of LN2 / LN1
};
static_assert(sizeof(BwdBlk) == sizeof(rgbnm_chain_bwd_block), "rgbnm_chain_bwd_block layout");
constexpr int MAX_DEPTH = 12;
struct BwdArgs {              // passed BY VALUE (kernel argument segment): nothing to upload, safe inside a stream capture
  BwdBlk blk[MAX_DEPTH]; bf16* dattn;      // dattn: [M, 192] scratch shared by all blocks (written and read by the same workgroup)
  int depth, nimg;
  float scale;
};

__device__ __forceinline__ int fswz(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}
struct Geo {
  int lane, l31, g, fl;
  unsigned tr0;
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
  f[0].v = pack8(r0, r1);
  f[1].v = pack8(r2, r3);
  f[2].v = pack8(r4, r5);
  f[3].v = pack8(r6, r7);
}
__device__ __forceinline__ Frag<bf16> pfrag(const float (&p)[16], int fi) {
  Frag<bf16> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (bf16)p[fi * 8 + j];
  return f;
}
__device__ __forceinline__ Frag<bf16> rowfrag_x(const unsigned char* arr, unsigned rb, int t, int c) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(arr + (rb ^ (unsigned)(c << 5)) + t * 32 * ROWB);
  return f;
}
// The eight MFMAs of one 32-row tile in the attention phases (scores and their gradient: two accumulators, K = 64 in four steps)
// with their eight row fragments in a PINNED order: four fragments ahead, every MFMA followed by the read of the fragment two steps
// on (into registers an earlier MFMA has read -- the sixteen the transposed fragments of the tile's second half take afterwards).
// Left alone the compiler emits read - wait - MFMA eight times with ONE fragment buffer: an exposed LDS latency per MFMA, 336 of
// them per block and wave.  XB_ROWPIPE=0: the plain loop (experiments).  Same order per accumulator: same bits.
#ifndef XB_ROWPIPE
#define XB_ROWPIPE 1
#endif
__device__ __forceinline__ void row_pair_mma(f32x16& sa, f32x16& da, const unsigned char* arrS, const unsigned char* arrD, unsigned rb,
                                             int t, const Frag<bf16> (&xs)[4], const Frag<bf16> (&xd)[4]) {
#if XB_ROWPIPE
#define XB_SB __builtin_amdgcn_sched_barrier(0)
  Frag<bf16> fs[4], fd[4];
  XB_SB;
  fs[0] = rowfrag_x(arrS, rb, t, 0);
  fd[0] = rowfrag_x(arrD, rb, t, 0);
  fs[1] = rowfrag_x(arrS, rb, t, 1);
  fd[1] = rowfrag_x(arrD, rb, t, 1);
  XB_SB;
  mma(sa, fs[0], xs[0]); XB_SB;
  fs[2] = rowfrag_x(arrS, rb, t, 2); XB_SB;
  mma(da, fd[0], xd[0]); XB_SB;
  fd[2] = rowfrag_x(arrD, rb, t, 2); XB_SB;
  mma(sa, fs[1], xs[1]); XB_SB;
  fs[3] = rowfrag_x(arrS, rb, t, 3); XB_SB;
  mma(da, fd[1], xd[1]); XB_SB;
  fd[3] = rowfrag_x(arrD, rb, t, 3); XB_SB;
  mma(sa, fs[2], xs[2]);
  mma(da, fd[2], xd[2]);
  mma(sa, fs[3], xs[3]);
  mma(da, fd[3], xd[3]);
  XB_SB;
#undef XB_SB
#else
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mma(sa, rowfrag_x(arrS, rb, t, c), xs[c]);
    mma(da, rowfrag_x(arrD, rb, t, c), xd[c]);
  }
#endif
}
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Global stores: XB_NT bit 0 = du (read again only by the weight-gradient launch) non-temporal, bit 1 = the gradient tiles the DMA
// wave writes (d(q, k, v): read back by phase X of the same workgroup), bit 2 = d(attention output) tiles (read back by phase A)
#ifndef XB_NT
#define XB_NT 1
#endif
// Loads: XB_NTLD bit 0 = the gelu' rows of phase M (read once) non-temporal, bit 1 = the K / V / Q / dO arrays of phase A (LDS-DMA aux = 2)
#ifndef XB_NTLD
#define XB_NTLD 0
#endif
template <bool NT, typename V, typename P>
__device__ __forceinline__ void gstore(P* ptr, const V& v) {
#ifdef XB_NOSTORE     // (experiments only, wrong gradients downstream: the kernel without the stores this helper issues)
  if ((XB_NOSTORE & 1) && NT) return;          // bit 0: du (the only non-temporal stream)
  if ((XB_NOSTORE & 2) && !NT) return;         // bit 1: the d(q, k, v) / d(attention output) tiles
#endif
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(ptr));
  else *reinterpret_cast<V*>(ptr) = v;
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
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2v v = {(bf16)a, (bf16)b};
  unsigned r = __builtin_bit_cast(unsigned, v);
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
struct Rows { u32x4 v[12]; };      // 12 operand fragments: features 16 c + 8 g + (0..7) of the lane's token (vit_chain.hip)

template <int NKB>
__device__ __forceinline__ void dma_linear(const unsigned char* src, unsigned char* dst, int lane) {
#pragma unroll
  for (int i = 0; i < NKB; ++i)
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + i * 1024 + lane * 16), (lds_ptr)(dst + i * 1024), 16, 0, 0);
}
// [N][64] bf16 matrix (row stride ld) -> LDS array, chunk q of row r at q ^ fswz(r): 28 pieces of 1 KB; rows >= N repeat row N - 1
__device__ __forceinline__ void dma_matrix_all(const bf16* __restrict__ src, int ld, unsigned char* dst, int lane) {
#pragma unroll 4
  for (int i = 0; i < 28; ++i) {
    const int row = 8 * i + (lane >> 3), pc = lane & 7;
    const int lc = pc ^ fswz(row);
    const int srow = row < NTOK ? row : NTOK - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + (size_t)srow * ld + lc * 8), (lds_ptr)(dst + i * 1024), 16, 0, (XB_NTLD & 2) ? 2 : 0);
  }
}
// the seven waves' own 32 x 64 pieces of a [N][ld] matrix (columns c0 .. c0 + 63) -> row tiles, chunk q of row r at q ^ (r & 7)
__device__ __forceinline__ void dma_row_tiles(const bf16* __restrict__ src, int ld, unsigned char* dst, int lane) {
#pragma unroll 4
  for (int i = 0; i < 28; ++i) {
    const int row = 8 * i + (lane >> 3), pc = lane & 7;
    const int lc = pc ^ (row & 7);
    const int srow = row < NTOK ? row : NTOK - 1;
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + (size_t)srow * ld + lc * 8), (lds_ptr)(dst + i * 1024), 16, 0, 0);
  }
}

// one 64-column piece of the wave's 32 rows -> private tile -> whole 128-byte row pieces -> global (vit_chain.hip tile_out)
__device__ __forceinline__ void tile_out(unsigned char* smem, unsigned stg, const u32x4& p0, const u32x4& p1, const u32x4& p2,
                                         const u32x4& p3, bf16* dst, int ld, int live) {
  const int ln = lane_id_here();
  const unsigned wo = stg + (unsigned)((ln & 31) * ROWB + (((ln >> 5) ^ (ln & 7)) << 4));
  *reinterpret_cast<u32x4*>(smem + wo) = p0;
  *reinterpret_cast<u32x4*>(smem + (wo ^ 32u)) = p1;
  *reinterpret_cast<u32x4*>(smem + (wo ^ 64u)) = p2;
  *reinterpret_cast<u32x4*>(smem + (wo ^ 96u)) = p3;
  own_tile_fence();
  const int rl = ln >> 3, seg = ln & 7;
  const unsigned ro = stg + (unsigned)(rl * ROWB + ((seg ^ rl) << 4));
  bf16* gp = dst + (size_t)rl * ld + seg * 8;
  u32x4 v[4];                     // (all four row pieces requested before the first store; full tiles store behind one uniform branch)
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(smem + ro + i * 8 * ROWB);
  if (live == 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) gstore<(XB_NT & 4) != 0>(gp + (size_t)i * 8 * ld, v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i * 8 + rl < live) gstore<(XB_NT & 4) != 0>(gp + (size_t)i * 8 * ld, v[i]);
  }
  own_tile_fence();
}

// acc (32 output rows x 32 tokens, swapped) += chunk rows [32 ht .. +31] (384 B rows, pchunk swizzle) . x
__device__ __forceinline__ void gemm_k192(f32x16& acc, const unsigned char* sW, int ht, const Rows& x, const Geo& L) {
  int wbase = L.l31 * (E * 2) + ((L.g ^ L.fl) << 4) + ht * 32 * (E * 2);
  asm volatile("" : "+v"(wbase));
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    Frag<bf16> fb, fx;
    fb.v = *reinterpret_cast<const bf16x8*>(sW + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
    fx.v = as_bf16x8(x.v[c]);
    mma(acc, fb, fx);
  }
}

// Both 32-row halves (ht = 0, 1) of a 64 x 192 chunk times x as ONE stream of 24 MFMAs with the weight fragments requested DEPTH
// MFMAs ahead (vit_chain.hip gemm_k192x2: left to itself the scheduler emits read -> wait -> MFMA with one fragment buffer, i.e.
// an exposed LDS latency per MFMA)
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
  __builtin_amdgcn_sched_barrier(0);
}
// transpose reads through the compiler's builtin: ordinary DS loads to the scheduler (requested ahead, waited for at the use)
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

// ---- gradient tiles of the attention backward (attention_v2.hip)
__device__ __forceinline__ void tile_park_private(unsigned char* stg, const f32x16 (&acc)[2], float mul, const Geo& L) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 v = {acc[dt][rq * 4 + 0], acc[dt][rq * 4 + 1], acc[dt][rq * 4 + 2], acc[dt][rq * 4 + 3]};
      store4<bf16>(reinterpret_cast<bf16*>(stg + L.l31 * STG_PITCH) + dt * 32 + rq * 8 + L.g * 4, v * mul);
    }
}
__device__ __forceinline__ void tile_park_rows(unsigned char* arr, int w, const f32x16 (&acc)[2], float mul) {
  const int ln = lane_id_here();
  const unsigned off0 = (unsigned)((w * 32 + (ln & 31)) * ROWB + ((ln & 7) << 4) + (ln >> 5) * 8);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 v = {acc[dt][rq * 4 + 0], acc[dt][rq * 4 + 1], acc[dt][rq * 4 + 2], acc[dt][rq * 4 + 3]};
      store4<bf16>(reinterpret_cast<bf16*>(arr + (off0 ^ (unsigned)((dt * 4 + rq) << 4))), v * mul);
    }
}
template <bool PRIVATE>
__device__ __forceinline__ void tiles_read(const unsigned char* src, int lane, u32x4 (&v)[NTILE][4]) {
  const int rl = lane >> 3, seg = lane & 7;
#pragma unroll
  for (int wv = 0; wv < NTILE; ++wv)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 8 + rl;
      v[wv][i] = PRIVATE ? *reinterpret_cast<const u32x4*>(src + wv * STG_WAVE + r * STG_PITCH + seg * 16)
                         : *reinterpret_cast<const u32x4*>(src + (wv * 32 + r) * ROWB + ((seg ^ (r & 7)) << 4));
    }
}
__device__ __forceinline__ void tiles_write(const u32x4 (&v)[NTILE][4], bf16* __restrict__ g0, size_t ld, int lane) {
  const int rl = lane >> 3, seg = lane & 7;
#pragma unroll
  for (int wv = 0; wv < NTILE; ++wv)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wv * 32 + i * 8 + rl;
      if (row < NTOK) gstore<(XB_NT & 2) != 0>(g0 + (size_t)row * ld + seg * 8, v[wv][i]);
    }
}

#ifdef CHAINB_PROF
__device__ unsigned long long g_chainb_prof[8 * 8 * 128];
#define CP_(i)                                                                                          \
  do {                                                                                                  \
    if (ib == CHAINB_PROF_BLK && blockIdx.x < 8) {                                                      \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                       \
      if ((threadIdx.x & 63) == 0) g_chainb_prof[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 128 + (i)] = t_; \
    }                                                                                                   \
  } while (0)
#ifndef CHAINB_PROF_BLK
#define CHAINB_PROF_BLK 6
#endif
#else
#define CP_(i) do {} while (0)
#endif
// barrier number s of a block: stamps 2 s (arrival) and 2 s + 1 (release)
#define BAR(s) do { CP_(2 * (s)); wg_barrier(); CP_(2 * (s) + 1); } while (0)
// barrier numbers inside a block
constexpr int B_M0 = 0;                 // 0..11   hidden chunks
constexpr int B_E2 = 12;                // 12..18  LN2-backward epilogue (7 barriers)
constexpr int B_P = 19;                 // 19: staging tile dead, 20..22 projection steps, 23: d(attn) stored
constexpr int B_A = 24;                 // 24: start, then 4 per head: 25..36
constexpr int B_X = 37;                 // 37..45  qkv steps
constexpr int B_E1 = 46;                // 46..52  LN1-backward epilogue
constexpr int B_END = 53;               // 53: staging tile dead (dy fragments of the next block are in registers)

// The LayerNorm-backward epilogue shared by M and X (the code of mlp_bwd_kernel's / gemm_nt_kpipe<LNBWD>'s epilogue): acc -> staging
// tile -> rows; writes dx to global AND back into the tile, from which the caller reads its own rows as operand fragments.
// Seven workgroup barriers: b0 .. b0 + 6 (the DMA wave mirrors them).  Compute threads only.
__device__ __forceinline__ void ln_bwd_request(LnBwd& lnb, const bf16* X, const float* mean, const float* rstd, const bf16* R, int img,
                                               int w) {
  const int tid = w * 64 + lane_id_here();
  lnb.request_x(X, E, mean, rstd, img * NTOK, NTOK, tid);
  lnb.request_res(R, E, img * NTOK, NTOK, tid);
}
template <int B0>
__device__ __forceinline__ void ln_bwd_epilogue(unsigned char* smem, const f32x16 (&acc)[6], const bf16* X, const float* mean,
                                                const float* rstd, const bf16* R, const float* gamma, bf16* DX,
                                                float* part, int img, int w, int ib) {
  const int lane_ = lane_id_here();
  const int tid = w * 64 + lane_, rloc = w * 32 + (lane_ & 31), g = lane_ >> 5;
  LnBwd lnb;
  lnb.request_x(X, E, mean, rstd, img * NTOK, NTOK, tid);
  lnb.request_res(R, E, img * NTOK, NTOK, tid);
  wait_lds();
  BAR(B0);                                // every wave is done with the phase's LDS: the staging tile takes its place
  bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = 32 * b + 8 * q + 4 * g;
      const f32x4 v = {acc[b][4 * q + 0], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]};
      store4<bf16>(Cs + rloc * CP + nl, v);
    }
  wait_lds();
  BAR(B0 + 1);
  lnb.run<true>(Cs, CP, DX, E, gamma, true, part, img, reinterpret_cast<float*>(smem + RED_OFF), img * NTOK, NTOK, tid, Cs,
          [&](int k) { if (k == 0) BAR(B0 + 2); else if (k == 1) BAR(B0 + 3); else if (k == 2) BAR(B0 + 4); else BAR(B0 + 5); });
  wait_lds();
  BAR(B0 + 6);                            // dx rows are back in the tile
}
__device__ __forceinline__ void read_own_rows(const unsigned char* smem, int w, Rows& x) {
  const int lane_ = lane_id_here();
  const unsigned char* rp = smem + (w * 32 + (lane_ & 31)) * (CP * 2) + 16 * (lane_ >> 5);      // 8-byte aligned (pitch 392 B)
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const u32x2 lo = *reinterpret_cast<const u32x2*>(rp + 32 * c), hi = *reinterpret_cast<const u32x2*>(rp + 32 * c + 8);
    x.v[c] = u32x4{lo[0], lo[1], hi[0], hi[1]};
  }
}

__global__ __launch_bounds__(NTHREADS) void vit_chain_bwd_kernel(BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int img = blockIdx.x;
  if (img >= p.nimg) return;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int depth = p.depth;

  if (w == NCW) {
#ifdef X_DMAPRIO
    __builtin_amdgcn_s_setprio(X_DMAPRIO);
#endif
    // ================================================================ DMA wave
    const int lane = threadIdx.x & 63;
    for (int ib = depth - 1; ib >= 0; --ib) {
      const BwdBlk& b = p.blk[ib];
      const unsigned char* wimg = reinterpret_cast<const unsigned char*>(b.wimg);
      // ---- M: hidden chunk c into stage c & 1, one chunk ahead
      dma_linear<48>(wimg, smem, lane);
      for (int c = 0; c < NCHUNK; ++c) {
        wait_vm<0>();
        BAR(B_M0 + c);
        if (c + 1 < NCHUNK) dma_linear<48>(wimg + (size_t)(c + 1) * STAGE, smem + ((c + 1) & 1) * STAGE, lane);
      }
      // ---- LN2-backward epilogue: first projection chunk behind the scratch while it runs
      BAR(B_E2);
      dma_linear<24>(wimg + (size_t)NCHUNK * STAGE, smem + P_SLOT0, lane);
#pragma unroll
      for (int k = 1; k < 7; ++k) BAR(B_E2 + k);
      // ---- P (and, behind it, the first head's K, V, Q: their arrays are free from here on)
      const bf16* qkv0 = b.qkv + (size_t)img * NTOK * LDQ;
      const bf16* do0 = p.dattn + (size_t)img * NTOK * INNER;
      bf16* dq0 = b.dqkv + (size_t)img * NTOK * LDQ;
      const unsigned char* stg0 = smem + A_STG;
      auto issue_kv = [&](int h) {
        dma_matrix_all(qkv0 + INNER + h * HD, LDQ, smem + A_K, lane);
        dma_matrix_all(qkv0 + 2 * INNER + h * HD, LDQ, smem + A_V, lane);
      };
      auto issue_q = [&](int h) { dma_matrix_all(qkv0 + h * HD, LDQ, smem + A_Q, lane); };
      auto issue_g = [&](int h) { dma_matrix_all(do0 + h * HD, INNER, smem + A_G, lane); };
      const unsigned char* xw = wimg + (size_t)NCHUNK * STAGE + 3 * SLOT;                 // the nine qkv chunks
      auto issue_xt = [&](int j) { dma_row_tiles(dq0 + j * 64, LDQ, smem + X_TILES + (j % 3) * X_TSET, lane); };
      wait_vm<0>();                                       // chunk 0 (fetched during the epilogue) has landed long ago
      BAR(B_P);                                           // the staging tile is dead
      BAR(B_P + 1);                                       // step 0 starts at once (its chunk landed long ago): the 80 DMA instructions
                                                          // below take this wave 4.4 k cycles to ISSUE, which the compute waves used to
                                                          // spend waiting at this barrier (profiles/r04_chain_bwd_stamps.txt, "P:frags")
      dma_linear<24>(wimg + (size_t)NCHUNK * STAGE + SLOT, smem + P_SLOT1, lane);
      issue_kv(0);
      wait_vm<56>(); BAR(B_P + 2);                        // chunk 1; step 0 is over: slot 0 takes chunk 2
      dma_linear<24>(wimg + (size_t)NCHUNK * STAGE + 2 * SLOT, smem + P_SLOT0, lane);
      wait_vm<0>(); BAR(B_P + 3);                         // chunk 2; step 1 is over: Q of head 0 over slot 1
      issue_q(0);
      BAR(B_P + 4);                                       // d(attention output) of this image is in L2
      // ---- A: the DMA wave of attn3_bwd_kernel for the three heads of this image
      {
        issue_g(0);
        wait_vm<0>();
        BAR(B_A);                                                          // start: K, V, Q, dO of head 0 are in LDS
        for (int h = 0; h < HEADS; ++h) {
          bf16* g0 = dq0 + h * HD;
          wait_vm<0>();                                   // Q, dO landed (and the previous head's stores are done)
          BAR(B_A + 1 + 4 * h);                           // mid
          u32x4 tq[NTILE][4];
          tiles_read<true>(stg0, lane, tq);               // dQ tiles
          wait_lds();
          BAR(B_A + 2 + 4 * h);                           // mid2
          tiles_write(tq, g0, LDQ, lane);
          if (h + 1 < HEADS) issue_kv(h + 1);
          else {                                          // K, V are dead for good: the first two qkv weight chunks take their place
            dma_linear<24>(xw, smem + X_SLOT0, lane);
            dma_linear<24>(xw + SLOT, smem + X_SLOT1, lane);
          }
          wait_vm<0>();
          BAR(B_A + 3 + 4 * h);                           // end
          tiles_read<true>(stg0, lane, tq);               // dK tiles
          wait_lds();
          BAR(B_A + 4 + 4 * h);                           // end2
          tiles_write(tq, g0 + INNER, LDQ, lane);
          tiles_read<false>(smem + A_G, lane, tq);        // dV
          wait_lds();
          if (h + 1 < HEADS) {
            issue_q(h + 1);
            issue_g(h + 1);
          } else {
            // every attention array is dead (the DMA wave is the last to arrive at end2, and the dV tiles are in registers): the
            // waves' own d(qkv) rows come back as row tiles; sets 0 .. 2 = dq of the three heads, stored long ago
            issue_xt(0);
          }
          tiles_write(tq, g0 + 2 * INNER, LDQ, lane);
        }
      }
      // ---- X: chunk j -> slot j % 3, the waves' own d(qkv) rows (columns 64 j ..) -> tile set j % 3; two steps ahead.  The wait in
      // front of barrier j leaves the 52 pieces of group j + 1 (at most) in flight: everything older -- group j, and the d(qkv)
      // stores a later group will read back -- is complete
      for (int j = 0; j < 9; ++j) {
        if (j == 0) wait_vm<28>();                        // tile set 0 (only the 28 dV stores are younger); chunk 0 landed long ago
        else if (j + 1 < 9) wait_vm<52>();
        else wait_vm<0>();
        BAR(B_X + j);
        if (j == 0) {                                     // what steps 1 and 2 need: behind the first barrier, not in front of it
          issue_xt(1);
          dma_linear<24>(xw + 2 * SLOT, smem + X_SLOT2, lane);
          issue_xt(2);
        }
        if (j + 2 < 9 && j >= 1) {                        // slot / tile set (j + 2) % 3 = (j - 1) % 3: free since step j - 1 ended
          dma_linear<24>(xw + (size_t)(j + 2) * SLOT, smem + ((j + 2) % 3 == 0 ? X_SLOT0 : ((j + 2) % 3 == 1 ? X_SLOT1 : X_SLOT2)), lane);
          issue_xt(j + 2);
        }
      }
      // ---- LN1-backward epilogue
#pragma unroll
      for (int k = 0; k < 7; ++k) BAR(B_E1 + k);
      BAR(B_END);
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
  const int live = NTOK - row0 < 32 ? NTOK - row0 : 32;
  const size_t grow0 = (size_t)img * NTOK + row0;
  const float LOG2E = 1.4426950408889634f;
  const float c2 = p.scale * LOG2E;

  Rows dyf;                                                       // dy rows of this lane's token as operand fragments
  {
    const int lane_ = lane_id_here();
    const int rloc = row0 + (lane_ & 31), tok = rloc < NTOK ? rloc : NTOK - 1;
    const bf16* yrow = p.blk[depth - 1].dy + ((size_t)img * NTOK + tok) * E + 8 * (lane_ >> 5);
#pragma unroll
    for (int c = 0; c < 12; ++c) dyf.v[c] = *reinterpret_cast<const u32x4*>(yrow + 16 * c);
  }

  for (int ib = depth - 1; ib >= 0; --ib) {
    const BwdBlk& b = p.blk[ib];
    // ================================================================ M: the loop of mlp_bwd_kernel
    f32x16 acc2[6];
#pragma unroll
    for (int bt = 0; bt < 6; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[bt][r] = 0.f;
    {
      const Geo L = fresh_geo();
      const int fl = L.fl;
      const int woff0 = L.l31 * (E * 2) + ((L.g ^ fl) << 4);
      const unsigned stg = (unsigned)(M_STG + w * 2 * TILE);      // tile 0: gelu' in, tile 1: du out
      bf16x8 gpraw[4];
      auto load_gp = [&](int chunk) {
        const int ln = lane_id_here();           // (row / segment addresses re-derived at every use: nothing to hoist and spill)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
          int rr = row0 + row;
          rr = rr < NTOK ? rr : NTOK - 1;
          const bf16x8* gsrc = reinterpret_cast<const bf16x8*>(b.gp + ((size_t)img * NTOK + rr) * HID + chunk * CH + vec * 8);
          gpraw[i] = (XB_NTLD & 1) ? __builtin_nontemporal_load(gsrc) : *gsrc;
        }
      };
      load_gp(0);
      for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        {
          const int ln = lane_id_here();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
            *reinterpret_cast<bf16x8*>(smem + stg + row * ROWB + ((vec ^ (row & 7)) << 4)) = gpraw[i];
          }
        }
        if (chunk + 1 < NCHUNK) load_gp(chunk + 1);
        wait_lds();
        BAR(B_M0 + chunk);
        const unsigned st_off = (unsigned)((chunk & 1) * STAGE);
        const unsigned char* sW1 = smem + st_off;
        const unsigned w2o = opaque(st_off + (unsigned)(SLOT + L.l31 * ROWB + ((L.g ^ fl) << 4)));
        const unsigned two = opaque(stg + (unsigned)(L.l31 * ROWB + ((L.g ^ (L.l31 & 7)) << 4)));
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
          f32x16 a1;
#pragma unroll
          for (int r = 0; r < 16; ++r) a1[r] = 0.f;
          int wbase = woff0 + ht * 32 * (E * 2);
          asm volatile("" : "+v"(wbase));
#ifdef X_NOPIPE
#pragma unroll
          for (int c = 0; c < 12; ++c) {
            Frag<bf16> fb, fx;
            fb.v = *reinterpret_cast<const bf16x8*>(sW1 + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
            fx.v = as_bf16x8(dyf.v[c]);
            mma(a1, fb, fx);                               // D rows = hidden (rows stored swap23-ed), D cols = tokens
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
              fx.v = as_bf16x8(dyf.v[c]);
              mma(a1, fb[c], fx);                          // D rows = hidden (rows stored swap23-ed), D cols = tokens
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
          // the six W1 fragments of the first 16 hidden columns travel under the gelu' product
          Frag<bf16> fw2[12];
#pragma unroll
          for (int bt = 0; bt < 6; ++bt)
            fw2[bt].v = *reinterpret_cast<const bf16x8*>(smem + (w2o ^ (unsigned)((2 * ht) << 5)) + bt * 32 * ROWB);
          __builtin_amdgcn_sched_barrier(0);
#endif
          Frag<bf16> pg[2];
#pragma unroll
          for (int hs = 0; hs < 2; ++hs) {
            const unsigned to = two ^ (unsigned)((2 * ht + hs) << 5);
            const bf16x8 gpv = *reinterpret_cast<const bf16x8*>(smem + to);
            bf16x8 dv;
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[j] = (bf16)((float)(bf16)a1[8 * hs + j] * (float)gpv[j]);
            pg[hs].v = dv;
            *reinterpret_cast<bf16x8*>(smem + to + TILE) = dv;
          }
#ifdef X_NOPIPE
#pragma unroll
          for (int hs = 0; hs < 2; ++hs) {
            const int s = 2 * ht + hs;
            Frag<bf16> fw[6];
#pragma unroll
            for (int bt = 0; bt < 6; ++bt)
              fw[bt].v = *reinterpret_cast<const bf16x8*>(smem + (w2o ^ (unsigned)(s << 5)) + bt * 32 * ROWB);
#pragma unroll
            for (int bt = 0; bt < 6; ++bt) mma(acc2[bt], fw[bt], pg[hs]);   // D rows = input features, D cols = tokens
          }
#else
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int bt = 0; bt < 6; ++bt)
            fw2[6 + bt].v = *reinterpret_cast<const bf16x8*>(smem + (w2o ^ (unsigned)((2 * ht + 1) << 5)) + bt * 32 * ROWB);
#pragma unroll
          for (int bt = 0; bt < 6; ++bt) mma(acc2[bt], fw2[bt], pg[0]);     // D rows = input features, D cols = tokens
#pragma unroll
          for (int bt = 0; bt < 6; ++bt) mma(acc2[bt], fw2[6 + bt], pg[1]);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // the chunk's du tile out as whole row pieces
        own_tile_fence();
        const int ln = lane_id_here();
        bf16x8 dv0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
          dv0[i] = *reinterpret_cast<const bf16x8*>(smem + stg + TILE + row * ROWB + ((vec ^ (row & 7)) << 4));
        }
        if (live == 32) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
            gstore<(XB_NT & 1) != 0>(b.du + (grow0 + row) * HID + chunk * CH + vec * 8, dv0[i]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
            if (row < live) gstore<(XB_NT & 1) != 0>(b.du + (grow0 + row) * HID + chunk * CH + vec * 8, dv0[i]);
          }
        }
      }
    }
    // ================================================================ d(x_mid) = dy + LN2'(dxn2)
    ln_bwd_epilogue<B_E2>(smem, acc2, b.x_mid, b.mean2, b.rstd2, b.dy, b.ln2_g, b.dx_mid, b.part2, img, w, ib);
    // ================================================================ P: d(attention output) = d(x_mid) . Wproj
    {
      const Geo L = fresh_geo();
      Rows dxf;
      read_own_rows(smem, w, dxf);
      wait_lds();
      BAR(B_P);
      const unsigned stg = (unsigned)(P_TILES + w * TILE);
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        BAR(B_P + 1 + h);
        const unsigned char* sW = smem + (h == 1 ? P_SLOT1 : P_SLOT0);
        f32x16 acc[2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ht][r] = 0.f;
#ifdef X_NOPIPE
        gemm_k192(acc[0], sW, 0, dxf, L);
        gemm_k192(acc[1], sW, 1, dxf, L);
#else
        gemm_k192x2<PIPE_P>(acc[0], acc[1], sW, dxf, L);
#endif
        u32x4 pc[4];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
          for (int hs = 0; hs < 2; ++hs)
#pragma unroll
            for (int d = 0; d < 4; ++d) pc[2 * ht + hs][d] = pack2(acc[ht][8 * hs + 2 * d], acc[ht][8 * hs + 2 * d + 1]);
        tile_out(smem, stg, pc[0], pc[1], pc[2], pc[3], p.dattn + grow0 * INNER + h * HD, INNER, live);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the rows are in L2 before anybody fetches them back
      BAR(B_P + 4);
    }
    // ================================================================ A: attn3_bwd_kernel over the three heads
    {
      unsigned char* Qs = smem + A_Q;
      unsigned char* Ks = smem + A_K;
      unsigned char* Vs = smem + A_V;
      unsigned char* Gs = smem + A_G;
      float* L2_s = reinterpret_cast<float*>(smem + A_L2);
      float* D_s = reinterpret_cast<float*>(smem + A_D);
      unsigned char* stg = smem + A_STG + w * STG_WAVE;
      const Geo L = fresh_geo();
      const int fl = L.fl;
      const unsigned rb0 = (unsigned)(L.l31 * ROWB + ((L.g ^ fl) << 4));
      const int row = row0 + L.l31;
      u32x4 qraw[4], graw[4], oraw[4];
      float lq = 0.f;
      const bf16* qkv0 = b.qkv + (size_t)img * NTOK * LDQ;
      const bf16* do0 = p.dattn + (size_t)img * NTOK * INNER;
      const bf16* o0 = b.attn + (size_t)img * NTOK * INNER;
      // (two requests: with dK and dV still in their accumulators all eight vectors do not fit -- the register allocator parked the
      // first one in scratch, i.e. WAITED for it right behind its own load: the prefetch became a blocking load per head, block and wave)
      auto load_own_g = [&](int h) {
        const int lane_ = lane_id_here();
        const int seg = lane_ & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int r = row0 + i * 8 + (lane_ >> 3);
          r = r < NTOK ? r : NTOK - 1;
          graw[i] = *reinterpret_cast<const u32x4*>(do0 + (size_t)r * INNER + h * HD + seg * 8);
        }
      };
      // ... and even then one vector too many is live: the first Q vector travels by LDS-DMA into a 1 KB side buffer of the wave (lane
      // l's 16 bytes at +16 l) and becomes a register only where it is used
      unsigned char* qside = smem + A_SIDE + w * 1024;
      auto load_own_q = [&](int h) {
        const int lane_ = lane_id_here();
        const int seg = lane_ & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int r = row0 + i * 8 + (lane_ >> 3);
          r = r < NTOK ? r : NTOK - 1;
          const bf16* src = qkv0 + (size_t)r * LDQ + h * HD + seg * 8;
          if (i == 0) __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)qside, 16, 0, 0);
          else qraw[i] = *reinterpret_cast<const u32x4*>(src);
        }
      };
      auto load_own_o = [&](int h) {
        const int lane_ = lane_id_here();
        const int seg = lane_ & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int r = row0 + i * 8 + (lane_ >> 3);
          r = r < NTOK ? r : NTOK - 1;
          oraw[i] = *reinterpret_cast<const u32x4*>(o0 + (size_t)r * INNER + h * HD + seg * 8);
        }
        int rq_ = row0 + (lane_ & 31);
        rq_ = rq_ < NTOK ? rq_ : NTOK - 1;
        lq = b.lse[((size_t)img * HEADS + h) * NTOK + rq_];
      };
      load_own_g(0);
      load_own_q(0);
      load_own_o(0);
      BAR(B_A);                                                   // start: K, V of head 0 are in LDS
      for (int h = 0; h < HEADS; ++h) {
        const bool more = h + 1 < HEADS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own rows
        // ---------------- phase A: wave = 32 queries -> dQ, D
        Frag<bf16> kf[4], vf[4];
        {
          f32x16 dq[2];
          Frag<bf16> qf[4], gf[4];
          const int rl = L.lane >> 3, seg = L.lane & 7;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf16x8 gv = __builtin_bit_cast(bf16x8, graw[i]), ov = __builtin_bit_cast(bf16x8, oraw[i]);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)gv[e] * (float)ov[e];
            d += lane_xor1(d);
            d += lane_xor2(d);
            d += lane_xor4(d);
            if (seg == 0) D_s[row0 + i * 8 + rl] = d;
          }
          qraw[0] = *reinterpret_cast<const u32x4*>(qside + lane_id_here() * 16);        // (landed: vmcnt(0) above)
#pragma unroll
          for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (i * 8 + rl) * STG_PITCH + seg * 16) = qraw[i];
          own_tile_fence();
#pragma unroll
          for (int c = 0; c < 4; ++c) qf[c].v = *reinterpret_cast<const bf16x8*>(stg + L.l31 * STG_PITCH + (2 * c + L.g) * 16);
          own_tile_fence();
#pragma unroll
          for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stg + (i * 8 + rl) * STG_PITCH + seg * 16) = graw[i];
          own_tile_fence();
#pragma unroll
          for (int c = 0; c < 4; ++c) gf[c].v = *reinterpret_cast<const bf16x8*>(stg + L.l31 * STG_PITCH + (2 * c + L.g) * 16);
          const float Dq = D_s[row];
          own_tile_fence();
          const float lq2 = lq * LOG2E;
          if (L.g == 0) L2_s[row] = lq2;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
          unsigned rb = rb0, tr = L.tr0;
          asm volatile("" : "+v"(rb), "+v"(tr));
          const unsigned kt = (unsigned)A_K + tr;
          TileLoop<NTILE>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
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
            if (t * 32 + 32 > NTOK) {
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (t * 32 + acc_row(r, L.lane) >= NTOK) ds[r] = 0.f;
            }
            Frag<bf16> kk[4];
            tfrag4<t>(kt, kk);
            Frag<bf16> sf = pfrag(ds, 0);
            mma(dq[0], kk[0], sf);
            mma(dq[1], kk[1], sf);
            sf = pfrag(ds, 1);
            mma(dq[0], kk[2], sf);
            mma(dq[1], kk[3], sf);
          });
          tile_park_private(stg, dq, p.scale, L);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            kf[c] = rowfrag_x(Ks, rb, w, c);
            vf[c] = rowfrag_x(Vs, rb, w, c);
          }
        }
        wait_lds();
        BAR(B_A + 1 + 4 * h);                                     // mid
        BAR(B_A + 2 + 4 * h);                                     // mid2
        // ---------------- phase B: wave = 32 keys -> dK, dV
        {
          f32x16 dk[2], dv[2];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
          unsigned rb = rb0, tr = L.tr0;
          asm volatile("" : "+v"(rb), "+v"(tr));
          const unsigned qt_ = (unsigned)A_Q + tr, gt_ = (unsigned)A_G + tr;
          TileLoop<NTILE>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            f32x16 sa, da;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
            row_pair_mma(sa, da, Qs, Gs, rb, t, kf, vf);
            float pp[16], ds[16];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
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
            if (t * 32 + 32 > NTOK) {
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (t * 32 + acc_row(r, L.lane) >= NTOK) { pp[r] = 0.f; ds[r] = 0.f; }
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
          });
          if (more) load_own_g(h + 1);
          tile_park_private(stg, dk, p.scale, L);
          if (more) load_own_q(h + 1);                            // (dK's 32 registers are free now)
          wait_lds();
          BAR(B_A + 3 + 4 * h);                                   // end
          tile_park_rows(Gs, w, dv, 1.0f);
        }
        if (more) load_own_o(h + 1);
        wait_lds();
        BAR(B_A + 4 + 4 * h);                                     // end2
      }
    }
    // ================================================================ X: dxn1 = d(qkv) . Wqkv
    f32x16 accx[6];
#pragma unroll
    for (int bt = 0; bt < 6; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[bt][r] = 0.f;
    for (int j = 0; j < 9; ++j) {
      BAR(B_X + j);
      const Geo L = fresh_geo();
      const int fl = L.fl;
      const int set = j % 3;
      const unsigned wo = opaque((unsigned)((set == 0 ? X_SLOT0 : (set == 1 ? X_SLOT1 : X_SLOT2)) + L.l31 * ROWB + ((L.g ^ fl) << 4)));
      const unsigned to = opaque((unsigned)(X_TILES + set * X_TSET + w * TILE + L.l31 * ROWB + ((L.g ^ (L.l31 & 7)) << 4)));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<bf16> fx, fw[6];
        fx.v = *reinterpret_cast<const bf16x8*>(smem + (to ^ (unsigned)(s << 5)));
#pragma unroll
        for (int bt = 0; bt < 6; ++bt)
          fw[bt].v = *reinterpret_cast<const bf16x8*>(smem + (wo ^ (unsigned)(s << 5)) + bt * 32 * ROWB);
#pragma unroll
        for (int bt = 0; bt < 6; ++bt) mma(accx[bt], fw[bt], fx);
      }
#ifndef X_NOPIPE
      // the 28 fragments of the step (4 x (own rows + 6 weight fragments)) PIPE_X reads ahead of the 24 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, PIPE_X, 0);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 28 - PIPE_X) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // ================================================================ dx = d(x_mid) + LN1'(dxn1); the next block's dy
    // (requesting the operand rows of this epilogue three steps earlier would hide their latency, but 104 more live registers in
    // the step loop spill: 29 dwords)
    ln_bwd_epilogue<B_E1>(smem, accx, b.x_in, b.mean1, b.rstd1, b.dx_mid, b.ln1_g, b.dx, b.part1, img, w, ib);
    read_own_rows(smem, w, dyf);
    wait_lds();
    BAR(B_END);
  }
}

}  // namespace

#ifdef CHAINB_PROF
extern "C" int rgbnm_chainb_prof_read(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_chainb_prof), sizeof(unsigned long long) * 8 * 8 * 128) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

size_t rgbnm_chain_bwd_block_bytes(void) { return sizeof(BwdBlk); }

// 1 = not eligible
int rgbnm_vit_chain_bwd(const rgbnm_vit_cfg* c, const rgbnm_chain_bwd_block* blocks, int depth, void* dattn, void* stream) {
  if (!c || !blocks || !dattn || depth <= 0) return RGBNM_EINVAL;
  if (c->dtype != RGBNM_DT_BF16 || c->E != E || c->heads != HEADS || c->N != NTOK || c->B < 1 || depth > MAX_DEPTH) return 1;
  BwdArgs p;
  memcpy(p.blk, blocks, sizeof(BwdBlk) * depth);
  p.dattn = (bf16*)dattn; p.depth = depth; p.nimg = c->B; p.scale = c->attn_scale;
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)vit_chain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  // algorithmic work per block and token (backward = 2 x forward per GEMM, recompute NOT counted): the dX products of
  // the four Linears and the four attention-gradient products dP, dV, dK, dQ = 0.23241 GFLOP per image and block (the kernel also
  // recomputes S = Q K^T: +6 % of MFMA work that no roofline figure is credited with); bytes that must cross HBM: gelu', x_mid, x_in, qkv, attn in, du, d(x_mid), d(qkv), dx
  // out (the d(attention output) scratch and the d(qkv) re-read are L2 hand-offs of one CU, not counted) + the weights once
  const double tok = (double)c->B * NTOK;
  const double flops = depth * tok * 2.0 * (2.0 * E * HID + INNER * E + 3.0 * INNER * E + 4.0 * NTOK * INNER);
  const double bytes = depth * (tok * ((2.0 * HID + 5.0 * E + 2.0 * 3.0 * INNER) * 2.0 + 4 * 4 + HEADS * 4) + 36.0 * SLOT) + tok * E * 2.0;
  const int slot = rgbnm_trace_begin(TR_CHAIN_BWD, flops, bytes, (hipStream_t)stream);
  hipLaunchKernelGGL(vit_chain_bwd_kernel, dim3(c->B), dim3(NTHREADS), SMEM, (hipStream_t)stream, p);
  rgbnm_trace_end(slot, (hipStream_t)stream);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // extern "C"
