// Weight-gradient GEMM, pipelined (bf16): part[s][No,Ki] = dY[tok range s, No]^T . X[tok range s, Ki]
// (reference: the dW of every nn.Linear backward, models/plainvit.py:195,441,443,487,490).
//
// HBM-bound op (reads every activation once, output is tiny), so the design is a memory pipeline:
//   * one 512-thread workgroup per CU, 128 x 192 output tile, 8 waves as 4(M) x 2(N), wave tile 32 x 96;
//   * token tiles of 64 rows stream HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip) into a
//     3-stage ring: two tiles (80 KB per CU, 20 MB chip-wide) are always in flight, one barrier per tile,
//     counted vmcnt (never 0 in the steady state);
//   * tiles keep their natural [token][feature] layout; fragments (reduction axis = tokens) come from
//     ds_read_b64_tr_b16.  64-byte segments are XOR-swizzled through the *source* address (A: seg ^ (row&3),
//     B: seg ^ ((row>>1)&1)) so the 4-row x 64-byte footprint of a 32-lane transpose read is bank-conflict free;
//   * bias gradient (column sums of dY) rides on the MFMA pipe: one extra MFMA per chunk against a ones fragment.
#include "common.h"
#include <string.h>
#include "internal.h"
#include "../../include/rgbnm.h"

// Operand streams: TN_NT bit 0 = the dY slices (read by ONE workgroup, once) non-temporal (aux = 2 of global_load_lds), bit 1 = the
// X tiles (shared by the tiles of a GEMM through their XCD's L2)
#ifndef TN_NT
#define TN_NT 0
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 64;                         // tokens per stage
constexpr int A_ROW = 256, B_ROW = 384;        // bytes per token row (128 / 192 bf16 features)
constexpr int A_STAGE = TK * A_ROW;            // 16 KB
constexpr int B_STAGE = TK * B_ROW;            // 24 KB
constexpr int STAGE = A_STAGE + B_STAGE;       // 40 KB
#ifndef TN_NSTAGE
#define TN_NSTAGE 3
#endif
constexpr int NSTAGE = TN_NSTAGE;              // 3: two tiles (80 KB) in flight; 4: three (120 KB), all 160 KB of LDS
constexpr int SMEM = NSTAGE * STAGE;           // 120 KB

struct TnPipe {
  const bf16* dY; const bf16* X; float* part; float* bpart;
  int ldy, ldx, M, No, Ki, S, kt_per_split, rtiles, ctiles;
};

// Several weight-gradient GEMMs over the same token axis share one launch: with T tiles in total every job is split
// S = 256 / T ways, so the fp32 partials (S x No x Ki per job: the traffic that bounds these kernels next to the operand
// reads) shrink with the number of jobs grouped, and the write burst at the end of the launch happens once.  The four GEMMs of
// one block (21 tiles) give S = 12; the 48 of all twelve blocks behind the one-launch backward (vit_chain_bwd.hip) 252 tiles and
// S = 1: no split at all.
constexpr int TN_MAXJOBS = 52;      // the 48 of twelve ViT blocks + the patch embedding's (4 more tiles: 256 in all) and room to spare
struct TnJobK {                   // per job, in the kernel argument segment
  const bf16* dY; const bf16* X; float* part; float* bpart;
  int ldy, ldx, No, Ki, ctiles, tile_end;       // tile_end: running sum of rtiles * ctiles
  int perm;                                     // > 0: output rows leave permuted (qkv de-interleave, reduce.hip qkv_row_r) -- direct writes only
};
struct TnGroup {
  TnJobK j[TN_MAXJOBS];
  int njobs, S, tiles, M, kt_per_split;
  // packed != 0 (launches without a token split): unit u runs tile unit_tile[u] of job unit_job[u] (255 = idle).  The tiles of one
  // job read the same operand rows, and the 32 units of an XCD share an L2: the host places every job inside ONE XCD's 32 units
  // (first fit, largest first) instead of letting jobs straddle the boundaries of a dense numbering
  int packed;
  unsigned char unit_job[256], unit_tile[256];
};

__device__ __forceinline__ bf16x8 pack8(u32x2 lo, u32x2 hi) {
  u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, v);
}

// all 8 transpose reads of chunk C (16 tokens) for this wave: 1 A fragment + 3 B fragments
template <int C>
__device__ __forceinline__ void tr_chunk(unsigned aA, unsigned aB0, unsigned aB1, unsigned aB2, Frag<bf16>& fa,
                                         Frag<bf16> (&fb)[3]) {
  u32x2 al, ah, b0l, b0h, b1l, b1h, b2l, b2h;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%14\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%15\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%14\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%15\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%14\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(al), "=&v"(ah), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h), "=&v"(b2l), "=&v"(b2h)
      : "v"(aA), "v"(aB0), "v"(aB1), "v"(aB2), "i"(C * 16 * A_ROW), "i"(C * 16 * A_ROW + 4 * A_ROW),
        "i"(C * 16 * B_ROW), "i"(C * 16 * B_ROW + 4 * B_ROW)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
  fa.v = pack8(al, ah);
  fb[0].v = pack8(b0l, b0h);
  fb[1].v = pack8(b1l, b1h);
  fb[2].v = pack8(b2l, b2h);
}

// Experiments (round 5, TN_FPIPE; default 0 = the loop above's read - wait - MFMA per 16 tokens, two waves per SIMD covering each
// other): 1 = the same eight reads through the builtin with the next chunk's reads under the current MFMAs -- MEASURED +50 % on the
// launch: the compiler drains vmcnt in front of every builtin LDS read that follows an LDS-DMA, which empties the ring;
// 2 = asm reads without the wait, lgkmcnt counted by hand: +-0.5 % (the kernel waits for its operand stream, not for LDS).
typedef bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4v* lds_b64_ptr;
__device__ __forceinline__ u32x2 tr_read(const unsigned char* smem, unsigned off) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_ptr)(smem + off)));
}
template <int C>
__device__ __forceinline__ void tr_chunk_b(const unsigned char* sp, unsigned oA, unsigned oB0, unsigned oB1, unsigned oB2,
                                           Frag<bf16>& fa, Frag<bf16> (&fb)[3]) {
  fa.v = pack8(tr_read(sp, oA + C * 16 * A_ROW), tr_read(sp, oA + C * 16 * A_ROW + 4 * A_ROW));
  fb[0].v = pack8(tr_read(sp, oB0 + C * 16 * B_ROW), tr_read(sp, oB0 + C * 16 * B_ROW + 4 * B_ROW));
  fb[1].v = pack8(tr_read(sp, oB1 + C * 16 * B_ROW), tr_read(sp, oB1 + C * 16 * B_ROW + 4 * B_ROW));
  fb[2].v = pack8(tr_read(sp, oB2 + C * 16 * B_ROW), tr_read(sp, oB2 + C * 16 * B_ROW + 4 * B_ROW));
}

#ifndef TN_FPIPE
#define TN_FPIPE 0
#endif
// ... and as one asm block WITHOUT the wait (TN_FPIPE == 2): the compiler drains vmcnt before a builtin LDS read that follows
// LDS-DMA (it cannot tell the ring stages apart), which empties the DMA pipeline; asm reads are invisible to that rule, so the waits
// (lgkmcnt counted by hand: LDS returns in order) are written out next to them
template <int C>
__device__ __forceinline__ void tr_chunk_nw(unsigned aA, unsigned aB0, unsigned aB1, unsigned aB2, u32x2 (&r)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%14\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%15\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%14\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%15\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%14\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%15"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(aA), "v"(aB0), "v"(aB1), "v"(aB2), "i"(C * 16 * A_ROW), "i"(C * 16 * A_ROW + 4 * A_ROW),
        "i"(C * 16 * B_ROW), "i"(C * 16 * B_ROW + 4 * B_ROW)
      : "memory");
}
template <bool V> struct BoolTag { static constexpr bool value = V; };

__global__ __launch_bounds__(512) void gemm_tn_pipe_kernel(TnGroup grp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // 256 workgroups, block b runs on XCD b % 8: unit u = (b % 8) * 32 + b / 8 keeps consecutive units -- the tiles of one
  // token range s, which read the same operand rows -- on one XCD's L2
  const int u = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);
  int s, tile, job = 0;
  if (grp.packed) {
    job = grp.unit_job[u];
    if (job == 255) return;
    tile = grp.unit_tile[u];
    s = 0;
  } else {
    if (u >= grp.S * grp.tiles) return;
    s = u / grp.tiles;
    tile = u % grp.tiles;
    while (tile >= grp.j[job].tile_end) ++job;          // (uniform: scalar loads from the argument segment)
    if (job) tile -= grp.j[job - 1].tile_end;
  }
  TnPipe p;
  int perm;
  {
    const TnJobK& q = grp.j[job];
    p.dY = q.dY; p.X = q.X; p.part = q.part; p.bpart = q.bpart; p.ldy = q.ldy; p.ldx = q.ldx; p.No = q.No; p.Ki = q.Ki;
    p.ctiles = q.ctiles; p.M = grp.M; p.S = grp.S; p.kt_per_split = grp.kt_per_split; p.rtiles = 0;
    perm = q.perm;
  }
  const int rt = tile / p.ctiles, ct = tile % p.ctiles;
  const int r0 = rt * 128, c0 = ct * 192;
  const int kt0 = s * p.kt_per_split;
  const int T = min(p.kt_per_split, p.M / TK - kt0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l31 = lane & 31, g = lane >> 5;

  // ---- global source offsets (elements) of this wave's 5 LDS-DMA instructions per token tile ----
  // A stage = 16 KB = 16 instructions of 1 KB (4 token rows each); B stage = 24 instructions (64/24 rows each).
  int offA[2], offB[3];
  const int vmaxA = (min(p.No - r0, 128) >> 3) - 1;     // clamp feature vectors of a ragged last row tile
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 4 * (w + 8 * j) + (lane >> 4), pv = lane & 15;
    int v = (((pv >> 2) ^ (row & 3)) << 2) | (pv & 3);
    v = v < vmaxA ? v : vmaxA;
    offA[j] = row * p.ldy + r0 + v * 8;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pidx = (w + 8 * j) * 64 + lane;
    const int row = pidx / 24, pv = pidx % 24;
    const int v = (((pv >> 2) ^ ((row >> 1) & 1)) << 2) | (pv & 3);
    offB[j] = row * p.ldx + c0 + v * 8;
  }
  const bf16* gA = p.dY + (size_t)kt0 * TK * p.ldy;
  const bf16* gB = p.X + (size_t)kt0 * TK * p.ldx;
  const size_t stepA = (size_t)TK * p.ldy, stepB = (size_t)TK * p.ldx;

  auto issue = [&](int stage) {
    unsigned char* st = smem + stage * STAGE + w * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(gA + offA[j]), (lds_ptr)(st + j * 8192), 16, 0, (TN_NT & 1) ? 2 : 0);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(gB + offB[j]), (lds_ptr)(st + A_STAGE + j * 8192), 16, 0, (TN_NT & 2) ? 2 : 0);
    gA += stepA;
    gB += stepB;
  };

  // ---- per-lane LDS byte offsets of the transpose reads (stage base and chunk offset added later) ----
  const int k = (lane >> 2) & 3;                              // row inside the 4-row group
  const int within = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int kb = (k >> 1) & 1;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned oA = (8 * g + k) * A_ROW + ((wm ^ k) << 6) + within;
  const unsigned oB0 = A_STAGE + (8 * g + k) * B_ROW + (((3 * wn + 0) ^ kb) << 6) + within;
  const unsigned oB1 = A_STAGE + (8 * g + k) * B_ROW + (((3 * wn + 1) ^ kb) << 6) + within;
  const unsigned oB2 = A_STAGE + (8 * g + k) * B_ROW + (((3 * wn + 2) ^ kb) << 6) + within;

  f32x16 acc[3], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; accb[r] = 0.f;
  }
  const bool do_bias = (p.bpart != nullptr) && (ct == 0) && (wn == 0);
  Frag<bf16> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.v[e] = (bf16)1.0f;

  int st_issue = 0, st_comp = 0;
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i)
    if (i < T) {
      issue(i);
      st_issue = i + 1 == NSTAGE ? 0 : i + 1;
    }
  // two copies of the tile loop, with and without the bias MFMA: a branch inside the chunk sequence would end the scheduling region
  // the fragment pipeline lives in
  auto tiles = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    for (int t = 0; t < T; ++t) {
      // tile t landed; NSTAGE - 2 younger tiles (5 DMA instructions each) may stay in flight
      if (NSTAGE >= 4 && t + 2 < T) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();     // tile t landed for every wave; every wave is done with tile t-1
      if (t + NSTAGE - 1 < T) {
        issue(st_issue);
        st_issue = st_issue == NSTAGE - 1 ? 0 : st_issue + 1;
      }
  #if TN_FPIPE == 2
      const unsigned sb = lds0 + st_comp * STAGE;
      st_comp = st_comp == NSTAGE - 1 ? 0 : st_comp + 1;
      const unsigned aA = sb + oA, aB0 = sb + oB0, aB1 = sb + oB1, aB2 = sb + oB2;
      u32x2 r[2][8];
      tr_chunk_nw<0>(aA, aB0, aB1, aB2, r[0]);
#define CHUNK(C)                                                                     \
      if (C < 3) {                                                                   \
        tr_chunk_nw<C + 1>(aA, aB0, aB1, aB2, r[(C + 1) & 1]);                       \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                           \
      } else {                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
      }                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                             \
      {                                                                              \
        Frag<bf16> fa, fb0, fb1, fb2;                                                \
        fa.v = pack8(r[C & 1][0], r[C & 1][1]);                                      \
        fb0.v = pack8(r[C & 1][2], r[C & 1][3]);                                     \
        fb1.v = pack8(r[C & 1][4], r[C & 1][5]);                                     \
        fb2.v = pack8(r[C & 1][6], r[C & 1][7]);                                     \
        mma(acc[0], fa, fb0);                                                        \
        mma(acc[1], fa, fb1);                                                        \
        mma(acc[2], fa, fb2);                                                        \
        if constexpr (BIAS) mma(accb, fa, ones);                                     \
      }                                                                              \
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(0) CHUNK(1) CHUNK(2) CHUNK(3)
#undef CHUNK
#elif TN_FPIPE
      // fragments of chunk C + 1 are read while the MFMAs of chunk C run (two register sets; LDS returns in order, so the compiler's
      // counted lgkmcnt waits let the next chunk's eight reads stay in flight)
      const unsigned char* sp = smem + st_comp * STAGE;
      st_comp = st_comp == NSTAGE - 1 ? 0 : st_comp + 1;
      Frag<bf16> fa[2], fb[2][3];
      tr_chunk_b<0>(sp, oA, oB0, oB1, oB2, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
#define CHUNK(C)                                                                     \
      if (C < 3) tr_chunk_b<C + 1>(sp, oA, oB0, oB1, oB2, fa[(C + 1) & 1], fb[(C + 1) & 1]);  \
      mma(acc[0], fa[C & 1], fb[C & 1][0]);                                            \
      mma(acc[1], fa[C & 1], fb[C & 1][1]);                                            \
      mma(acc[2], fa[C & 1], fb[C & 1][2]);                                            \
      if constexpr (BIAS) mma(accb, fa[C & 1], ones);                                  \
      if (C < 3) {            /* the next chunk's reads trickle out between this chunk's MFMAs */ \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                             \
        if constexpr (BIAS) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
      }
      CHUNK(0) CHUNK(1) CHUNK(2) CHUNK(3)
#undef CHUNK
#else
      const unsigned sb = lds0 + st_comp * STAGE;
      st_comp = st_comp == NSTAGE - 1 ? 0 : st_comp + 1;
      const unsigned aA = sb + oA, aB0 = sb + oB0, aB1 = sb + oB1, aB2 = sb + oB2;
      Frag<bf16> fa, fb[3];
  #define CHUNK(C)                                   \
      tr_chunk<C>(aA, aB0, aB1, aB2, fa, fb);        \
      mma(acc[0], fa, fb[0]);                        \
      mma(acc[1], fa, fb[1]);                        \
      mma(acc[2], fa, fb[2]);                        \
      if constexpr (BIAS) mma(accb, fa, ones);
      CHUNK(0) CHUNK(1) CHUNK(2) CHUNK(3)
  #undef CHUNK
  #endif
    }
  };
  if (do_bias) tiles(BoolTag<true>());
  else tiles(BoolTag<false>());

  // (perm > 0 only in a launch without a token split, where part / bpart ARE the gradient tensors: the rows leave in the order
  // the reduction would have given them)
  float* part = p.part + (size_t)s * p.No * p.Ki;
  int orow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + wm * 32 + acc_row(r, lane);
    int o = row;
    if (perm > 0) {
      const int inner = perm * 64, s3 = row / inner, rem = row % inner;
      o = (rem / 64) * 192 + (rem % 64) * 3 + s3;
    }
    orow[r] = row < p.No ? o : -1;
  }
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int col = c0 + wn * 96 + b * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (orow[r] >= 0) part[(size_t)orow[r] * p.Ki + col] = acc[b][r];
  }
  if (do_bias && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (orow[r] >= 0) p.bpart[(size_t)s * p.No + orow[r]] = accb[r];
  }
}

// ------------------------------------------------------------------------------------------------ wide tiles (round 6)
// Every workgroup of the kernel above takes in ~18 bytes per cycle through the LDS-DMA path whatever it computes (JPEG-Ti's one
// launch: 256 tiles x 50 176 tokens x 640 B in 770 us = 10.7 TB/s; a JPEG-S pair launch: the same 11 TB/s), and a 128 x 192 tile
// does 77 flops per byte taken in: at E = 192 the operands are read from HBM once and that is the bound, at E = 384 each operand
// row is shared by 2 - 12 tiles through L2, the HBM side idles at 2.8 TB/s and the MFMA pipe at 0.30 -- the intake is the bound.
// Same pipeline with a 192 x 384 tile (128 flops per byte): 8 waves as 2 (M) x 4 (N), wave tile 96 x 96 = 3 x 3 MFMA tiles
// (9 MFMAs per 6 fragment reads instead of 3 per 4), token tiles of 32 rows (36 KB) in a 4-stage ring (108 KB in flight per CU).
// The A tile has the 384-byte rows of the narrow kernel's B tile and its swizzle; the B tile's 768-byte rows put all four rows of a
// transpose read's footprint into the same 256-byte bank window, so its 64-byte segments are XOR-ed with (row & 3) inside
// their group of four, like the narrow A tile.  Eligible: No % 192 == 0 and Ki % 384 == 0 (E = 384 / 768 / 1536 Linears).
constexpr int WTK = 32;
constexpr int WA_ROW = 384, WB_ROW = 768;
constexpr int WA_STAGE = WTK * WA_ROW;          // 12 KB
constexpr int WB_STAGE = WTK * WB_ROW;          // 24 KB
constexpr int WSTAGE = WA_STAGE + WB_STAGE;     // 36 KB = 36 DMA instructions of 1 KB
constexpr int WNSTAGE = 4;
constexpr int WSMEM = WNSTAGE * WSTAGE;         // 144 KB

// the 12 transpose reads of chunk C (16 tokens): 3 A fragments + 3 B fragments, no wait
template <int C>
__device__ __forceinline__ void trw_chunk(unsigned aA0, unsigned aA1, unsigned aA2, unsigned aB0, unsigned aB1, unsigned aB2,
                                          u32x2 (&r)[12]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %12 offset:%18\n\t"
      "ds_read_b64_tr_b16 %1, %12 offset:%19\n\t"
      "ds_read_b64_tr_b16 %2, %13 offset:%18\n\t"
      "ds_read_b64_tr_b16 %3, %13 offset:%19\n\t"
      "ds_read_b64_tr_b16 %4, %14 offset:%18\n\t"
      "ds_read_b64_tr_b16 %5, %14 offset:%19\n\t"
      "ds_read_b64_tr_b16 %6, %15 offset:%20\n\t"
      "ds_read_b64_tr_b16 %7, %15 offset:%21\n\t"
      "ds_read_b64_tr_b16 %8, %16 offset:%20\n\t"
      "ds_read_b64_tr_b16 %9, %16 offset:%21\n\t"
      "ds_read_b64_tr_b16 %10, %17 offset:%20\n\t"
      "ds_read_b64_tr_b16 %11, %17 offset:%21\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
        "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11])
      : "v"(aA0), "v"(aA1), "v"(aA2), "v"(aB0), "v"(aB1), "v"(aB2), "i"(C * 16 * WA_ROW), "i"(C * 16 * WA_ROW + 4 * WA_ROW),
        "i"(C * 16 * WB_ROW), "i"(C * 16 * WB_ROW + 4 * WB_ROW)
      : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512) void gemm_tn_wide_kernel(TnGroup grp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int u = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);        // XCD-contiguous unit numbering (see the narrow kernel)
  int s, tile, job = 0;
  if (grp.packed) {
    job = grp.unit_job[u];
    if (job == 255) return;
    tile = grp.unit_tile[u];
    s = 0;
  } else {
    if (u >= grp.S * grp.tiles) return;
    s = u / grp.tiles;
    tile = u % grp.tiles;
    while (tile >= grp.j[job].tile_end) ++job;
    if (job) tile -= grp.j[job - 1].tile_end;
  }
  const TnJobK& q = grp.j[job];
  const bf16* dY = q.dY; const bf16* X = q.X;
  const int ldy = q.ldy, ldx = q.ldx, No = q.No, Ki = q.Ki, perm = q.perm;
  float* bpart = q.bpart;
  const int rt = tile / q.ctiles, ct = tile % q.ctiles;
  const int r0 = rt * 192, c0 = ct * 384;
  const int kt0 = s * grp.kt_per_split;
  const int T = min(grp.kt_per_split, grp.M / WTK - kt0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int l31 = lane & 31, g = lane >> 5;

  // ---- global source offsets (elements) of the stage's 36 LDS-DMA instructions: 0 - 11 the A tile, 12 - 35 the B tile; wave w
  // issues instructions w, w + 8, w + 16, w + 24 of every tile and instruction 32 + (w & 3) of the tiles with (t & 1) == (w >> 2):
  // 4 or 5 per tile, NINE per two consecutive tiles -- the count the waits below are written for
  auto src_off = [&](int qi) -> int {
    if (qi < 12) {
      const int pidx = qi * 64 + lane;                            // 16-byte pieces, 24 per 384-byte row
      const int row = pidx / 24, pv = pidx % 24;
      const int v = (((pv >> 2) ^ ((row >> 1) & 1)) << 2) | (pv & 3);
      return row * ldy + r0 + v * 8;
    }
    const int pidx = (qi - 12) * 64 + lane;                       // 48 per 768-byte row
    const int row = pidx / 48, pv = pidx % 48;
    const int sg = pv >> 2;
    const int v = ((((sg & ~3) | ((sg & 3) ^ (row & 3)))) << 2) | (pv & 3);
    return row * ldx + c0 + v * 8;
  };
  int off[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) off[j] = src_off(w + 8 * j);
  off[4] = src_off(32 + (w & 3));
  const bf16* gA = dY + (size_t)kt0 * WTK * ldy;
  const bf16* gB = X + (size_t)kt0 * WTK * ldx;
  const size_t stepA = (size_t)WTK * ldy, stepB = (size_t)WTK * ldx;

  auto issue = [&](int stage, int t) {
    unsigned char* st = smem + stage * WSTAGE;
    // instruction w: A tile for every wave (w < 12); w + 8: A for waves 0 - 3, B for the others; the rest: B
    __builtin_amdgcn_global_load_lds((glb_ptr)(gA + off[0]), (lds_ptr)(st + w * 1024), 16, 0, (TN_NT & 1) ? 2 : 0);
    if (w < 4) __builtin_amdgcn_global_load_lds((glb_ptr)(gA + off[1]), (lds_ptr)(st + (w + 8) * 1024), 16, 0, (TN_NT & 1) ? 2 : 0);
    else __builtin_amdgcn_global_load_lds((glb_ptr)(gB + off[1]), (lds_ptr)(st + (w + 8) * 1024), 16, 0, (TN_NT & 2) ? 2 : 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(gB + off[2]), (lds_ptr)(st + (w + 16) * 1024), 16, 0, (TN_NT & 2) ? 2 : 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(gB + off[3]), (lds_ptr)(st + (w + 24) * 1024), 16, 0, (TN_NT & 2) ? 2 : 0);
    if (((t ^ (w >> 2)) & 1) == 0)
      __builtin_amdgcn_global_load_lds((glb_ptr)(gB + off[4]), (lds_ptr)(st + (32 + (w & 3)) * 1024), 16, 0, (TN_NT & 2) ? 2 : 0);
    gA += stepA;
    gB += stepB;
  };

  // ---- per-lane LDS byte offsets of the transpose reads ----
  const int k = (lane >> 2) & 3;
  const int within = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int kb = (k >> 1) & 1;
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned oA[3], oB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    oA[i] = (8 * g + k) * WA_ROW + (((3 * wm + i) ^ kb) << 6) + within;
    const int sg = 3 * wn + i;
    oB[i] = WA_STAGE + (8 * g + k) * WB_ROW + (((sg & ~3) | ((sg & 3) ^ k)) << 6) + within;
  }

  f32x16 acc[3][3], accb[3];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      accb[i][r] = 0.f;
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[i][b][r] = 0.f;
    }
  }
  const bool do_bias = (bpart != nullptr) && (ct == 0) && (wn == 0);
  Frag<bf16> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.v[e] = (bf16)1.0f;

  int st_issue = 0, st_comp = 0;
#pragma unroll
  for (int i = 0; i < WNSTAGE - 1; ++i)
    if (i < T) {
      issue(i, i);
      st_issue = i + 1 == WNSTAGE ? 0 : i + 1;
    }
  auto tiles = [&](auto bias_tag) {
    constexpr bool BIAS = decltype(bias_tag)::value;
    for (int t = 0; t < T; ++t) {
      // tile t landed; the two younger tiles (nine DMA instructions of this wave together) may stay in flight; towards the end
      // one younger tile (four or five: four is the safe count), then none
      if (t + 2 < T) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if (t + 1 < T) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();     // tile t landed for every wave; every wave is done with tile t - 1
      if (t + WNSTAGE - 1 < T) {
        issue(st_issue, t + WNSTAGE - 1);
        st_issue = st_issue == WNSTAGE - 1 ? 0 : st_issue + 1;
      }
      const unsigned sb = lds0 + st_comp * WSTAGE;
      st_comp = st_comp == WNSTAGE - 1 ? 0 : st_comp + 1;
      u32x2 r[12];
#define WCHUNK(C)                                                                                     \
      trw_chunk<C>(sb + oA[0], sb + oA[1], sb + oA[2], sb + oB[0], sb + oB[1], sb + oB[2], r);        \
      {                                                                                               \
        Frag<bf16> fa[3], fb[3];                                                                      \
        fa[0].v = pack8(r[0], r[1]); fa[1].v = pack8(r[2], r[3]); fa[2].v = pack8(r[4], r[5]);        \
        fb[0].v = pack8(r[6], r[7]); fb[1].v = pack8(r[8], r[9]); fb[2].v = pack8(r[10], r[11]);      \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                               \
          _Pragma("unroll") for (int b = 0; b < 3; ++b) mma(acc[i][b], fa[i], fb[b]);                 \
          if constexpr (BIAS) mma(accb[i], fa[i], ones);                                              \
        }                                                                                             \
      }
      WCHUNK(0) WCHUNK(1)
#undef WCHUNK
    }
  };
  if (do_bias) tiles(BoolTag<true>());
  else tiles(BoolTag<false>());

  float* part = q.part + (size_t)s * No * Ki;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int orow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + wm * 96 + i * 32 + acc_row(r, lane);
      int o = row;
      if (perm > 0) {
        const int inner = perm * 64, s3 = row / inner, rem = row % inner;
        o = (rem / 64) * 192 + (rem % 64) * 3 + s3;
      }
      orow[r] = o;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int col = c0 + wn * 96 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(size_t)orow[r] * Ki + col] = acc[i][b][r];
    }
    if (do_bias && l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bpart[(size_t)s * No + orow[r]] = accb[i][r];
    }
  }
}

// (Round 6 pruned the 192 x 192 / 6-wave variant, option tn_square: measured no faster at B = 256 and slower in its reduction.)

}  // namespace

namespace {

int tn_fill(TnPipe& p, const void* dY, int ldy, const void* X, int ldx, float* part, float* bpart, int M, int No, int Ki) {
  if (M % TK || Ki % 192 || No % 8 || ldy % 8 || ldx % 8 || M < TK) return 1;
  p.dY = (const bf16*)dY; p.X = (const bf16*)X; p.part = part; p.bpart = bpart;
  p.ldy = ldy; p.ldx = ldx; p.M = M; p.No = No; p.Ki = Ki;
  p.rtiles = cdiv(No, 128);
  p.ctiles = Ki / 192;
  return 0;
}

}  // namespace

// jobs[0..n): same M; returns 1 when a job is not eligible (nothing launched), S (common split count) through S_out
int rgbnm_launch_tn_pipe_group(const RgbnmTnJob* jobs, int n, int* S_out, hipStream_t st, int* direct_out) {
  if (direct_out) *direct_out = 0;
  if (n < 1 || n > TN_MAXJOBS) return RGBNM_EINVAL;
  static DevOnce attr_set;
  if (attr_set.need()) {
    if (hipFuncSetAttribute((const void*)gemm_tn_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr_set.done();
  }
  TnGroup g;
  int tiles = 0;
  double flops = 0, bytes = 0;
  // wide tiles (192 x 384) when every job of the launch has the shape for them
  bool wide = rgbnm_get_option("tn_wide") != 0;
  for (int i = 0; i < n && wide; ++i) wide = jobs[i].No % 192 == 0 && jobs[i].Ki % 384 == 0 && jobs[i].M % WTK == 0;
  if (wide) {
    // ... and the launch fills the chip with them: a job that brought a workspace for few token slices (sized by its 128 x 192
    // tile count) may leave most CUs without a unit when its tiles get three times as large
    auto units = [&](bool w) {
      int tl = 0, cap = RGBNM_TN_MAX_SPLIT;
      for (int i = 0; i < n; ++i) {
        tl += w ? (jobs[i].No / 192) * (jobs[i].Ki / 384) : cdiv(jobs[i].No, 128) * (jobs[i].Ki / 192);
        if (jobs[i].smax > 0 && jobs[i].smax < cap) cap = jobs[i].smax;
      }
      if (tl > 256) return 0;
      int S = 256 / tl;
      if (S > cap) S = cap;
      const int kt = jobs[0].M / (w ? WTK : TK);
      if (S > kt) S = kt;
      return S * tl;
    };
    const int uw = units(true), un = units(false);
    wide = uw > 0 && 4 * uw >= 3 * un;
  }
  if (wide) {
    static DevOnce attr_w;
    if (attr_w.need()) {
      if (hipFuncSetAttribute((const void*)gemm_tn_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WSMEM) != hipSuccess)
        return RGBNM_ELAUNCH;
      attr_w.done();
    }
  }
  for (int i = 0; i < n; ++i) {
    const RgbnmTnJob& j = jobs[i];
    if (j.M != jobs[0].M) return 1;
    TnPipe t;
    if (tn_fill(t, j.dY, j.ldy, j.X, j.ldx, j.part, j.bpart, j.M, j.No, j.Ki)) return 1;
    if (wide) { t.rtiles = j.No / 192; t.ctiles = j.Ki / 384; }
    tiles += t.rtiles * t.ctiles;
    TnJobK& q = g.j[i];
    q.dY = t.dY; q.X = t.X; q.part = t.part; q.bpart = t.bpart; q.ldy = t.ldy; q.ldx = t.ldx; q.No = t.No; q.Ki = t.Ki;
    q.ctiles = t.ctiles; q.tile_end = tiles; q.perm = 0;
    flops += 2.0 * j.M * (double)j.No * j.Ki;
    bytes += ((double)j.M * j.No + (double)j.M * j.Ki) * 2.0 + (double)j.No * j.Ki * 4.0;
  }
  for (int i = n; i < TN_MAXJOBS; ++i) { g.j[i] = g.j[0]; g.j[i].tile_end = tiles; }
  if (tiles > 256) return 1;
  const int ktiles = jobs[0].M / (wide ? WTK : TK);
  // one workgroup per CU (120 KB LDS) and at most 256 of them: a 257th would wait for a whole first round
  int S = 256 / tiles;
  if (S > RGBNM_TN_MAX_SPLIT) S = RGBNM_TN_MAX_SPLIT;
  for (int i = 0; i < n; ++i)
    if (jobs[i].smax > 0 && S > jobs[i].smax) S = jobs[i].smax;      // a job that brought a workspace for fewer slices
  if (S > ktiles) S = ktiles;
  const int kt_per = cdiv(ktiles, S);
  S = cdiv(ktiles, kt_per);
  g.njobs = n; g.S = S; g.tiles = tiles; g.M = jobs[0].M; g.kt_per_split = kt_per;
  g.packed = 0;
  *S_out = S;
  if (S == 1 && n > 1 && rgbnm_get_option("tn_pack")) {
    // first fit, largest job first, into the eight XCDs' 32 units each; keep the dense numbering when something does not fit
    int order[TN_MAXJOBS], cnt[TN_MAXJOBS], fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) { order[i] = i; cnt[i] = g.j[i].tile_end - (i ? g.j[i - 1].tile_end : 0); }
    for (int i = 1; i < n; ++i)
      for (int k = i; k > 0 && cnt[order[k]] > cnt[order[k - 1]]; --k) { const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t; }
    memset(g.unit_job, 255, sizeof(g.unit_job));
    memset(g.unit_tile, 0, sizeof(g.unit_tile));
    bool ok = true;
    for (int oi = 0; oi < n && ok; ++oi) {
      const int i = order[oi];
      int x = 0;
      while (x < 8 && fill[x] + cnt[i] > 32) ++x;
      if (x == 8) { ok = false; break; }
      for (int t = 0; t < cnt[i]; ++t) {
        g.unit_job[x * 32 + fill[x] + t] = (unsigned char)i;
        g.unit_tile[x * 32 + fill[x] + t] = (unsigned char)t;
      }
      fill[x] += cnt[i];
    }
    g.packed = ok ? 1 : 0;
  }
  if (S == 1 && direct_out) {
    // no token split: the "partials" are the result.  Write the gradient tensors from the kernel (row permutation included) and
    // spare the reduction launch its copy of every weight gradient (22.6 MB read + written per JPEG-Ti step)
    bool ok = true;
    for (int i = 0; i < n; ++i)
      ok = ok && jobs[i].dW && !jobs[i].accumulate && (jobs[i].bpart == nullptr || jobs[i].db) &&
           (jobs[i].perm_heads <= 0 || jobs[i].No == 3 * jobs[i].perm_heads * 64);
    if (ok) {
      for (int i = 0; i < n; ++i) {
        g.j[i].part = jobs[i].dW;
        g.j[i].bpart = jobs[i].bpart ? jobs[i].db : nullptr;
        g.j[i].perm = jobs[i].perm_heads > 0 ? jobs[i].perm_heads : 0;
      }
      for (int i = n; i < TN_MAXJOBS; ++i) g.j[i] = g.j[0];
      for (int i = n; i < TN_MAXJOBS; ++i) g.j[i].tile_end = tiles;
      *direct_out = 1;
    }
  }
  const int slot = rgbnm_trace_begin(TR_TN, flops, bytes, st);
  if (wide) hipLaunchKernelGGL(gemm_tn_wide_kernel, dim3(256), dim3(512), WSMEM, st, g);
  else hipLaunchKernelGGL(gemm_tn_pipe_kernel, dim3(256), dim3(512), SMEM, st, g);
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_launch_tn_pipe(const void* dY, int ldy, const void* X, int ldx, float* part, float* bpart, int M, int No,
                         int Ki, int* S_out, hipStream_t st, int smax) {
  RgbnmTnJob j;
  j.dY = dY; j.X = X; j.part = part; j.bpart = bpart; j.ldy = ldy; j.ldx = ldx; j.M = M; j.No = No; j.Ki = Ki;
  j.dW = nullptr; j.db = nullptr; j.perm_heads = 0; j.accumulate = 0; j.smax = smax;
  return rgbnm_launch_tn_pipe_group(&j, 1, S_out, st);
}
