// Fused FeedForwardBlock forward of JPEG-Ti (reference: models/plainvit.py:481-491 inside ResidualAdd :475-479, followed
// by the next LayerNorm :513/:522), bf16, E = 192, hidden = 768:
//     x_out = x_mid + fc2(gelu(fc1(xn2)))      [ ; xn_next = LayerNorm(x_out) ]
// in ONE launch: the [M,768] hidden activation is produced and consumed on chip; what leaves the CU are the two tensors the
// backward needs (gelu(u) for dW2, gelu'(u) for dX) and the 192-wide outputs.
//
// Why: measured on MI355X (tools/nt_probe.py) a CU ingests ~25 GB/s through the vector-memory path whether the line comes
// from HBM, the Infinity Cache or its own XCD's L2 -- 256 CUs x 25 GB/s is the whole chip's ~6.4 TB/s -- so a kernel's time
// is (bytes its CU loads + bytes it stores) and re-reading an intermediate costs exactly what an HBM read costs.  Separate
// fc1 / fc2 launches load 1044 KB and store 752 KB per image; this kernel loads 740 KB (150 KB activations + both weight
// matrices, streamed once per image as 48 KB chunks) and stores the same 752 KB minus nothing re-read: the 301 KB read of
// gelu(u) by fc2 and fc2's second weight/activation pipeline disappear, and one launch boundary per block with them.
//
// Structure: one workgroup per image-sized row panel (M / 256 rows, <= 224) = 7 compute waves x 32 rows + 1 DMA wave.
//   * the wave's 32 x 192 LayerNorm-ed input rows live in registers as 12 MFMA operand fragments for the whole kernel;
//   * the DMA wave streams hidden chunks of 64: W1[64 x 192] (24 KB) and W2[192 x 64] (24 KB) into a 2-stage LDS ring,
//     one chunk ahead, swizzled through the DMA source address (conflict-free ds_read_b128, same maps as gemm_nt_wres /
//     gemm_nt_kpipe), one workgroup barrier per chunk;
//   * per chunk a wave runs 24 MFMAs of fc1 with swapped operands (D rows = hidden, D cols = tokens), adds the bias,
//     rounds to bf16 as the unfused path does, evaluates gelu / gelu' once, and feeds gelu(u) STRAIGHT FROM REGISTERS into
//     the 24 fc2 MFMAs: a lane owns one token and, because the rows of the W1 chunk are stored with index bits 2 and 3
//     swapped, its 8 accumulator registers of a k-step are 8 consecutive hidden units = one B-operand fragment;
//   * gelu(u) / gelu'(u) leave through a private 2 x 4 KB staging tile per wave as whole 128-byte row pieces;
//   * epilogue = the residual + LayerNorm epilogue of gemm_nt_kpipe (same arithmetic, same bits as ln_fwd_kernel).
// Results are bit-identical to the two-launch path (same operand rounding, same fp32 summation order):
// tests/test_fastpath_model.py::test_fused_mlp_forward_equals_the_two_gemm_path.
//
// Measured (B = 256): 78 us against 48 + 37 us for the two launches; the weight stream is free (a
// variant that fetches only two chunks: -1 us).  Why 78 and not the ~35 us of its MFMAs or its VALU work alone -- per-wave
// s_memtime stamps (build with RGBNM_HIPCC_FLAGS=-DMLP_TRACE, tools/mlp_trace.py), SQ counters (tools/gpu_stalls.sh) and a
// two-wave micro-benchmark (tools/pipe_probe.py) agree:
//   * one wave's GELU arithmetic already saturates its SIMD's VALU issue (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.35 cycles per
//     instruction; two waves running the GELU code on one SIMD take exactly twice as long as one);
//   * the packed-fp32 and bf16-convert instructions the GELU is made of run 1.8-1.9 x slower while the other wave of the SIMD
//     issues MFMAs (v_fma_f32: 1.2 x), and an MFMA wave next to an older GELU wave is slowed 1.65 x: MFMA and VALU work on one
//     SIMD overlap by only ~12 %, so a workgroup's time is close to the SUM of its MFMA, VALU and store phases however the waves
//     are arranged;
//   * the older wave of a SIMD wins every arbitration: per 64-unit chunk waves 0-3 run their two halves in 2 x 2850 ticks
//     and then wait ~2800 at the barrier while waves 4-6, starved until then, finish (first half 6000 ticks, second 2650).
// Variants built on the opposite assumption, all bit-identical and all measured slower, are not kept: 32-wide granules in a
// 4-stage ring with fc1 of granule h + 1 issued next to the GELU of granule h (82 us); the same with the 24 MFMAs woven
// into the GELU instruction stream by hand (82 us: the lone dependent GELU chain of a weave step issues at half rate);
// a barrier-free version (LDS flags, waves drifting up to three granules apart: 85 us); the two waves of a SIMD in opposite
// roles per barrier-separated slot, one on GELU while the other runs fc2 / fc1 / stores (84 us: the GELU slot stretches from
// 1700 to 2600 ticks next to the MFMA wave); scalar instead of packed GELU arithmetic (88 us).
#include "common.h"
#include <mutex>
#include "internal.h"
#include "ln_bwd_rows.h"
#include "../../include/rgbnm.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int E = 192, H = 768, CH = 64, NCHUNK = H / CH;
constexpr int NCW = 7, NTHREADS = 64 * (NCW + 1), CTHREADS = 64 * NCW, BM = 32 * NCW;   // 224 panel rows
constexpr int W1_STAGE = CH * E * 2;                // 24 KB: [64 hidden rows][384 B]
constexpr int W2_STAGE = E * CH * 2;                // 24 KB: [192 feature rows][128 B]
constexpr int STAGE = W1_STAGE + W2_STAGE, NSTAGE = 2;
constexpr int STG_OFF = NSTAGE * STAGE;             // per-wave staging: gelu tile 4 KB | gelu' tile 4 KB
constexpr int STG_TILE = 32 * CH * 2, STG_WAVE = 2 * STG_TILE;
constexpr int B1_OFF = STG_OFF + NCW * STG_WAVE;    // 768 floats
constexpr int B2_OFF = B1_OFF + H * 4;              // 192 floats
constexpr int FLAG_OFF = B2_OFF + E * 4;            // one int: the last chunk whose tiles of waves 4-6 the DMA wave has taken
constexpr int SMEM = FLAG_OFF + 16;                 // 159,504 B
constexpr int CP = E + 4;                           // final staging pitch (elements)
// ---- forward kernel: [GELU table image | ] weight ring | per-wave staging tiles sized by the LIVE rows | bias ring | b2 | flag.
// With 32 x 7 = 224 staging rows but 196 live ones (one image per workgroup at B = 256) the seventh wave needs 4 rows, not
// 32: sizing the tiles by the rows that exist frees 7 KB, the fc1 bias arriving per chunk (256 B with each weight chunk)
// instead of resident (3 KB) frees the rest of what the 13 KB table image needs.
constexpr int F_TAB_BYTES = 13328;                  // GELU table image at LDS offset 0 (16-bit packed byte offsets address it directly)
constexpr int F_TAB_ROWS = 196;                     // most panel rows the table leaves room for
constexpr int F_B1R = 2 * CH * 4;                   // bias ring: one chunk's 64 floats per stage
__host__ __device__ constexpr int f_smem(int tab, int rows) {
  return (tab ? F_TAB_BYTES : 0) + NSTAGE * STAGE + rows * 2 * CH * 2 + F_B1R + E * 4 + 16;
}
static_assert(f_smem(1, F_TAB_ROWS) <= 160 * 1024, "LDS");
constexpr int LN_GROUPS = CTHREADS / 8, LN_ITERS = BM / LN_GROUPS;      // 56 row groups of 8 lanes, 4 rounds
static_assert(SMEM <= 160 * 1024, "LDS");
static_assert(BM * CP * 2 <= NSTAGE * STAGE, "final staging tile lives in the ring");

struct MlpArgs {
  const bf16* X; const bf16* W1; const float* b1; const bf16* W2; const float* b2; const bf16* R;
  bf16* G; bf16* GP; bf16* Y;
  int ldx, ldr, ldg, ldy;
  int M, rows_per_wg, npanels, cold;       // cold: 1 = nt stores, 2 = sc1 (write-through) stores for the saved tensors
  int offload;                             // 1: the DMA wave stores the gelu / gelu' tiles of waves 4-6 (option mlp_dmast)
  const unsigned* tab_img;                 // g_gelu_img (device symbol), TAB kernels only
  unsigned kneg, kpos, klo, koff, ksgn;    // packed-key constants of the table window
  const float* gamma; const float* beta; bf16* Y2; float* mean_o; float* rstd_o; float eps; int ldy2;   // gamma == null: no LN
};

__device__ __forceinline__ int fswz(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}
__device__ __forceinline__ int pchunk(int lc, int row) { return (lc & ~7) | ((lc & 7) ^ fswz(row)); }

// x_out = acc2 + b2 + R [, LayerNorm of x_out]: the arithmetic of gemm_nt_kpipe's EPI_RES_LN (same bits as ln_fwd_kernel).
// Called by the compute waves only (the DMA wave has ended); uses the weight ring as the staging tile.
__device__ __forceinline__ void mlp_epilogue(const MlpArgs& p, unsigned char* smem, const float* B2s, f32x16 (&acc2)[6], int m0,
                                             int rows, int tid, int w, int l31, int g) {
  const int rloc = 32 * w + l31;
  // ---------------- epilogue: x_out = acc2 + b2 + R [, LayerNorm]: the arithmetic of gemm_nt_kpipe's EPI_RES_LN
  const int ctid = tid;                                   // compute threads are 0 .. 447
  const int l8 = ctid & 7, grp = ctid >> 3;
  bf16x8 lr[LN_ITERS][3];
#pragma unroll
  for (int it = 0; it < LN_ITERS; ++it) {
    const int row = it * LN_GROUPS + grp, rr = m0 + (row < rows ? row : rows - 1);
#pragma unroll
    for (int v = 0; v < 3; ++v) lr[it][v] = *reinterpret_cast<const bf16x8*>(p.R + (size_t)rr * p.ldr + v * 64 + l8 * 8);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave is done with the ring: it becomes the staging tile
  bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = 32 * b + 8 * q + 4 * g;
      f32x4 v = {acc2[b][4 * q + 0], acc2[b][4 * q + 1], acc2[b][4 * q + 2], acc2[b][4 * q + 3]};
      v += *reinterpret_cast<const f32x4*>(B2s + nl);
      store4<bf16>(Cs + rloc * CP + nl, v);
    }
  const bool do_ln = p.gamma != nullptr;
  f32x4 gm[3][2], bt[3][2];
  if (do_ln) {
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        gm[v][hf] = *reinterpret_cast<const f32x4*>(p.gamma + v * 64 + l8 * 8 + hf * 4);
        bt[v][hf] = *reinterpret_cast<const f32x4*>(p.beta + v * 64 + l8 * 8 + hf * 4);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
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
          xb[i] = (bf16)((float)(i < 4 ? c0[i & 3] : c1[i & 3]) + (float)lr[it][v][i]);
          xv[v][i] = (float)xb[i];
        }
        *reinterpret_cast<bf16x8*>(p.Y + (size_t)(m0 + row) * p.ldy + v * 64 + l8 * 8) = xb;
        s0 += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
        s1 += xv[v][4] + xv[v][5] + xv[v][6] + xv[v][7];
      }
      if (do_ln) {
        const float mu = group8_pair_sum(s0, s1) * (1.f / E);
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d0 = xv[v][i] - mu, d1 = xv[v][4 + i] - mu;
            q0 = __builtin_fmaf(d0, d0, q0);
            q1 = __builtin_fmaf(d1, d1, q1);
          }
        const float rs = rsqrtf(__builtin_fmaf(group8_pair_sum(q0, q1), 1.f / E, p.eps));
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
  }
  }
}

#ifdef MLP_TRACE
// experiments only (RGBNM_HIPCC_FLAGS=-DMLP_TRACE): per-wave s_memtime stamps of the first workgroups, read by tools/mlp_trace.py
__device__ unsigned long long g_mlp_trace[16 * 8 * 80];
#define MLP_STAMP(i)                                                                                  \
  do {                                                                                                \
    if (blockIdx.x < 16) {                                                                            \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                     \
      if ((threadIdx.x & 63) == 0) g_mlp_trace[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 80 + (i)] = t_; \
    }                                                                                                 \
  } while (0)
#else
#define MLP_STAMP(i)
#endif
// -DMLP_TRACE=2 stamps the backward kernel instead of the forward (same slots, same reader)
#if defined(MLP_TRACE) && MLP_TRACE == 2
#define MLP_FSTAMP(i)
#define MLP_BSTAMP(i) MLP_STAMP(i)
#else
#define MLP_FSTAMP(i) MLP_STAMP(i)
#define MLP_BSTAMP(i)
#endif

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// TAB = false: the arithmetic GELU, fc1 bias resident, 32-row staging tiles (any panel height up to 224 rows).
// TAB = true : GELU by table lookup; LDS = table image | ring | staging tiles of the LIVE rows | bias ring | b2 | flag
//              (panel height <= F_TAB_ROWS; the host passes the window constants it read back once in rgbnm_gelu_table_init).
template <bool TAB>
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int panel = blockIdx.x;
  if (panel >= p.npanels) return;
  const int m0 = panel * p.rows_per_wg;
  const int rows = min(p.rows_per_wg, p.M - m0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  constexpr bool tab = TAB;
  unsigned char* ring = smem + (TAB ? F_TAB_BYTES : 0);
  unsigned char* stg_all = ring + NSTAGE * STAGE;          // TAB: wave v has 2 tiles of live(v) rows at 256 * min(32 v, rows_per_wg)
  float* B1r = reinterpret_cast<float*>(TAB ? stg_all + p.rows_per_wg * (2 * CH * 2) : smem + B1_OFF);   // TAB: [2][64]; else [768]
  float* B2s = TAB ? B1r + 2 * CH : reinterpret_cast<float*>(smem + B2_OFF);
  volatile int* flag = reinterpret_cast<volatile int*>(TAB ? reinterpret_cast<unsigned char*>(B2s + E) : smem + FLAG_OFF);
  auto live_rows = [&](int v) { return TAB ? max(0, min(32, p.rows_per_wg - 32 * v)) : 32; };
  auto stg_of = [&](int v) { return TAB ? stg_all + min(32 * v, p.rows_per_wg) * (2 * CH * 2) : smem + STG_OFF + v * STG_WAVE; };
  MLP_FSTAMP(0);
#if defined(MLP_TRACE) && MLP_TRACE != 2
  if (blockIdx.x < 16 && (threadIdx.x & 63) == 0) g_mlp_trace[(blockIdx.x * 8 + w) * 80 + 79] = __builtin_amdgcn_s_memrealtime();
#endif

  if (w == NCW) {
    // ---------------- DMA wave: table image and b2 once, then one 48 KB weight chunk (+ its 64 fc1 biases) per barrier, one
    // chunk ahead of the math
    if (tab) {
      const int npiece = F_TAB_BYTES / 16;                            // 833 pieces of 16 bytes
      for (int i = 0; i * 64 < npiece; ++i)
        if (i * 64 + lane < npiece)
          __builtin_amdgcn_global_load_lds((glb_ptr)(p.tab_img + (i * 64 + lane) * 4), (lds_ptr)(smem + i * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < H / 64; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr)(p.b1 + 64 * i + lane), (lds_ptr)(B1r + 64 * i), 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < E / 64; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr)(p.b2 + 64 * i + lane), (lds_ptr)(B2s + 64 * i), 4, 0, 0);
    const int rl = lane >> 3, pc = lane & 7;
    auto issue = [&](int chunk) {
      unsigned char* st = ring + (chunk & 1) * STAGE;
      const bf16* w1 = p.W1 + (size_t)chunk * CH * E;
#pragma unroll
      for (int i = 0; i < W1_STAGE / 1024; ++i) {
        const int pidx = 64 * i + lane, row = pidx / 24, c24 = pidx % 24;
        const int hrow = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);     // LDS row r holds hidden unit swap23(r)
        __builtin_amdgcn_global_load_lds((glb_ptr)(w1 + hrow * E + pchunk(c24, row) * 8), (lds_ptr)(st + i * 1024), 16, 0, 0);
      }
      const bf16* w2 = p.W2 + chunk * CH;
#pragma unroll
      for (int i = 0; i < W2_STAGE / 1024; ++i) {
        const int r8 = 8 * i + rl;
        __builtin_amdgcn_global_load_lds((glb_ptr)(w2 + (size_t)r8 * H + ((pc ^ fswz(r8)) * 8)),
                                         (lds_ptr)(st + W1_STAGE + i * 1024), 16, 0, 0);
      }
      if (tab) __builtin_amdgcn_global_load_lds((glb_ptr)(p.b1 + chunk * CH + lane), (lds_ptr)(B1r + (chunk & 1) * CH), 4, 0, 0);
    };
    // offload: the younger wave of each SIMD pair (4, 5, 6) is the critical path of a chunk (stamps: it is starved while the
    // older one runs, then finishes alone and stores); its tile stores are taken over by this wave, which is idle between two
    // weight chunks.  After barrier c + 1 the tiles of chunk c are complete: read them into registers, raise the flag (the
    // owners poll it before they overwrite the tiles), issue the stores, then the next weight chunk (it has a whole chunk to land).
    bf16x8 tv[3][2][4];
    const int trow = lane >> 3, tvec = lane & 7;
    auto take_tiles = [&]() {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned char* sq = stg_of(4 + q);
        const int lr = live_rows(4 + q);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = trow + 8 * i;
            if (row < lr)
              tv[q][t][i] = *reinterpret_cast<const bf16x8*>(sq + t * lr * (CH * 2) + row * (CH * 2) + ((tvec ^ (row & 7)) << 4));
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto store_tiles = [&](int chunk) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 32 * (4 + q) + trow + 8 * i;
          if (row < rows) {
            const size_t go = (size_t)(m0 + row) * p.ldg + chunk * CH + tvec * 8;
            *reinterpret_cast<bf16x8*>(p.G + go) = tv[q][0][i];
            *reinterpret_cast<bf16x8*>(p.GP + go) = tv[q][1][i];
          }
        }
    };
    if (p.offload && lane == 0) *flag = 0;
    issue(0);
    for (int c = 0; c < NCHUNK; ++c) {
      // chunk c (and, the first time, the table and b2) landed -- and the tile stores issued before it (vmcnt retires in order;
      // the number of store instructions depends on the valid rows, so no counted wait)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      MLP_FSTAMP(2 + 5 * c);
      __builtin_amdgcn_s_barrier();                       // ... and every compute wave is done with chunk c - 1
      MLP_FSTAMP(3 + 5 * c);
      if (p.offload && c >= 1) {
        take_tiles();
        if (lane == 0) *flag = c;
        store_tiles(c - 1);
      }
      if (c + 1 < NCHUNK) issue(c + 1);
      MLP_FSTAMP(4 + 5 * c);
    }
    MLP_FSTAMP(62);
#if defined(MLP_TRACE) && MLP_TRACE != 2
    if (blockIdx.x < 16 && (threadIdx.x & 63) == 0) g_mlp_trace[(blockIdx.x * 8 + w) * 80 + 78] = __builtin_amdgcn_s_memrealtime();
#endif
    if (p.offload) {                                      // the last chunk's tiles: after the epilogue's first barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      take_tiles();
      store_tiles(NCHUNK - 1);
    }
    return;                                               // ended waves drop out of the workgroup barrier
  }

  // ---------------- compute waves
  const int rloc = 32 * w + l31;
  const bf16* xrow = p.X + (size_t)(m0 + (rloc < rows ? rloc : rows - 1)) * p.ldx;
  bf16x8 fa[E / 16];
#pragma unroll
  for (int c = 0; c < E / 16; ++c) fa[c] = *reinterpret_cast<const bf16x8*>(xrow + (2 * c + g) * 8);

  f32x16 acc2[6];
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[b][r] = 0.f;

  const int fl = fswz(l31);
  const int woff0 = l31 * (E * 2) + ((g ^ fl) << 4);      // W1 fragment c of LDS row l31: (woff0 ^ ((c % 4) << 5)) + 128 (c / 4)
  unsigned char* stg = stg_of(w);
  const int lrw = live_rows(w);                            // rows of this wave's two tiles (gelu | gelu')
  const bool lane_live = TAB ? l31 < lrw : true;
  // table constants (SGPRs, from the host): splats for the packed 16-bit key arithmetic
  //   kneg: u16 min -- negative magnitudes stop at N1 (gelu = -0, gelu' constant beyond)
  //   kpos: i16 min -- positive magnitudes stop at P1 (key only)      klo: u16 max -- everything tiny shares the entry below the window
  //   koff: byte offset 4 (a - (A0 - 1)) mod 2^16                      ksgn: the negative half of the image starts here
  const unsigned kneg = p.kneg, kpos = p.kpos, klo = p.klo, koff = p.koff, ksgn = p.ksgn;
  unsigned k4v = 0x00040004u;                              // (a VGPR: an instruction takes one scalar operand)
  asm volatile("" : "+v"(k4v));
#if defined(MLP_TRACE) && MLP_TRACE != 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MLP_FSTAMP(1);

  const bool handed = p.offload && w >= 4;                // this wave's tiles leave through the DMA wave
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    MLP_FSTAMP(2 + 5 * chunk);
    if (p.offload) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tile writes of the previous chunk are in LDS
    __builtin_amdgcn_s_barrier();
    MLP_FSTAMP(3 + 5 * chunk);
    const unsigned char* sW1 = ring + (chunk & 1) * STAGE;
    const unsigned char* sW2 = sW1 + W1_STAGE;
    const float* bch = TAB ? B1r + (chunk & 1) * CH : B1r + chunk * CH;
#pragma unroll
    for (int ht = 0; ht < 2; ++ht) {
      f32x16 a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.f;
      int wbase = woff0 + ht * 32 * (E * 2);
      asm volatile("" : "+v"(wbase));
#pragma unroll
      for (int c = 0; c < E / 16; ++c) {
        Frag<bf16> fb, fx;
        fb.v = *reinterpret_cast<const bf16x8*>(sW1 + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
        fx.v = fa[c];
        mma(a1, fb, fx);                                   // D rows = hidden (LDS row order), D cols = tokens
      }
      Frag<bf16> pg[2];
      if (handed && ht == 0 && chunk > 0) {               // the DMA wave has taken the previous chunk's tiles (normally long ago)
        while (__builtin_amdgcn_readfirstlane(*flag) < chunk) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        // registers 8 hs .. 8 hs + 7 of this lane = hidden units h0 .. h0 + 7 of the token l31 (rows were stored swap23-ed)
        const int hl = 32 * ht + 16 * hs + 8 * g;
        const float* bp = bch + hl;
        const f32x4 bl = *reinterpret_cast<const f32x4*>(bp), bh = *reinterpret_cast<const f32x4*>(bp + 4);
        bf16x8 gv, dv;
        if constexpr (TAB) {
          // gelu / gelu' of the eight elements: bf16 bits p of two elements -> packed 16-bit keys -> one ds_read_b32 of
          // {D | gelu' << 16} per element (all eight in flight, one wait) -> |gelu| = max(a, 0x80) - D  (see the table comment
          // above gelu_full_kernel).  Packed 16-bit VALU by inline asm: the compiler's own selection of the same arithmetic
          // needed 28 instructions per pair against 17 here.
          // Non-finite pre-activations: the unsigned clamp at N1 also catches sign-bit-set magnitudes of exponent row 255, so a
          // NEGATIVE-signed NaN or -inf gives gelu = -0 with the finite large-negative gelu' (positive NaN / +inf pass through
          // as gelu = u, gelu' = 1).  A NaN pre-activation comes from a NaN LayerNorm output or weight, which also reaches the
          // residual stream and the logits through the other operands, so overflow checks on the loss / gradients still see it;
          // an exact patch costs three packed instructions per element pair in the loop that bounds this kernel.
          unsigned pb[4], agv[4], alo[4], ahi[4], e0[4], e1[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = 2 * jj;
            const f32x2 uu = f32x2{a1[8 * hs + j], a1[8 * hs + j + 1]} + (j < 4 ? f32x2{bl[j], bl[j + 1]} : f32x2{bh[j - 4], bh[j - 3]});
            const bf16x2v pbv = {(bf16)uu[0], (bf16)uu[1]};                            // the unfused path rounds u to bf16 before GELU
            pb[jj] = __builtin_bit_cast(unsigned, pbv);
            unsigned p1, p2, ak, i4, sg;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(p1) : "v"(pb[jj]), "s"(kneg));
            asm("v_pk_min_i16 %0, %1, %2" : "=v"(p2) : "v"(p1), "s"(kpos));
            p1 &= 0x7FFF7FFFu;
            p2 &= 0x7FFF7FFFu;
            asm("v_pk_max_u16 %0, %1, %2" : "=v"(agv[jj]) : "v"(p1), "s"(0x00800080u));
            asm("v_pk_max_u16 %0, %1, %2" : "=v"(ak) : "v"(p2), "s"(klo));
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(i4) : "v"(ak), "v"(k4v), "s"(koff));
            asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(sg) : "s"(0x000F000Fu), "v"(pb[jj]));   // (a literal 15 would shift the low half only)
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(i4) : "v"(sg), "s"(ksgn), "v"(i4));
            alo[jj] = i4 & 0xffffu;
            ahi[jj] = i4 >> 16;
          }
          asm volatile(
              "ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
              "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(e0[0]), "=&v"(e1[0]), "=&v"(e0[1]), "=&v"(e1[1]), "=&v"(e0[2]), "=&v"(e1[2]), "=&v"(e0[3]), "=&v"(e1[3])
              : "v"(alo[0]), "v"(ahi[0]), "v"(alo[1]), "v"(ahi[1]), "v"(alo[2]), "v"(ahi[2]), "v"(alo[3]), "v"(ahi[3])
              : "memory");
          u32x4v gq, dq;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const unsigned dpair = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x05040100u);
            unsigned gm;
            asm("v_pk_sub_u16 %0, %1, %2" : "=v"(gm) : "v"(agv[jj]), "v"(dpair));
            gq[jj] = (pb[jj] & 0x80008000u) | gm;
            dq[jj] = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x07060302u);
          }
          gv = __builtin_bit_cast(bf16x8, gq);
          dv = __builtin_bit_cast(bf16x8, dq);
        } else {
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const float u0 = a1[8 * hs + j] + (j < 4 ? bl[j] : bh[j - 4]);
            const float u1 = a1[8 * hs + j + 1] + (j < 4 ? bl[j + 1] : bh[j - 3]);
            const f32x2 u = {(float)(bf16)u0, (float)(bf16)u1};          // the unfused path rounds u to bf16 before GELU
            f32x2 ge, dg;
            gelu_pair_fast(u, ge, dg);
            gv[j] = (bf16)ge[0];
            gv[j + 1] = (bf16)ge[1];
            dv[j] = (bf16)dg[0];
            dv[j + 1] = (bf16)dg[1];
          }
        }
        pg[hs].v = gv;
        if (lane_live) {                                   // rows >= live(w) have no tile row (and are never stored)
          const int pcx = ((2 * (2 * ht + hs) + g) ^ (l31 & 7)) << 4;
          *reinterpret_cast<bf16x8*>(stg + l31 * (CH * 2) + pcx) = gv;
          *reinterpret_cast<bf16x8*>(stg + lrw * (CH * 2) + l31 * (CH * 2) + pcx) = dv;
        }
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        const int s = 2 * ht + hs;
        Frag<bf16> fw[6];
#pragma unroll
        for (int b = 0; b < 6; ++b)
          fw[b].v = *reinterpret_cast<const bf16x8*>(sW2 + (32 * b + l31) * (CH * 2) + (((2 * s + g) ^ fl) << 4));
#pragma unroll
        for (int b = 0; b < 6; ++b) mma(acc2[b], fw[b], pg[hs]);     // D rows = output features, D cols = tokens
      }
      if (ht == 0) MLP_FSTAMP(4 + 5 * chunk);
      else MLP_FSTAMP(5 + 5 * chunk);
    }
    // ---- the chunk's gelu / gelu' tiles: live rows x 128 B each, out as whole row pieces (LDS runs a wave in order)
    const int ln = lane_id_here();
    if (!handed)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
      if (row < lrw && 32 * w + row < rows) {
        const int so = row * (CH * 2) + ((vec ^ (row & 7)) << 4);
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(stg + so);
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(stg + lrw * (CH * 2) + so);
        const size_t go = (size_t)(m0 + 32 * w + row) * p.ldg + chunk * CH + vec * 8;
        // gelu(u) / gelu'(u) are read again only in the backward, seconds of traffic later: with the non-temporal hint they do
        // not push the tensors the next kernels are about to read out of the Infinity Cache
        if (p.cold == 1) {
          __builtin_nontemporal_store(v0, reinterpret_cast<bf16x8*>(p.G + go));
          __builtin_nontemporal_store(v1, reinterpret_cast<bf16x8*>(p.GP + go));
        } else {
          *reinterpret_cast<bf16x8*>(p.G + go) = v0;
          *reinterpret_cast<bf16x8*>(p.GP + go) = v1;
        }
      }
    }
    MLP_FSTAMP(6 + 5 * chunk);
  }
  MLP_FSTAMP(62);
  mlp_epilogue(p, ring, B2s, acc2, m0, rows, tid, w, l31, g);
  MLP_FSTAMP(63);
#if defined(MLP_TRACE) && MLP_TRACE != 2
  if (blockIdx.x < 16 && (threadIdx.x & 63) == 0) g_mlp_trace[(blockIdx.x * 8 + w) * 80 + 78] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// The FeedForwardBlock BACKWARD data path in one launch: du = (dy . W2) * gelu'(u)  ->  dx = dy + LayerNorm'(du . W1)
// (reference: models/plainvit.py:481-529 run backwards by autograd).  Until now that was gemm_nt_wres<DGELU> writing du
// [M,768] and gemm_nt_kpipe<LNBWD> reading it back: here a wave's 32 x 64 du tile of every hidden chunk feeds the second
// product's MFMAs straight from registers, exactly as gelu(u) does in mlp_fwd_kernel, and du goes to HBM once, for the dW1
// GEMM only.  Same skeleton as the forward -- the dy rows of the wave live in registers as 12 operand fragments, a DMA wave
// streams W2^T[64 x 192] / W1^T[192 x 64] chunks through the 2-stage ring with the same swizzles -- with these differences:
//   * no biases; instead of evaluating GELU, the chunk's gelu'(u) tile (32 rows x 128 B per wave) arrives by 4 coalesced
//     loads per lane issued one chunk ahead, is parked in the wave's first staging tile and read back as 16-byte fragments;
//   * du = bf16(bf16(acc) * gelu') -- the rounding sequence of the DGELU epilogues -- leaves through the second staging tile;
//   * epilogue = gemm_nt_kpipe's EPI_LNBWD (same arithmetic, 16 lanes per row, panel partial sums of dgamma / dbeta for
//     reduce.hip), with its operand rows requested at the start of the epilogue (there is no register room in the loop).
// Bit-identical to the two-launch path: tests/test_fastpath_model.py::test_fused_mlp_backward_equals_the_two_gemm_path.
struct MlpBwdArgs {
  const bf16* DY; const bf16* W2T; const bf16* W1T; const bf16* GP; bf16* DU;
  int lddy, ldg, ldu;
  int M, rows_per_wg, npanels;
  const bf16* X; int ldx; const float* gamma; const float* mean; const float* rstd;     // LayerNorm input rows + saved statistics
  bf16* DX; int lddx; float* part;                                                      // part: [npanels][2][192]
  int offload;                                                                          // option mlp_dmast, see MlpArgs
};
constexpr int RED_OFF = (BM * CP * 2 + 1023) / 1024 * 1024;              // column-reduction scratch behind the staging tile
typedef rgbnm::LnBwdRows<CTHREADS, BM> LnBwd;
static_assert(RED_OFF + LnBwd::RED_BYTES <= SMEM, "LN-backward scratch: the ring and the (by then dead) staging tiles");


__global__ __launch_bounds__(NTHREADS) void mlp_bwd_kernel(MlpBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int panel = blockIdx.x;
  if (panel >= p.npanels) return;
  const int m0 = panel * p.rows_per_wg;
  const int rows = min(p.rows_per_wg, p.M - m0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;

  if (w == NCW) {
    // ---------------- DMA wave: one 48 KB weight chunk per barrier, one chunk ahead of the math (as in mlp_fwd_kernel)
    const int rl = lane >> 3, pc = lane & 7;
    auto issue = [&](int chunk) {
      unsigned char* st = smem + (chunk & 1) * STAGE;
      const bf16* w1 = p.W2T + (size_t)chunk * CH * E;
#pragma unroll
      for (int i = 0; i < W1_STAGE / 1024; ++i) {
        const int pidx = 64 * i + lane, row = pidx / 24, c24 = pidx % 24;
        const int hrow = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);     // LDS row r holds hidden unit swap23(r)
        __builtin_amdgcn_global_load_lds((glb_ptr)(w1 + hrow * E + pchunk(c24, row) * 8), (lds_ptr)(st + i * 1024), 16, 0, 0);
      }
      const bf16* w2 = p.W1T + chunk * CH;
#pragma unroll
      for (int i = 0; i < W2_STAGE / 1024; ++i) {
        const int r8 = 8 * i + rl;
        __builtin_amdgcn_global_load_lds((glb_ptr)(w2 + (size_t)r8 * H + ((pc ^ fswz(r8)) * 8)),
                                         (lds_ptr)(st + W1_STAGE + i * 1024), 16, 0, 0);
      }
    };
    // offload (as in mlp_fwd_kernel): the du tiles of waves 4-6, the younger and therefore critical wave of each SIMD pair,
    // leave through this wave: registers after barrier c + 1, flag, 12 stores, next weight chunk
    volatile int* flag = reinterpret_cast<volatile int*>(smem + FLAG_OFF);
    bf16x8 tv[3][4];
    const int trow = lane >> 3, tvec = lane & 7;
    auto take_tiles = [&]() {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = trow + 8 * i;
          tv[q][i] = *reinterpret_cast<const bf16x8*>(smem + STG_OFF + (4 + q) * STG_WAVE + STG_TILE + row * (CH * 2) +
                                                      ((tvec ^ (row & 7)) << 4));
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto store_tiles = [&](int chunk) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 32 * (4 + q) + trow + 8 * i;
          if (row < rows) *reinterpret_cast<bf16x8*>(p.DU + (size_t)(m0 + row) * p.ldu + chunk * CH + tvec * 8) = tv[q][i];
        }
    };
    if (p.offload && lane == 0) *flag = 0;
    issue(0);
    for (int c = 0; c < NCHUNK; ++c) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // chunk c landed (and the tile stores issued before it)
      __builtin_amdgcn_s_barrier();                       // ... and every compute wave is done with chunk c - 1
      if (p.offload && c >= 1) {
        take_tiles();
        if (lane == 0) *flag = c;
        store_tiles(c - 1);
      }
      if (c + 1 < NCHUNK) issue(c + 1);
    }
    if (p.offload) {                                      // the last chunk's tiles: after the epilogue's first barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      take_tiles();
      store_tiles(NCHUNK - 1);
    }
    return;                                               // ended waves drop out of the workgroup barrier
  }

  // ---------------- compute waves
  const int rloc = 32 * w + l31;
  const bf16* yrow = p.DY + (size_t)(m0 + (rloc < rows ? rloc : rows - 1)) * p.lddy;
  bf16x8 fa[E / 16];
#pragma unroll
  for (int c = 0; c < E / 16; ++c) fa[c] = *reinterpret_cast<const bf16x8*>(yrow + (2 * c + g) * 8);

  f32x16 acc2[6];
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[b][r] = 0.f;

  const int fl = fswz(l31);
  const int woff0 = l31 * (E * 2) + ((g ^ fl) << 4);
  unsigned char* stg = smem + STG_OFF + w * STG_WAVE;     // tile 0: gelu' in, tile 1: du out

  // the chunk's gelu' tile of this wave, row-major pieces: lane = (row i*8 + lane/8, 16-byte segment lane%8)
  const int ln = lane_id_here();
  bf16x8 gpraw[4];
  auto load_gp = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
      int rr = 32 * w + row;
      rr = rr < rows ? rr : rows - 1;
      gpraw[i] = *reinterpret_cast<const bf16x8*>(p.GP + (size_t)(m0 + rr) * p.ldg + chunk * CH + vec * 8);
    }
  };
  load_gp(0);
  MLP_BSTAMP(0);
  MLP_BSTAMP(1);

  const bool handed = p.offload && w >= 4;                // this wave's du tiles leave through the DMA wave
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
      *reinterpret_cast<bf16x8*>(stg + row * (CH * 2) + ((vec ^ (row & 7)) << 4)) = gpraw[i];
    }
    if (chunk + 1 < NCHUNK) load_gp(chunk + 1);
    if (p.offload) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // du tile writes of the previous chunk are in LDS
    MLP_BSTAMP(2 + 5 * chunk);
    __builtin_amdgcn_s_barrier();
    MLP_BSTAMP(3 + 5 * chunk);
    const unsigned char* sW1 = smem + (chunk & 1) * STAGE;
    const unsigned char* sW2 = sW1 + W1_STAGE;
#pragma unroll
    for (int ht = 0; ht < 2; ++ht) {
      f32x16 a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.f;
      int wbase = woff0 + ht * 32 * (E * 2);
      asm volatile("" : "+v"(wbase));
#pragma unroll
      for (int c = 0; c < E / 16; ++c) {
        Frag<bf16> fb, fx;
        fb.v = *reinterpret_cast<const bf16x8*>(sW1 + ((wbase ^ ((c % 4) << 5)) + 128 * (c / 4)));
        fx.v = fa[c];
        mma(a1, fb, fx);                                   // D rows = hidden (LDS row order), D cols = tokens
      }
      Frag<bf16> pg[2];
      if (handed && ht == 0 && chunk > 0) {               // the DMA wave has taken the previous chunk's du tile
        const volatile int* flag = reinterpret_cast<const volatile int*>(smem + FLAG_OFF);
        while (__builtin_amdgcn_readfirstlane(*flag) < chunk) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        // registers 8 hs .. 8 hs + 7 of this lane = hidden units h0 .. h0 + 7 of the token l31 (rows were stored swap23-ed)
        const int pcx = ((2 * (2 * ht + hs) + g) ^ (l31 & 7)) << 4;
        const bf16x8 gpv = *reinterpret_cast<const bf16x8*>(stg + l31 * (CH * 2) + pcx);
        bf16x8 dv;
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = (bf16)((float)(bf16)a1[8 * hs + j] * (float)gpv[j]);
        pg[hs].v = dv;
        *reinterpret_cast<bf16x8*>(stg + STG_TILE + l31 * (CH * 2) + pcx) = dv;
      }
#pragma unroll
      for (int hs = 0; hs < 2; ++hs) {
        const int s = 2 * ht + hs;
        Frag<bf16> fw[6];
#pragma unroll
        for (int b = 0; b < 6; ++b)
          fw[b].v = *reinterpret_cast<const bf16x8*>(sW2 + (32 * b + l31) * (CH * 2) + (((2 * s + g) ^ fl) << 4));
#pragma unroll
        for (int b = 0; b < 6; ++b) mma(acc2[b], fw[b], pg[hs]);     // D rows = input features, D cols = tokens
      }
      if (ht == 0) MLP_BSTAMP(4 + 5 * chunk);
      else MLP_BSTAMP(5 + 5 * chunk);
    }
    // ---- the chunk's du tile: 32 rows x 128 B, out as whole row pieces
    if (!handed)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = ln + 64 * i, row = idx >> 3, vec = idx & 7;
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(stg + STG_TILE + row * (CH * 2) + ((vec ^ (row & 7)) << 4));
      if (32 * w + row < rows)
        *reinterpret_cast<bf16x8*>(p.DU + (size_t)(m0 + 32 * w + row) * p.ldu + chunk * CH + vec * 8) = v0;
    }
    MLP_BSTAMP(6 + 5 * chunk);
  }
  MLP_BSTAMP(62);

  // ---------------- epilogue: dx = dy + LayerNorm'(acc2): gemm_nt_kpipe's EPI_LNBWD (ln_bwd_rows.h, 8 lanes per row)
  LnBwd lnb;
  lnb.request_x(p.X, p.ldx, p.mean, p.rstd, m0, rows, tid);
  lnb.request_res(p.DY, p.lddy, m0, rows, tid);
  MLP_BSTAMP(64);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave is done with the ring: it becomes the staging tile
  MLP_BSTAMP(65);
  bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = 32 * b + 8 * q + 4 * g;
      const f32x4 v = {acc2[b][4 * q + 0], acc2[b][4 * q + 1], acc2[b][4 * q + 2], acc2[b][4 * q + 3]};
      store4<bf16>(Cs + rloc * CP + nl, v);
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  MLP_BSTAMP(66);
  lnb.run(Cs, CP, p.DX, p.lddx, p.gamma, true, p.part, panel, reinterpret_cast<float*>(smem + RED_OFF), m0, rows, tid);
  MLP_BSTAMP(63);
}

// ------------------------------------------------------------------------------------------------------------------
// Table GELU.  In the bf16 path u is rounded to bf16 BEFORE the GELU (as the unfused epilogue does), so gelu(u) and gelu'(u)
// are functions of 16 bits.  g_gelu_full holds the arithmetic of gelu_pair_fast for every bf16 input; the forward kernel looks
// the pair up in a compact LDS image of it instead of spending ~35 VALU instructions (4 of them transcendental) per two
// elements.  Outside a window of magnitudes the function has closed forms, all in terms of the 15-bit magnitude `a` of u and
// ONE 16-bit subtraction  |gelu(u)| = max(a, 0x80) - D[key]  that the clamped keys at both ends of the window share:
//     a <  A0 (tiny)          : gelu = u / 2  (exponent - 1: D = 0x80),  gelu' = 0.5
//     a >= P1, u > 0 (large)  : gelu = u      (D = 0),                   gelu' = 1
//     a >= N1, u < 0          : gelu = -0     (a clamped to N1, D = N1), gelu' = its constant there
// (sign(gelu(u)) = sign(u) throughout).  A0, P1, N1 are not assumed: gelu_scan_kernel finds the widest tiny / narrowest large
// regions in which the closed forms reproduce g_gelu_full bit for bit and writes the LDS image; if the window does not fit
// the kernel's LDS the table is marked invalid and the arithmetic runs.  The only inputs whose result differs from the
// arithmetic are those whose u or u / 2 is a bf16 DENORMAL (|u| < 2.36e-38): gelu = a denormal of the right sign, not u / 2.
// Measured on MI355X: A0 = 2^-9, P1 = 4, N1 = 16 -> 3076 dwords = 12.0 KB.
constexpr int GELU_IMG_MAX = 3336;                 // dwords of LDS the forward kernel can spare (13 344 B)
__device__ unsigned g_gelu_full[65536];            // [bits of u] -> gelu(u) | gelu'(u) << 16 (bf16 each)
__device__ unsigned g_gelu_img[GELU_IMG_MAX];      // [0] tiny | [1 .. NP] a = A0 .. P1-1 | [NP+1] large+ | [NP+2] tiny | a = A0 .. N1-1 | large-
__device__ int g_gelu_win[16];                     // [0] valid [1] A0 [2] P1 [3] N1 [4] image dwords [5] gelu'(-large) bits [6..] splat constants

__global__ __launch_bounds__(256) void gelu_full_kernel() {
  const unsigned k = blockIdx.x * 256 + threadIdx.x;
  const float u = (float)__builtin_bit_cast(bf16, (unsigned short)k);
  const f32x2 x = {u, u};
  f32x2 ge, dg;
  gelu_pair_fast(x, ge, dg);
  g_gelu_full[k] = (unsigned)__builtin_bit_cast(unsigned short, (bf16)ge[0]) |
                   ((unsigned)__builtin_bit_cast(unsigned short, (bf16)dg[0]) << 16);
}

__global__ __launch_bounds__(1024) void gelu_scan_kernel() {
  __shared__ int tiny_ok[256], pos_ok[256], neg_ok[256], win[4];
  const int tid = threadIdx.x;
  if (tid < 256) { tiny_ok[tid] = 1; pos_ok[tid] = 1; neg_ok[tid] = 1; }
  __syncthreads();
  const unsigned gpn = g_gelu_full[0xFF7F] >> 16;                  // gelu'(most negative finite)
  {
    const int row = tid >> 2;                                       // exponent row: 128 magnitudes
    bool t_ok = true, p_ok = true, n_ok = true;
    for (int m = (tid & 3) * 32; m < (tid & 3) * 32 + 32; ++m) {
      const unsigned a = row * 128 + m;
      const unsigned ep = g_gelu_full[a], en = g_gelu_full[0x8000 | a];
      if (a >= 0x80) {
        t_ok = t_ok && ep == ((a - 0x80) | (0x3F00u << 16)) && en == ((0x8000u | (a - 0x80)) | (0x3F00u << 16));
      }
      p_ok = p_ok && ep == (a | (0x3F80u << 16));
      n_ok = n_ok && en == (0x8000u | (gpn << 16));
    }
    if (!t_ok) atomicAnd(&tiny_ok[row], 0);
    if (!p_ok) atomicAnd(&pos_ok[row], 0);
    if (!n_ok) atomicAnd(&neg_ok[row], 0);
  }
  __syncthreads();
  if (tid == 0) {
    int e0 = 2;                                     // rows 2 .. e0-1 are tiny (rows 0, 1: u or u / 2 is a denormal -- flushed)
    while (e0 < 255 && tiny_ok[e0]) ++e0;
    int ep = 255, en = 255;                                         // rows ep .. 254 / en .. 254 are large (255 = inf / nan)
    while (ep > e0 && pos_ok[ep - 1]) --ep;
    while (en > e0 && neg_ok[en - 1]) --en;
    const int A0 = e0 * 128, P1 = ep * 128, N1 = en * 128;
    const int ndw = (P1 - A0 + 2) + (N1 - A0 + 2);
    win[0] = A0; win[1] = P1; win[2] = N1; win[3] = ndw;
    g_gelu_win[1] = A0; g_gelu_win[2] = P1; g_gelu_win[3] = N1; g_gelu_win[4] = ndw; g_gelu_win[5] = (int)gpn;
  }
  __syncthreads();
  const int A0 = win[0], P1 = win[1], N1 = win[2], ndw = win[3];
  if (ndw <= GELU_IMG_MAX) {
    const int NP = P1 - A0, NN = N1 - A0;
    for (int i = tid; i < ndw; i += 1024) {
      unsigned e;
      if (i == 0 || i == NP + 2) e = 0x80u | (0x3F00u << 16);                       // tiny
      else if (i <= NP) { const unsigned a = A0 + i - 1, f = g_gelu_full[a]; e = (a - (f & 0x7FFF)) | (f & 0xFFFF0000u); }
      else if (i == NP + 1) e = 0u | (0x3F80u << 16);                               // large positive
      else if (i <= NP + 2 + NN) { const unsigned a = A0 + (i - NP - 2) - 1, f = g_gelu_full[0x8000 | a]; e = (a - (f & 0x7FFF)) | (f & 0xFFFF0000u); }
      else e = (unsigned)N1 | (gpn << 16);                                          // large negative
      g_gelu_img[i] = e;
    }
  }
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    g_gelu_win[0] = ndw <= GELU_IMG_MAX ? 1 : 0;
  }
}

}  // namespace

#ifdef MLP_TRACE
extern "C" int rgbnm_mlp_trace_read(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mlp_trace), sizeof(unsigned long long) * 16 * 8 * 80) == hipSuccess ? 0 : -1;
}
#endif

// Host-side record of the device table (per device): 0 = not built, 1 = valid, -1 = built but unusable
struct GeluTabHost { std::atomic<int> state{0}; int A0 = 0, P1 = 0, N1 = 0; const unsigned* img = nullptr; };
static GeluTabHost g_tab_host[64];

// Builds the tables on `stream`, waits for them and reads the window back: a set-up call (SYNCHRONISES; not capturable).
// Idempotent per device.  The forward launcher uses the table only on devices where this has been called.
extern "C" int rgbnm_gelu_table_init(void* stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  GeluTabHost& T = g_tab_host[dev & 63];
  if (T.state.load(std::memory_order_acquire) != 0) return RGBNM_OK;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (T.state.load(std::memory_order_acquire) != 0) return RGBNM_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gelu_full_kernel, dim3(256), dim3(256), 0, st);
  hipLaunchKernelGGL(gelu_scan_kernel, dim3(1), dim3(1024), 0, st);
  LAUNCH_CHECK();
  int win[16];
  void* img = nullptr;
  if (hipStreamSynchronize(st) != hipSuccess || hipMemcpyFromSymbol(win, HIP_SYMBOL(g_gelu_win), sizeof(win)) != hipSuccess ||
      hipGetSymbolAddress(&img, HIP_SYMBOL(g_gelu_img)) != hipSuccess)
    return RGBNM_ELAUNCH;
  T.A0 = win[1]; T.P1 = win[2]; T.N1 = win[3]; T.img = (const unsigned*)img;
  const bool ok = win[0] == 1 && win[4] * 4 <= F_TAB_BYTES && T.A0 >= 0x100 && T.P1 > T.A0 && T.N1 > T.A0;
  T.state.store(ok ? 1 : -1, std::memory_order_release);
  return RGBNM_OK;
}
// Other translation units (vit_chain.hip): the device image and its window on the current device; returns the table state
// (1 = usable, 0 = rgbnm_gelu_table_init not called, -1 = built but unusable)
int rgbnm_gelu_table_query(const unsigned** img, int* A0, int* P1, int* N1, int* ndw) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const GeluTabHost& T = g_tab_host[dev & 63];
  const int state = T.state.load(std::memory_order_acquire);
  if (state == 1) {
    *img = T.img; *A0 = T.A0; *P1 = T.P1; *N1 = T.N1;
    *ndw = (T.P1 - T.A0 + 2) + (T.N1 - T.A0 + 2);
  }
  return state;
}
// Test / diagnostics: the window descriptor (16 ints) and, optionally, the full table (65536 dwords) on the host.  Synchronises.
extern "C" int rgbnm_gelu_table_info(int* win16, unsigned* full65536) {
  if (hipDeviceSynchronize() != hipSuccess) return RGBNM_ELAUNCH;
  if (win16 && hipMemcpyFromSymbol(win16, HIP_SYMBOL(g_gelu_win), 16 * sizeof(int)) != hipSuccess) return RGBNM_ELAUNCH;
  if (full65536 && hipMemcpyFromSymbol(full65536, HIP_SYMBOL(g_gelu_full), 65536 * sizeof(unsigned)) != hipSuccess) return RGBNM_ELAUNCH;
  return RGBNM_OK;
}

// 1 = shape not eligible (the caller runs fc1 and fc2 as two GEMM launches).
int rgbnm_launch_mlp_fwd(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2,
                         const void* R, int ldr, void* G, void* GP, int ldg, void* Y, int ldy, const float* gamma,
                         const float* beta, void* Y2, int ldy2, float* mean, float* rstd, float eps, int M, int Edim,
                         int Hdim, hipStream_t st) {
  if (Edim != E || Hdim != H || M < 8192 || ldx % 8 || ldr % 8 || ldg % 8 || ldy % 8 || (gamma && ldy2 % 8)) return 1;
  if (!X || !W1 || !b1 || !W2 || !b2 || !R || !G || !GP || !Y) return RGBNM_EINVAL;
  if (gamma && (!beta || !Y2 || !mean || !rstd)) return RGBNM_EINVAL;
  MlpArgs p;
  p.X = (const bf16*)X; p.W1 = (const bf16*)W1; p.b1 = b1; p.W2 = (const bf16*)W2; p.b2 = b2; p.R = (const bf16*)R;
  p.G = (bf16*)G; p.GP = (bf16*)GP; p.Y = (bf16*)Y;
  p.ldx = ldx; p.ldr = ldr; p.ldg = ldg; p.ldy = ldy; p.M = M;
  p.cold = rgbnm_get_option("nt_cold");
  p.offload = rgbnm_get_option("mlp_dmast");
  p.gamma = gamma; p.beta = beta; p.Y2 = (bf16*)Y2; p.mean_o = mean; p.rstd_o = rstd; p.eps = eps; p.ldy2 = ldy2;
  int rows = cdiv(M, 256);
  if (rows > BM) rows = BM;
  p.rows_per_wg = rows;
  p.npanels = cdiv(M, rows);
  // table GELU (option gelu_table): on devices where rgbnm_gelu_table_init found a usable table, for panels whose staging tiles
  // leave the LDS for it
  int dev = 0;
  (void)hipGetDevice(&dev);
  const GeluTabHost& T = g_tab_host[dev & 63];
  const bool table = rgbnm_get_option("gelu_table") && rows <= F_TAB_ROWS && T.state.load(std::memory_order_acquire) == 1;
  p.tab_img = nullptr; p.kneg = p.kpos = p.klo = p.koff = p.ksgn = 0;
  if (table) {
    p.tab_img = T.img;
    p.kneg = 0x00010001u * (unsigned)(0x8000 | T.N1);
    p.kpos = 0x00010001u * (unsigned)T.P1;
    p.klo = 0x00010001u * (unsigned)(T.A0 - 1);
    p.koff = 0x00010001u * (unsigned)((0x10000 - 4 * (T.A0 - 1)) & 0xffff);
    p.ksgn = 0x00010001u * (unsigned)(4 * (T.P1 - T.A0 + 2));
  }
  const int smem_bytes = table ? f_smem(1, rows) : SMEM;
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess ||
        hipFuncSetAttribute((const void*)mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, f_smem(1, F_TAB_ROWS)) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  const double me = (double)M * E, mh = (double)M * H;
  // algorithmic bytes: xn2 + x_mid in, gelu + gelu' + x_out (+ xn_next) out, both weight matrices once
  const int slot = rgbnm_trace_begin(TR_NT, 4.0 * mh * E, (me * (gamma ? 4.0 : 3.0) + mh * 2.0) * 2.0 + 4.0 * E * H, st);
  if (table) hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3(p.npanels), dim3(NTHREADS), smem_bytes, st, p);
  else hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3(p.npanels), dim3(NTHREADS), smem_bytes, st, p);
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

// 1 = shape not eligible (the caller runs the DGELU GEMM and the LayerNorm-backward GEMM as two launches).
int rgbnm_launch_mlp_bwd(const void* DY, int lddy, const void* W2T, const void* W1T, const void* GP, int ldg, void* DU, int ldu,
                         const void* X, int ldx, const float* gamma, const float* mean, const float* rstd, void* DX, int lddx,
                         float* part, int* npanels_out, int M, int Edim, int Hdim, hipStream_t st) {
  if (Edim != E || Hdim != H || M < 8192 || lddy % 8 || ldg % 8 || ldu % 8 || ldx % 8 || lddx % 8) return 1;
  if (!DY || !W2T || !W1T || !GP || !DU || !X || !gamma || !mean || !rstd || !DX || !part || !npanels_out) return RGBNM_EINVAL;
  MlpBwdArgs p;
  p.DY = (const bf16*)DY; p.W2T = (const bf16*)W2T; p.W1T = (const bf16*)W1T; p.GP = (const bf16*)GP; p.DU = (bf16*)DU;
  p.lddy = lddy; p.ldg = ldg; p.ldu = ldu; p.M = M;
  p.X = (const bf16*)X; p.ldx = ldx; p.gamma = gamma; p.mean = mean; p.rstd = rstd; p.DX = (bf16*)DX; p.lddx = lddx; p.part = part;
  p.offload = rgbnm_get_option("mlp_dmast");
  int rows = cdiv(M, 256);
  if (rows > BM) rows = BM;
  p.rows_per_wg = rows;
  p.npanels = cdiv(M, rows);
  *npanels_out = p.npanels;
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  const double me = (double)M * E, mh = (double)M * H;
  // algorithmic bytes: dy + x_mid in, gelu' in, du + dx out, both weight matrices once
  const int slot = rgbnm_trace_begin(TR_NT, 4.0 * mh * E, (me * 3.0 + mh * 2.0) * 2.0 + 4.0 * E * H, st);
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(p.npanels), dim3(NTHREADS), SMEM, st, p);
  rgbnm_trace_end(slot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}
