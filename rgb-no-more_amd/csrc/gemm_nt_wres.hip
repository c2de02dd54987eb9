// Weight-resident NT GEMM for the K = 192 Linear layers of JPEG-Ti (qkv, fc1, the dGELU product, attention dX):
//   C[M, N] = epi(A[M,192] . W[N,192]^T), bf16.
// These GEMMs read 19 MB and write 58-154 MB at B = 256: HBM bound, and with a tile-per-workgroup kernel also
// latency bound (every workgroup re-fetches its 72 KB weight tile from L2 and has one 16 KB activation tile in
// flight).  Here one persistent 448-thread workgroup per CU keeps a 192-column weight tile (72 KB, LDS-DMA'd once,
// XOR-swizzled for conflict-free ds_read_b128) for its whole life.  Each of its 7 waves then works ALONE on
// 32-token x 192-feature tiles with a private 12 KB LDS buffer: the tile's 12 KB arrive as 12 fully coalesced 1 KB
// loads issued one whole tile ahead of the math, are parked in the private buffer (same swizzle), feed 72 MFMAs, and
// the result goes back through the same buffer (two 96-column halves) to leave as coalesced 16-byte rows with the
// epilogue (bias, GELU + GELU', dGELU product, residual) fused.  Tiles are handed out by an LDS counter; no
// workgroup barrier after start-up: waves drift apart, so one wave's erf polynomial overlaps another's loads/MFMAs.
// Measured on the way here (B = 256, qkv GEMM): L2 -> CU bytes cost about as much as HBM bytes on this chip
// (aggregate ~6 TB/s), so A must be re-read as few times as possible (192 columns per workgroup: 3x, not 6x);
// fragment-shaped global loads (32 B per row per instruction) cost 25 % more than coalesced loads + LDS; an LDS-DMA
// ring with only 2 waves per CU was 2x slower (too few waves to overlap the phases).
#include "common.h"
#include "internal.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int K = 192;
constexpr int ROWB = K * 2;                 // 384 B per row
constexpr int BN = 192, BNH = 96, BMT = 32; // one wave-step: 32 tokens x 192 features, staged as two 96-column halves
constexpr int NWAVES = 7, NTHREADS = 64 * NWAVES;
constexpr int W_BYTES = BN * ROWB;          // 72 KB resident
constexpr int A_WAVE = BMT * ROWB;          // 12 KB private tile buffer per wave (re-used as the output staging tile)
constexpr int CP = BNH + 4;                 // staging pitch (elements)
constexpr int BIAS_BYTES = BN * 4;
constexpr int SMEM = W_BYTES + BIAS_BYTES + 16 + NWAVES * A_WAVE;   // 160,528 B
constexpr int W_INSTR = W_BYTES / 1024;     // 72 DMA instructions
constexpr int NCH = K / 16;                 // 12 reduction chunks (one MFMA fragment each) = 12 x 1 KB loads per tile
constexpr int NVEC = BMT * (BNH / 8) / 64;  // 6 output vectors (16 B) per lane per half tile
static_assert(BMT * CP * 2 <= A_WAVE, "staging tile must fit the private buffer");
static_assert(SMEM <= 160 * 1024, "LDS");

enum { EPI_NONE = 0, EPI_RES = 1, EPI_GELU = 2, EPI_DGELU = 4 };

struct WresArgs {
  const bf16* A; const bf16* W; bf16* C; const float* bias; const bf16* R; bf16* C2;
  int lda, ldw, ldc, ldr, ldc2;
  int M, N, ntiles, msplit, tiles_per_split;
};

__device__ __forceinline__ int fswz(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}
// physical 16-byte chunk (0..23) of logical chunk lc in `row` (an involution)
__device__ __forceinline__ int pchunk(int lc, int row) { return (lc & ~7) | ((lc & 7) ^ fswz(row)); }

#ifdef WRES_PROF
__device__ unsigned long long g_wres_prof[320 * 8 * 32];
#define PROF(i) do { if (lane == 0 && (i) < 32) g_wres_prof[(blockIdx.x * 8 + w) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PROF(i) do {} while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_wres_kernel(WresArgs p) {
  constexpr bool HASR = (EPI == EPI_RES || EPI == EPI_DGELU);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem;
  float* Bs = reinterpret_cast<float*>(smem + W_BYTES);
  int* next_tile = reinterpret_cast<int*>(smem + W_BYTES + BIAS_BYTES);

  // XCD-aware: the ntiles workgroups that stream the same token range sit on one XCD (A re-reads hit its L2)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int nt = jj % p.ntiles, ms = (jj / p.ntiles) * 8 + xcd;
  if (ms >= p.msplit) return;
  const int n0 = nt * BN;
  const int mtiles = p.M / BMT;
  const int t0 = ms * p.tiles_per_split;
  const int tend = min(t0 + p.tiles_per_split, mtiles);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  unsigned char* Aw = smem + W_BYTES + BIAS_BYTES + 16 + w * A_WAVE;
  bf16* Cs = reinterpret_cast<bf16*>(Aw);

  PROF(0);
  // ---- start-up: bias and the resident weight tile by LDS-DMA; tiles t0 .. t0+6 are pre-assigned to the 7 waves
  if (tid == 0) *next_tile = t0 + NWAVES;
  if (w < BN / 64) {
    if (p.bias) __builtin_amdgcn_global_load_lds((glb_ptr)(p.bias + n0 + 64 * w + lane), (lds_ptr)(Bs + 64 * w), 4, 0, 0);
    else Bs[64 * w + lane] = 0.f;
  }
  for (int i = w; i < W_INSTR; i += NWAVES) {
    const int pidx = i * 64 + lane, row = pidx / 24, pc = pidx % 24;
    __builtin_amdgcn_global_load_lds((glb_ptr)(p.W + (size_t)(n0 + row) * p.ldw + pchunk(pc, row) * 8),
                                     (lds_ptr)(Ws + i * 1024), 16, 0, 0);
  }

  // ---- per-lane addressing.  Load j of a tile: 16-byte chunk pidx = 64 j + lane -> (row, chunk) = (pidx/24, pidx%24)
  // (3 loads = 8 rows: the source pattern repeats every 3 loads, the swizzled destination every 6)
  int a_dst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pidx = 64 * j + lane, row = pidx / 24, pc = pidx % 24;
    a_dst[j] = row * ROWB + (pchunk(pc, row) << 4);
  }
  // output vector i of a lane: idx = lane + 64 i -> (row, vec) = (idx / 12, idx % 12); i + 3 is 16 rows further down
  int r_lane[3], c_lane[3], c2_lane[3], s_lane[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = lane + 64 * i, row = idx / (BNH / 8), vec = idx % (BNH / 8);
    r_lane[i] = row * p.ldr + vec * 8;
    c_lane[i] = row * p.ldc + vec * 8;
    c2_lane[i] = row * p.ldc2 + vec * 8;
    s_lane[i] = row * CP + vec * 8;
  }
  const int fl = fswz(l31);
  // fragment c of row l31 sits at l31 * ROWB + (((2c + g) ^ fl) << 4) = woff0 ^ (c << 5) (c < 4; + 128 per 4 chunks):
  // one base, chunk selected by XOR with a constant, re-derived per tile so that four hoisted copies per LDS array do
  // not sit in registers through the epilogue (the DGELU / RES variants spilled one of them and reloaded it in the MFMA
  // phase behind an s_waitcnt that drained the next tile's prefetch)
  const int woff0 = l31 * ROWB + ((g ^ fl) << 4);

  bf16x8 a[NCH];                                       // the NEXT tile, in flight while the current one is computed
  auto load_tile = [&](int tt) {
    const bf16* ab = p.A + (size_t)tt * BMT * p.lda;
    // the three source offsets are recomputed from lane_id_here(): as loop invariants they get hoisted, and in the
    // register-heavy epilogue variants SPILLED -- the reload then sits in front of these loads behind an s_waitcnt
    // vmcnt(0) that drains the wave's in-flight stores and R loads on every tile
    const int ln = lane_id_here();
    int src[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pidx = 64 * j + ln;
      src[j] = (pidx / 24) * p.lda + (pidx % 24) * 8;
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) a[j] = *reinterpret_cast<const bf16x8*>(ab + src[j % 3] + (j / 3) * 8 * p.lda);
  };
  auto grab = [&]() -> int {                           // next unclaimed tile of this workgroup (wave-uniform)
    int v = 0;
    if (lane == 0) v = atomicAdd(next_tile, 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  // second operand (residual / GELU').  vmcnt retires in order, so WHEN a load is issued decides what it waits for:
  // the first 96-column half of the NEXT tile is requested at the end of a step (behind that tile's A loads, which
  // are needed first anyway); the second half right after the MFMA phase (registers are free then; by the time it is
  // consumed the prefetch issued before it has long landed).
  bf16x8 r_cur[2][NVEC];
  auto load_r = [&](int tt, int h, bf16x8 (&r)[NVEC]) {
    const bf16* rb = p.R + (size_t)tt * BMT * p.ldr + n0 + BNH * h;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) r[i] = *reinterpret_cast<const bf16x8*>(rb + r_lane[i % 3] + (i / 3) * 16 * p.ldr);
  };
  int t = t0 + w;
  if (t < tend) {
    load_tile(t);
    if (HASR) load_r(t, 0, r_cur[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA share has landed (first-tile loads too: one-off cost)
  __syncthreads();                                    // whole weight tile + bias + counter visible; the only barrier
  PROF(1);
#ifdef WRES_PROF
  int pstep = 0;
#endif

#ifdef WRES_SIXWAVES
  if (w == 6) return;
#endif
  while (t < tend) {
    // park the tile in the private buffer (previous tile's staging reads are older LDS instructions of this wave:
    // LDS executes a wave in order), then immediately refill the registers with the tile after it
#pragma unroll
    for (int j = 0; j < NCH; ++j) *reinterpret_cast<bf16x8*>(Aw + a_dst[j % 6] + (j / 6) * 16 * ROWB) = a[j];
    const int tn = grab();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tn < tend) load_tile(tn);
    PROF(2 + 4 * pstep);

    f32x16 acc[6];
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    int wbase = woff0;
    asm volatile("" : "+v"(wbase));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      Frag<bf16> fa, fb[6];
      const int wo = (wbase ^ ((c % 4) << 5)) + 128 * (c / 4);
      fa.v = *reinterpret_cast<const bf16x8*>(Aw + wo);
#pragma unroll
      for (int b = 0; b < 6; ++b) fb[b].v = *reinterpret_cast<const bf16x8*>(Ws + 32 * b * ROWB + wo);
#pragma unroll
      for (int b = 0; b < 6; ++b) mma(acc[b], fb[b], fa);     // swapped: D rows <-> features, D cols <-> tokens
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // fragment reads done before the buffer becomes staging
    PROF(3 + 4 * pstep);
    if (HASR) load_r(t, 1, r_cur[1]);

#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int nh = n0 + BNH * h;
      // ---- pass 1: lane = token l31, register quad = 4 consecutive features -> staging tile (+ bias)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = 32 * b + 8 * q + 4 * g;
          const f32x16& ac = acc[3 * h + b];
          f32x4 v = {ac[4 * q + 0], ac[4 * q + 1], ac[4 * q + 2], ac[4 * q + 3]};
          v += *reinterpret_cast<const f32x4*>(Bs + BNH * h + nl);
          store4<bf16>(Cs + l31 * CP + nl, v);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // ---- pass 2: 32 rows x 12 vectors of 8 features, 6 per lane, coalesced 192-byte row pieces
      bf16* cbase = p.C + (size_t)t * BMT * p.ldc + nh;
      bf16* c2base = EPI == EPI_GELU ? p.C2 + (size_t)t * BMT * p.ldc2 + nh : nullptr;
#pragma unroll
      for (int i = 0; i < NVEC; ++i) {
        const int so = s_lane[i % 3] + (i / 3) * 16 * CP;
        const bf16x4 c0 = *reinterpret_cast<const bf16x4*>(Cs + so);
        const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(Cs + so + 4);
        bf16x8 cv = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        if (EPI == EPI_GELU) {
          bf16x8 dv;
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2 u = {(float)cv[e], (float)cv[e + 1]};
            f32x2 gv, dgv;
#ifdef WRES_NOGELU
            gv = u; dgv = u * 0.5f;
#else
            gelu_pair_fast(u, gv, dgv);
#endif
            dv[e] = (bf16)dgv[0];
            dv[e + 1] = (bf16)dgv[1];
            cv[e] = (bf16)gv[0];
            cv[e + 1] = (bf16)gv[1];
          }
          store_c2(c2base + c2_lane[i % 3] + (i / 3) * 16 * p.ldc2, dv);
        }
        if (HASR) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = (float)cv[e];
            if (EPI == EPI_RES) v += (float)r_cur[h][i][e];
            else v *= (float)r_cur[h][i][e];
            cv[e] = (bf16)v;
          }
        }
        *reinterpret_cast<bf16x8*>(cbase + c_lane[i % 3] + (i / 3) * 16 * p.ldc) = cv;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging reads done before the buffer is rewritten
    }
#ifdef WRES_PROF
    PROF(4 + 4 * pstep);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROF(5 + 4 * pstep);
    ++pstep;
#endif
    if (HASR && tn < tend) load_r(tn, 0, r_cur[0]);
    t = tn;
  }
}

template <int EPI>
int launch(const WresArgs& p, int grid, hipStream_t st) {
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)gemm_nt_wres_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) !=
        hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  hipLaunchKernelGGL((gemm_nt_wres_kernel<EPI>), dim3(grid), dim3(NTHREADS), SMEM, st, p);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // namespace

// returns 1 when the shape is not eligible (caller falls back to the tile-per-workgroup kernel)
int rgbnm_launch_nt_wres(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                         const void* R, int ldr, void* C2, int ldc2, int M, int N, int Kdim, hipStream_t st) {
  if (Kdim != K || N % BN || M % BMT || lda % 8 || ldw % 8 || ldc % 8 || M < 4096) return 1;
  if (epi != EPI_NONE && epi != EPI_RES && epi != EPI_GELU && epi != EPI_DGELU) return 1;
  if ((epi == EPI_RES || epi == EPI_DGELU) && (!R || ldr % 8)) return 1;
  if (epi == EPI_GELU && (!C2 || ldc2 % 8)) return 1;
  WresArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = (bf16*)C; p.bias = bias; p.R = (const bf16*)R; p.C2 = (bf16*)C2;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldc2 = ldc2; p.M = M; p.N = N;
  p.ntiles = N / BN;
  const int mtiles = M / BMT;
  // one workgroup per CU and never more than 256 of them INCLUDING the padding of the XCD map: a 257th workgroup
  // would wait for a whole first round to finish (1 workgroup fits per CU) and double the kernel time
  int msplit = (256 / p.ntiles) / 8 * 8;
  if (msplit < 8) return 1;
  if (msplit > cdiv(mtiles, NWAVES)) msplit = cdiv(mtiles, NWAVES);
  p.tiles_per_split = cdiv(mtiles, msplit);
  p.msplit = cdiv(mtiles, p.tiles_per_split);
  const int grid = p.ntiles * ((p.msplit + 7) / 8) * 8;
  const double mn = (double)M * N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * K, ((double)M * K + (double)N * K) * 2.0 + mn * 2.0 +
                                                              (epi != EPI_NONE ? mn * 2.0 : 0.0), st);
  int rc;
  switch (epi) {
    case EPI_NONE: rc = launch<EPI_NONE>(p, grid, st); break;
    case EPI_RES: rc = launch<EPI_RES>(p, grid, st); break;
    case EPI_GELU: rc = launch<EPI_GELU>(p, grid, st); break;
    default: rc = launch<EPI_DGELU>(p, grid, st); break;
  }
  rgbnm_trace_end(slot, st);
  return rc;
}

#ifdef WRES_PROF
extern "C" int rgbnm_debug_wres_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wres_prof), sizeof(unsigned long long) * 320 * 8 * 32) == hipSuccess ? 0 : -1;
}
#endif
