// Streaming row-panel NT GEMM (bf16) for the wide Linear layers with several 192-column tiles (E = 384: qkv, fc1 + GELU, the dX
// GEMMs):   C[M,N] = epi(A[M,K] . W[N,K]^T + bias)      (reference: nn.Linear + nn.GELU, models/plainvit.py:467-491)
//
// What the persistent kernel of gemm_nt_kpipe.hip loses (cycle stamps per unit, tools/kpipe_prof_persist.py, fc1 + GELU at E = 384):
// 42 % of a workgroup's time is the k-loop, 13 % the staging pass, 44 % the pass that applies GELU and stores -- one after the
// other, and the k-loop of the NEXT unit starts by waiting for the stores (vmcnt is one in-order queue per wave).  Here nothing of
// a unit's epilogue has a phase of its own:
//   * seven waves (32 rows each, a 224 x 192 unit) stream 32-wide k-tiles through a 4-stage LDS-DMA ring (three tiles =
//     78 KB in flight per CU) and, when a unit's reduction ends, PARK its tile (+ bias) as packed bf16 in 48 registers and start the
//     next unit's reduction at once -- the ring never drains between units;
//   * the parked tile leaves in six 32-column slices during the first twelve k-tiles of the next unit: on an even k-tile a wave
//     applies the epilogue to its slice IN THE ACCUMULATOR LAYOUT (lane = token; GELU runs under the MFMAs) and writes it to its own
//     32-row scratch tile in LDS (64 B per row, pitch 80); on the odd k-tile it reads the rows back (16 B per lane) and stores them,
//     64 B per row and slice.  Two to four store instructions per wave every other k-tile: the store traffic is spread over the whole
//     launch instead of one burst per unit, and the in-order vmcnt queue gives every store three k-tiles to be acknowledged before
//     a wait reaches it (the waits count the stores younger than the k-tile they need).  (A first version gave all stores to an
//     eighth wave: one wave issues a 1 KB store every ~200 cycles, and the whole workgroup waited for it at the barriers.)
//   * one workgroup barrier per k-tile orders the ring stages; the scratch tiles are wave-private.
// Same MFMA order per output element as gemm_nt_kpipe.hip (k ascending in steps of 16), bias added in fp32, one rounding to bf16,
// GELU on the rounded value: the same bits.
#include "common.h"
#include "internal.h"
#include "../../include/rgbnm.h"

int rgbnm_gelu_table_query(const unsigned** img, int* A0, int* P1, int* N1, int* ndw);   // mlp_fused.hip

namespace ks {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int BN = 192, NCW = 7, BM = 32 * NCW, NTHREADS = 64 * NCW;
constexpr int TK = 32, TKB = 2 * TK;            // k-tile: 32 bf16 = 64 bytes per row
constexpr int A_STAGE = BM * TKB;               // 14 KB
constexpr int W_STAGE = BN * TKB;               // 12 KB
constexpr int STAGE = A_STAGE + W_STAGE;        // 26 KB
constexpr int NSTAGE = 4;
constexpr int NSLOT = STAGE / 1024;             // 1 KB DMA instructions per k-tile: 26 (16 rows x 64 B each)
constexpr int ASLOT = A_STAGE / 1024;           // of which A rows: 14
constexpr int NDMA = (NSLOT + NCW - 1) / NCW;   // per compute wave per k-tile: 4 (28 issued for 26: 2 repeats)
constexpr int SP = 80;                          // scratch row pitch (bytes): 64 B of payload, ds_write_b64 of 16 lanes conflict free
constexpr int SCR = BM * SP;                    // 17.5 KB per scratch tile
constexpr int OFF_SCR = NSTAGE * STAGE;         // two scratch tiles: gelu (or the plain output) | gelu'
constexpr int OFF_BIAS = OFF_SCR + 2 * SCR;
constexpr int SMEM = OFF_BIAS + BN * 4;         // 139.75 KB
constexpr int NSLICE = BN / 32;                 // 6
constexpr int TAB_BYTES = 13328, TAB_RESERVE = 14336;     // GELU table image (mlp_fused.hip F_TAB_BYTES)
static_assert(SMEM + TAB_RESERVE <= 160 * 1024, "LDS");

enum { EPI_NONE = 0, EPI_GELU = 2 };            // numbering of gemm.hip

struct KsArgs {
  const bf16* A; const bf16* W; bf16* C; const float* bias; bf16* C2;
  int lda, ldw, ldc, ldc2;
  int M, K, npanels, ntiles;
  const unsigned* tab_img; unsigned kneg, kpos, klo, koff, ksgn;     // table GELU (mlp_fused.hip: the image and the key constants)
};

typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
#ifdef KS_PROF     // experiments only: per wave, cycles summed over all k-tiles: vmcnt wait | barrier | DMA issue | fragments + MFMAs + slices
__device__ unsigned long long g_ks_prof[256 * 8 * 6];
#define KSTAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define KSTAMP(v) do { } while (0)
#endif
template <int V> struct IntTag { static constexpr int value = V; };

// 16-byte chunk swizzle of the 64-byte LDS rows: chunk ^ ((row >> 2) & 3).  A ds_read_b128 is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): the four rows of a group that share row % 4 have four different row >> 2 & 3.
__device__ __forceinline__ int sw64(int row) { return (row >> 2) & 3; }

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef bf16 bf16x2v __attribute__((ext_vector_type(2)));
  const bf16x2v v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

template <int EPI, bool TAB>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kstream_kernel(KsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem0[];
  // TAB: the GELU table image (mlp_fused.hip, 13 328 bytes) sits at LDS offset 0 -- its 16-bit packed byte offsets address it
  // directly -- and everything else 14 KB higher
  unsigned char* const smem = smem0 + (TAB ? TAB_RESERVE : 0);
  float* Bs = reinterpret_cast<float*>(smem + OFF_BIAS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  // Unit order as in gemm_nt_kpipe.hip's persistent kernel: workgroup b is slot b / 8 of XCD b % 8; the XCD's units are numbered
  // jj = (its panel index) * ntiles + tile and slot s takes jj = s, s + slots, ...: the slots of an XCD work on consecutive jj --
  // the column tiles of the same few row panels -- so a panel's k-tiles come from HBM once
  const int xcd = blockIdx.x & 7, slots = gridDim.x >> 3;
  const int J = ((p.npanels + 7) >> 3) * p.ntiles;
  auto unit_ok = [&](int jj) { return jj < J && (jj / p.ntiles) * 8 + xcd < p.npanels; };
  auto next_unit = [&](int u) {
    int un = u + slots;
    while (un < J && !unit_ok(un)) un += slots;
    return un;
  };
  int u0 = (int)(blockIdx.x >> 3);
  while (u0 < J && !unit_ok(u0)) u0 += slots;
  if (u0 >= J) return;
  const int T2 = p.K / TK;                                   // k-tiles per unit (>= 2 NSLICE: host)
  auto unit_geo = [&](int u, int& m0, int& n0, int& rows) {
    const int panel = (u / p.ntiles) * 8 + xcd;
    n0 = (u % p.ntiles) * BN;
    m0 = panel * BM;
    rows = min(BM, p.M - m0);
  };

  // DMA slot i of a k-tile covers 16 rows x 64 B; i < ASLOT: A rows, else W rows (i is wave-uniform).  Lane -> row lane / 4, chunk
  // lane % 4 of the slot; the swizzle goes into the SOURCE address (the LDS side of an LDS-DMA is lane * 16 behind M0)
  unsigned offI[NDMA];
  int iu = u0, ikt = 0, stI = 0;                              // the DMA stream: next k-tile to request = (unit iu, k-tile ikt)
  auto set_off = [&](int u) {
    int m0, n0, rows;
    unit_geo(u, m0, n0, rows);
    const int ln = lane_id_here();
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      int i = w + NCW * j;
      i = i < NSLOT ? i : i - NSLOT;
      const int r16 = (i < ASLOT ? i : i - ASLOT) * 16 + (ln >> 2);
      const int lc = ((ln & 3) ^ sw64(r16)) * 8;
      if (i < ASLOT) offI[j] = (unsigned)(m0 + (r16 < rows ? r16 : rows - 1)) * (unsigned)p.lda + lc;
      else offI[j] = (unsigned)(n0 + r16) * (unsigned)p.ldw + lc;
    }
  };
  // The request stream never stops: behind the last k-tile of the last unit it repeats that tile into the stage just freed (never
  // read again) -- four straight-line DMA instructions per k-tile that the scheduling pipeline below can place between MFMAs, and
  // a vmcnt arithmetic without a tail case
  auto issue4 = [&]() {
    unsigned char* st = smem + stI * STAGE;
    const int k0 = ikt * TK;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      int i = w + NCW * j;
      i = i < NSLOT ? i : i - NSLOT;
      const bf16* base = i < ASLOT ? p.A : p.W;
      __builtin_amdgcn_global_load_lds((glb_ptr)(base + (size_t)offI[j] + k0), (lds_ptr)(st + i * 1024), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    stI = (stI + 1) & (NSTAGE - 1);
    if (iu < J && ++ikt == T2) {
      iu = next_unit(iu);
      if (iu < J) { ikt = 0; set_off(iu); }
      else ikt = T2 - 1;
    }
  };
  // fragment offsets inside a stage: lane (l31, g) reads k-elements 16 c + 8 g .. + 7 of row l31 (chunk 2 c + g, swizzled)
  int foff[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) foff[c] = l31 * TKB + (((2 * c + g) ^ sw64(l31)) << 4);
  const int a_row = 32 * w * TKB;

  if (!p.bias && w < BN / 64) Bs[64 * w + lane] = 0.f;
  if (TAB && w == 0) {                                       // (older than this wave's first k-tile in its vmcnt queue: landed with it)
    const int npiece = TAB_BYTES / 16;
    for (int i = 0; i * 64 < npiece; ++i)
      if (i * 64 + lane < npiece)
        __builtin_amdgcn_global_load_lds((glb_ptr)(p.tab_img + (i * 64 + lane) * 4), (lds_ptr)(smem0 + i * 1024), 16, 0, 0);
  }
  set_off(u0);
  issue4(); advance();
  issue4(); advance();
  issue4(); advance();

  f32x16 acc[6];
#pragma unroll
  for (int b = 0; b < 6; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  unsigned parked[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) parked[i] = 0u;
  bool have = false;
  int stC = 0;
  const int srow = (32 * w + l31) * SP + 8 * g;               // this lane's scratch row + the 8 bytes of its feature quad

  // slice S of the parked tile: epilogue in the accumulator layout, then into the scratch tile(s)
  const unsigned kneg = p.kneg, kpos = p.kpos, klo = p.klo, koff = p.koff, ksgn = p.ksgn;
  unsigned k4v = 0x00040004u;                                 // (a VGPR: an instruction takes one scalar operand)
  asm volatile("" : "+v"(k4v));
  auto put_slice = [&](auto tag) {
    constexpr int S = decltype(tag)::value;
    unsigned char* sc = smem + OFF_SCR + srow;
    if constexpr (EPI == EPI_GELU && TAB) {
      // gelu / gelu' of 16 elements by table, as mlp_fused.hip's forward: bf16 bits of an element pair -> packed 16-bit keys -> one
      // ds_read_b32 of {D | gelu' << 16} per element -> |gelu| = max(a, 0x80) - D (the table comment above gelu_full_kernel there)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned pb[4], agv[4], alo[4], ahi[4], e0[4], e1[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          pb[jj] = parked[S * 8 + 4 * h + jj];
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
        asm volatile(
            "ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
            "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(e0[0]), "=&v"(e1[0]), "=&v"(e0[1]), "=&v"(e1[1]), "=&v"(e0[2]), "=&v"(e1[2]), "=&v"(e0[3]), "=&v"(e1[3])
            : "v"(alo[0]), "v"(ahi[0]), "v"(alo[1]), "v"(ahi[1]), "v"(alo[2]), "v"(ahi[2]), "v"(alo[3]), "v"(ahi[3])
            : "memory");
        unsigned gq[4], dq[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const unsigned dpair = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x05040100u);
          unsigned gm;
          asm("v_pk_sub_u16 %0, %1, %2" : "=v"(gm) : "v"(agv[jj]), "v"(dpair));
          gq[jj] = (pb[jj] & 0x80008000u) | gm;
          dq[jj] = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x07060302u);
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const uint2v gv = {gq[2 * qq], gq[2 * qq + 1]}, dv = {dq[2 * qq], dq[2 * qq + 1]};
          *reinterpret_cast<uint2v*>(sc + 16 * (2 * h + qq)) = gv;
          *reinterpret_cast<uint2v*>(sc + SCR + 16 * (2 * h + qq)) = dv;
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned v0 = parked[S * 8 + 2 * q], v1 = parked[S * 8 + 2 * q + 1];
      if (EPI == EPI_GELU) {
        f32x2 x0 = {__builtin_bit_cast(float, v0 << 16), __builtin_bit_cast(float, v0 & 0xffff0000u)};
        f32x2 x1 = {__builtin_bit_cast(float, v1 << 16), __builtin_bit_cast(float, v1 & 0xffff0000u)};
        f32x2 g0, d0, g1, d1;
        gelu_pair_fast(x0, g0, d0);
        gelu_pair_fast(x1, g1, d1);
        uint2v gv = {pack2(g0[0], g0[1]), pack2(g1[0], g1[1])};
        uint2v dv = {pack2(d0[0], d0[1]), pack2(d1[0], d1[1])};
        *reinterpret_cast<uint2v*>(sc + 16 * q) = gv;
        *reinterpret_cast<uint2v*>(sc + SCR + 16 * q) = dv;
      } else {
        uint2v gv = {v0, v1};
        *reinterpret_cast<uint2v*>(sc + 16 * q) = gv;
      }
    }
  };
  // slice S back from the scratch tile, row-major, and out: lane -> (row lane / 4 (+ 16), 16-byte chunk lane % 4).  Returns the number of
  // store instructions issued (wave-uniform)
  int pm0 = 0, pn0 = 0, prows = 0;
  auto get_slice = [&](auto tag) {
    constexpr int S = decltype(tag)::value;
    const int ln = lane_id_here();
    const unsigned char* sc = smem + OFF_SCR + 32 * w * SP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rl = (ln >> 2) + 16 * i, ch = ln & 3, row = 32 * w + rl;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(sc + rl * SP + ch * 16);
      bf16x8 d;
      if (EPI == EPI_GELU) d = *reinterpret_cast<const bf16x8*>(sc + SCR + rl * SP + ch * 16);
      if (row < prows) {
        *reinterpret_cast<bf16x8*>(p.C + (size_t)(pm0 + row) * p.ldc + pn0 + 32 * S + ch * 8) = v;
        if (EPI == EPI_GELU) store_c2(p.C2 + (size_t)(pm0 + row) * p.ldc2 + pn0 + 32 * S + ch * 8, d);
      }
    }
    return EPI == EPI_GELU ? 4 : 2;
  };
  // vmcnt is ONE in-order queue: a wait for k-tile j must allow for everything issued after j's DMA -- the two younger k-tiles and
  // the stores of the last three k-tiles (s1, s2, s3; counted exactly: an immediate is all s_waitcnt takes, hence the switch)
  int s1 = 0, s2 = 0, s3 = 0;
#ifdef KS_PROF
  unsigned long long pw[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tstart = __builtin_readcyclecounter();
#endif
#ifndef KS_LATE
#define KS_LATE 1
#endif
  const bool late = KS_LATE && w >= 4;
  auto wait_vm = [&](int allow) {
    switch (allow >> 1) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
  };
  // one k-tile: wait for it, barrier, request the tile three ahead, [slice into scratch], 12 MFMAs, [slice out of scratch -> stores]
  auto ktile = [&](auto put_tag, auto get_tag, int extra_vm) {
    constexpr int PS = decltype(put_tag)::value, GS = decltype(get_tag)::value;      // slice to write / to store, -1: none
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KSTAMP(t0);
    wait_vm(2 * NDMA + s1 + s2 + (late ? 0 : s3));
    KSTAMP(t1);
    __builtin_amdgcn_s_barrier();
    KSTAMP(t2);
    // The two waves of a SIMD leave the barrier together.  A slice's epilogue is VALU + LDS work (table GELU: ~130 instructions and
    // two LDS round trips per wave), its way out 2 - 4 store instructions, the MFMA block in between leaves the VALU idle: waves
    // 0 - 3 run slice-in, MFMAs, slice-out; their SIMD partners 4 - 6 ("late") slice-out (of the slice before), MFMAs, slice-in --
    // one wave's slice work sits under the other's MFMAs.  (Stores first: the late waves' vmcnt allowance has no s3 term.)
    int sn = extra_vm;
    if constexpr (GS >= 0) {
      if (have && late) sn += get_slice(get_tag);
    }
    if constexpr (PS >= 0) {
      if (have && !late) put_slice(put_tag);
    }
#ifdef KS_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    KSTAMP(t3);
    const unsigned char* sA = smem + stC * STAGE + a_row;
    const unsigned char* sW = smem + stC * STAGE + A_STAGE;
    stC = (stC + 1) & (NSTAGE - 1);
    // One scheduling region: the seven fragments of the first 16 k, then one MFMA at a time with a fragment of the second 16 k (into
    // the registers the MFMA just read) and, four times, a DMA request of the k-tile three ahead behind it.  Left to itself the
    // compiler reads two fragments, waits, issues their MFMAs, reads the next two ... (seven exposed LDS latencies per k-tile), and
    // all waves of the CU pushed their DMA requests into the texture addresser in one burst right behind the barrier.
    __builtin_amdgcn_sched_barrier(0);
    {
#define SB __builtin_amdgcn_sched_barrier(0)
      Frag<bf16> fa0, fa1, fb0[6], fb1[6];
      unsigned char* stD = smem + stI * STAGE;
      const int k0 = ikt * TK;
      auto dma = [&](int j) {
        int i = w + NCW * j;
        i = i < NSLOT ? i : i - NSLOT;
        const bf16* base = i < ASLOT ? p.A : p.W;
        __builtin_amdgcn_global_load_lds((glb_ptr)(base + (size_t)offI[j] + k0), (lds_ptr)(stD + i * 1024), 16, 0, 0);
      };
      fa0.v = *reinterpret_cast<const bf16x8*>(sA + foff[0]);
#pragma unroll
      for (int b = 0; b < 6; ++b) fb0[b].v = *reinterpret_cast<const bf16x8*>(sW + 32 * b * TKB + foff[0]);
      SB;
      mma(acc[0], fb0[0], fa0); SB;
      fa1.v = *reinterpret_cast<const bf16x8*>(sA + foff[1]);
      fb1[0].v = *reinterpret_cast<const bf16x8*>(sW + 32 * 0 * TKB + foff[1]); SB;
      mma(acc[1], fb0[1], fa0); SB;
      fb1[1].v = *reinterpret_cast<const bf16x8*>(sW + 32 * 1 * TKB + foff[1]); dma(0); SB;
      mma(acc[2], fb0[2], fa0); SB;
      fb1[2].v = *reinterpret_cast<const bf16x8*>(sW + 32 * 2 * TKB + foff[1]); dma(1); SB;
      mma(acc[3], fb0[3], fa0); SB;
      fb1[3].v = *reinterpret_cast<const bf16x8*>(sW + 32 * 3 * TKB + foff[1]); dma(2); SB;
      mma(acc[4], fb0[4], fa0); SB;
      fb1[4].v = *reinterpret_cast<const bf16x8*>(sW + 32 * 4 * TKB + foff[1]); dma(3); SB;
      mma(acc[5], fb0[5], fa0); SB;
      fb1[5].v = *reinterpret_cast<const bf16x8*>(sW + 32 * 5 * TKB + foff[1]); SB;
#pragma unroll
      for (int b = 0; b < 6; ++b) mma(acc[b], fb1[b], fa1);
#undef SB
    }
    __builtin_amdgcn_sched_barrier(0);
    advance();
#ifdef KS_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KSTAMP(t3b);
#endif
    if constexpr (PS >= 0) {
      if (have && late) put_slice(put_tag);
    }
    if constexpr (GS >= 0) {
      if (have && !late) sn += get_slice(get_tag);
    }
    s3 = s2; s2 = s1; s1 = sn;
#ifdef KS_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KSTAMP(t4);
    pw[0] += t1 - t0; pw[1] += t2 - t1; pw[2] += t3 - t2; pw[3] += t3b - t3; pw[5] += t4 - t3b; pw[4] += 1;
#endif
  };

#pragma unroll 1
  for (int u = u0; u < J; u = next_unit(u)) {
    int m0, n0, rows;
    unit_geo(u, m0, n0, rows);
    // k-tile 0 (its barrier is behind the previous unit's parking, which read Bs): this unit's bias follows it into LDS
    const bool bias_dma = p.bias && w < BN / 64;
    // Slice schedule.  The epilogue arithmetic of a slice (GELU: ~200 VALU instructions per wave) and a k-tile's MFMAs come from
    // the same instruction stream, so inside one wave they add up; a SIMD's two waves overlap them only when one is in its slice
    // while the other is in its MFMAs: waves 0 - 3 take their slices on the even k-tiles (stores on the odd ones), their SIMD
    // partners 4 - 6 on the odd ones (slice, MFMAs, stores in one k-tile)
#ifndef KS_STAGGER
#define KS_STAGGER 0
#endif
    if (!KS_STAGGER || w < 4) {
      ktile(IntTag<0>(), IntTag<-1>(), 0);
      if (bias_dma) {
        __builtin_amdgcn_global_load_lds((glb_ptr)(p.bias + n0 + 64 * w + lane), (lds_ptr)(Bs + 64 * w), 4, 0, 0);
        s1 += 1;
      }
      ktile(IntTag<-1>(), IntTag<0>(), 0);
      ktile(IntTag<1>(), IntTag<-1>(), 0); ktile(IntTag<-1>(), IntTag<1>(), 0);
      ktile(IntTag<2>(), IntTag<-1>(), 0); ktile(IntTag<-1>(), IntTag<2>(), 0);
      ktile(IntTag<3>(), IntTag<-1>(), 0); ktile(IntTag<-1>(), IntTag<3>(), 0);
      ktile(IntTag<4>(), IntTag<-1>(), 0); ktile(IntTag<-1>(), IntTag<4>(), 0);
      ktile(IntTag<5>(), IntTag<-1>(), 0); ktile(IntTag<-1>(), IntTag<5>(), 0);
    } else {
      ktile(IntTag<-1>(), IntTag<-1>(), 0); ktile(IntTag<0>(), IntTag<0>(), 0);
      ktile(IntTag<-1>(), IntTag<-1>(), 0); ktile(IntTag<1>(), IntTag<1>(), 0);
      ktile(IntTag<-1>(), IntTag<-1>(), 0); ktile(IntTag<2>(), IntTag<2>(), 0);
      ktile(IntTag<-1>(), IntTag<-1>(), 0); ktile(IntTag<3>(), IntTag<3>(), 0);
      ktile(IntTag<-1>(), IntTag<-1>(), 0); ktile(IntTag<4>(), IntTag<4>(), 0);
      ktile(IntTag<-1>(), IntTag<-1>(), 0); ktile(IntTag<5>(), IntTag<5>(), 0);
    }
#pragma unroll 1
    for (int kt = 2 * NSLICE; kt < T2; ++kt) ktile(IntTag<-1>(), IntTag<-1>(), 0);
    // park: + bias, one rounding to bf16; the accumulators start the next unit
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + 32 * b + 8 * q + 4 * g);
        parked[b * 8 + 2 * q] = pack2(acc[b][4 * q + 0] + bv[0], acc[b][4 * q + 1] + bv[1]);
        parked[b * 8 + 2 * q + 1] = pack2(acc[b][4 * q + 2] + bv[2], acc[b][4 * q + 3] + bv[3]);
        acc[b][4 * q + 0] = 0.f; acc[b][4 * q + 1] = 0.f; acc[b][4 * q + 2] = 0.f; acc[b][4 * q + 3] = 0.f;
      }
    have = true;
    pm0 = m0; pn0 = n0; prows = rows;
  }
  // the last unit's tile: wave-private scratch, no barrier needed
  auto drain = [&](auto tag) {
    put_slice(tag);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    get_slice(tag);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  drain(IntTag<0>()); drain(IntTag<1>()); drain(IntTag<2>()); drain(IntTag<3>()); drain(IntTag<4>()); drain(IntTag<5>());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the repeated requests behind the stream's end still target this LDS
#ifdef KS_PROF
  if (lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_ks_prof + (blockIdx.x * 8 + w) * 6;
    o[0] = pw[0]; o[1] = pw[1]; o[2] = pw[2]; o[3] = pw[3]; o[4] = pw[4]; o[5] = pw[5];
  }
#endif
}

template <int EPI, bool TAB>
int launch(const KsArgs& p, hipStream_t st) {
  constexpr int smem_bytes = SMEM + (TAB ? TAB_RESERVE : 0);
  static DevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute((const void*)gemm_nt_kstream_kernel<EPI, TAB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes) != hipSuccess)
      return RGBNM_ELAUNCH;
    attr.done();
  }
  const int J = ((p.npanels + 7) / 8) * p.ntiles;             // units per XCD
  const int slots = J < 32 ? J : 32;                          // 32 CUs per XCD, one workgroup each
  hipLaunchKernelGGL((gemm_nt_kstream_kernel<EPI, TAB>), dim3(8 * slots), dim3(NTHREADS), smem_bytes, st, p);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // namespace ks

#ifdef KS_PROF
extern "C" int rgbnm_debug_ks_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ks::g_ks_prof), sizeof(unsigned long long) * 256 * 8 * 6) == hipSuccess ? 0 : -1;
}
#endif

// returns 1 when the shape / epilogue is not eligible (the caller goes on to gemm_nt_kpipe.hip)
int rgbnm_launch_nt_kstream(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                            const void* R, int ldr, void* C2, int ldc2, int M, int N, int K, hipStream_t st) {
  using namespace ks;
  (void)R; (void)ldr;
  if (epi != EPI_NONE && epi != EPI_GELU) return 1;
  if (N % BN || N < 2 * BN || K % TK || K < 2 * NSLICE * TK || lda % 8 || ldw % 8 || ldc % 8 || M < 8192) return 1;
  if (epi == EPI_GELU && (!C2 || ldc2 % 8)) return 1;
  if ((long long)M * (lda > ldw ? lda : ldw) >= (1LL << 32) || (long long)N * ldw >= (1LL << 32)) return 1;   // 32-bit element offsets
  KsArgs p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = (bf16*)C; p.bias = bias; p.C2 = (bf16*)C2;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldc2 = ldc2; p.M = M; p.K = K;
  p.ntiles = N / BN;
  p.npanels = cdiv(M, BM);
  const double mn = (double)M * N;
  const int slot = rgbnm_trace_begin(TR_NT, 2.0 * mn * K, ((double)M * K + (double)N * K) * 2.0 + mn * 2.0 + (epi != 0 ? mn * 2.0 : 0.0), st);
  // table GELU (option gelu_table) on devices where rgbnm_gelu_table_init found a usable table: bit-identical to the arithmetic
  // form for every bf16 input (tests/test_gelu_table.py), a third of its instructions
  p.tab_img = nullptr; p.kneg = p.kpos = p.klo = p.koff = p.ksgn = 0;
  bool table = false;
  if (epi == EPI_GELU && rgbnm_get_option("gelu_table")) {
    int A0 = 0, P1 = 0, N1 = 0, ndw = 0;
    const unsigned* img = nullptr;
    if (rgbnm_gelu_table_query(&img, &A0, &P1, &N1, &ndw) == 1 && ndw * 4 <= TAB_BYTES) {
      table = true;
      p.tab_img = img;
      p.kneg = 0x00010001u * (unsigned)(0x8000 | N1);
      p.kpos = 0x00010001u * (unsigned)P1;
      p.klo = 0x00010001u * (unsigned)(A0 - 1);
      p.koff = 0x00010001u * (unsigned)((0x10000 - 4 * (A0 - 1)) & 0xffff);
      p.ksgn = 0x00010001u * (unsigned)(4 * (P1 - A0 + 2));
    }
  }
  const int rc = epi == EPI_GELU ? (table ? launch<EPI_GELU, true>(p, st) : launch<EPI_GELU, false>(p, st)) : launch<EPI_NONE, false>(p, st);
  rgbnm_trace_end(slot, st);
  return rc;
}
