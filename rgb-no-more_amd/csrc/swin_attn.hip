// SwinV2 window attention on the MFMA pipe (reference: WindowAttention.forward + the shift / partition plumbing of
// SwinTransformerBlock.forward, models/swinv2.py:143-182, 273-300; BASELINE config 5 "windowed-attention HIP kernel").
//
//   logits[i][j] = cos(q_i, k_j) * exp(min(logit_scale_h, ln 100)) + bias_h[i][j] + shift_mask[i][j],  8x8 windows,
//   head_dim 32, softmax over the 64 keys of the window, O = P V.
//
// One wave per (window, head).  Gather: the head slice of a token is 64 contiguous bytes (bf16), so LPT = 4 lanes fetch one
// token with a 16-byte load each -- 16 tokens per load instruction, 4 instructions per operand (a lane-per-token gather with
// 8-byte loads touched 64 cache lines per instruction, 40 instructions per window in the backward: the texture-address path,
// not HBM, set the pace: 1.1 TB/s).  Cyclic shift + window partition are index arithmetic (token = f(window, position, shift));
// outputs are scattered back the same way, so no rolled / partitioned copy of the activations exists.  |q|, |k| and
// D = dO . O are 4-lane DPP sums; the L2-normalised rows are parked in wave-private LDS tiles (row-major for operands reduced
// over d, transposed for operands reduced over tokens).  The math then follows the ViT attention kernels (attention.hip):
// swapped orientation S^T = K_n Q_n^T so a lane owns one query and 16 keys per 32-key tile in registers - softmax is
// register-local plus one lane^32 exchange, and the probabilities are the next MFMA's B operand straight from registers.
// 16 MFMAs forward, 56 backward per (window, head) in bf16 (32x32x16).
// Launch: persistent.  Every workgroup belongs to ONE head: that head's 64 x 64 position bias sits in LDS (pitch 68: row reads
// for the query-major phase, column reads for the key-major phase -- no transposed copy, no per-lane global row reads), and
// each wave walks a contiguous range of windows, the next window's operands requested before the current one is computed
// (bf16; registers are free at one wave per SIMD, which the LDS tiles dictate).  Backward: the wave keeps its head's d(bias)
// in 64 registers (accumulator layout) over all its windows; it leaves as one partial slice per wave, summed by the batched
// deterministic reduction (no atomics); d(logit_scale) leaves as one partial per (window, head).
// Templated on T in {float, bf16} (fp32 = exact 32x32x2 MFMA, the parity mode).
#include "common.h"
#include <type_traits>
#include "../../include/rgbnm.h"
#include "internal.h"

namespace {

#ifdef WIN_PROF   // experiments only: cycle stamps of every wave's second window (tools/winattn_prof.py); 1 = backward, 2 = forward
__device__ unsigned long long g_win_prof[1024 * 4 * 8];
#define WPROF_(i) do { if (lane == 0 && win == win_lo + 1 && blockIdx.x < 1024) g_win_prof[(blockIdx.x * 4 + w) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#if WIN_PROF == 2
#define WPROF(i) do { } while (0)
#define WPROFF(i) WPROF_(i)
#else
#define WPROF(i) WPROF_(i)
#define WPROFF(i) do { } while (0)
#endif
#else
#define WPROF(i) do { } while (0)
#define WPROFF(i) do { } while (0)
#endif

constexpr int WS = 8, WT = 64, HD = 32;
constexpr int RP = HD + 8;        // row-major tile pitch (elements): 16-byte aligned rows, staggered banks
constexpr int TP = WT + 4;        // transposed tile pitch
constexpr int BP = WT + 4;        // position-bias pitch in LDS (floats): float4 row reads and column reads both conflict free
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct WA {
  static constexpr int EPL = Frag<T>::EPL;
  static constexpr int NCH = HD * (int)sizeof(T) / 32;   // 32-byte chunks along d: 2 (bf16) / 4 (f32)
  static constexpr int CH = 32 / (int)sizeof(T);
  static constexpr int FPT = 16 / EPL;                   // fragments per 32-token tile per lane: 2 / 4
  static constexpr int ROW_T = WT * RP * (int)sizeof(T);  // bytes of a row-major tile
  static constexpr int TR_T = HD * TP * (int)sizeof(T);   // bytes of a transposed tile
  static constexpr int SMALL = 5 * WT * 4;                // lse, D, |q|, |k|, mask id
  // bf16 reads operands whose reduction axis is the token axis (V^T, K^T, Q^T, dO^T) out of the ROW-major tiles with
  // ds_read_b64_tr_b16; fp32 has no transposing read and parks transposed copies
  static constexpr bool TRREAD = sizeof(T) == 2;
  static constexpr int FWD_WAVE = 2 * ROW_T + (TRREAD ? ROW_T : TR_T) + WT * 4;   // Qn, Kn, V (bf16) / V^T (fp32), mask id
  // backward: Qn, Kn, V, dO rows (+ bf16: a staging tile for dq; dk / dv are staged in their own dead Kn / V rows)
  static constexpr bool STAGE_DQ = sizeof(T) == 2;
  static constexpr int BWD_WAVE = 4 * ROW_T + (TRREAD ? 0 : 3 * TR_T) + (STAGE_DQ ? ROW_T : 0) + SMALL;
  static constexpr int FWD_WAVES = 4;
  static constexpr int BWD_WAVES = sizeof(T) == 2 ? 4 : 2;
  static constexpr int BIAS = WT * BP * 4;                // the workgroup's head: bias[64][BP] fp32
  static constexpr int FWD_LDS = BIAS + FWD_WAVES * FWD_WAVE;
  static constexpr int BWD_LDS = BIAS + BWD_WAVES * BWD_WAVE;
  // gather geometry: LPT lanes fetch one token's head slice (HD elements) as 16-byte pieces of EP elements
  static constexpr int LPT = HD * (int)sizeof(T) / 16;    // 4 / 8
  static constexpr int EP = 16 / (int)sizeof(T);          // 8 / 4
  static constexpr int TPI = 64 / LPT;                    // tokens per load instruction: 16 / 8
  static constexpr int NI = WT / TPI;                     // load instructions per operand: 4 / 8
  static constexpr bool PREFETCH = sizeof(T) == 2;        // next window's operands in registers during the math (80 VGPRs)
};

template <typename T> __device__ __forceinline__ void unpack16(const u32x4& r, float (&f)[WA<T>::EP]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __builtin_bit_cast(float, r[i] << 16);
      f[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
    }
  } else {
    const f32x4 t = __builtin_bit_cast(f32x4, r);
    f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3];
  }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float (&f)[WA<T>::EP]) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    return __builtin_bit_cast(u32x4, v);
  } else {
    return (u32x4){__builtin_bit_cast(unsigned, f[0]), __builtin_bit_cast(unsigned, f[1]), __builtin_bit_cast(unsigned, f[2]),
                   __builtin_bit_cast(unsigned, f[3])};
  }
}
// sum over the LPT consecutive lanes that share a token
template <typename T> __device__ __forceinline__ float token_sum(float v) {
  v += lane_xor1(v);
  v += lane_xor2(v);
  if constexpr (WA<T>::LPT == 8) v += lane_xor4(v);
  return v;
}
// Softmax in base 2: logits are kept as log2(e) * (scale cos + bias) -- the bias is scaled once when it is staged, the scale and
// lse once per wave / window -- so a probability is ONE v_exp_f32 of a difference (exp(x) costs a multiply more per element).
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
// Workgroup -> (head, slot among the head's workgroups).  gpx = 0: head-major (bph workgroups per head).  gpx > 0: XCD-aware --
// workgroup b runs on XCD b % 8, whose L2 it shares; the heads of one window range sit on ONE XCD (slot s of the XCD: head
// s % heads, group s / heads, gpx groups per XCD), so the half of a 128-byte line that belongs to the neighbouring head is an
// L2 hit for that head's workgroup instead of a second HBM fetch (a head's slice of a token is 64 bytes).  Returns false for the
// few workgroups of an XCD that are left over when its slots do not divide by the head count.
__device__ __forceinline__ bool head_slot(int bph, int gpx, int heads, int& h, int& slot, int& nslots) {
  if (gpx == 0) {
    h = blockIdx.x / bph;
    slot = blockIdx.x % bph;
    nslots = bph;
    return true;
  }
  const int c = blockIdx.x & 7, s = blockIdx.x >> 3;
  if (s >= gpx * heads) return false;
  h = s % heads;
  slot = c * gpx + s / heads;
  nslots = 8 * gpx;
  return true;
}
// the workgroup's head bias [64][64] -> LDS [64][BP], times log2(e)
__device__ __forceinline__ void stage_bias(float* Bs, const float* __restrict__ bias_h) {
  for (int i = threadIdx.x; i < WT * WT / 4; i += blockDim.x)
    *reinterpret_cast<f32x4*>(Bs + (i >> 4) * BP + (i & 15) * 4) = *reinterpret_cast<const f32x4*>(bias_h + i * 4) * LOG2E;
}

__device__ __forceinline__ int region(int s, int res, int shift) { return s < res - WS ? 0 : (s < res - shift ? 1 : 2); }
// token index (in the un-shifted image) and mask id of local position i of window (wy, wx)
__device__ __forceinline__ int win_token(int i, int wy, int wx, int res, int shift, int& mid) {
  const int sy = wy * WS + (i >> 3), sx = wx * WS + (i & 7);            // coordinates in the shifted frame
  mid = shift ? 3 * region(sy, res, shift) + region(sx, res, shift) : 0;
  int yy = sy + shift, xx = sx + shift;                                  // shifted[y] = x[(y + shift) % res]
  yy = yy >= res ? yy - res : yy;
  xx = xx >= res ? xx - res : xx;
  return yy * res + xx;
}

// window coordinates (image, window row, window column), walked incrementally: a 64-bit `win % nw` per window and per
// prefetch cost more than the window's MFMAs
struct WinPos {
  int b, wy, wx;
  __device__ __forceinline__ void set(long long win, int nw) {
    wx = (int)(win % nw);
    wy = (int)((win / nw) % nw);
    b = (int)(win / ((long long)nw * nw));
  }
  __device__ __forceinline__ void next(int nw) {
    if (++wx == nw) {
      wx = 0;
      if (++wy == nw) {
        wy = 0;
        ++b;
      }
    }
  }
};
// 1 / max(|x|, 1e-12) from |x|^2  (F.normalize's eps; v_rsq_f32 is 1 ulp)
__device__ __forceinline__ float inv_norm(float n2) { return fminf(__builtin_amdgcn_rsqf(n2), 1e12f); }

template <typename T> __device__ __forceinline__ Frag<T> rowfrag(const T* tile, int row, int c, int g) {
  return load_frag<T>(tile + row * RP + c * WA<T>::CH + g * WA<T>::EPL);
}
// The score / score-gradient MFMAs of one 32-row tile with ALL their row fragments requested first (pinned by sched_barrier): left
// alone the compiler emits read - wait - MFMA per fragment with one buffer -- an exposed LDS latency per MFMA in a kernel that runs
// one wave per SIMD (nothing else to hide it).  Same MFMA order per accumulator: same bits.
template <typename T, int NCH>
__device__ __forceinline__ void row_pair_mma(f32x16& sa, f32x16& da, const T* tileS, const T* tileD, int row, int g,
                                             const Frag<T> (&xs)[NCH], const Frag<T> (&xd)[NCH]) {
  if constexpr (sizeof(T) == 2) {
    Frag<T> fs[NCH], fd[NCH];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      fs[c] = rowfrag<T>(tileS, row, c, g);
      fd[c] = rowfrag<T>(tileD, row, c, g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      mma(sa, fs[c], xs[c]);
      mma(da, fd[c], xd[c]);
    }
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      mma(sa, rowfrag<T>(tileS, row, c, g), xs[c]);
      mma(da, rowfrag<T>(tileD, row, c, g), xd[c]);
    }
  }
}
// A-operand fragment from a transposed tile [d][TP]: row d, tokens of fragment `fi` of 32-token tile `t`, lane group g
template <typename T> __device__ __forceinline__ Frag<T> tfrag(const T* img, int d, int t, int fi, int g) {
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    const T* p = img + d * TP + 32 * t + 16 * fi + 4 * g;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 8);
    f.v = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  } else {
    f.v = *reinterpret_cast<const f32x4*>(img + d * TP + 32 * t + 8 * fi + 4 * g);
  }
  return f;
}
// bf16: the same fragment as tfrag(img, d = lane & 31, t, fi, g) read from the row-major tile [token][RP].  A 16-lane group
// (fixed g, G1) hands ds_read_b64_tr_b16 a 4-row x 16-column block -- lane (k, l3) points at row 4 g + k, columns
// 16 G1 + 4 l3 .. +3 -- and gets back column 16 G1 + 4 k + l3 (= lane & 31) of rows 4 g .. 4 g + 3; the second read takes the
// rows 8 further down (tfrag's `hi` half).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned tr_base(const bf16* tile, int lane) {
  const int g = lane >> 5, G1 = (lane >> 4) & 1, k = (lane >> 2) & 3, l3 = lane & 3;
  return (unsigned)(size_t)tile + (unsigned)((4 * g + k) * RP * 2 + (16 * G1 + 4 * l3) * 2);
}
__device__ __forceinline__ Frag<bf16> trfrag(unsigned a) {   // a = tr_base + byte offset of fragment (t, fi): (32 t + 16 fi) rows
  u32x2 lo, hi;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %2\n\t"
      "ds_read_b64_tr_b16 %1, %2 offset:%3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(lo), "=&v"(hi)
      : "v"(a), "i"(8 * RP * 2)
      : "memory");
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
  return f;
}
// fragment (t, fi) of a token-reduced operand: transposing read of the row tile (bf16) or plain read of the transposed copy
template <typename T>
__device__ __forceinline__ Frag<T> tok_frag(unsigned trb, const T* timg, int l31, int t, int fi, int g) {
  if constexpr (sizeof(T) == 2) return trfrag(trb + (unsigned)((32 * t + 16 * fi) * RP * 2));
  else return tfrag<T>(timg, l31, t, fi, g);
}

template <typename T> __device__ __forceinline__ Frag<T> pfrag(const float (&p)[16], int fi) {
  Frag<T> f;
#pragma unroll
  for (int j = 0; j < Frag<T>::EPL; ++j) f.v[j] = from_f32<T>(p[fi * Frag<T>::EPL + j]);
  return f;
}
template <typename T> __device__ __forceinline__ void put_row(T* tile, int row, const float (&v)[HD], float s) {
#pragma unroll
  for (int d = 0; d < HD; d += 4) store4<T>(tile + row * RP + d, (f32x4){v[d] * s, v[d + 1] * s, v[d + 2] * s, v[d + 3] * s});
}
template <typename T> __device__ __forceinline__ void put_col(T* img, int col, const float (&v)[HD], float s) {
#pragma unroll
  for (int d = 0; d < HD; ++d) img[d * TP + col] = from_f32<T>(v[d] * s);
}

// ------------------------------------------------------------------------------------------------ forward
// raw 16-byte pieces of one window's q, k, v rows for this lane: piece n covers token n * TPI + lane / LPT, elements (lane % LPT) * EP ..
template <typename T> struct RawQKV { u32x4 q[WA<T>::NI], k[WA<T>::NI], v[WA<T>::NI]; };

template <typename T>
__device__ __forceinline__ void fetch_qkv(RawQKV<T>& r, const T* __restrict__ qkv, const WinPos& wp, int h, int res, int C,
                                          int shift, int lane) {
  using A = WA<T>;
  const int wx = wp.wx, wy = wp.wy, b = wp.b;
#pragma unroll
  for (int n = 0; n < A::NI; ++n) {
    int mid;
    const int tok = win_token(n * A::TPI + lane / A::LPT, wy, wx, res, shift, mid);
    const T* row = qkv + ((size_t)b * res * res + tok) * 3 * C + h * HD + (lane % A::LPT) * A::EP;
    r.q[n] = *reinterpret_cast<const u32x4*>(row);
    r.k[n] = *reinterpret_cast<const u32x4*>(row + C);
    r.v[n] = *reinterpret_cast<const u32x4*>(row + 2 * C);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void win_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                                           const float* __restrict__ scale, T* __restrict__ out,
                                                           float* __restrict__ lse, int B, int res, int C, int heads,
                                                           int shift, int bph, int gpx) {
  using A = WA<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char win_smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* Bs = reinterpret_cast<float*>(win_smem);
  unsigned char* base = win_smem + A::BIAS + w * A::FWD_WAVE;
  T* Qn = reinterpret_cast<T*>(base);
  T* Kn = reinterpret_cast<T*>(base + A::ROW_T);
  T* Vt = reinterpret_cast<T*>(base + 2 * A::ROW_T);          // bf16: V row-major [token][RP]; fp32: V^T [d][TP]
  int* Mid = reinterpret_cast<int*>(base + A::FWD_WAVE - WT * 4);
  unsigned trV = 0;
  if constexpr (A::TRREAD) trV = tr_base(Vt, lane);
  int h, slot, nslots;
  if (!head_slot(bph, gpx, heads, h, slot, nslots)) return;
  stage_bias(Bs, bias + (size_t)h * WT * WT);
  __syncthreads();
  const int nw = res / WS;
  const long long nwin = (long long)B * nw * nw;
  const int wph = nslots * A::FWD_WAVES, wih = slot * A::FWD_WAVES + w;      // waves of this head, index among them
  const long long win_lo = nwin * wih / wph, win_hi = nwin * (wih + 1) / wph;
  const int l31 = lane & 31, g = lane >> 5;
  const float sc = scale[h] * LOG2E;                 // logits in base-2 units (stage_bias)
  RawQKV<T> raw;
  WinPos nxt;
  nxt.set(win_lo, nw);
  if (A::PREFETCH && win_lo < win_hi) fetch_qkv<T>(raw, qkv, nxt, h, res, C, shift, lane);
#pragma unroll 1
  for (long long win = win_lo; win < win_hi; ++win) {
    const long long unit = win * heads + h;
    const int wx = nxt.wx, wy = nxt.wy, b = nxt.b;
    const bool masked = shift && (wy == nw - 1 || wx == nw - 1);      // only the last row / column of windows mixes regions
    WPROFF(0);
    if (!A::PREFETCH) fetch_qkv<T>(raw, qkv, nxt, h, res, C, shift, lane);
    nxt.next(nw);
    __builtin_amdgcn_wave_barrier();                   // the previous window's LDS reads are done (same wave, in order)
#pragma unroll
    for (int n = 0; n < A::NI; ++n) {
      const int tk = n * A::TPI + lane / A::LPT, ch = lane % A::LPT;
      float q[A::EP], k[A::EP], v[A::EP];
      unpack16<T>(raw.q[n], q);
      unpack16<T>(raw.k[n], k);
      unpack16<T>(raw.v[n], v);
      float nq = 0.f, nk = 0.f;
#pragma unroll
      for (int e = 0; e < A::EP; ++e) {
        nq += q[e] * q[e];
        nk += k[e] * k[e];
      }
      const float iq = inv_norm(token_sum<T>(nq)), ik = inv_norm(token_sum<T>(nk));   // F.normalize(eps = 1e-12)
#pragma unroll
      for (int e = 0; e < A::EP; ++e) {
        q[e] *= iq;
        k[e] *= ik;
      }
      *reinterpret_cast<u32x4*>(Qn + tk * RP + ch * A::EP) = pack16<T>(q);
      *reinterpret_cast<u32x4*>(Kn + tk * RP + ch * A::EP) = pack16<T>(k);
      if constexpr (A::TRREAD) {
        *reinterpret_cast<u32x4*>(Vt + tk * RP + ch * A::EP) = raw.v[n];
      } else {
#pragma unroll
        for (int e = 0; e < A::EP; ++e) Vt[(ch * A::EP + e) * TP + tk] = from_f32<T>(v[e]);
      }
      if (ch == 0) {
        int mid;
        (void)win_token(tk, wy, wx, res, shift, mid);
        Mid[tk] = mid;
      }
    }
    WPROFF(1);
    if (A::PREFETCH && win + 1 < win_hi) fetch_qkv<T>(raw, qkv, nxt, h, res, C, shift, lane);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WPROFF(2);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int q = 32 * i + l31;
      int midq;
      (void)win_token(q, wy, wx, res, shift, midq);
      Frag<T> qf[A::NCH];
#pragma unroll
      for (int c = 0; c < A::NCH; ++c) qf[c] = rowfrag<T>(Qn, q, c, g);
      const float* brow = Bs + q * BP;
      float s[2][16];
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < A::NCH; ++c) mma(acc, rowfrag<T>(Kn, 32 * t + l31, c, g), qf[c]);
        auto logits = [&](auto MK) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {               // registers 4 q4 .. +3 <-> keys 32 t + 8 q4 + 4 g + 0..3
            const int k0 = 32 * t + 8 * q4 + 4 * g;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + k0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[4 * q4 + e] * sc + b4[e];
              if constexpr (decltype(MK)::value) {
                if (Mid[k0 + e] != midq) v += -100.f * LOG2E;
              }
              s[t][4 * q4 + e] = v;
              m = fmaxf(m, v);
            }
          }
        };
        if (masked) logits(std::true_type{});
        else logits(std::false_type{});
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m);
          sum += s[t][r];
        }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      if (g == 0) lse[unit * WT + q] = (m + __log2f(sum)) * LN2;      // stored in natural units
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] *= inv;
#pragma unroll
        for (int fi = 0; fi < A::FPT; ++fi) mma(o, tok_frag<T>(trV, Vt, l31, t, fi, g), pfrag<T>(s[t], fi));
      }
      // o[r] = O[q][acc_row(r)] -> the tile's own (now dead) Qn rows; HBM gets whole 64-byte token slices below
      T* orow = Qn + q * RP;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        store4<T>(orow + rq * 8 + g * 4, (f32x4){o[rq * 4 + 0], o[rq * 4 + 1], o[rq * 4 + 2], o[rq * 4 + 3]});
      WPROFF(3 + i);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < A::NI; ++n) {                  // the gather's geometry backwards: 16-byte pieces, LPT lanes per token
      const int tk = n * A::TPI + lane / A::LPT, ch = lane % A::LPT;
      int mid;
      const int tok = win_token(tk, wy, wx, res, shift, mid);
      *reinterpret_cast<u32x4*>(out + ((size_t)b * res * res + tok) * C + h * HD + ch * A::EP) =
          *reinterpret_cast<const u32x4*>(Qn + tk * RP + ch * A::EP);
    }
    WPROFF(5);
  }
}

// ------------------------------------------------------------------------------------------------ backward
template <typename T> struct RawBwd { u32x4 q[WA<T>::NI], k[WA<T>::NI], v[WA<T>::NI], g[WA<T>::NI], o[WA<T>::NI]; float lse; };

template <typename T>
__device__ __forceinline__ void fetch_bwd(RawBwd<T>& r, const T* __restrict__ qkv, const T* __restrict__ out,
                                          const T* __restrict__ dout, const float* __restrict__ lse_unit, const WinPos& wp, int h,
                                          int res, int C, int shift, int lane) {
  using A = WA<T>;
  const int wx = wp.wx, wy = wp.wy, b = wp.b;
  r.lse = lse_unit[lane];                              // with the operands: asked for inside the park it was an exposed round trip
#pragma unroll
  for (int n = 0; n < A::NI; ++n) {
    int mid;
    const int tok = win_token(n * A::TPI + lane / A::LPT, wy, wx, res, shift, mid);
    const size_t trow = (size_t)b * res * res + tok;
    const int col = h * HD + (lane % A::LPT) * A::EP;
    const T* row = qkv + trow * 3 * C + col;
    r.q[n] = *reinterpret_cast<const u32x4*>(row);
    r.k[n] = *reinterpret_cast<const u32x4*>(row + C);
    r.v[n] = *reinterpret_cast<const u32x4*>(row + 2 * C);
    r.g[n] = *reinterpret_cast<const u32x4*>(dout + trow * C + col);
    r.o[n] = *reinterpret_cast<const u32x4*>(out + trow * C + col);
  }
}

template <typename T>
__global__ __launch_bounds__(64 * WA<T>::BWD_WAVES) void win_attn_bwd_kernel(
    const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ lse, T* __restrict__ dqkv, float* __restrict__ dpart,
    float* __restrict__ dscale_part, int B, int res, int C, int heads, int shift, int bph, int gpx) {
  using A = WA<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char win_smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* Bs = reinterpret_cast<float*>(win_smem);
  unsigned char* base = win_smem + A::BIAS + w * A::BWD_WAVE;
  T* Qn = reinterpret_cast<T*>(base);
  T* Kn = reinterpret_cast<T*>(base + A::ROW_T);
  T* Vr = reinterpret_cast<T*>(base + 2 * A::ROW_T);
  T* Gr = reinterpret_cast<T*>(base + 3 * A::ROW_T);
  T* Knt = reinterpret_cast<T*>(base + 4 * A::ROW_T);          // transposed copies: fp32 only
  T* Qnt = reinterpret_cast<T*>(base + 4 * A::ROW_T + A::TR_T);
  T* Gt = reinterpret_cast<T*>(base + 4 * A::ROW_T + 2 * A::TR_T);
  T* St = reinterpret_cast<T*>(base + A::BWD_WAVE - A::SMALL - A::ROW_T);      // bf16 only
  float* Ls = reinterpret_cast<float*>(base + A::BWD_WAVE - A::SMALL);
  unsigned trK = 0, trQ = 0, trG = 0;
  if constexpr (A::TRREAD) {
    trK = tr_base(Kn, lane);
    trQ = tr_base(Qn, lane);
    trG = tr_base(Gr, lane);
  }
  float* Ds = Ls + WT;
  float* Rq = Ds + WT;
  float* Rk = Rq + WT;
  int* Mid = reinterpret_cast<int*>(Rk + WT);
  int h, slot, nslots;
  if (!head_slot(bph, gpx, heads, h, slot, nslots)) return;
  stage_bias(Bs, bias + (size_t)h * WT * WT);
  __syncthreads();
  const int nw = res / WS;
  const long long nwin = (long long)B * nw * nw;
  const int wph = nslots * A::BWD_WAVES, wih = slot * A::BWD_WAVES + w;      // waves of this head, index among them
  const long long win_lo = nwin * wih / wph, win_hi = nwin * (wih + 1) / wph;
  const int l31 = lane & 31, g = lane >> 5;
  const float sc = scale[h], sc2 = sc * LOG2E;       // sc2: logits in base-2 units (stage_bias); sc: the chain rule's factor
  float dbacc[2][2][16];                    // this head's d(bias) in accumulator layout: [query tile][key tile][r]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dbacc[i][t][r] = 0.f;

  RawBwd<T> raw;
  WinPos nxt;
  nxt.set(win_lo, nw);
  if (A::PREFETCH && win_lo < win_hi) fetch_bwd<T>(raw, qkv, out, dout, lse + (win_lo * heads + h) * WT, nxt, h, res, C, shift, lane);
#pragma unroll 1
  for (long long win = win_lo; win < win_hi; ++win) {
    const long long unit = win * heads + h;            // (window, head) index of lse / dscale_part (forward's order)
    const int wx = nxt.wx, wy = nxt.wy, b = nxt.b;
    const bool masked = shift && (wy == nw - 1 || wx == nw - 1);      // only the last row / column of windows mixes regions
    WPROF(0);
    if (!A::PREFETCH) fetch_bwd<T>(raw, qkv, out, dout, lse + unit * WT, nxt, h, res, C, shift, lane);
    nxt.next(nw);
    __builtin_amdgcn_wave_barrier();                   // the previous window's LDS reads are done (same wave, in order)
#pragma unroll
    for (int n = 0; n < A::NI; ++n) {
      const int tk = n * A::TPI + lane / A::LPT, ch = lane % A::LPT;
      float q[A::EP], k[A::EP], gg[A::EP], oo[A::EP];
      unpack16<T>(raw.q[n], q);
      unpack16<T>(raw.k[n], k);
      unpack16<T>(raw.g[n], gg);
      unpack16<T>(raw.o[n], oo);
      float nq = 0.f, nk = 0.f, Dq = 0.f;
#pragma unroll
      for (int e = 0; e < A::EP; ++e) {
        nq += q[e] * q[e];
        nk += k[e] * k[e];
        Dq += gg[e] * oo[e];                           // D_i = sum_j p_ij dP_ij = dO_i . O_i
      }
      const float iq = inv_norm(token_sum<T>(nq)), ik = inv_norm(token_sum<T>(nk));
      Dq = token_sum<T>(Dq);
#pragma unroll
      for (int e = 0; e < A::EP; ++e) {
        q[e] *= iq;
        k[e] *= ik;
      }
      *reinterpret_cast<u32x4*>(Qn + tk * RP + ch * A::EP) = pack16<T>(q);
      *reinterpret_cast<u32x4*>(Kn + tk * RP + ch * A::EP) = pack16<T>(k);
      *reinterpret_cast<u32x4*>(Vr + tk * RP + ch * A::EP) = raw.v[n];
      *reinterpret_cast<u32x4*>(Gr + tk * RP + ch * A::EP) = raw.g[n];
      if constexpr (!A::TRREAD) {
#pragma unroll
        for (int e = 0; e < A::EP; ++e) {
          Qnt[(ch * A::EP + e) * TP + tk] = from_f32<T>(q[e]);
          Knt[(ch * A::EP + e) * TP + tk] = from_f32<T>(k[e]);
          Gt[(ch * A::EP + e) * TP + tk] = from_f32<T>(gg[e]);
        }
      }
      if (ch == 0) {
        int mid;
        (void)win_token(tk, wy, wx, res, shift, mid);
        Ds[tk] = Dq;
        Rq[tk] = iq;                                   // 1 / |q|, 1 / |k| for the normalisation backward
        Rk[tk] = ik;
        Mid[tk] = mid;
      }
    }
    Ls[lane] = raw.lse * LOG2E;
    WPROF(1);
    if (A::PREFETCH && win + 1 < win_hi) fetch_bwd<T>(raw, qkv, out, dout, lse + (unit + heads) * WT, nxt, h, res, C, shift, lane);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WPROF(2);
    // ================= phase A: lane = query (two 32-query tiles) -> dq, d(bias), d(scale) =================
    f32x2 dsc2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = 32 * i + l31;
      int midq;
      const int tokq = win_token(q, wy, wx, res, shift, midq);
      Frag<T> qf[A::NCH], gf[A::NCH];
#pragma unroll
      for (int c = 0; c < A::NCH; ++c) {
        qf[c] = rowfrag<T>(Qn, q, c, g);
        gf[c] = rowfrag<T>(Gr, q, c, g);
      }
      const float lq = Ls[q], Dq = Ds[q];
      const float* brow = Bs + q * BP;
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 sa, da;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
        row_pair_mma<T, A::NCH>(sa, da, Kn, Vr, 32 * t + l31, g, qf, gf);          // rows = keys, cols = queries
        float dss[16];
        auto softmax_bwd = [&](auto MK) {          // element pairs: packed fp32 VALU (v_pk_fma / v_pk_mul / v_pk_add)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int k0 = 32 * t + 8 * q4 + 4 * g;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + k0);
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              const int r = 4 * q4 + 2 * hp;
              const f32x2 s2 = {sa[r], sa[r + 1]};
              f32x2 lg = s2 * sc2 + (f32x2){b4[2 * hp], b4[2 * hp + 1]};
              if constexpr (decltype(MK)::value) {
                if (Mid[k0 + 2 * hp] != midq) lg[0] += -100.f * LOG2E;
                if (Mid[k0 + 2 * hp + 1] != midq) lg[1] += -100.f * LOG2E;
              }
              const f32x2 x = lg - lq;
              const f32x2 p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
              const f32x2 ds = p * ((f32x2){da[r], da[r + 1]} - Dq);
              dbacc[i][t][r] += ds[0];
              dbacc[i][t][r + 1] += ds[1];
              dsc2 = ds * s2 + dsc2;
              const f32x2 o = ds * sc;
              dss[r] = o[0];
              dss[r + 1] = o[1];
            }
          }
        };
        if (masked) softmax_bwd(std::true_type{});
        else softmax_bwd(std::false_type{});
#pragma unroll
        for (int fi = 0; fi < A::FPT; ++fi) mma(dq, tok_frag<T>(trK, Knt, l31, t, fi, g), pfrag<T>(dss, fi));   // rows = d
      }
      // d(x/|x|) = (I - n n^T) dy / |x| ; dq[r] belongs to d = acc_row(r), the other 16 d live in lane ^ 32
      float qn[16];
      float proj = 0.f;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 n4 = load4<T>(Qn + q * RP + rq * 8 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qn[rq * 4 + e] = n4[e];
          proj += n4[e] * dq[rq * 4 + e];
        }
      }
      proj += __shfl_xor(proj, 32, 64);
      const float ir = Rq[q];
      // bf16: rows go to the staging tile and leave as whole 64-byte token slices (16 lines per store instruction instead of 64)
      T* drow = A::STAGE_DQ ? St + q * RP : dqkv + ((size_t)b * res * res + tokq) * 3 * C + h * HD;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (dq[rq * 4 + e] - qn[rq * 4 + e] * proj) * ir;
        store4<T>(drow + rq * 8 + g * 4, o);
      }
    }
    const float dsc = wave_sum(dsc2[0] + dsc2[1]);
    if (lane == 0) dscale_part[unit] = dsc;
    if constexpr (A::STAGE_DQ) {
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int n = 0; n < A::NI; ++n) {
        const int tk = n * A::TPI + lane / A::LPT, ch = lane % A::LPT;
        int mid;
        const int tok = win_token(tk, wy, wx, res, shift, mid);
        *reinterpret_cast<u32x4*>(dqkv + ((size_t)b * res * res + tok) * 3 * C + h * HD + ch * A::EP) =
            *reinterpret_cast<const u32x4*>(St + tk * RP + ch * A::EP);
      }
    }
    WPROF(3);
    // ================= phase B: lane = key (two 32-key tiles) -> dk, dv =================
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = 32 * j + l31;
      int midk;
      const int tokk = win_token(key, wy, wx, res, shift, midk);
      Frag<T> kf[A::NCH], vf[A::NCH];
#pragma unroll
      for (int c = 0; c < A::NCH; ++c) {
        kf[c] = rowfrag<T>(Kn, key, c, g);
        vf[c] = rowfrag<T>(Vr, key, c, g);
      }
      const float* bcol = Bs + key;                                    // bias[h][query][key]: a column of the LDS copy
      f32x16 dk, dv;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x16 sa, da;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
        row_pair_mma<T, A::NCH>(sa, da, Qn, Gr, 32 * i + l31, g, kf, vf);          // rows = queries, cols = keys
        float pp[16], dss[16];
        auto softmax_bwd = [&](auto MK) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int q0 = 32 * i + 8 * q4 + 4 * g;
            const f32x4 b4 = {bcol[q0 * BP], bcol[(q0 + 1) * BP], bcol[(q0 + 2) * BP], bcol[(q0 + 3) * BP]};
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + q0), d4 = *reinterpret_cast<const f32x4*>(Ds + q0);
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              const int r = 4 * q4 + 2 * hp;
              f32x2 lg = (f32x2){sa[r], sa[r + 1]} * sc2 + (f32x2){b4[2 * hp], b4[2 * hp + 1]};
              if constexpr (decltype(MK)::value) {
                if (Mid[q0 + 2 * hp] != midk) lg[0] += -100.f * LOG2E;
                if (Mid[q0 + 2 * hp + 1] != midk) lg[1] += -100.f * LOG2E;
              }
              const f32x2 x = lg - (f32x2){l4[2 * hp], l4[2 * hp + 1]};
              const f32x2 p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
              const f32x2 o = p * ((f32x2){da[r], da[r + 1]} - (f32x2){d4[2 * hp], d4[2 * hp + 1]}) * sc;
              pp[r] = p[0];
              pp[r + 1] = p[1];
              dss[r] = o[0];
              dss[r + 1] = o[1];
            }
          }
        };
        if (masked) softmax_bwd(std::true_type{});
        else softmax_bwd(std::false_type{});
#pragma unroll
        for (int fi = 0; fi < A::FPT; ++fi) {
          mma(dv, tok_frag<T>(trG, Gt, l31, i, fi, g), pfrag<T>(pp, fi));
          mma(dk, tok_frag<T>(trQ, Qnt, l31, i, fi, g), pfrag<T>(dss, fi));
        }
      }
      float kn[16];
      float proj = 0.f;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 n4 = load4<T>(Kn + key * RP + rq * 8 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kn[rq * 4 + e] = n4[e];
          proj += n4[e] * dk[rq * 4 + e];
        }
      }
      proj += __shfl_xor(proj, 32, 64);
      const float ir = Rk[key];
      // dk / dv rows replace the key tile's own Kn / V rows (kf, vf and kn are in registers; nobody else reads these rows in
      // this phase) and leave as whole token slices after both tiles
      (void)tokk;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 o, o2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (dk[rq * 4 + e] - kn[rq * 4 + e] * proj) * ir;
          o2[e] = dv[rq * 4 + e];
        }
        store4<T>(Kn + key * RP + rq * 8 + g * 4, o);
        store4<T>(Vr + key * RP + rq * 8 + g * 4, o2);
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < A::NI; ++n) {
      const int tk = n * A::TPI + lane / A::LPT, ch = lane % A::LPT;
      int mid;
      const int tok = win_token(tk, wy, wx, res, shift, mid);
      T* drow = dqkv + ((size_t)b * res * res + tok) * 3 * C + h * HD + ch * A::EP;
      *reinterpret_cast<u32x4*>(drow + C) = *reinterpret_cast<const u32x4*>(Kn + tk * RP + ch * A::EP);
      *reinterpret_cast<u32x4*>(drow + 2 * C) = *reinterpret_cast<const u32x4*>(Vr + tk * RP + ch * A::EP);
    }
    WPROF(4);
  }
  // d(bias)[h][query][key] of this wave's windows -> its own partial slice [wave of the head][head][64][64]; summed over waves
  // by the batched deterministic reduction (reduce.hip).  (Global atomics here cost 5 ms of a 20 ms step: every
  // window of the batch adds into the same 64 x 64 x heads addresses.)
  float* dp = dpart + (((size_t)wih * heads + h) * (size_t)(WT * WT));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[(32 * i + l31) * WT + 32 * t + acc_row(r, lane)] = dbacc[i][t][r];
}

template <typename T> int set_attrs() {
  static DevOnce done;
  if (!done.need()) return RGBNM_OK;
  if (hipFuncSetAttribute((const void*)win_attn_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, WA<T>::FWD_LDS) != hipSuccess ||
      hipFuncSetAttribute((const void*)win_attn_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, WA<T>::BWD_LDS) != hipSuccess)
    return RGBNM_ELAUNCH;
  done.done();
  return RGBNM_OK;
}

int num_cus() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  int c = cached[dev].load(std::memory_order_acquire);
  if (c > 0) return c;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
  cached[dev].store(c, std::memory_order_release);
  return c;
}
// workgroups per head: one resident round of the chip split evenly between the heads (a workgroup holds ONE head's bias);
// `per_cu` = workgroups of this kernel that fit a CU (LDS).  Never more waves than windows.
struct HeadGrid { int bph, gpx, grid, slots; };     // slots = workgroups per head that do work (= partial d(bias) slices / waves)
HeadGrid head_grid(long long nwin, int heads, int per_cu, int waves) {
  HeadGrid g;
  const int cus = num_cus();
  long long bph = (long long)cus * per_cu / heads;
  const long long cap = (nwin + waves - 1) / waves;
  const bool capped = bph > cap;
  if (capped) bph = cap;
  if (bph < 1) bph = 1;
  g.bph = (int)bph; g.gpx = 0; g.grid = g.bph * heads; g.slots = g.bph;
  // XCD-aware mapping (head_slot) when the slots of an XCD divide (nearly) evenly by the head count
  const int spx = cus / 8 * per_cu, gpx = spx / heads;
  if (!capped && cus % 8 == 0 && gpx >= 1 && rgbnm_get_option("win_xcd") && (spx - gpx * heads) * 100 <= 7 * spx) {
    g.gpx = gpx; g.grid = 8 * spx; g.slots = 8 * gpx;
  }
  return g;
}

template <typename T>
int launch_fwd(const void* qkv, const float* bias, const float* scale, void* out, float* lse, int B, int res, int C,
               int heads, int shift, hipStream_t st) {
  if (set_attrs<T>() != RGBNM_OK) return RGBNM_ELAUNCH;
  const long long nwin = (long long)B * (res / WS) * (res / WS);
  const HeadGrid g = head_grid(nwin, heads, 160 * 1024 / WA<T>::FWD_LDS, WA<T>::FWD_WAVES);
  hipLaunchKernelGGL(win_attn_fwd_kernel<T>, dim3(g.grid), dim3(64 * WA<T>::FWD_WAVES), WA<T>::FWD_LDS, st,
                     (const T*)qkv, bias, scale, (T*)out, lse, B, res, C, heads, shift, g.bph, g.gpx);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

template <typename T> HeadGrid bwd_grid(long long nwin, int heads) {
  return head_grid(nwin, heads, 160 * 1024 / WA<T>::BWD_LDS, WA<T>::BWD_WAVES);
}

template <typename T>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* bias, const float* scale, const float* lse,
               void* dqkv, float* dbias, float* dscale_part, float* dscale, float* dpart, int B, int res, int C, int heads, int shift,
               hipStream_t st) {
  if (set_attrs<T>() != RGBNM_OK) return RGBNM_ELAUNCH;
  const long long nwin = (long long)B * (res / WS) * (res / WS);
  const HeadGrid g = bwd_grid<T>(nwin, heads);
  hipLaunchKernelGGL(win_attn_bwd_kernel<T>, dim3(g.grid), dim3(64 * WA<T>::BWD_WAVES), WA<T>::BWD_LDS, st,
                     (const T*)qkv, (const T*)out, (const T*)dout, bias, scale, lse, (T*)dqkv, dpart, dscale_part, B, res, C,
                     heads, shift, g.bph, g.gpx);
  LAUNCH_CHECK();
  RgbnmReduceJob j;
  j.part = dpart; j.stride = (long long)heads * WT * WT; j.out = dbias; j.n = heads * WT * WT; j.S = g.slots * WA<T>::BWD_WAVES;
  j.cols = 1; j.perm_heads = 0; j.accumulate = 0; j.epw = 8;
  const int rc = rgbnm_reduce_submit(j, st);
  if (rc != RGBNM_OK || !dscale) return rc;
  // d(scale)[h] = sum over the (window, head) partials: another job of the same batched, fixed-order reduction
  j.part = dscale_part; j.stride = heads; j.out = dscale; j.n = heads; j.S = (int)((long long)B * (res / WS) * (res / WS));
  return rgbnm_reduce_submit(j, st);
}

}  // namespace

extern "C" {

int rgbnm_window_attention_fwd(int dtype, const void* qkv, const float* bias, const float* scale, void* out, float* lse,
                               int B, int res, int C, int heads, int shift, void* stream) {
  if (!qkv || !bias || !scale || !out || !lse || B <= 0 || res % WS || C != heads * HD || shift < 0 || shift >= WS)
    return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_BF16) return launch_fwd<bf16>(qkv, bias, scale, out, lse, B, res, C, heads, shift, st);
  if (dtype == DT_F32) return launch_fwd<float>(qkv, bias, scale, out, lse, B, res, C, heads, shift, st);
  return RGBNM_EINVAL;
}

size_t rgbnm_window_attention_bwd_workspace(int B, int res, int heads) {
  // one d(bias) slice per wave: the fp32 geometry (2 waves per workgroup, as many workgroups) bounds both dtypes
  const long long nwin = (long long)B * (res / WS) * (res / WS);
  // (head-major slots >= XCD-aware slots, so the head-major count bounds both mappings)
  const long long waves_bf = (long long)bwd_grid<bf16>(nwin, heads).bph * WA<bf16>::BWD_WAVES;
  const long long waves_f = (long long)bwd_grid<float>(nwin, heads).bph * WA<float>::BWD_WAVES;
  return (size_t)(waves_bf > waves_f ? waves_bf : waves_f) * heads * WT * WT * sizeof(float);
}

int rgbnm_window_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* bias,
                               float* dscale, const float* scale, const float* lse, void* dqkv, float* dbias,
                               float* dscale_part, int B, int res, int C, int heads, int shift, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!qkv || !out || !dout || !bias || !scale || !lse || !dqkv || !dbias || !dscale_part || !workspace ||
      B <= 0 || res % WS || C != heads * HD || shift < 0 || shift >= WS)
    return RGBNM_EINVAL;
  if (workspace_bytes < rgbnm_window_attention_bwd_workspace(B, res, heads)) return RGBNM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* dpart = (float*)workspace;
  if (dtype == DT_BF16)
    return launch_bwd<bf16>(qkv, out, dout, bias, scale, lse, dqkv, dbias, dscale_part, dscale, dpart, B, res, C, heads, shift, st);
  if (dtype == DT_F32)
    return launch_bwd<float>(qkv, out, dout, bias, scale, lse, dqkv, dbias, dscale_part, dscale, dpart, B, res, C, heads, shift, st);
  return RGBNM_EINVAL;
}

#ifdef WIN_PROF
int rgbnm_debug_win_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_win_prof), sizeof(unsigned long long) * 1024 * 4 * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
