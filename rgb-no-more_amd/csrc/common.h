// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of rgb-no-more_amd.
// Written for MI355X only: no portability layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Timing experiments that produce WRONG numbers (kernels without their stores, their GELU, their LayerNorm ...: the sensitivity
// builds of DESIGN_NOTES.md 9) exist as -D switches in the kernel sources; a build that defines one must also say
// -DRGBNM_EXPERIMENTS, so that no release or test build can carry one by accident.
#if (defined(X_NOLN) || defined(X_NOGELU) || defined(X_NOEXP) || defined(X_NOSAVE) || defined(X_HALFW) || defined(X_NOATTN) || \
     defined(X_NOMLP) || defined(X_NOPIPE) || defined(KPX_NOGELU) || defined(KPX_NOC2) || defined(KPX_NOSTORE)) &&             \
    !defined(RGBNM_EXPERIMENTS)
#error "a wrong-numbers timing switch (X_NO* / X_HALFW / KPX_NO*) is defined without -DRGBNM_EXPERIMENTS"
#endif

#define RGBNM_OK 0
#define RGBNM_EINVAL (-1)
#define RGBNM_ELAUNCH (-2)
#define RGBNM_EWORKSPACE (-3)

enum { DT_F32 = 0, DT_BF16 = 1 };

// "hipFuncSetAttribute already done for this kernel on this device": per function and per device, lock-free.  Racing
// host threads may both set the (idempotent) attribute; nobody launches before it is set.  Keeps the launchers re-entrant.
#include <atomic>
struct DevOnce {
  std::atomic<unsigned long long> mask{0};
  static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
  bool need() const { return !((mask.load(std::memory_order_acquire) >> dev()) & 1ULL); }
  void done() { mask.fetch_or(1ULL << dev(), std::memory_order_release); }
};

#define LAUNCH_CHECK()                                  \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) return RGBNM_ELAUNCH;        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// MFMA operand fragments.  One fragment = 16 bytes per lane = EPL elements of T along the reduction
// axis.  Lane l supplies row/col (l & 31) and reduction slots [g*EPL, (g+1)*EPL) of a 32-byte chunk,
// g = l >> 5.  `mma` computes acc[i][n] += sum over both lane groups and all slots of A*B; because A and
// B fragments are always built with the same slot<->index assignment, any consistent assignment is
// valid (the dot product is permutation invariant) -- kernels exploit this (see attention.hip).
// C/D layout of every 32x32 MFMA on gfx950: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16> {
  bf16x8 v;
  static constexpr int EPL = 8;
};
template <> struct Frag<float> {
  f32x4 v;
  static constexpr int EPL = 4;
};

__device__ __forceinline__ void mma(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile held by this lane in register r
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <typename T> __device__ __forceinline__ Frag<T> load_frag(const T* p);  // 16-byte aligned
template <> __device__ __forceinline__ Frag<bf16> load_frag<bf16>(const bf16* p) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(p);
  return f;
}
template <> __device__ __forceinline__ Frag<float> load_frag<float>(const float* p) {
  Frag<float> f;
  f.v = *reinterpret_cast<const f32x4*>(p);
  return f;
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// 4 consecutive elements <-> 4 floats (8-byte / 16-byte accesses)
template <typename T> struct Vec4;
template <> struct Vec4<bf16> { typedef bf16x4 type; };
template <> struct Vec4<float> { typedef f32x4 type; };

template <typename T> __device__ __forceinline__ f32x4 load4(const T* p) {
  typename Vec4<T>::type v = *reinterpret_cast<const typename Vec4<T>::type*>(p);
  f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return o;
}
// Lane id recomputed on the spot (2 VALU ops) and opaque to the optimiser.  Per-lane address arithmetic written against
// it stays where it is used; written against a long-lived `lane` it is hoisted out of the main loop as an invariant and,
// in register-heavy kernels, SPILLED: every reload is a scratch round trip behind an s_waitcnt vmcnt(0), which also
// drains whatever prefetch / store traffic the wave had in flight (measured: the difference between 0 and 13 k cycles per
// (image, head) pair in the attention backward).
__device__ __forceinline__ int lane_id_here() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v) {
  typename Vec4<T>::type o;
  o[0] = (T)v[0]; o[1] = (T)v[1]; o[2] = (T)v[2]; o[3] = (T)v[3];
  *reinterpret_cast<typename Vec4<T>::type*>(p) = o;
}

// bf16-path GELU: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution) with the hardware
// exp/rcp; one exponential serves both the cdf and the pdf (exp(-(x/sqrt2)^2) = exp(-x^2/2)).
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf_x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float e = __expf(-ax * ax);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erfa = fmaf(-poly * t, e, 1.0f);        // erf(|x|/sqrt2)
  cdf = 0.5f * (1.0f + copysignf(erfa, x));
  pdf_x = 0.39894228040143267794f * e * x;            // x * phi(x)
}
// Two elements at a time so the FMA chain compiles to packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32): the GELU
// epilogue of the fc1 GEMM is VALU bound (measured: 3x its MFMA phase).  Same A&S 7.1.26 erfc as above, constants
// folded: z = |x| sqrt(log2 e / 2) so that exp(-x^2/2) = exp2(-z^2);  Q = erfc(|x|/sqrt2)/2.
//   gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x), Phi = x >= 0 ? 1 - Q : Q.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_pair_fast(f32x2 x, f32x2& g, f32x2& dg) {
  const float kz = 0.84932180028801904272f;                       // sqrt(log2(e) / 2)
  const float kp = 0.3275911f * 0.70710678118654752440f / kz;     // p |x|/sqrt2 expressed in z
  f32x2 z;
  z[0] = fabsf(x[0]) * kz;
  z[1] = fabsf(x[1]) * kz;
  const f32x2 d = z * kp + 1.0f;
  f32x2 t, e;
  t[0] = __builtin_amdgcn_rcpf(d[0]);
  t[1] = __builtin_amdgcn_rcpf(d[1]);
  const f32x2 w = z * z;
  e[0] = __builtin_amdgcn_exp2f(-w[0]);
  e[1] = __builtin_amdgcn_exp2f(-w[1]);
  f32x2 poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  poly = poly * t + (0.5f * 1.421413741f);
  poly = poly * t + (0.5f * -0.284496736f);
  poly = poly * t + (0.5f * 0.254829592f);
  const f32x2 q = (poly * t) * e;                                  // erfc(|x|/sqrt2) / 2
  f32x2 cdf;
  cdf[0] = x[0] >= 0.f ? 1.0f - q[0] : q[0];
  cdf[1] = x[1] >= 0.f ? 1.0f - q[1] : q[1];
  const f32x2 px = (e * 0.39894228040143267794f) * x;              // x phi(x)
  g = x * cdf;
  dg = cdf + px;
}
__device__ __forceinline__ float gelu_fast(float x) {
  float c, p;
  gelu_parts_fast(x, c, p);
  return x * c;
}
__device__ __forceinline__ float dgelu_fast(float x) {
  float c, p;
  gelu_parts_fast(x, c, p);
  return c + p;
}

// gelu'(u), the second output of the fc1 + GELU epilogues: nobody reads it before the backward pass -- a non-temporal store keeps
// it from pushing the rows the NEXT launch reads (gelu(u)) out of L2.  -DNT_C2=0: plain store (experiments).
#ifndef NT_C2
#define NT_C2 1
#endif
__device__ __forceinline__ void store_c2(bf16* ptr, const bf16x8& v) {
#if NT_C2
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(ptr));
#else
  *reinterpret_cast<bf16x8*>(ptr) = v;
#endif
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// The per-element steps of the LayerNorm backward with every FMA explicit and every product that must NOT be fused into a
// following add made opaque (an empty asm "modifies" it; `#pragma clang fp contract(off)` in an inlined helper did not stop
// -ffp-contract=fast from fusing across the inlining): the three kernels that run them (ln_bwd_kernel, gemm_nt_kpipe
// EPI_LNBWD, mlp_bwd_kernel) then round identically.  acc: xhat and g*dy of one element, row sums s1 = sum(g dy),
// s2 = sum(g dy xhat), column sums dgamma / dbeta.  dx = rs * (g dy - c1 - xhat * c2) + residual gradient (res = 0 if none).
__device__ __forceinline__ void ln_bwd_acc(float dv, float x, float mu, float rs, float gm, float& xh, float& gv, float& s1,
                                           float& s2, float& dg, float& db) {
  xh = (x - mu) * rs;
  gv = dv * gm;
  asm volatile("" : "+v"(xh), "+v"(gv));
  s1 += gv;
  s2 = __builtin_fmaf(gv, xh, s2);
  dg = __builtin_fmaf(dv, xh, dg);
  db += dv;
}
__device__ __forceinline__ float ln_bwd_dx(float rs, float gv, float c1, float xh, float c2, float res) {
  float t = gv - c1;
  asm volatile("" : "+v"(t));
  return __builtin_fmaf(rs, __builtin_fmaf(-xh, c2, t), res);
}

// ---- lane exchanges inside a row of 16 lanes as DPP operand modifiers.  __shfl_xor compiles to ds_bpermute_b32 (an address
// computation plus an LDS round trip of ~100 cycles); the row sums of the LayerNorm epilogues are chains of 3-4 of them per
// row and made up most of those loops' time (mlp_bwd_kernel: 64 dependent exchanges, 9.3 k of its 26 k epilogue cycles).
// A DPP exchange is one VALU instruction (usually folded into the add).  lane_xorN(v) returns v of lane (id ^ N).
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_take(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf,
                                                               BANK, false));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_take<0xB1, 0xf>(v, v); }      // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_take<0x4E, 0xf>(v, v); }      // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor8(float v) { return dpp_take<0x128, 0xf>(v, v); }     // row_ror:8
// lanes 0-3 and 8-11 of a row take lane + 4 (row_shl:4 on banks 0, 2), lanes 4-7 and 12-15 lane - 4 (row_shr:4 on banks 1, 3)
__device__ __forceinline__ float lane_xor4(float v) { return dpp_take<0x114, 0xa>(dpp_take<0x104, 0x5>(v, v), v); }
// v of lane (id + 4) mod 16 of the row: equals lane_xor4 when v repeats with period 8 inside the row (after a lane_xor8 step)
__device__ __forceinline__ float lane_ror4(float v) { return dpp_take<0x124, 0xf>(v, v); }

// butterfly sums in the order 32, 16, 8, 4, 2, 1 (the order -- and therefore the bits -- of the __shfl_xor loops they replace)
__device__ __forceinline__ float group16_sum(float v) {     // over the 16 lanes of a row
  v += lane_xor8(v);
  v += lane_ror4(v);
  v += lane_xor2(v);
  v += lane_xor1(v);
  return v;
}
// sum over the 16 virtual lanes of a LayerNorm row held by 8 lanes: s0 / s1 = partial sums of virtual lanes 2 l8 / 2 l8 + 1
__device__ __forceinline__ float group8_pair_sum(float s0, float s1) {
  s0 += lane_xor4(s0);
  s1 += lane_xor4(s1);
  s0 += lane_xor2(s0);
  s1 += lane_xor2(s1);
  s0 += lane_xor1(s0);
  s1 += lane_xor1(s1);
  return s0 + s1;
}
__device__ __forceinline__ float group8_sum(float v) {      // over the 8 lanes of a half row
  v += lane_xor4(v);
  v += lane_xor2(v);
  v += lane_xor1(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32, 64);
  v += __shfl_xor(v, 16, 64);
  return group16_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, lane_xor8(v));
  v = fmaxf(v, lane_ror4(v));
  v = fmaxf(v, lane_xor2(v));
  return fmaxf(v, lane_xor1(v));
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
