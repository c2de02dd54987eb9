// Fused multi-head attention for the JPEG-ViT (reference: MultiHeadAttention.forward,
// models/plainvit.py:445-464).  N = 196 tokens, head dim 64, softmax(Q K^T / sqrt(emb_size)) -- the scale
// is sqrt(EMB), not sqrt(head_dim) (plainvit.py:455-457).  q|k|v arrive as three contiguous column
// blocks [heads*64] of one [B*N, 3*heads*64] buffer (the qkv GEMM de-interleaves the reference's
// '(h d qkv)' feature order on the weight side, see gemm.hip prep_weights).
//
// One workgroup per (image, head); 7 waves, each owning a 32-row block of queries (fwd, bwd-dq) or keys
// (bwd-dkv).  Everything is computed in the "swapped" orientation S^T = K.Q^T so that a lane owns ONE
// query (column lane&31) and holds its scores for keys {(r&3)+8(r>>2)+4g} of each 32-key tile in
// registers: the softmax row reduction is register-local + one exchange with lane^32, and the
// probabilities feed the next MFMA as the B operand without leaving registers.  The A operand of that
// second MFMA (V^T, K^T, Q^T, dO^T) is read from an LDS image transposed at staging time, using the same
// key<->slot assignment (the MFMA reduction is permutation invariant).
// Row-major operands (Q, K, V, dO rows) are read straight from global/L2 as 16-byte fragments.
#include "common.h"
#include "../../include/rgbnm.h"
#include "internal.h"

namespace {

constexpr int HD = 64;        // head dim
// NT = number of 32-token tiles a workgroup covers: 7 (224 >= 196 tokens: embed_type 1 / 2) or 10 (320 >= 294 tokens:
// embed_type 3, PatchEmbedding_DCT_Concat, plainvit.py:353-410).  One wave per tile.
template <int NT> struct AN {
  static constexpr int NTILE = NT;
  static constexpr int NPAD = NT * 32;
  static constexpr int TP = NT * 32 + 36;   // LDS pitch (elements) of transposed images: 130 / 162 dwords (bf16), 260 / 356 (f32)
  static constexpr int NTHREADS = NT * 64;
};
constexpr int NT_MAX = 10;

template <typename T> struct AT {
  static constexpr int EPL = Frag<T>::EPL;
  static constexpr int NCH = HD * (int)sizeof(T) / 32;   // 32-byte chunks along d: 4 (bf16) / 8 (f32)
  static constexpr int CH = 32 / (int)sizeof(T);         // elements per chunk
  static constexpr int QPF = EPL / 4;                    // register quads (4 keys) per fragment: 2 / 1
  static constexpr int FPT = 16 / EPL;                   // fragments per 32-key tile per lane: 2 / 4
};

// stage src[tok][HD] (row stride ld) transposed into dst[d][TP]; tokens >= N are zero-filled.
template <typename T, int NT>
__device__ __forceinline__ void stage_transposed(const T* __restrict__ src, int ld, int N, T* dst) {
  constexpr int NPAD = AN<NT>::NPAD, TP = AN<NT>::TP, NTHREADS = AN<NT>::NTHREADS;
  constexpr int EPV = 16 / (int)sizeof(T);
  constexpr int VPR = HD / EPV;  // vectors per row
  for (int idx = threadIdx.x; idx < NPAD * VPR; idx += NTHREADS) {
    const int tok = idx % NPAD, v = idx / NPAD;   // consecutive lanes -> consecutive tokens (conflict-free writes)
    Frag<T> f;
    if (tok < N) f = load_frag<T>(src + (size_t)tok * ld + v * EPV);
    else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) f.v[e] = (T)0.f;
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) dst[(v * EPV + e) * TP + tok] = f.v[e];
  }
}

// A-operand fragment from a transposed LDS image: row d, keys of fragment `fi` of tile `t` for lane group g.
template <typename T, int NT>
__device__ __forceinline__ Frag<T> tfrag(const T* img, int d, int t, int fi, int g) {
  constexpr int TP = AN<NT>::TP;
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    // regs 8*fi .. 8*fi+7  <->  keys 32t + 16fi + 4g + {0..3}  and  + 8 + {0..3}
    const T* p = img + d * TP + 32 * t + 16 * fi + 4 * g;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 8);
    f.v = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  } else {
    // regs 4*fi .. 4*fi+3  <->  keys 32t + 8fi + 4g + {0..3}
    f.v = *reinterpret_cast<const f32x4*>(img + d * TP + 32 * t + 8 * fi + 4 * g);
  }
  return f;
}

// B-operand fragment from 16 accumulator-layout values of one tile.
template <typename T> __device__ __forceinline__ Frag<T> pfrag(const float (&p)[16], int fi) {
  Frag<T> f;
#pragma unroll
  for (int j = 0; j < Frag<T>::EPL; ++j) f.v[j] = from_f32<T>(p[fi * Frag<T>::EPL + j]);
  return f;
}

// ------------------------------------------------------------------------------------------- forward
template <typename T, int NT>
__global__ __launch_bounds__(AN<NT>::NTHREADS) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                            float* __restrict__ lse, int N, int heads, float scale) {
  using A = AT<T>;
  constexpr int NTILE = AN<NT>::NTILE, NPAD = AN<NT>::NPAD, TP = AN<NT>::TP, NTHREADS = AN<NT>::NTHREADS;
  (void)NPAD; (void)NTHREADS; (void)TP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Vt = reinterpret_cast<T*>(smem_raw);   // [HD][TP]
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int inner = heads * HD, ld = 3 * inner;
  const T* Q = qkv + (size_t)b * N * ld + h * HD;
  const T* K = Q + inner;
  const T* V = Q + 2 * inner;

  stage_transposed<T, NT>(V, ld, N, Vt);
  __syncthreads();

  const int lane = threadIdx.x & 63, qt = threadIdx.x >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int q = qt * 32 + l31;
  if (qt * 32 >= N) return;
  const int qc = q < N ? q : N - 1;

  Frag<T> qf[A::NCH];
#pragma unroll
  for (int c = 0; c < A::NCH; ++c) qf[c] = load_frag<T>(Q + (size_t)qc * ld + c * A::CH + g * A::EPL);

  // S^T tile t of this wave's 32 queries: rows = keys, cols = queries; keys >= N come back as -inf
  auto score_tile = [&](int t, float (&sc)[16]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int key = t * 32 + l31;
    key = key < N ? key : N - 1;
#pragma unroll
    for (int c = 0; c < A::NCH; ++c) {
      const Frag<T> kf = load_frag<T>(K + (size_t)key * ld + c * A::CH + g * A::EPL);
      mma(acc, kf, qf[c]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = t * 32 + acc_row(r, lane);
      sc[r] = kk < N ? acc[r] : -INFINITY;
    }
  };
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  if constexpr (NT <= 7) {
    // all scores of the query stay in registers (7 x 16): one pass
    float s[NTILE][16];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      score_tile(t, s[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, s[t][r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[t][r] = __expf((s[t][r] - m) * scale);
        sum += s[t][r];
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (g == 0 && q < N) lse[(size_t)bh * N + q] = m * scale + __logf(sum);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] *= inv;
#pragma unroll
      for (int fi = 0; fi < A::FPT; ++fi) {
        const Frag<T> pf = pfrag<T>(s[t], fi);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma(o[dt], tfrag<T, NT>(Vt, dt * 32 + l31, t, fi, g), pf);
      }
    }
  } else {
    // 10 tiles do not fit the register file next to the accumulators: the row maximum first, then the scores are
    // recomputed tile by tile; P is normalised BEFORE it becomes an MFMA operand, exactly as in the one-pass form
    float m = -INFINITY;
#pragma unroll 1
    for (int t = 0; t < NTILE; ++t) {
      if (t * 32 >= N) break;
      float sc[16];
      score_tile(t, sc);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll 1
    for (int t = 0; t < NTILE; ++t) {
      if (t * 32 >= N) break;
      float sc[16];
      score_tile(t, sc);
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += __expf((sc[r] - m) * scale);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (g == 0 && q < N) lse[(size_t)bh * N + q] = m * scale + __logf(sum);
#pragma unroll 1
    for (int t = 0; t < NTILE; ++t) {
      if (t * 32 >= N) break;
      float sc[16];
      score_tile(t, sc);
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = __expf((sc[r] - m) * scale) * inv;
#pragma unroll
      for (int fi = 0; fi < A::FPT; ++fi) {
        const Frag<T> pf = pfrag<T>(sc, fi);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma(o[dt], tfrag<T, NT>(Vt, dt * 32 + l31, t, fi, g), pf);
      }
    }
  }
  // o[dt][r] = O[q][dt*32 + acc_row(r)] : 4 consecutive d per register quad
  if (q < N) {
    T* orow = out + ((size_t)b * N + q) * inner + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {o[dt][rq * 4 + 0], o[dt][rq * 4 + 1], o[dt][rq * 4 + 2], o[dt][rq * 4 + 3]};
        store4<T>(orow + dt * 32 + rq * 8 + g * 4, v);
      }
  }
}

// ------------------------------------------------------------------------------------- backward: dQ
// wave = 32 queries.  P recomputed from lse; dP^T = V.dO^T; dS = P*(dP - D)*scale; dQ^T = K^T.dS.
template <typename T, int NT>
__global__ __launch_bounds__(AN<NT>::NTHREADS) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                               const T* __restrict__ dout,
                                                               const float* __restrict__ lse, T* __restrict__ dqkv,
                                                               int N, int heads, float scale) {
  using A = AT<T>;
  constexpr int NTILE = AN<NT>::NTILE, NPAD = AN<NT>::NPAD, TP = AN<NT>::TP, NTHREADS = AN<NT>::NTHREADS;
  (void)NPAD; (void)NTHREADS; (void)TP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Kt = reinterpret_cast<T*>(smem_raw);   // [HD][TP]
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int inner = heads * HD, ld = 3 * inner;
  const T* Q = qkv + (size_t)b * N * ld + h * HD;
  const T* K = Q + inner;
  const T* V = Q + 2 * inner;
  const T* O = out + (size_t)b * N * inner + h * HD;
  const T* dO = dout + (size_t)b * N * inner + h * HD;

  stage_transposed<T, NT>(K, ld, N, Kt);
  __syncthreads();

  const int lane = threadIdx.x & 63, qt = threadIdx.x >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int q = qt * 32 + l31;
  if (qt * 32 >= N) return;
  const int qc = q < N ? q : N - 1;

  Frag<T> qf[A::NCH], dof[A::NCH];
  float Dq = 0.f;
#pragma unroll
  for (int c = 0; c < A::NCH; ++c) {
    qf[c] = load_frag<T>(Q + (size_t)qc * ld + c * A::CH + g * A::EPL);
    dof[c] = load_frag<T>(dO + (size_t)qc * inner + c * A::CH + g * A::EPL);
    const Frag<T> of = load_frag<T>(O + (size_t)qc * inner + c * A::CH + g * A::EPL);
#pragma unroll
    for (int e = 0; e < A::EPL; ++e) Dq += to_f32(dof[c].v[e]) * to_f32(of.v[e]);
  }
  Dq += __shfl_xor(Dq, 32, 64);
  const float lq = lse[(size_t)bh * N + qc];

  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

#pragma unroll 1
  for (int t = 0; t < NTILE; ++t) {
    if (t * 32 >= N) break;
    f32x16 sa, da;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
    int key = t * 32 + l31;
    key = key < N ? key : N - 1;
#pragma unroll
    for (int c = 0; c < A::NCH; ++c) {
      const Frag<T> kf = load_frag<T>(K + (size_t)key * ld + c * A::CH + g * A::EPL);
      const Frag<T> vf = load_frag<T>(V + (size_t)key * ld + c * A::CH + g * A::EPL);
      mma(sa, kf, qf[c]);
      mma(da, vf, dof[c]);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = t * 32 + acc_row(r, lane);
      const float p = kk < N ? __expf(sa[r] * scale - lq) : 0.f;
      ds[r] = p * (da[r] - Dq) * scale;
    }
#pragma unroll
    for (int fi = 0; fi < A::FPT; ++fi) {
      const Frag<T> sf = pfrag<T>(ds, fi);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) mma(dq[dt], tfrag<T, NT>(Kt, dt * 32 + l31, t, fi, g), sf);
    }
  }
  if (q < N) {
    T* drow = dqkv + ((size_t)b * N + q) * ld + h * HD;   // q block of dqkv
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v = {dq[dt][rq * 4 + 0], dq[dt][rq * 4 + 1], dq[dt][rq * 4 + 2], dq[dt][rq * 4 + 3]};
        store4<T>(drow + dt * 32 + rq * 8 + g * 4, v);
      }
  }
}

// ---------------------------------------------------------------------------------- backward: dK, dV
// wave = 32 keys (lane owns key lane&31), loop over query tiles; S[q][key] = Q.K^T (un-swapped), so the
// lane's registers run over queries.  dV^T = dO^T.P ; dK^T = Q^T.dS  with dO^T / Q^T from transposed LDS.
template <typename T, int NT, int WHICH>
__global__ __launch_bounds__(AN<NT>::NTHREADS) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                const T* __restrict__ dout,
                                                                const float* __restrict__ lse, T* __restrict__ dqkv,
                                                                int N, int heads, float scale) {
  using A = AT<T>;
  constexpr int NTILE = AN<NT>::NTILE, NPAD = AN<NT>::NPAD, TP = AN<NT>::TP, NTHREADS = AN<NT>::NTHREADS;
  (void)NPAD; (void)NTHREADS; (void)TP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // WHICH: 0 = dK and dV (two transposed images), 1 = dV only (dO^T), 2 = dK only (Q^T): fp32 with 10 tiles needs
  // 2 x 91 KB for both images, more than the CU has, so that shape runs the kernel twice
  constexpr bool DO_V = WHICH != 2, DO_K = WHICH != 1;
  T* Qt = reinterpret_cast<T*>(smem_raw);                        // [HD][TP] (DO_K)
  T* dOt = Qt + (DO_K ? HD * TP : 0);                            // [HD][TP] (DO_V)
  float* lse_s = reinterpret_cast<float*>(dOt + (DO_V ? HD * TP : 0));   // [NPAD]
  float* D_s = lse_s + NPAD;                                     // [NPAD]
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int inner = heads * HD, ld = 3 * inner;
  const T* Q = qkv + (size_t)b * N * ld + h * HD;
  const T* K = Q + inner;
  const T* V = Q + 2 * inner;
  const T* O = out + (size_t)b * N * inner + h * HD;
  const T* dO = dout + (size_t)b * N * inner + h * HD;

  if (DO_K) stage_transposed<T, NT>(Q, ld, N, Qt);
  if (DO_V) stage_transposed<T, NT>(dO, inner, N, dOt);
  for (int i = threadIdx.x; i < NPAD; i += NTHREADS) {
    float d = 0.f, l = 0.f;
    if (i < N) {
      l = lse[(size_t)bh * N + i];
      for (int c = 0; c < HD; c += 4) {
        const f32x4 a = load4<T>(dO + (size_t)i * inner + c), o4 = load4<T>(O + (size_t)i * inner + c);
        d += a[0] * o4[0] + a[1] * o4[1] + a[2] * o4[2] + a[3] * o4[3];
      }
    }
    lse_s[i] = l;
    D_s[i] = d;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, kt = threadIdx.x >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int key = kt * 32 + l31;
  if (kt * 32 >= N) return;
  const int kc = key < N ? key : N - 1;

  Frag<T> kf[A::NCH], vf[A::NCH];
#pragma unroll
  for (int c = 0; c < A::NCH; ++c) {
    kf[c] = load_frag<T>(K + (size_t)kc * ld + c * A::CH + g * A::EPL);
    vf[c] = load_frag<T>(V + (size_t)kc * ld + c * A::CH + g * A::EPL);
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

#pragma unroll 1
  for (int t = 0; t < NTILE; ++t) {
    if (t * 32 >= N) break;
    f32x16 sa, da;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
    int qi = t * 32 + l31;
    qi = qi < N ? qi : N - 1;
#pragma unroll
    for (int c = 0; c < A::NCH; ++c) {
      const Frag<T> qf = load_frag<T>(Q + (size_t)qi * ld + c * A::CH + g * A::EPL);
      const Frag<T> df = load_frag<T>(dO + (size_t)qi * inner + c * A::CH + g * A::EPL);
      mma(sa, qf, kf[c]);   // rows = queries, cols = keys
      mma(da, df, vf[c]);
    }
    float pp[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = t * 32 + acc_row(r, lane);
      const float p = qq < N ? __expf(sa[r] * scale - lse_s[qq]) : 0.f;
      pp[r] = p;
      ds[r] = p * (da[r] - D_s[qq]) * scale;
    }
#pragma unroll
    for (int fi = 0; fi < A::FPT; ++fi) {
      const Frag<T> pf = pfrag<T>(pp, fi), sf = pfrag<T>(ds, fi);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        if (DO_V) mma(dv[dt], tfrag<T, NT>(dOt, dt * 32 + l31, t, fi, g), pf);
        if (DO_K) mma(dk[dt], tfrag<T, NT>(Qt, dt * 32 + l31, t, fi, g), sf);
      }
    }
  }
  if (key < N) {
    T* krow = dqkv + ((size_t)b * N + key) * ld + inner + h * HD;
    T* vrow = krow + inner;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 a = {dk[dt][rq * 4 + 0], dk[dt][rq * 4 + 1], dk[dt][rq * 4 + 2], dk[dt][rq * 4 + 3]};
        f32x4 c = {dv[dt][rq * 4 + 0], dv[dt][rq * 4 + 1], dv[dt][rq * 4 + 2], dv[dt][rq * 4 + 3]};
        if (DO_K) store4<T>(krow + dt * 32 + rq * 8 + g * 4, a);
        if (DO_V) store4<T>(vrow + dt * 32 + rq * 8 + g * 4, c);
      }
  }
}

template <typename T, int NT>
int attn_fwd_t(const void* qkv, void* out, float* lse, int B, int N, int heads, float scale, hipStream_t st) {
  const size_t smem = (size_t)HD * AN<NT>::TP * sizeof(T);
  if (hipFuncSetAttribute((const void*)attn_fwd_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return RGBNM_ELAUNCH;
  hipLaunchKernelGGL((attn_fwd_kernel<T, NT>), dim3(B * heads), dim3(AN<NT>::NTHREADS), smem, st, (const T*)qkv, (T*)out,
                     lse, N, heads, scale);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

template <typename T, int NT, int WHICH>
int attn_dkv_launch(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N,
                    int heads, float scale, hipStream_t st) {
  const size_t smem = (size_t)(WHICH == 0 ? 2 : 1) * HD * AN<NT>::TP * sizeof(T) + 2 * AN<NT>::NPAD * sizeof(float);
  if (hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T, NT, WHICH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return RGBNM_ELAUNCH;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, NT, WHICH>), dim3(B * heads), dim3(AN<NT>::NTHREADS), smem, st,
                     (const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, N, heads, scale);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

template <typename T, int NT>
int attn_bwd_t(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N,
               int heads, float scale, hipStream_t st) {
  const size_t smem1 = (size_t)HD * AN<NT>::TP * sizeof(T);
  if (hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1) != hipSuccess) return RGBNM_ELAUNCH;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, NT>), dim3(B * heads), dim3(AN<NT>::NTHREADS), smem1, st, (const T*)qkv,
                     (const T*)out, (const T*)dout, lse, (T*)dqkv, N, heads, scale);
  LAUNCH_CHECK();
  if ((size_t)2 * HD * AN<NT>::TP * sizeof(T) + 2 * AN<NT>::NPAD * sizeof(float) <= 160 * 1024)
    return attn_dkv_launch<T, NT, 0>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
  const int rc = attn_dkv_launch<T, NT, 1>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
  if (rc != RGBNM_OK) return rc;
  return attn_dkv_launch<T, NT, 2>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
}

}  // namespace

extern "C" {

int rgbnm_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int heads, float scale,
                        void* stream) {
  if (!qkv || !out || !lse || B <= 0 || heads <= 0 || N <= 0 || N > AN<NT_MAX>::NPAD) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (N > AN<7>::NPAD) {       // 225 .. 320 tokens (embed_type 3: 294): generic kernels with 10 tiles
    if (dtype == DT_BF16) return attn_fwd_t<bf16, 10>(qkv, out, lse, B, N, heads, scale, st);
    if (dtype == DT_F32) return attn_fwd_t<float, 10>(qkv, out, lse, B, N, heads, scale, st);
    return RGBNM_EINVAL;
  }
  if (dtype == DT_BF16 && rgbnm_get_option("attn_v2")) return rgbnm_launch_attn2_fwd(qkv, out, lse, B, N, heads, scale, st);
  if (dtype == DT_BF16) return attn_fwd_t<bf16, 7>(qkv, out, lse, B, N, heads, scale, st);
  if (dtype == DT_F32) return attn_fwd_t<float, 7>(qkv, out, lse, B, N, heads, scale, st);
  return RGBNM_EINVAL;
}

int rgbnm_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                        int B, int N, int heads, float scale, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || B <= 0 || heads <= 0 || N <= 0 || N > AN<NT_MAX>::NPAD) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (N > AN<7>::NPAD) {
    if (dtype == DT_BF16) return attn_bwd_t<bf16, 10>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
    if (dtype == DT_F32) return attn_bwd_t<float, 10>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
    return RGBNM_EINVAL;
  }
  if (dtype == DT_BF16 && rgbnm_get_option("attn_v2"))
    return rgbnm_launch_attn2_bwd(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
  if (dtype == DT_BF16) return attn_bwd_t<bf16, 7>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
  if (dtype == DT_F32) return attn_bwd_t<float, 7>(qkv, out, dout, lse, dqkv, B, N, heads, scale, st);
  return RGBNM_EINVAL;
}

}  // extern "C"
