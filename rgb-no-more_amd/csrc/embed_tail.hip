// Entry and exit of the train step around the encoder (HBM / launch bound, no MFMA):
//   subblock_embed : 2x2 8x8 luma DCT blocks -> one 16x16 DCT block (A.X.A^T) + Cb|Cr concat, the
//                    "sub-block reshuffle" (models/plainvit.py:71-88,50-69,192,200-216)
//   softxent       : soft- or hard-label cross entropy, loss + dlogits (pipeline_utils.py:535)
//   mixup          : roll-by-one batch mixup of (Y, CbCr) and labels (utils/cls_transforms.py:163-176)
//   clip_adamw_wd  : global-norm clip + AdamW(wd=0) + schedule-relative WeightDecay in one pass over flat
//                    fp32 buffers (train.py:163-165, utils/custom_optims.py:37-42)
#include "common.h"
#include "internal.h"
#include "../../include/rgbnm.h"

namespace {

// ------------------------------------------------------------------------------------------------
// One wave per 16x16 patch.  X is gathered block-major ('(pdh p1) (pdw p2)', plainvit.py:83):
// X[8*pdh + p1][8*pdw + p2] = y[b,0,2*ph+pdh,2*pw+pdw,p1,p2].  Output row layout 'b h w (c i j)':
// [0,256) = Z row-major, [256,320) = Cb, [320,384) = Cr.
// ------------------------------------------------------------------------------------------------
// The two 16 x 16 x 16 products run on the matrix pipe in exact fp32 (v_mfma_f32_16x16x4_f32: 8 instructions per patch instead of
// 160 scalar LDS reads + 128 FMAs per lane -- the scalar version was instruction-bound at 2 TB/s).  With g = lane / 16, c = lane % 16:
//   T^T = X^T . A^T :  A-operand X[4 kk + g][c] (from the wave's LDS copy of the patch), B-operand A[c][4 kk + g]
//                      -> the lane holds T[c][4 g + r], r = 0..3
//   Z^T = A . T^T   :  reduction index taken in the order o = 4 g + kk, so that the B-operand of step kk IS register r = kk;
//                      A-operand A[c][4 g + kk]  -> the lane holds Z[c][4 g + r]: four consecutive outputs of row c
// lam != null (round 6): the batch is mixed on the way in -- RandomMixup_DCT's roll-by-one (cls_transforms.py:163-176) applied to
// the values as they are loaded: x[b] <- round_TI(lam[0] x[b] + lam[1] x[b - 1]), exactly what rgbnm_mixup writes (mix2 below is
// its arithmetic, the rounding its output type), so that the mixed batch never exists in memory.
__device__ __forceinline__ float mix2(float a, float c, float l0, float l1) { return fmaf(c, l1, a * l0); }

template <typename TI, typename T>
__global__ __launch_bounds__(256, 8) void subblock_embed_kernel(const TI* __restrict__ y, const TI* __restrict__ cbcr,
                                                             const float* __restrict__ A, T* __restrict__ feat,
                                                             int B, int Hb, int Wb, int transpose_a,
                                                             const float* __restrict__ lam) {
  __shared__ float Xs[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  float a1[4], a2[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    a1[kk] = transpose_a ? A[(4 * kk + g) * 16 + c] : A[c * 16 + 4 * kk + g];
    a2[kk] = transpose_a ? A[(4 * g + kk) * 16 + c] : A[c * 16 + 4 * g + kk];
  }
  const int ph_n = Hb / 2, pw_n = Wb / 2, npos = ph_n * pw_n;
  const long long npatch = (long long)B * npos;
  // A wave owns a CONTIGUOUS run of the patch list taken position-major, image-minor (q = pos * B + b): the eight A operands are
  // loaded once per wave, the inputs of the next THREE patches are in flight while one is converted (the kernel is bound by bytes in
  // flight: 768 B per patch and wave; one patch ahead ran at 2 TB/s), and with mixing on a patch's rolled partner (image b - 1, same
  // position) is the patch the wave converted just before -- every input is loaded once.
  const int pdh = lane >> 5, pdw = (lane >> 4) & 1, l16 = lane & 15;
  const int p1 = l16 >> 1, p2 = (l16 & 1) * 4;
  const int e0 = lane * 2, cc = e0 >> 6, k = e0 & 63;
  const bool mixed = lam != nullptr;
  const float l0 = mixed ? lam[0] : 1.f, l1 = mixed ? lam[1] : 0.f;
  const size_t ypi = (size_t)Hb * Wb * 64, cpi = (size_t)2 * npos * 64;      // elements per image
  // luma: 4 values per lane; chroma: 128 values, 2 per lane (identity sub-block conversion for 8x8 chroma patches)
  // (inputs in flight are kept as loaded -- 3 registers per bf16 patch -- and widened where they are used)
  struct alignas(2 * sizeof(TI)) TI2 { TI a, b; };
  struct Raw { typename Vec4<TI>::type v; TI2 c; };
  auto fetch = [&](int pos, int b, Raw& r) {
    const int ph = pos / pw_n, pw = pos % pw_n;
    r.v = *reinterpret_cast<const typename Vec4<TI>::type*>(y + b * ypi + ((size_t)(2 * ph + pdh) * Wb + 2 * pw + pdw) * 64 + l16 * 4);
    r.c = *reinterpret_cast<const TI2*>(cbcr + b * cpi + (((size_t)cc * ph_n + ph) * pw_n + pw) * 64 + k);
  };
  // (32-bit index arithmetic throughout, the launcher checks the sizes: 64-bit divisions by run-time values were a third of the
  // kernel's instructions; the fetch cursor (fpos, fb) and the consume cursor (pos, b) are advanced, not re-derived)
  const int nwaves = gridDim.x * 4, per = ((int)npatch + nwaves - 1) / nwaves;
  const int q0 = (blockIdx.x * 4 + w) * per, q1 = min(q0 + per, (int)npatch);
  if (q0 >= q1) return;
  int pos = q0 / B, b = q0 - pos * B;
  int fpos = pos, fb = b, fq = q0;
  auto fetch_next = [&](Raw& r) {
    if (fq < q1) fetch(fpos, fb, r);
    ++fq;
    if (++fb == B) { fb = 0; ++fpos; }
  };
  Raw r0 = {}, r1 = {}, r2 = {}, prev = {};
  fetch_next(r0);
  fetch_next(r1);
  fetch_next(r2);
  for (int q = q0; q < q1; ++q) {
    const int patch = b * npos + pos;                           // row of feat: 'b h w'
    Raw cur = r0;
    r0 = r1;
    r1 = r2;
    fetch_next(r2);
    if (mixed && (q == q0 || b == 0)) fetch(pos, b == 0 ? B - 1 : b - 1, prev);   // no predecessor in this run: load the partner
    f32x4 v = {to_f32(cur.v[0]), to_f32(cur.v[1]), to_f32(cur.v[2]), to_f32(cur.v[3])};
    float c0 = to_f32(cur.c.a), c1 = to_f32(cur.c.b);
    if (mixed) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = to_f32(from_f32<TI>(mix2(v[e], to_f32(prev.v[e]), l0, l1)));
      c0 = to_f32(from_f32<TI>(mix2(c0, to_f32(prev.c.a), l0, l1)));
      c1 = to_f32(from_f32<TI>(mix2(c1, to_f32(prev.c.b), l0, l1)));
      prev = cur;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) Xs[w][8 * pdh + p1][8 * pdw + p2 + e] = v[e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the wave's own LDS rows (no other wave touches Xs[w])
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) t = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[w][4 * kk + g][c], a1[kk], t, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the next patch's rows overwrite Xs[w] only after these reads)
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) z = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kk], t[kk], z, 0, 0, 0);
    T* dst = feat + (size_t)patch * 384;
    store4<T>(dst + c * 16 + 4 * g, z);
    {                                                            // the lane's two chroma values as ONE store
      struct alignas(2 * sizeof(T)) Pair { T a, b; };
      Pair pr;
      pr.a = from_f32<T>(c0);
      pr.b = from_f32<T>(c1);
      *reinterpret_cast<Pair*>(dst + 256 + e0) = pr;
    }
    if (++b == B) { b = 0; ++pos; }
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softxent_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                                                       const long long* __restrict__ hard, float* __restrict__ loss_rows,
                                                       T* __restrict__ dlogits, int C, float gscale) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* z = logits + (size_t)b * C;
  const long long lab = hard ? hard[b] : -1;
  float m = -INFINITY;
  for (int c = tid; c < C; c += 256) m = fmaxf(m, z[c]);
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f, st = 0.f, stz = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float t = hard ? (c == lab ? 1.f : 0.f) : soft[(size_t)b * C + c];
    se += __expf(z[c] - m);
    st += t;
    stz += t * z[c];
  }
  float vals[3] = {se, st, stz};
  for (int k = 0; k < 3; ++k) {
    const float v = wave_sum(vals[k]);
    if (lane == 0) red[w] = v;
    __syncthreads();
    vals[k] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
  const float lse = m + __logf(vals[0]);
  if (tid == 0) loss_rows[b] = lse * vals[1] - vals[2];
  if (dlogits) {
    for (int c = tid; c < C; c += 256) {
      const float t = hard ? (c == lab ? 1.f : 0.f) : soft[(size_t)b * C + c];
      dlogits[(size_t)b * C + c] = from_f32<T>((__expf(z[c] - lse) * vals[1] - t) * gscale);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The loss as ONE launch forward and ONE backward (round 6; rgbnm_softxent above stays for callers that want dlogits at once):
// forward: per-row loss, log-sum-exp and target mass; the workgroup that finishes LAST (a ticket in caller-owned memory) sums the
// rows in the fixed order of mean_kernel -- same bits, no second launch.  backward: dlogits = (softmax * sum(t) - t) * gscale *
// gout[0] straight from the saved row statistics, gout read on the device (the autograd output gradient: no host sync, no
// element-wise launches to scale / convert it).
__global__ __launch_bounds__(256) void softxent_loss_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                                                            const long long* __restrict__ hard, float* __restrict__ rows,
                                                            float* __restrict__ stat, float* __restrict__ loss,
                                                            unsigned* __restrict__ ticket, int C,
                                                            const float* __restrict__ mixlam = nullptr) {
  __shared__ float red[4];
  __shared__ int last_s;
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* z = logits + (size_t)b * C;
  const long long lab = hard ? hard[b] : -1;
  // mixlam (with hard labels): the roll-by-one mixup target of cls_transforms.RandomMixup_DCT built on the fly -- the expression of
  // mixup_target_kernel, so the dense [B, C] target never exists (LazyTarget)
  const long long lab2 = (hard && mixlam) ? hard[(b + B - 1) % B] : -1;
  const float l0 = mixlam ? mixlam[0] : 1.f, l1 = mixlam ? mixlam[1] : 0.f;
  float m = -INFINITY;
  for (int c = tid; c < C; c += 256) m = fmaxf(m, z[c]);
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f, st = 0.f, stz = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float t = hard ? (mixlam ? (c == lab ? l0 : 0.f) + (c == lab2 ? l1 : 0.f) : (c == lab ? 1.f : 0.f)) : soft[(size_t)b * C + c];
    se += __expf(z[c] - m);
    st += t;
    stz += t * z[c];
  }
  float vals[3] = {se, st, stz};
  for (int k = 0; k < 3; ++k) {
    const float v = wave_sum(vals[k]);
    if (lane == 0) red[w] = v;
    __syncthreads();
    vals[k] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
  const float lse = m + __logf(vals[0]);
  if (tid == 0) {
    __hip_atomic_store(rows + b, lse * vals[1] - vals[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stat[2 * b] = lse;
    stat[2 * b + 1] = vals[1];
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (t == (unsigned)B - 1u);
  }
  __syncthreads();
  if (!last_s) return;
  float a = 0.f;                                   // mean_kernel's order: stride-256 partial sums, wave sums, four partials
  for (int i = tid; i < B; i += 256) a += __hip_atomic_load(rows + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a = wave_sum(a);
  if (lane == 0) red[w] = a;
  __syncthreads();
  if (tid == 0) {
    loss[0] = (red[0] + red[1] + red[2] + red[3]) / B;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softxent_grad_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                                                            const long long* __restrict__ hard, const float* __restrict__ stat,
                                                            const float* __restrict__ gout, T* __restrict__ dlogits, int C,
                                                            float gscale, const float* __restrict__ mixlam = nullptr) {
  const int b = blockIdx.x, tid = threadIdx.x, B = gridDim.x;
  const float* z = logits + (size_t)b * C;
  const long long lab = hard ? hard[b] : -1;
  const long long lab2 = (hard && mixlam) ? hard[(b + B - 1) % B] : -1;
  const float l0 = mixlam ? mixlam[0] : 1.f, l1 = mixlam ? mixlam[1] : 0.f;
  const float lse = stat[2 * b], sumt = stat[2 * b + 1];
  const float g = gout ? gscale * gout[0] : gscale;
  for (int c = tid; c < C; c += 256) {
    const float t = hard ? (mixlam ? (c == lab ? l0 : 0.f) + (c == lab2 ? l1 : 0.f) : (c == lab ? 1.f : 0.f)) : soft[(size_t)b * C + c];
    dlogits[(size_t)b * C + c] = from_f32<T>((__expf(z[c] - lse) * sumt - t) * g);
  }
}

__global__ void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / n;
}

// ------------------------------------------------------------------------------------------------
// out[b] = lam[0]*in[b] + lam[1]*in[(b-1) mod B]   (torch.roll(1, 0)); lam lives on the device (no sync)
template <typename TI, typename TO>
__global__ void mixup_kernel(const TI* __restrict__ in, TO* __restrict__ out, const float* __restrict__ lam, int B,
                             long long per) {
  const float l0 = lam[0], l1 = lam[1];
  const long long n = (long long)B * per;
  for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4; i < n;
       i += (long long)gridDim.x * blockDim.x * 4) {
    const long long b = i / per, r = i % per;
    const long long pb = (b + B - 1) % B;
    const f32x4 a = load4<TI>(in + i), c = load4<TI>(in + pb * per + r);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = mix2(a[e], c[e], l0, l1);
    store4<TO>(out + i, o);
  }
}

__global__ void mixup_target_kernel(const long long* __restrict__ lab, float* __restrict__ tgt,
                                    const float* __restrict__ lam, int B, int C) {
  const float l0 = lam[0], l1 = lam[1];
  const long long n = (long long)B * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / C), c = (int)(i % C);
    const int pb = (b + B - 1) % B;
    tgt[i] = (lab[b] == c ? l0 : 0.f) + (lab[pb] == c ? l1 : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  // four strides per turn with their loads issued together (one workgroup per CU: a load per turn was 22 exposed latencies);
  // the terms are still added in ascending order: the same bits
  constexpr long long STRIDE = (long long)NORM_BLOCKS * 256 * 4;
  for (long long i = (blockIdx.x * 256LL + threadIdx.x) * 4; i < n; i += 4 * STRIDE) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * STRIDE;
      v[u] = j < n ? *reinterpret_cast<const f32x4*>(g + j) : f32x4{0.f, 0.f, 0.f, 0.f};   // n is a multiple of 256 (padded segments)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * STRIDE < n) a += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v; const unsigned char* wd_flag; const float* part; float* norm_out;
  long long n;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, wd_factor, max_norm;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  __shared__ float red[4];
  __shared__ float coef_s;
  {
    float t = a.part[threadIdx.x];   // NORM_BLOCKS == blockDim.x
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float total = sqrtf(red[0] + red[1] + red[2] + red[3]);
      float c = a.max_norm > 0.f ? a.max_norm / (total + 1e-6f) : 1.f;
      coef_s = c < 1.f ? c : 1.f;
      if (blockIdx.x == 0 && a.norm_out) a.norm_out[0] = total;
    }
    __syncthreads();
  }
  const float coef = coef_s;
  const float step = a.lr / a.bc1;
  for (long long chunk = blockIdx.x; chunk * 256 < a.n; chunk += gridDim.x) {
    const long long i = chunk * 256 + threadIdx.x;
    const float g = a.g[i] * coef;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    float p = a.p[i];
    p -= step * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
    if (a.wd_flag[chunk]) p -= a.wd_factor * p;
    a.p[i] = p;
  }
}

}  // namespace

extern "C" {

int rgbnm_subblock_embed_mix(int in_dtype, int out_dtype, const void* y, const void* cbcr, const float* lam_dev, const float* conv16,
                             void* feat, int B, int Hb, int Wb, int transpose_a, void* stream) {
  if (!y || !cbcr || !conv16 || !feat || B <= 0 || Hb <= 0 || Wb <= 0 || (Hb & 1) || (Wb & 1)) return RGBNM_EINVAL;
  const long long npatch = (long long)B * (Hb / 2) * (Wb / 2);
  if (npatch * 384 >= (1LL << 31)) return RGBNM_EINVAL;                    // 32-bit index arithmetic in the kernel
  const dim3 grid((unsigned)min(cdivl(npatch, 4), 2048LL)), blk(256);      // 256 CUs x 8 workgroups: one round, waves loop
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes: the 28 x 28 + 2 x 14 x 14 blocks in, 196 x 384 features out, per image
  const int tslot = rgbnm_trace_begin(TR_EMBED, 4.0 * npatch * 16 * 16 * 16, (double)npatch * 384 * ((in_dtype == DT_F32 ? 4.0 : 2.0) + (out_dtype == DT_F32 ? 4.0 : 2.0)), st);
#define SB(TI, TO) hipLaunchKernelGGL((subblock_embed_kernel<TI, TO>), grid, blk, 0, st, (const TI*)y, (const TI*)cbcr, conv16, (TO*)feat, B, Hb, Wb, transpose_a, lam_dev)
  if (in_dtype == DT_F32 && out_dtype == DT_F32) SB(float, float);
  else if (in_dtype == DT_F32 && out_dtype == DT_BF16) SB(float, bf16);
  else if (in_dtype == DT_BF16 && out_dtype == DT_BF16) SB(bf16, bf16);
  else if (in_dtype == DT_BF16 && out_dtype == DT_F32) SB(bf16, float);
  else return RGBNM_EINVAL;
#undef SB
  rgbnm_trace_end(tslot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_subblock_embed(int in_dtype, int out_dtype, const void* y, const void* cbcr, const float* conv16,
                         void* feat, int B, int Hb, int Wb, int transpose_a, void* stream) {
  return rgbnm_subblock_embed_mix(in_dtype, out_dtype, y, cbcr, nullptr, conv16, feat, B, Hb, Wb, transpose_a, stream);
}

int rgbnm_softxent(int dl_dtype, const float* logits, const float* soft_target, const long long* hard_target,
                   float* loss_rows, float* loss, void* dlogits, int B, int C, float grad_scale, void* stream) {
  if (!logits || (!soft_target && !hard_target) || !loss_rows || !loss || B <= 0 || C <= 0) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dl_dtype == DT_BF16)
    hipLaunchKernelGGL((softxent_kernel<bf16>), dim3(B), dim3(256), 0, st, logits, soft_target, hard_target, loss_rows, (bf16*)dlogits, C, grad_scale);
  else if (dl_dtype == DT_F32)
    hipLaunchKernelGGL((softxent_kernel<float>), dim3(B), dim3(256), 0, st, logits, soft_target, hard_target, loss_rows, (float*)dlogits, C, grad_scale);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, loss_rows, loss, B);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_softxent_loss(const float* logits, const float* soft_target, const long long* hard_target, float* loss_rows,
                        float* row_stats, float* loss, unsigned* ticket, int B, int C, void* stream) {
  if (!logits || (!soft_target && !hard_target) || !loss_rows || !row_stats || !loss || !ticket || B <= 0 || C <= 0) return RGBNM_EINVAL;
  hipLaunchKernelGGL(softxent_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, soft_target, hard_target, loss_rows,
                     row_stats, loss, ticket, C);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_softxent_grad(int dl_dtype, const float* logits, const float* soft_target, const long long* hard_target,
                        const float* row_stats, const float* gout_dev, void* dlogits, int B, int C, float grad_scale, void* stream) {
  if (!logits || (!soft_target && !hard_target) || !row_stats || !dlogits || B <= 0 || C <= 0) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dl_dtype == DT_BF16)
    hipLaunchKernelGGL((softxent_grad_kernel<bf16>), dim3(B), dim3(256), 0, st, logits, soft_target, hard_target, row_stats, gout_dev, (bf16*)dlogits, C, grad_scale);
  else if (dl_dtype == DT_F32)
    hipLaunchKernelGGL((softxent_grad_kernel<float>), dim3(B), dim3(256), 0, st, logits, soft_target, hard_target, row_stats, gout_dev, (float*)dlogits, C, grad_scale);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_softxent_loss_mix(const float* logits, const long long* labels, const float* mix_lam, float* loss_rows, float* row_stats,
                            float* loss, unsigned* ticket, int B, int C, void* stream) {
  if (!logits || !labels || !mix_lam || !loss_rows || !row_stats || !loss || !ticket || B <= 0 || C <= 0) return RGBNM_EINVAL;
  hipLaunchKernelGGL(softxent_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, (const float*)nullptr, labels, loss_rows,
                     row_stats, loss, ticket, C, mix_lam);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_softxent_grad_mix(int dl_dtype, const float* logits, const long long* labels, const float* mix_lam, const float* row_stats,
                            const float* gout_dev, void* dlogits, int B, int C, float grad_scale, void* stream) {
  if (!logits || !labels || !mix_lam || !row_stats || !dlogits || B <= 0 || C <= 0) return RGBNM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dl_dtype == DT_BF16)
    hipLaunchKernelGGL((softxent_grad_kernel<bf16>), dim3(B), dim3(256), 0, st, logits, (const float*)nullptr, labels, row_stats, gout_dev, (bf16*)dlogits, C, grad_scale, mix_lam);
  else if (dl_dtype == DT_F32)
    hipLaunchKernelGGL((softxent_grad_kernel<float>), dim3(B), dim3(256), 0, st, logits, (const float*)nullptr, labels, row_stats, gout_dev, (float*)dlogits, C, grad_scale, mix_lam);
  else return RGBNM_EINVAL;
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_mixup(int in_dtype, int out_dtype, const void* in, void* out, const float* lam_dev, int B,
                long long per_sample, void* stream) {
  if (!in || !out || !lam_dev || B <= 0 || per_sample <= 0 || (per_sample & 3)) return RGBNM_EINVAL;
  const long long n = (long long)B * per_sample;
  const dim3 grid((unsigned)min(4096LL, cdivl(n, 1024))), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define MX(TI, TO) hipLaunchKernelGGL((mixup_kernel<TI, TO>), grid, blk, 0, st, (const TI*)in, (TO*)out, lam_dev, B, per_sample)
  if (in_dtype == DT_F32 && out_dtype == DT_F32) MX(float, float);
  else if (in_dtype == DT_F32 && out_dtype == DT_BF16) MX(float, bf16);
  else if (in_dtype == DT_BF16 && out_dtype == DT_BF16) MX(bf16, bf16);
  else return RGBNM_EINVAL;
#undef MX
  LAUNCH_CHECK();
  return RGBNM_OK;
}

int rgbnm_mixup_target(const long long* labels, float* target, const float* lam_dev, int B, int C, void* stream) {
  if (!labels || !target || !lam_dev || B <= 0 || C <= 0) return RGBNM_EINVAL;
  hipLaunchKernelGGL(mixup_target_kernel, dim3((unsigned)min(2048LL, cdivl((long long)B * C, 256))), dim3(256), 0,
                     (hipStream_t)stream, labels, target, lam_dev, B, C);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

size_t rgbnm_clip_adamw_wd_workspace(void) { return NORM_BLOCKS * sizeof(float); }

int rgbnm_clip_adamw_wd_step(float* p, const float* g, float* m, float* v, const unsigned char* wd_flag_per_256,
                             long long n, float lr, float beta1, float beta2, float eps, int step, float wd_factor,
                             float max_norm, float* norm_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!p || !g || !m || !v || !wd_flag_per_256 || !workspace || n <= 0 || (n & 255) || step < 1) return RGBNM_EINVAL;
  if (workspace_bytes < NORM_BLOCKS * sizeof(float)) return RGBNM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  // algorithmic bytes: 28 B per parameter (p, g, m, v read; p, m, v written) + the gradient once more for the norm
  const int tslot = rgbnm_trace_begin(TR_OPT, 0.0, 32.0 * (double)n, st);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, st, g, n, part);
  LAUNCH_CHECK();
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.wd_flag = wd_flag_per_256; a.part = part; a.norm_out = norm_out; a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.wd_factor = wd_factor; a.max_norm = max_norm;
  const int grid = (int)min(4096LL, n / 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, st, a);
  rgbnm_trace_end(tslot, st);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // extern "C"
