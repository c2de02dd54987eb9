// Batched deterministic reduction of split partials (weight / bias gradients of the TN GEMMs, LayerNorm gamma / beta
// gradients).  Every producer leaves partial slices part[s][n] in its own workspace region and SUBMITS a job; a
// composite (rgbnm_vit_block_bwd) brackets its producers with defer_begin / defer_flush so that the twelve small
// reductions of one transformer block run as ONE launch instead of six latency-bound kernels (measured: 6 launches,
// 41 us per block -> 1 launch).  Outside a bracket a submit launches immediately: same kernel, same summation order.
#include <cstddef>
#include <cstring>
#include "common.h"
#include "internal.h"

namespace {

constexpr int MAXJOBS = 16;
struct Jobs {
  RgbnmReduceJob j[MAXJOBS];
};

// the defer bracket opens and closes inside ONE C-ABI call on one host thread: per-thread queues make concurrent calls
// from other host threads / streams (a second model, an eval pass) independent
thread_local bool g_defer = false;
thread_local int g_njobs = 0;
thread_local Jobs g_jobs;

__device__ __forceinline__ int qkv_row_r(int n, int heads) {     // same map as gemm.hip's qkv_row
  const int inner = heads * 64;
  const int s3 = n / inner, rem = n % inner;
  return (rem / 64) * 192 + (rem % 64) * 3 + s3;
}

// 256 threads = epw elements x (256 / epw) partial-groups; the group sums are combined in a fixed order.
__global__ __launch_bounds__(256) void reduce_multi_kernel(Jobs J) {
  const RgbnmReduceJob jb = J.j[blockIdx.y];
  __shared__ float red[256];
  const int epw = jb.epw, nsg = 256 / epw;
  const int el = threadIdx.x % epw, sg = threadIdx.x / epw;
  for (int base = blockIdx.x * epw; base < jb.n; base += gridDim.x * epw) {
    const int i = base + el;
    float a = 0.f;
    if (i < jb.n) {
      const float* src = jb.part + i;
#pragma unroll 8
      for (int s = sg; s < jb.S; s += nsg) a += src[(size_t)s * jb.stride];
    }
    red[sg * epw + el] = a;
    __syncthreads();
    if (sg == 0 && i < jb.n) {
      float t;
      if (nsg == 4) {
        t = (red[el] + red[epw + el]) + (red[2 * epw + el] + red[3 * epw + el]);
      } else {
        t = 0.f;
        for (int g = 0; g < nsg; ++g) t += red[g * epw + el];
      }
      int o = i;
      if (jb.perm_heads > 0) {
        const int r = i / jb.cols, c = i % jb.cols;
        o = qkv_row_r(r, jb.perm_heads) * jb.cols + c;
      }
      jb.out[o] = jb.accumulate ? (jb.out[o] + t) : t;
    }
    __syncthreads();
  }
}

// ---- held reductions: a caller that needs no gradient before the END of a backward pass (one GPU, or one all-reduce after
// the backward) brackets the pass with rgbnm_reduce_hold_begin / _end; every reduction submitted in between -- the twelve of
// each encoder block, the head's, the patch embedding's -- is collected and run as ONE launch at the end (12 + launches of
// ~8.5 us, each latency bound, become one bandwidth-bound pass over the partials).  The job table lives in a device buffer of
// the caller and is uploaded only when it changed (the same pointers come back every step).  Same workgroup-to-element
// scheme and summation order as reduce_multi_kernel: the same bits.
constexpr int MAXHELD = 384;
struct HeldTable {
  int njobs, total;
  int first[MAXHELD + 1];      // job k owns workgroups first[k] .. first[k + 1] - 1
  unsigned char vec[MAXHELD];  // 1: element count, slice stride and base pointer of job k allow 16-byte moves
  RgbnmReduceJob j[MAXHELD];
};
thread_local bool g_hold = false;
thread_local HeldTable g_held;                 // being collected

// VEC = 4: an element slot is four consecutive elements moved as one 16-byte load per partial slice (the held pass reads
// partials that have long left the caches: wide loads are what an HBM-bound pass wants); same partial -> group assignment and
// the same summation order per element as VEC = 1 and as reduce_multi_kernel: the same bits.
template <int VEC>
__device__ __forceinline__ void reduce_job(const RgbnmReduceJob& jb, int bx, int nbx, float* red) {
  const int epw = jb.epw, nsg = 256 / epw;
  const int el = threadIdx.x % epw, sg = threadIdx.x / epw;
  for (int base = bx * epw * VEC; base < jb.n; base += nbx * epw * VEC) {
    const int i = base + el * VEC;
    float a[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[v] = 0.f;
    if (i < jb.n) {
      const float* src = jb.part + i;
#pragma unroll 8
      for (int s = sg; s < jb.S; s += nsg) {
        if (VEC == 4) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(src + (size_t)s * jb.stride);
#pragma unroll
          for (int v = 0; v < VEC; ++v) a[v] += q[v];
        } else {
          a[0] += src[(size_t)s * jb.stride];
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[(sg * epw + el) * VEC + v] = a[v];
    __syncthreads();
    if (sg == 0 && i < jb.n) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float t;
        if (nsg == 4) {
          t = (red[el * VEC + v] + red[(epw + el) * VEC + v]) + (red[(2 * epw + el) * VEC + v] + red[(3 * epw + el) * VEC + v]);
        } else {
          t = 0.f;
          for (int g = 0; g < nsg; ++g) t += red[(g * epw + el) * VEC + v];
        }
        int o = i + v;
        if (jb.perm_heads > 0) {
          const int r = o / jb.cols, c = o % jb.cols;
          o = qkv_row_r(r, jb.perm_heads) * jb.cols + c;
        }
        jb.out[o] = jb.accumulate ? (jb.out[o] + t) : t;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void reduce_table_kernel(const HeldTable* __restrict__ T) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  int lo = 0, hi = T->njobs - 1;               // largest k with first[k] <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (T->first[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const RgbnmReduceJob jb = T->j[lo];
  const int bx = blockIdx.x - T->first[lo], nbx = T->first[lo + 1] - T->first[lo];
  if (T->vec[lo]) reduce_job<4>(jb, bx, nbx, red);
  else reduce_job<1>(jb, bx, nbx, red);
}

int launch(const Jobs& J, int njobs, hipStream_t st) {
  int gx = 1;
  for (int k = 0; k < njobs; ++k) {
    const int need = cdiv(J.j[k].n, J.j[k].epw);
    gx = need > gx ? need : gx;
  }
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(reduce_multi_kernel, dim3(gx, njobs), dim3(256), 0, st, J);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

}  // namespace

bool rgbnm_reduce_defer_active() { return g_defer; }
void rgbnm_reduce_defer_begin() {
  g_defer = true;
  g_njobs = 0;
}

int rgbnm_reduce_defer_flush(hipStream_t st) {
  g_defer = false;
  const int n = g_njobs;
  g_njobs = 0;
  return n > 0 ? launch(g_jobs, n, st) : RGBNM_OK;
}

int rgbnm_reduce_submit(const RgbnmReduceJob& job, hipStream_t st) {
  if (job.n <= 0 || job.S <= 0 || (job.epw != 64 && job.epw != 8)) return RGBNM_EINVAL;
  if (g_hold && g_held.njobs < MAXHELD) {                  // (a full table: the job runs the ordinary way, in order)
    g_held.j[g_held.njobs++] = job;
    return RGBNM_OK;
  }
  if (g_defer) {
    if (g_njobs == MAXJOBS) {          // queue full: run what is there, keep queueing
      const int rc = launch(g_jobs, g_njobs, st);
      if (rc != RGBNM_OK) return rc;
      g_njobs = 0;
    }
    g_jobs.j[g_njobs++] = job;
    return RGBNM_OK;
  }
  Jobs one;
  one.j[0] = job;
  return launch(one, 1, st);
}

extern "C" {

int rgbnm_reduce_hold_begin(void) {
  if (g_hold) return RGBNM_EINVAL;              // one bracket per host thread: a second begin would drop the jobs collected so far
  g_hold = true;
  g_held.njobs = 0;
  return RGBNM_OK;
}

void rgbnm_reduce_hold_cancel(void) {
  g_hold = false;
  g_held.njobs = 0;
}

int rgbnm_reduce_hold_end(void* table_dev, void* table_host, size_t table_bytes, void* stream) {
  if (!g_hold) return RGBNM_OK;
  g_hold = false;
  const int n = g_held.njobs;
  if (n == 0) return RGBNM_OK;
  if (!table_dev || !table_host || table_bytes < sizeof(HeldTable)) return RGBNM_EWORKSPACE;
  int total = 0;
  for (int k = 0; k < n; ++k) {
    const RgbnmReduceJob& j = g_held.j[k];
    const bool vec = j.n % 4 == 0 && j.stride % 4 == 0 && ((size_t)j.part & 15) == 0 && (j.perm_heads <= 0 || j.cols % 4 == 0);
    g_held.vec[k] = vec ? 1 : 0;
    g_held.first[k] = total;
    const int need = cdiv(j.n, j.epw * (vec ? 4 : 1));
    total += need > 1024 ? 1024 : need;
  }
  for (int k = n; k <= MAXHELD; ++k) g_held.first[k] = total;
  g_held.total = total;
  hipStream_t st = (hipStream_t)stream;
  const size_t used = offsetof(HeldTable, j) + sizeof(RgbnmReduceJob) * (size_t)n;
  // What the device table holds is recorded in `table_host`, a host buffer the caller allocates (zeroed) TOGETHER with the
  // device table and frees with it: a device address handed out again by the allocator after its owner died comes with a fresh,
  // zeroed record and is uploaded again, whichever host thread gets here.
  if (memcmp(table_host, &g_held, used) != 0) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { g_held.njobs = 0; return RGBNM_ELAUNCH; }
    if (cap != hipStreamCaptureStatusNone) {
      // inside a graph capture the upload becomes a node of the graph whose SOURCE is table_host itself: that needs a page-locked
      // table_host that stays as it is while the graph lives (rgbnm.h) -- callers whose buffers differ between the eager passes
      // and the capture (swinv2.py: fresh tensors from the graph's pool).  A pageable table_host cannot be captured: EINVAL.
      memcpy(table_host, &g_held, used);
      if (hipMemcpyAsync(table_dev, table_host, used, hipMemcpyHostToDevice, st) != hipSuccess) {
        (void)hipGetLastError();
        memset(table_host, 0, used);
        g_held.njobs = 0;
        return RGBNM_EINVAL;                     // run one eager pass with the same buffers first, or pin table_host
      }
    } else {
      // pageable source: the runtime stages it before returning, so g_held may be reused at once; stream-ordered on st
      if (hipMemcpyAsync(table_dev, &g_held, used, hipMemcpyHostToDevice, st) != hipSuccess) {
        g_held.njobs = 0;
        return RGBNM_ELAUNCH;
      }
      memcpy(table_host, &g_held, used);
    }
  }
  g_held.njobs = 0;
  hipLaunchKernelGGL(reduce_table_kernel, dim3(total), dim3(256), 0, st, (const HeldTable*)table_dev);
  LAUNCH_CHECK();
  return RGBNM_OK;
}

size_t rgbnm_reduce_hold_table_bytes(void) { return sizeof(HeldTable); }

}  // extern "C"
