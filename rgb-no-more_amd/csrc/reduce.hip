// Batched deterministic reduction of split partials (weight / bias gradients of the TN GEMMs, LayerNorm gamma / beta
// gradients).  Every producer leaves partial slices part[s][n] in its own workspace region and SUBMITS a job; a
// composite (rgbnm_vit_block_bwd) brackets its producers with defer_begin / defer_flush so that the twelve small
// reductions of one transformer block run as ONE launch instead of six latency-bound kernels (measured: 6 launches,
// 41 us per block -> 1 launch).  Outside a bracket a submit launches immediately: same kernel, same summation order.
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
