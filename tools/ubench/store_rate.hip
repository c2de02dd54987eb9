// What does a global store cost the wave that issues it?  gfx950: W waves per CU (one workgroup per CU, 256 CUs) each issue a stream
// of 1 KB stores (global_store_dwordx4, 64 lanes x 16 B, consecutive 1 KB lines of a private region) straight from registers,
// `burst` stores back to back, then `gap` cycles of s_sleep (0: a continuous stream).  Printed: cycles per store instruction seen by
// the issuing wave (s_memtime around the whole stream, the final drain excluded and included), GB/s per CU and TB/s for the chip.
// Variants: plain / non-temporal stores.
// Build + run on the box:  hipcc --offload-arch=gfx950 -O2 store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(1024) void k_store(unsigned char* buf, size_t per_wave, int nstore, int burst, int gap, unsigned long long* out, int work = 0) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned char* base = buf + ((size_t)blockIdx.x * nw + w) * per_wave + lane * 16;
  u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  asm volatile("" : "+v"(v));
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fa = 1.0001f, fb = 0.5f;
  asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(fa), "+v"(fb));
  const size_t wrap = per_wave / 1024;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long in_burst = 0;
  int i = 0;
  while (i < nstore) {
    const unsigned long long b0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < burst && i < nstore; ++b, ++i) {
      u32x4* p = reinterpret_cast<u32x4*>(base + (size_t)(i % wrap) * 1024);
      if (NT) __builtin_nontemporal_store(v, p);
      else *p = v;
      for (int k = 0; k < work; ++k)                       // independent VALU work between two stores: does the store's cost hide under it?
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));
    }
    in_burst += __builtin_amdgcn_s_memtime() - b0;
    for (int g = 0; g < gap; g += 64) __builtin_amdgcn_s_sleep(1);      // s_sleep 1 = 64 cycles
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  const unsigned long long t1 = t0 + in_burst;
  if (f0 + f1 + f2 + f3 == 12345.f) out[0] = 1;
  if (lane == 0) {
    out[(blockIdx.x * 16 + w) * 2] = t1 - t0;
    out[(blockIdx.x * 16 + w) * 2 + 1] = t2 - t0;
  }
}

// store width: one wave per CU, a continuous stream of dword / dwordx2 / dwordx4 stores (256 / 512 / 1024 bytes per instruction)
template <int W>
__global__ __launch_bounds__(64) void k_width(unsigned char* buf, size_t per_wave, int nstore, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  unsigned char* base = buf + (size_t)blockIdx.x * 16 * per_wave + lane * 4 * W;
  u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  asm volatile("" : "+v"(v));
  const size_t wrap = per_wave / (256 * W);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < nstore; ++i) {
    unsigned char* p = base + (size_t)(i % wrap) * 256 * W;
    if (W == 1) *reinterpret_cast<unsigned*>(p) = v[0];
    else if (W == 2) { typedef unsigned int u32x2v __attribute__((ext_vector_type(2))); u32x2v t = {v[0], v[1]}; *reinterpret_cast<u32x2v*>(p) = t; }
    else *reinterpret_cast<u32x4*>(p) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[blockIdx.x * 32] = t2 - t0; out[blockIdx.x * 32 + 1] = t2 - t0; }
}

__global__ __launch_bounds__(1024) void k_work(int n, unsigned long long* out, int work) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fa = 1.0001f, fb = 0.5f;
  asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(fa), "+v"(fb));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < work; ++k)
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (f0 + f1 + f2 + f3 == 12345.f) out[0] = 1;
  if (lane == 0) { out[(blockIdx.x * 16 + w) * 2] = t2 - t0; out[(blockIdx.x * 16 + w) * 2 + 1] = t2 - t0; }
}

int main() {
  const int ncu = 256;
  const size_t per_wave = 4u << 20;                       // 4 MB per wave: 16 GB for 16 waves x 256 CUs would be too much -> cap below
  unsigned char* buf;
  unsigned long long* out;
  const size_t total = (size_t)ncu * 16 * per_wave;       // 16 GB
  if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, sizeof(unsigned long long) * ncu * 16 * 2);
  std::vector<unsigned long long> h(ncu * 16 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("waves/CU  burst  gap  nt | cycles/store issued (100 MHz ticks x f)  | kernel us | GB/s per CU | TB/s chip\n");
  for (int nt = 0; nt < 2; ++nt)
    for (int waves : {1, 2, 4, 8, 16})
      for (int cfg = 0; cfg < 3; ++cfg) {
        const int burst = cfg == 0 ? 1 << 30 : (cfg == 1 ? 4 : 24), gap = cfg == 0 ? 0 : (cfg == 1 ? 1024 : 8192);
        const int nstore = 2048;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (nt) hipLaunchKernelGGL(k_store<true>, dim3(ncu), dim3(64 * waves), 0, 0, buf, per_wave, nstore, burst, gap, out);
          else hipLaunchKernelGGL(k_store<false>, dim3(ncu), dim3(64 * waves), 0, 0, buf, per_wave, nstore, burst, gap, out);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double issue = 0, drain = 0;
        int n = 0;
        for (int b = 0; b < ncu; ++b)
          for (int w = 0; w < waves; ++w) { issue += (double)h[(b * 16 + w) * 2]; drain += (double)h[(b * 16 + w) * 2 + 1]; ++n; }
        issue /= n; drain /= n;
        const double bytes = (double)ncu * waves * nstore * 1024.0;
        // s_memtime ticks at 100 MHz on this part: report ticks per store and the wall-clock figures
        printf("%8d %6d %4d  %2d | in bursts %7.1f ticks/store  whole life %7.1f (%.0f ticks/us) | %8.1f | %7.1f | %6.2f\n", waves, burst > 1000 ? 0 : burst, gap, nt,
               issue / nstore, drain / nstore, drain / (ms * 1e3), ms * 1e3, bytes / (ms * 1e-3) / ncu / 1e9, bytes / (ms * 1e-3) / 1e12);
      }
  // ---- VALU work between the stores of a continuous stream (one and two waves per CU): work = groups of 4 independent v_fma
  printf("\nwaves/CU  valu/store | ticks per (store + work)   [the same work alone] | GB/s per CU\n");
  for (int waves : {1, 2})
    for (int work : {0, 8, 16, 32, 64, 128}) {
      double res[2] = {0, 0};
      float mss[2] = {0, 0};
      for (int mode = 0; mode < 2; ++mode) {            // mode 1: nstore = 0 stores ... measured as burst of work only: use gap-less stream with stores disabled via nstore trick
        const int nstore = 1024;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k_store<false>, dim3(ncu), dim3(64 * waves), 0, 0, buf, per_wave, nstore, 1 << 30, 0, out, work);
          else hipLaunchKernelGGL(k_work, dim3(ncu), dim3(64 * waves), 0, 0, nstore, out, work);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&mss[mode], e0, e1);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double t = 0; int n = 0;
        for (int b = 0; b < ncu; ++b) for (int w = 0; w < waves; ++w) { t += (double)h[(b * 16 + w) * 2 + 1]; ++n; }
        res[mode] = t / n / nstore;
      }
      printf("%8d %10d | %8.1f   [%8.1f] | %7.1f\n", waves, work * 4, res[0], res[1], (double)ncu * waves * 1024 * 1024.0 / (mss[0] * 1e-3) / ncu / 1e9);
    }
  printf("\nstore width, one wave per CU: bytes/instruction | ticks per store | GB/s per CU\n");
  for (int wd : {1, 2, 4}) {
    const int nstore = 2048;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (wd == 1) hipLaunchKernelGGL(k_width<1>, dim3(ncu), dim3(64), 0, 0, buf, per_wave, nstore, out);
      else if (wd == 2) hipLaunchKernelGGL(k_width<2>, dim3(ncu), dim3(64), 0, 0, buf, per_wave, nstore, out);
      else hipLaunchKernelGGL(k_width<4>, dim3(ncu), dim3(64), 0, 0, buf, per_wave, nstore, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double t = 0;
    for (int b = 0; b < ncu; ++b) t += (double)h[b * 32];
    printf("%6d | %8.1f | %7.1f\n", 256 * wd, t / ncu / nstore, (double)nstore * 256 * wd / (ms * 1e-3) / 1e9);
  }
  return 0;
}
