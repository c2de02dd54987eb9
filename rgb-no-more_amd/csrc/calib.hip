// Calibration micro-kernels: the attainable MFMA and HBM peaks on the box the bench runs on (SURVEY.md §8d asks for the
// measured figures next to the nominal ones).  They are not on the product path; tools/calib.py times them with HIP events.
#include "common.h"
#include "../../include/rgbnm.h"

namespace rgbnm {

// Every wave issues `iters` rounds of 4 independent 32x32x16 bf16 MFMAs (no memory traffic at all).
__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, float* sink) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(float)(threadIdx.x & 7);
        b[i] = (__bf16)(float)((threadIdx.x >> 3) & 7);
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[0] = s;          // keeps the chain live; never true
}

// mode 0: dst = src; mode 1: read only (sum into sink); mode 2: write only.  16 B per lane, grid-stride.
__global__ __launch_bounds__(256) void calib_stream_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16,
                                                           int mode, unsigned* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    if (mode == 0)
        for (; i < n16; i += step) dst[i] = src[i];
    else if (mode == 1) {
        for (; i < n16; i += step) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        if (acc == 0x9e3779b9u) sink[0] = acc;
    } else {
        uint4 v = {1u, 2u, 3u, 4u};
        for (; i < n16; i += step) dst[i] = v;
    }
}

}  // namespace rgbnm

extern "C" int rgbnm_calib_mfma_bf16(int workgroups, int iters, float* sink, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (workgroups <= 0 || iters <= 0 || !sink) return RGBNM_EINVAL;
    hipLaunchKernelGGL(rgbnm::calib_mfma_kernel, dim3(workgroups), dim3(256), 0, stream, iters, sink);
    return hipGetLastError() == hipSuccess ? RGBNM_OK : RGBNM_ELAUNCH;
}

extern "C" int rgbnm_calib_stream(const void* src, void* dst, size_t bytes, int mode, int workgroups, void* sink,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (bytes % 16 || workgroups <= 0 || mode < 0 || mode > 2 || !sink) return RGBNM_EINVAL;
    if ((mode != 2 && !src) || (mode != 1 && !dst)) return RGBNM_EINVAL;
    hipLaunchKernelGGL(rgbnm::calib_stream_kernel, dim3(workgroups), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst,
                       bytes / 16, mode, (unsigned*)sink);
    return hipGetLastError() == hipSuccess ? RGBNM_OK : RGBNM_ELAUNCH;
}

// ---- how fast can ONE wave issue back-to-back 1 KB vmem instructions?  (experiment behind the DMA-wave design of the
// attention backward).  mode 0: global_store_dwordx4, 8 rows x 128 B per instruction, row stride `ld` bytes;
// mode 1: global_load_dwordx4 of the same footprint.  out[wg*waves + w] = {cycles to issue 16, cycles until vmcnt(0)}.
namespace rgbnm {
__global__ __launch_bounds__(512) void calib_vmem_issue_kernel(int mode, unsigned char* buf, size_t wave_bytes, int ld,
                                                               unsigned long long* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned char* base = buf + ((size_t)blockIdx.x * nw + w) * wave_bytes + (size_t)(lane >> 3) * ld + (lane & 7) * 16;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 v[16];
    for (int i = 0; i < 16; ++i) v[i] = u4{(unsigned)lane, (unsigned)i, (unsigned)w, 7u};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<u4*>(base + (size_t)i * 8 * ld) = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(base + (size_t)i * 8 * ld));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    unsigned acc = 0;
    for (int i = 0; i < 16; ++i) acc += v[i][0];
    if (lane == 0) {
        out[((size_t)blockIdx.x * nw + w) * 2 + 0] = t1 - t0 + (acc == 0x7fffffffu);
        out[((size_t)blockIdx.x * nw + w) * 2 + 1] = t2 - t0;
    }
}
}  // namespace rgbnm

extern "C" int rgbnm_calib_vmem_issue(int mode, int workgroups, int waves, void* buf, size_t wave_bytes, int ld,
                                      unsigned long long* out, void* stream) {
    if (mode < 0 || mode > 1 || workgroups <= 0 || waves < 1 || waves > 8 || !buf || !out || ld < 128) return RGBNM_EINVAL;
    if (wave_bytes < (size_t)128 * ld) return RGBNM_EINVAL;
    hipLaunchKernelGGL(rgbnm::calib_vmem_issue_kernel, dim3(workgroups), dim3(64 * waves), 0, (hipStream_t)stream, mode,
                       (unsigned char*)buf, wave_bytes, ld, out);
    return hipGetLastError() == hipSuccess ? RGBNM_OK : RGBNM_ELAUNCH;
}

// ---- what can ONE CU pull through its vector-memory path when every line is an L2 hit?  Each workgroup re-reads its OWN
// slice (slice_bytes, bigger than the 32 KB L1, all slices of an XCD together smaller than its 4 MB L2) `iters` times.
// mode 0: global_load_dwordx4 into registers (8 in flight per lane); mode 1: LDS-DMA (global_load_lds_dwordx4, 8 pieces in
// flight per wave).  This prices the weight re-streaming of the row-panel kernels.
namespace rgbnm {
typedef __attribute__((address_space(3))) void* clds_ptr;
typedef const __attribute__((address_space(1))) void* cglb_ptr;
__global__ __launch_bounds__(1024) void calib_l2_kernel(const unsigned char* __restrict__ buf, size_t slice_bytes, int iters,
                                                        int mode, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned char* base = buf + (size_t)blockIdx.x * slice_bytes;
    const size_t per_wave = slice_bytes / nw;                 // multiple of 8 KB
    const unsigned char* wb = base + (size_t)w * per_wave;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (size_t off = 0; off < per_wave; off += 8192) {
            if (mode == 0) {
                u4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u4*>(wb + off + k * 1024 + lane * 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k][0] ^ v[k][3];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_global_load_lds((cglb_ptr)(wb + off + k * 1024 + lane * 16),
                                                     (clds_ptr)(smem + (size_t)w * 8192 + k * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    if (mode == 1) acc = *reinterpret_cast<unsigned*>(smem + (size_t)w * 8192 + lane * 4);
    if (acc == 0x9e3779b9u) sink[0] = acc;
}
}  // namespace rgbnm

extern "C" int rgbnm_calib_l2(const void* buf, size_t slice_bytes, int iters, int mode, int workgroups, int waves, void* sink,
                              void* stream) {
    if (!buf || !sink || waves < 1 || waves > 16 || workgroups <= 0 || iters <= 0 || mode < 0 || mode > 1) return RGBNM_EINVAL;
    if (slice_bytes % ((size_t)waves * 8192)) return RGBNM_EINVAL;
    const size_t smem = mode == 1 ? (size_t)waves * 8192 : 0;
    if (hipFuncSetAttribute((const void*)rgbnm::calib_l2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192) != hipSuccess)
        return RGBNM_ELAUNCH;
    hipLaunchKernelGGL(rgbnm::calib_l2_kernel, dim3(workgroups), dim3(64 * waves), smem, (hipStream_t)stream,
                       (const unsigned char*)buf, slice_bytes, iters, mode, (unsigned*)sink);
    return hipGetLastError() == hipSuccess ? RGBNM_OK : RGBNM_ELAUNCH;
}

// ---- pipe-overlap probe: what do two waves of one SIMD share?  Workgroup of `waves` waves (waves w and w + 4 sit on the same
// SIMD).  role[w] decides what wave w runs for `iters` rounds:  0 idle, 1 = 8 independent-accumulator 32x32x16 bf16 MFMAs per
// round, 2 = 64 dependent-free v_fma_f32 per round, 3 = 32 v_pk_fma_f32, 4 = 16 v_exp_f32 + 16 v_rcp_f32, 5 = 32
// v_cvt_pk_bf16_f32, 6 = one wave doing both 8 MFMAs and 64 v_fma_f32 per round in one instruction stream.
// out[(wg * waves + w)] = cycles the wave's loop took (s_memtime).
namespace rgbnm {
__global__ __launch_bounds__(512) void calib_pipes_kernel(const int* __restrict__ role, int iters, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int r = __builtin_amdgcn_readfirstlane(role[w]);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((lane + i) & 3); b[i] = (__bf16)(float)((lane >> 2) & 3); }
    f32x16 c[8];
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) c[k][i] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + 1e-3f * (float)(lane + i);
    const float m = 1.0000001f, d = 1e-7f;
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.25f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (r == 1 || r == 6) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
        }
        if (r == 2 || r == 6) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(d));
        }
        if (r == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f32x2 x = {v[i], v[i + 1]};
                    const f32x2 mm = {m, m}, dd = {d, d};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(mm), "v"(dd));
                    v[i] = x[0]; v[i + 1] = x[1];
                }
        }
        if (r == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])); }
        }
        if (r == 7) {                                   // the real GELU / GELU' pair code of the GEMM epilogues, 8 pairs
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 x = {v[i], v[i + 1]}, ge, dg;
                gelu_pair_fast(x, ge, dg);
                v[i] = ge[0] + dg[1];
                v[i + 1] = ge[1] + dg[0];
            }
        }
        if (r == 8) {                                   // 8 MFMAs, each fed by a ds_read_b128 of its A fragment
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(lds + ((k * 1024 + lane * 16 + it * 16) & 0x3fff));
                c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b, c[k], 0, 0, 0);
            }
        }
        if (r == 5) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) { unsigned pk; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(pk) : "v"(v[i])); v[i] = __builtin_bit_cast(float, pk); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) s += c[k][i];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[(size_t)blockIdx.x * nw + w] = t1 - t0;
}
}  // namespace rgbnm

extern "C" int rgbnm_calib_pipes(const int* role_dev, int waves, int iters, int workgroups, unsigned long long* out, float* sink,
                                 void* stream) {
    if (!role_dev || !out || !sink || waves < 1 || waves > 8 || iters <= 0 || workgroups <= 0) return RGBNM_EINVAL;
    hipLaunchKernelGGL(rgbnm::calib_pipes_kernel, dim3(workgroups), dim3(64 * waves), 0, (hipStream_t)stream, role_dev, iters, out, sink);
    return hipGetLastError() == hipSuccess ? RGBNM_OK : RGBNM_ELAUNCH;
}

// ---- the stand-in for an RCCL channel: `workgroups` workgroups of 256 threads that each hold `lds_bytes` of LDS (so that a
// one-workgroup-per-CU kernel cannot be placed next to them when lds_bytes is large, and can when it is small) and stay resident
// until `ticks` s_memtime ticks have passed or *stop becomes non-zero, re-reading an L2-sized slice of `buf` meanwhile (mode 1) or
// just sleeping (mode 0).  tools/cu_steal_probe.py times the chain kernels next to it; tests/test_chain_soak.py uses it as the timing
// perturber under which the one race of round 4 showed.
namespace rgbnm {
__global__ __launch_bounds__(256) void calib_occupy_kernel(const unsigned char* __restrict__ buf, size_t slice_bytes, long long ticks,
                                                           int mode, const volatile int* stop, unsigned* sink,
                                                           unsigned long long* log) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (log && threadIdx.x == 0) {      // residency record of this workgroup: start (100 MHz), [end], hardware id (XCC, SE, CU)
        log[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));       // HW_REG_XCC_ID, bits 0..3
        log[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    unsigned acc = 0;
    const unsigned char* base = buf + (size_t)blockIdx.x * slice_bytes;
    size_t off = (size_t)threadIdx.x * 16;
    if (threadIdx.x == 0) smem[0] = 1;                       // (the allocation is real even if nobody reads it)
    for (;;) {
        if (mode == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u4 v = *reinterpret_cast<const u4*>(base + off);
                acc += v[0] ^ v[3];
                off += 4096;
                if (off >= slice_bytes) off = (size_t)threadIdx.x * 16;
            }
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
        if ((long long)(__builtin_amdgcn_s_memtime() - t0) >= ticks) break;
        if (stop && *stop) break;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc + smem[0];
    if (log && threadIdx.x == 0) log[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace rgbnm

extern "C" int rgbnm_calib_occupy_log(const void* buf, size_t slice_bytes, int workgroups, int lds_bytes, long long ticks, int mode,
                                      const int* stop, void* sink, unsigned long long* log, void* stream);
extern "C" int rgbnm_calib_occupy(const void* buf, size_t slice_bytes, int workgroups, int lds_bytes, long long ticks, int mode,
                                  const int* stop, void* sink, void* stream) {
    return rgbnm_calib_occupy_log(buf, slice_bytes, workgroups, lds_bytes, ticks, mode, stop, sink, nullptr, stream);
}
// the same with a residency log: log[3 wg] = s_memrealtime (100 MHz) at the workgroup's start, [3 wg + 1] at its end,
// [3 wg + 2] = (XCC id << 32) | HW_ID -- so that a probe can tell WHERE and WHEN the stand-ins really were resident
extern "C" int rgbnm_calib_occupy_log(const void* buf, size_t slice_bytes, int workgroups, int lds_bytes, long long ticks, int mode,
                                      const int* stop, void* sink, unsigned long long* log, void* stream) {
    if (workgroups <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks <= 0 || mode < 0 || mode > 1 || !sink) return RGBNM_EINVAL;
    if (mode == 1 && (!buf || slice_bytes < 4096 || slice_bytes % 4096)) return RGBNM_EINVAL;
    if (hipFuncSetAttribute((const void*)rgbnm::calib_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return RGBNM_ELAUNCH;
    hipLaunchKernelGGL(rgbnm::calib_occupy_kernel, dim3(workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                       (const unsigned char*)buf, slice_bytes, ticks, mode, (const volatile int*)stop, (unsigned*)sink, log);
    return hipGetLastError() == hipSuccess ? RGBNM_OK : RGBNM_ELAUNCH;
}
