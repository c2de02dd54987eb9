// Issue cost of VALU opcodes on gfx950 (cycles per wave64 instruction and SIMD), one wave per SIMD and two: long streams of independent
// instructions (8 register chains), s_memtime around them.  Build + run on the box:  hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, body)                                                                                   \
  __global__ __launch_bounds__(512) void k_##name(unsigned long long* out, int iters) {                      \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    unsigned b = 0x3f803f80u, c = 0x00010001u;                                                               \
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b), "+v"(c)); \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < iters; ++i) {                                                                        \
      asm volatile(REP16(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }                                                                                                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                         \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345u) out[0] = 0;                                       \
  }

// one "body" = 8 independent instructions (one per chain)
#define B8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n\t"
#define AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define CVT(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %8\n\t"
#define PKMAXU(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n\t"
#define PKMADU(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n\t"
#define PKMINI(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n\t"
#define PKSUBU(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n\t"
#define PKLSHR(i) "v_pk_lshrrev_b16 %" #i ", 1, %" #i "\n\t"
#define ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n\t"
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n\t"
#define ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 7\n\t"
#define LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n\t"
#define MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define DOT2(i) "v_dot2_f32_bf16 %" #i ", %8, %9, %" #i "\n\t"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"

KERNEL(fma, B8(FMA))
KERNEL(addf, B8(ADDF))
KERNEL(mulf, B8(MULF))
KERNEL(and, B8(AND))
KERNEL(lshl, B8(LSHL))
KERNEL(perm, B8(PERM))
KERNEL(cvtpk, B8(CVT))
KERNEL(pkmaxu16, B8(PKMAXU))
KERNEL(pkmadu16, B8(PKMADU))
KERNEL(pkmini16, B8(PKMINI))
KERNEL(pksubu16, B8(PKSUBU))
KERNEL(pklshr16, B8(PKLSHR))
KERNEL(andor, B8(ANDOR))
KERNEL(max3, B8(MAX3))
KERNEL(exp, B8(EXP))
KERNEL(rcp, B8(RCP))
KERNEL(xad, B8(XAD))
KERNEL(add3, B8(ADD3))
KERNEL(mov, B8(MOV))
KERNEL(bfe, B8(BFE))
KERNEL(lshladd, B8(LSHLADD))
KERNEL(dot2bf16, B8(DOT2))
KERNEL(cndmask, B8(CNDMASK))

// packed fp32: 64-bit register pairs
#define KERNEL2(name, opstr)                                                                                 \
  __global__ __launch_bounds__(512) void k_##name(unsigned long long* out, int iters) {                      \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                    \
    f2 a0 = {1.f + threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f}; \
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b), "+v"(c));                             \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < iters; ++i) {                                                                        \
      asm volatile(REP16(opstr " %0, %0, %4, %5\n\t" opstr " %1, %1, %4, %5\n\t" opstr " %2, %2, %4, %5\n\t" opstr " %3, %3, %4, %5\n\t" \
                         opstr " %0, %0, %4, %5\n\t" opstr " %1, %1, %4, %5\n\t" opstr " %2, %2, %4, %5\n\t" opstr " %3, %3, %4, %5\n\t") \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));                               \
    }                                                                                                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                         \
    if (a0[0] + a1[0] + a2[0] + a3[1] == 0.12345f) out[0] = 0;                                               \
  }
KERNEL2(pkfma, "v_pk_fma_f32")
#define KERNEL2B(name, opstr)                                                                                \
  __global__ __launch_bounds__(512) void k_##name(unsigned long long* out, int iters) {                      \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                    \
    f2 a0 = {1.f + threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.9999f};   \
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b));                                      \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
    for (int i = 0; i < iters; ++i) {                                                                        \
      asm volatile(REP16(opstr " %0, %0, %4\n\t" opstr " %1, %1, %4\n\t" opstr " %2, %2, %4\n\t" opstr " %3, %3, %4\n\t" \
                         opstr " %0, %0, %4\n\t" opstr " %1, %1, %4\n\t" opstr " %2, %2, %4\n\t" opstr " %3, %3, %4\n\t") \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));                                       \
    }                                                                                                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                         \
    if (a0[0] + a1[0] + a2[0] + a3[1] == 0.12345f) out[0] = 0;                                               \
  }
KERNEL2B(pkadd, "v_pk_add_f32")
KERNEL2B(pkmul, "v_pk_mul_f32")

typedef void (*kern_t)(unsigned long long*, int);
struct Case { const char* name; kern_t k; };

int main() {
  const Case cases[] = {{"v_fma_f32", k_fma}, {"v_add_f32", k_addf}, {"v_mul_f32", k_mulf}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl},
                        {"v_mov_b32", k_mov}, {"v_bfe_u32", k_bfe}, {"v_lshl_add_u32", k_lshladd}, {"v_xad_u32", k_xad}, {"v_add3_u32", k_add3},
                        {"v_and_or_b32", k_andor}, {"v_cndmask_b32", k_cndmask}, {"v_perm_b32", k_perm}, {"v_cvt_pk_bf16_f32", k_cvtpk},
                        {"v_pk_max_u16", k_pkmaxu16}, {"v_pk_min_i16", k_pkmini16}, {"v_pk_sub_u16", k_pksubu16}, {"v_pk_lshrrev_b16", k_pklshr16},
                        {"v_pk_mad_u16", k_pkmadu16}, {"v_max3_f32", k_max3}, {"v_dot2_f32_bf16", k_dot2bf16}, {"v_pk_fma_f32", k_pkfma},
                        {"v_pk_add_f32", k_pkadd}, {"v_pk_mul_f32", k_pkmul}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp}};
  unsigned long long* d;
  hipMalloc(&d, 4096 * 8);
  const int iters = 200, per_iter = 16 * 8;
  printf("%-22s %12s %12s   (cycles of s_memtime per wave64 instruction and SIMD)\n", "opcode", "1 wave/SIMD", "2 waves/SIMD");
  for (const Case& c : cases) {
    double r[2];
    for (int two = 0; two < 2; ++two) {
      const int threads = two ? 512 : 256;
      hipLaunchKernelGGL(c.k, dim3(256), dim3(threads), 0, 0, d, iters);      // warm
      hipLaunchKernelGGL(c.k, dim3(256), dim3(threads), 0, 0, d, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * 8);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0; int n = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 8 + w]; ++n; }
      // cycles per instruction ISSUED BY THE SIMD: a wave's span / its instructions, divided by the waves sharing the SIMD
      r[two] = s / n / ((double)iters * per_iter) / (two ? 2.0 : 1.0);
    }
    printf("%-22s %12.2f %12.2f\n", c.name, r[0], r[1]);
  }
  return 0;
}
