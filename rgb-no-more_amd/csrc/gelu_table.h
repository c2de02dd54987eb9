// Table GELU for epilogues that hold bf16 pre-activations as packed pairs (mlp_fused.hip builds the table and documents it above
// gelu_full_kernel: the library's own GELU arithmetic evaluated for all 65 536 bf16 inputs, a compact image of it -- 13 328 bytes --
// at LDS offset 0, closed forms outside the image's window; tests/test_gelu_table.py).  Shared by the row-panel GEMM epilogues
// (gemm_nt_kpipe_body.inc); the fused MLP and the chain kernels carry their own copy of
// the same instruction sequence.
#pragma once
#include "common.h"
#include "../../include/rgbnm.h"

int rgbnm_gelu_table_query(const unsigned** img, int* A0, int* P1, int* N1, int* ndw);   // mlp_fused.hip

namespace rgbnm {

typedef unsigned int tab_u32x4 __attribute__((ext_vector_type(4)));

constexpr int GELU_TAB_BYTES = 13328;        // image size (mlp_fused.hip F_TAB_BYTES)
constexpr int GELU_TAB_RESERVE = 14336;      // LDS bytes to keep free at offset 0

struct GeluTabKeys {                         // kernel argument part: the image and the splat constants of the packed 16-bit key arithmetic
  const unsigned* img;
  unsigned kneg, kpos, klo, koff, ksgn;
};

// host: keys of the current device's table; false when option gelu_table is off or rgbnm_gelu_table_init has not found a usable table
inline bool gelu_table_keys(GeluTabKeys& k) {
  k.img = nullptr; k.kneg = k.kpos = k.klo = k.koff = k.ksgn = 0;
  if (!rgbnm_get_option("gelu_table")) return false;
  int A0 = 0, P1 = 0, N1 = 0, ndw = 0;
  const unsigned* img = nullptr;
  if (rgbnm_gelu_table_query(&img, &A0, &P1, &N1, &ndw) != 1 || ndw * 4 > GELU_TAB_BYTES) return false;
  k.img = img;
  k.kneg = 0x00010001u * (unsigned)(0x8000 | N1);
  k.kpos = 0x00010001u * (unsigned)P1;
  k.klo = 0x00010001u * (unsigned)(A0 - 1);
  k.koff = 0x00010001u * (unsigned)((0x10000 - 4 * (A0 - 1)) & 0xffff);
  k.ksgn = 0x00010001u * (unsigned)(4 * (P1 - A0 + 2));
  return true;
}

// device: LDS-DMA the image to LDS offset 0; wave `w` of `nwaves` takes every nwaves-th 1 KB piece (the caller waits vmcnt + barrier)
__device__ __forceinline__ void gelu_table_dma(const unsigned* img, unsigned char* lds0, int w, int nwaves, int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr_;
  typedef const __attribute__((address_space(1))) void* glb_ptr_;
  const int npiece = GELU_TAB_BYTES / 16;                              // 833 pieces of 16 bytes
  for (int i = w; i * 64 < npiece; i += nwaves)
    if (i * 64 + lane < npiece)
      __builtin_amdgcn_global_load_lds((glb_ptr_)(img + (i * 64 + lane) * 4), (lds_ptr_)(lds0 + i * 1024), 16, 0, 0);
}

// device: gelu / gelu' of four packed bf16 pairs (eight elements): bits of a pair -> packed 16-bit keys -> one ds_read_b32 of
// {D | gelu' << 16} per element (all eight in flight, one wait) -> |gelu| = max(a, 0x80) - D.  k4v: 0x00040004 in a VGPR.
__device__ __forceinline__ void gelu_table_pairs4(const unsigned (&pb)[4], unsigned (&gq)[4], unsigned (&dq)[4], const GeluTabKeys& k,
                                                  unsigned k4v) {
  unsigned agv[4], alo[4], ahi[4], e0[4], e1[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    unsigned p1, p2, ak, i4, sg;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(p1) : "v"(pb[jj]), "s"(k.kneg));
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(p2) : "v"(p1), "s"(k.kpos));
    p1 &= 0x7FFF7FFFu;
    p2 &= 0x7FFF7FFFu;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(agv[jj]) : "v"(p1), "s"(0x00800080u));
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(ak) : "v"(p2), "s"(k.klo));
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(i4) : "v"(ak), "v"(k4v), "s"(k.koff));
    asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(sg) : "s"(0x000F000Fu), "v"(pb[jj]));
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(i4) : "v"(sg), "s"(k.ksgn), "v"(i4));
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
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const unsigned dpair = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x05040100u);
    unsigned gm;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(gm) : "v"(agv[jj]), "v"(dpair));
    gq[jj] = (pb[jj] & 0x80008000u) | gm;
    dq[jj] = __builtin_amdgcn_perm(e1[jj], e0[jj], 0x07060302u);
  }
}

}  // namespace rgbnm
