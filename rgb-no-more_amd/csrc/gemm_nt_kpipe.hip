// Row-panel NT GEMM for the N = 192 Linear layers with a long reduction (fc2: K = 768 + bias + residual; the dX
// GEMMs of fc1 and qkv: K = 768 / 576), bf16:   C[M,192] = epi(A[M,K] . W[192,K]^T).
// These read 58-77 MB of activations to produce 19 MB: HBM-read bound.  One 448-thread workgroup per CU owns ONE
// contiguous panel of rows (M / 256 = 196 tokens at B = 256: exactly one workgroup per CU, a single balanced round;
// 7 waves x 32 rows, the last 28 rows are padding) and all 192 output columns, and streams the reduction through a
// 3-stage LDS-DMA ring of 64-wide k-tiles (A panel 28 KB + W 24 KB per stage; two tiles = 104 KB in flight per CU,
// hand-counted vmcnt, one raw s_barrier per k-tile).  Rows are 128 bytes in LDS; 16-byte chunks are XOR-swizzled
// (chunk ^ rot3((row>>1)&7)) through the DMA source address so ds_read_b128 over 32 rows is conflict free.
// Epilogue: accumulators (MFMA issued with swapped operands: a lane owns one token) -> bf16 staging tile in the ring
// memory -> coalesced 16-byte row pieces, bias and the residual (prefetched into registers before the loop) fused.
#include "common.h"
#include "internal.h"
#include "ln_bwd_rows.h"
#include "gelu_table.h"
#include "../../include/rgbnm.h"

#define KP_NS kp7
#define KP_NWAVES 7
#define KP_NSTAGE 3
#define KP_BIAS_LDS true
#include "gemm_nt_kpipe_body.inc"
#undef KP_NS
#undef KP_NWAVES
#undef KP_NSTAGE
#undef KP_BIAS_LDS

// kp8: 8 waves x 32 rows = 256-row panels, 2-stage ring (112 KB), plain epilogues only.  For row counts that are multiples of
// 256 but not of 224 -- the SwinV2-T stages at B = 256 (M = 2^20, 2^18, 2^16, 2^14) -- every round of workgroups is full:
// at M = 16384, N = 768 the 224-row panels made 296 workgroups (two rounds on 256 CUs) where 256 of these make one.
#define KP_NS kp8
#define KP_NWAVES 8
#define KP_NSTAGE 2
#define KP_BIAS_LDS true
#include "gemm_nt_kpipe_body.inc"
#undef KP_NS
#undef KP_NWAVES
#undef KP_NSTAGE
#undef KP_BIAS_LDS

// kp8 when its rounds of workgroups cost less than kp7's: rounds x rows per panel, kp8 charged 5 % for its shallower ring
// (measured: equal at M = 2^18 where both geometries fill their rounds, 17 % / 28 % faster at M = 2^16 / 2^14).
static bool use_kp8(int M, int N, int) {
  if (M % 256 || N % 192 || !rgbnm_get_option("kp8")) return false;
  // row counts that suit both geometries (M = 50176 = 224 x 224 = 196 x 256, the ViT batch): the persistent 7-wave kernel.  (Round 4
  // measured the plain GEMMs 2 - 10 % faster on the 2-D wave tiles of kp8; with the pinned k-tile order of ktile_mma the
  // persistent kernel is ahead again: 55.5 vs 57.7 us per launch over the 48 plain GEMMs of a JPEG-S step.)
  if (M % 224 == 0) return false;
  const long long nt = N / 192, cus = 256;
  const long long r7 = ((long long)((M + 223) / 224) * nt + cus - 1) / cus * 224, r8 = ((long long)(M / 256) * nt + cus - 1) / cus * 256;
  return r8 * 105 < r7 * 100;
}

// x = A . W^T + bias + R ; y = LayerNorm(x): fc2 (+ the next block's LN1) and proj (+ LN2) in one launch each.
int rgbnm_launch_nt_kpipe_res_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* R,
                                 int ldr, void* x, int ldc, const float* gamma, const float* beta, void* y, int ldy,
                                 float* mean, float* rstd, float eps, int M, int N, int K, hipStream_t st) {
  return kp7::launch_res_ln(A, lda, W, ldw, bias, R, ldr, x, ldc, gamma, beta, y, ldy, mean, rstd, eps, M, N, K, st);
}

// dx = [dres +] LayerNorm'(A . W^T): the dX GEMM of fc1 / qkv with the LayerNorm backward fused into its epilogue
// (saves writing and re-reading the [M,192] gradient and one launch).  part: [npanels][2][192] partial dgamma/dbeta;
// *npanels_out tells the caller how many slices to reduce.  Returns 1 when the shape is not eligible.
int rgbnm_launch_nt_kpipe_lnbwd(const void* A, int lda, const void* W, int ldw, const void* X, int ldx,
                                const float* gamma, const float* mean, const float* rstd, const void* dres, int ldr,
                                void* dx, int ldc, float* part, int* npanels_out, int M, int N, int K, hipStream_t st) {
  return kp7::launch_lnbwd(A, lda, W, ldw, X, ldx, gamma, mean, rstd, dres, ldr, dx, ldc, part, npanels_out, M, N, K, st);
}

// returns 1 when the shape is not eligible (caller falls back to the tile-per-workgroup kernel)
int rgbnm_launch_nt_kpipe(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                          const void* R, int ldr, void* C2, int ldc2, int M, int N, int K, hipStream_t st) {
  if (use_kp8(M, N, epi)) return kp8::launch_plain(epi, A, lda, W, ldw, C, ldc, bias, R, ldr, C2, ldc2, M, N, K, st);
  return kp7::launch_plain(epi, A, lda, W, ldw, C, ldc, bias, R, ldr, C2, ldc2, M, N, K, st);
}

#ifdef KP_PROF
extern "C" int rgbnm_debug_kp_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kp7::g_kp_prof), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : -1;
}
#endif
