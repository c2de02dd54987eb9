// Cross-translation-unit launch helpers inside librgbnm.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

constexpr int RGBNM_TN_MAX_SPLIT = 128;   // token-axis splits of a weight-gradient GEMM (workspace is sized for it)
// Pipelined bf16 weight-gradient GEMM (gemm_tn_pipe.hip).  Returns RGBNM_OK, or 1 if the shape is not eligible
// (caller falls back to the generic kernel), or a negative error.
// One weight-gradient GEMM of a grouped launch (gemm_tn_pipe.hip): part[s][No][Ki], bpart[s][No] (or null).
struct RgbnmTnJob {
  const void* dY; const void* X; float* part; float* bpart;
  int ldy, ldx, M, No, Ki;
  // where the reduced result goes (may be null: always through part / bpart): when the launch ends up WITHOUT a token split
  // (S == 1) and accumulate == 0 the kernel writes dW / db itself -- rows permuted as the reduction would (perm_heads) -- instead
  // of partials that a reduction launch only copies
  float* dW; float* db; int perm_heads, accumulate;
  int smax;      // > 0: the job's workspace holds at most this many split slices (part[s], bpart[s]): the launch splits no further
};
// *direct_out (may be null) = 1 when the kernel wrote dW / db of every job itself: the caller submits no reductions
int rgbnm_launch_tn_pipe_group(const RgbnmTnJob* jobs, int n, int* S_out, hipStream_t st, int* direct_out = nullptr);
// Queue the eligible bf16 rgbnm_gemm_tn calls that follow and run them as one grouped launch at flush (or when 4 are
// queued); their partial reductions are submitted at flush.  Used by rgbnm_vit_block_bwd to pair fc2/fc1 and proj/qkv.
void rgbnm_tn_defer_begin();
void rgbnm_tn_defer_begin_n(int max_jobs);     // the same with room for up to max_jobs (<= 48) jobs per launch: the GEMMs of several blocks
int rgbnm_tn_defer_flush(hipStream_t st);
int rgbnm_launch_tn_pipe(const void* dY, int ldy, const void* X, int ldx, float* part, float* bpart, int M, int No,
                         int Ki, int* S_out, hipStream_t st, int smax = 0);

// bf16 attention with LDS-DMA tiles and transpose reads (attention_v2.hip)
int rgbnm_launch_attn2_fwd(const void* qkv, void* out, float* lse, int B, int N, int heads, float scale,
                           hipStream_t st);
int rgbnm_launch_attn2_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                           int N, int heads, float scale, hipStream_t st);

// per-kernel HIP-event tracing (vit.hip); tags: 1 gemm_nt, 2 gemm_tn, 3 attention fwd, 4 attention bwd, 5 / 6 the one-launch encoder
// forward / backward (vit_chain.hip, vit_chain_bwd.hip); the HBM-bound classes of SURVEY 8d: 7 augment stage (dct_resize +
// dct_randaug as one bracket), 8 sub-block embed, 9 clip + AdamW + WeightDecay (sqnorm + adamw as one bracket)
enum { TR_NT = 1, TR_TN = 2, TR_ATTN_FWD = 3, TR_ATTN_BWD = 4, TR_CHAIN_FWD = 5, TR_CHAIN_BWD = 6, TR_AUG = 7, TR_EMBED = 8, TR_OPT = 9 };
// Weight-resident K = 192 bf16 NT GEMM (gemm_nt_wres.hip).  Returns RGBNM_OK / error, or 1 if the shape is not eligible.
int rgbnm_launch_nt_wres(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                         const void* R, int ldr, void* C2, int ldc2, int M, int N, int K, hipStream_t st);
// Row-panel N = 192 bf16 NT GEMM with a pipelined reduction (gemm_nt_kpipe.hip).  Same return convention.
int rgbnm_launch_nt_kpipe(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                          const void* R, int ldr, void* C2, int ldc2, int M, int N, int K, hipStream_t st);
// Linear + bias + residual + LayerNorm of the result in one launch (gemm_nt_kpipe.hip); 1 = not eligible.
int rgbnm_launch_nt_kpipe_res_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* R,
                                 int ldr, void* x, int ldc, const float* gamma, const float* beta, void* y, int ldy,
                                 float* mean, float* rstd, float eps, int M, int N, int K, hipStream_t st);
// Fused FeedForwardBlock forward (mlp_fused.hip): x_out = x_mid + fc2(gelu(fc1(xn2))) [+ LayerNorm of x_out], saving
// gelu(u) -> G and gelu'(u) -> GP for the backward; 1 = not eligible (E != 192, hidden != 768, small M).
int rgbnm_launch_mlp_fwd(const void* X, int ldx, const void* W1, const float* b1, const void* W2, const float* b2,
                         const void* R, int ldr, void* G, void* GP, int ldg, void* Y, int ldy, const float* gamma,
                         const float* beta, void* Y2, int ldy2, float* mean, float* rstd, float eps, int M, int E, int H,
                         hipStream_t st);
// NT GEMM for few rows (M <= 512: the classification head), bf16 operands, bf16 or fp32 output, epilogues none / tanh /
// (1 - h^2) product (gemm_nt_small.hip); 1 = not eligible.
int rgbnm_launch_nt_small(int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const float* bias,
                          const void* R, int ldr, int c_f32, int M, int N, int K, hipStream_t st);
// The backward of the same block's data path in one launch (mlp_fused.hip): du = (dy . W2) * gelu'(u) -> DU (for the dW1 GEMM),
// dx = dy + LayerNorm'(du . W1) -> DX, panel partial sums of dgamma / dbeta -> part [npanels][2][192]; 1 = not eligible.
int rgbnm_launch_mlp_bwd(const void* DY, int lddy, const void* W2T, const void* W1T, const void* GP, int ldg, void* DU, int ldu,
                         const void* X, int ldx, const float* gamma, const float* mean, const float* rstd, void* DX, int lddx,
                         float* part, int* npanels_out, int M, int E, int H, hipStream_t st);
// fc1 / qkv dX GEMM with the LayerNorm backward fused into the epilogue (gemm_nt_kpipe.hip); 1 = not eligible.
int rgbnm_launch_nt_kpipe_lnbwd(const void* A, int lda, const void* W, int ldw, const void* X, int ldx,
                                const float* gamma, const float* mean, const float* rstd, const void* dres, int ldr,
                                void* dx, int ldc, float* part, int* npanels_out, int M, int N, int K, hipStream_t st);
// Batched reduction of split partials (reduce.hip): out[map(i)] (+)= sum_s part[s * stride + i], i < n.
// epw = elements per workgroup: 64 (4 partial-groups; S up to ~64) or 8 (32 partial-groups; S in the hundreds).
struct RgbnmReduceJob {
  const float* part; long long stride; float* out;
  int n, S, cols, perm_heads, accumulate, epw;      // perm_heads > 0: row i / cols is a de-interleaved qkv row
};
bool rgbnm_reduce_defer_active();                    // inside a bracket opened further up the call chain?
void rgbnm_reduce_defer_begin();                     // queue the following submits ...
int rgbnm_reduce_defer_flush(hipStream_t st);        // ... and run them as one launch
int rgbnm_reduce_submit(const RgbnmReduceJob& job, hipStream_t st);
int rgbnm_trace_begin(int tag, double flops, double bytes, hipStream_t st);
void rgbnm_trace_end(int slot, hipStream_t st);
