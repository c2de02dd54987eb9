/* rgbnm.h -- C ABI of librgbnm.so: the MI355X (gfx950) hot path of RGB-no-more.
 *
 * Drop-in boundary.  The reference has no native FFI for this path except
 * dct_manip (pybind11); everything else is PyTorch ops reached from Python, so the binding a maintainer
 * adds is a ctypes stub.  Each entry point below names the reference code it
 * replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers + sizes; device pointers unless the name says host; no C++/torch types.
 *   - returns 0 on success, negative RGBNM_E* otherwise; never throws; never allocates device memory;
 *     keeps no per-call state; all work is enqueued on `stream` (a hipStream_t passed as void*), no sync.
 *   - dtype: 0 = fp32 ("strict" mode, exact-fp32 MFMA), 1 = bf16 (fp32 accumulate).  Activations and
 *     MFMA operands use that type; parameters, gradients, statistics and optimizer state are fp32.
 *   - tensors are dense row-major; "ld*" are row strides in elements.
 */
#ifndef RGBNM_H
#define RGBNM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBNM_ABI_VERSION 3
#define RGBNM_DT_F32 0
#define RGBNM_DT_BF16 1
#define RGBNM_DT_I16 2   /* only as out_dtype of rgbnm_dct_augment[_ex] */

/* epilogues of rgbnm_gemm_nt */
#define RGBNM_EPI_NONE 0   /* C = A.W^T (+bias)                                         */
#define RGBNM_EPI_RES 1    /* C = A.W^T + bias + R            (ResidualAdd, plainvit.py:475-479)       */
#define RGBNM_EPI_GELU 2   /* u = A.W^T + bias ; C = gelu_erf(u), C2 = gelu_erf'(u)  (FeedForwardBlock :487-488) */
#define RGBNM_EPI_POS 3    /* C = A.W^T + bias + pos[row % period]      (SinCosEmbedding :97-121)      */
#define RGBNM_EPI_DGELU 4  /* C = (A.W^T) * R, R = the C2 (gelu') saved by RGBNM_EPI_GELU         */
#define RGBNM_EPI_TANH 5   /* C = tanh(A.W^T + bias)                    (ClassificationHead :553-554)  */
#define RGBNM_EPI_DTANH 6  /* C = (A.W^T) * (1 - R^2)                                             */

int rgbnm_abi_version(void);
/* Runtime switches: every one selects between parity-tested kernels (tests/ run both sides); defaults in the middle column.
 * Round 6 removed the switches whose alternative had measured slower in every configuration and nothing else used
 * (tn_square, nt_dmawave, kp_split, nt_kstream, attn_proj -- numbers in DESIGN.md 7).
 *   GEMM  C = A . W^T (nn.Linear forward, dX)
 *   "nt_staged"    1  coalesced LDS-staged epilogue of the generic tile kernel (0: direct fragment stores)
 *   "nt_wres"      1  K = 192 layers on the weight-resident persistent kernel (csrc/gemm_nt_wres.hip)
 *   "nt_kpipe"     1  N % 192 == 0, K >= 256 on the row-panel kernel with the k-tile DMA ring (csrc/gemm_nt_kpipe.hip)
 *   "kp8"          1  ... its 8-wave / 256-row geometry for row counts that are multiples of 256 (the SwinV2 stages)
 *   "kp_persist"   1  ... its persistent form for several column tiles (JPEG-S: E = 384)
 *   "nt_small"     1  at most 512 rows (the classification head) on 32 x 32 tiles with an in-workgroup split of K
 *   "ln_fuse"      1  LayerNorm forward / backward in the row-panel kernel's epilogues (E = 192)
 *   GEMM  dW = dY^T . X (weight gradients)
 *   "tn_pipe"      1  pipelined kernel (csrc/gemm_tn_pipe.hip; 0: the generic one)     "tn_tr"  1  its ds_read_b64_tr_b16 fragments
 *   "tn_group"     2  dW GEMMs of a ViT block: 0 one per launch, 1 pairs, 2 all four in one launch
 *   "tn_direct"    1  a grouped launch that ends up without a token split writes dW / db itself (no partial sums, no reduction)
 *   "tn_pack"      1  ... and places its tiles so that no GEMM straddles two XCDs (L2 locality)
 *   "tn_wide"      1  weight-gradient launches whose GEMMs all have No % 192 == 0 and Ki % 384 == 0 (E = 384 and wider) run 192 x 384
 *                     tiles (128 flops per byte taken in instead of 77: the kernel is bound by its intake there, not by HBM)
 *   "tn_wgs"     512  workgroup budget of the generic kernel's token split
 *   attention
 *   "attn_v2"      1  LDS-DMA / transpose-read attention kernels (0: first-generation kernels, also used for 294 tokens)
 *   "attn_persist" 1  persistent forward / backward with a DMA wave (>= 256 (image, head) pairs)
 *   FeedForwardBlock, E = 192
 *   "mlp_fuse"     1  fc1 + GELU + fc2 + residual [+ next LayerNorm] as one launch          "mlp_bwd"  1  its backward data path as one
 *   "mlp_dmast"    1  ... the DMA wave stores the saved-tensor tiles of waves 4-6           "nt_cold"  0  ... non-temporal hint on them
 *   "gelu_table"   1  table GELU in the bf16 kernels once rgbnm_gelu_table_init has run (0: the erf arithmetic; same bits)
 *   the one-launch encoder (E = 192, 3 heads, 196 tokens, bf16)
 *   "fwd_chain"    1  rgbnm_vit_chain_fwd eligible        "bwd_chain"  1  rgbnm_vit_chain_bwd eligible (0: the per-operation kernels)
 *   SwinV2
 *   "ln_rows"      1  lanes-per-row LayerNorm kernels for the four stage widths            "win_xcd"  1  XCD-aware window walk
 *   measurement
 *   "trace"        0  bit mask of kernel classes to bracket with HIP events, see rgbnm_trace_collect */
int rgbnm_set_option(const char* name, int value);
int rgbnm_get_option(const char* name);
/* With option "trace" = (1 << tag) the launchers bracket kernels of that class with HIP events recorded on the launch
 * stream (tags: 1 gemm_nt, 2 gemm_tn, 3 attention fwd, 4 attention bwd, 5 one-launch encoder forward, 6 one-launch encoder backward,
 * 7 augment stage, 8 sub-block embed, 9 clip + AdamW + WeightDecay).  collect() synchronises those events and
 * returns their summed elapsed ms plus the algorithmic FLOPs / bytes of the bracketed launches, then forgets them. */
int rgbnm_trace_collect(int tag, double* ms_total, double* flops_total, double* bytes_total, int* count);
/* Create the events of the next n_events / 2 bracketed launches NOW (call outside a timed region: the runtime grows its signal
 * pool in chunks, and the chunk that a traced step happens to trigger costs the host tens of milliseconds -- 70 - 85 ms measured in
 * the 9th traced step of bench.py, enough to drain a 16-step queue). */
int rgbnm_trace_reserve(int n_events);
/* human readable text for a negative return code */
const char* rgbnm_strerror(int code);

/* ---------------------------------------------------------------------------------------------
 * GEMMs (replace nn.Linear forward/backward: plainvit.py:195,441,443,487,490,553,555)
 * ------------------------------------------------------------------------------------------- */
/* C[M,N] = epi(A[M,K] . W[N,K]^T).  bias fp32 [N] or NULL.  R/C2/pos as required by `epi`.
 * c_f32 != 0 stores C as fp32 regardless of dtype (logits). */
int rgbnm_gemm_nt(int dtype, int epi, const void* A, int lda, const void* W, int ldw, void* C, int ldc,
                  const float* bias, const void* R, int ldr, void* C2, int ldc2, const float* pos, int pos_period,
                  int M, int N, int K, int c_f32, void* stream);

/* dW[No,Ki] (fp32) = dY[M,No]^T . X[M,Ki]; db[No] = column sums of dY (db may be NULL).
 * perm_heads > 0: rows of dW/db are written in the reference's interleaved '(h d qkv)' order
 * (plainvit.py:447) although dY's columns are [q|k|v] blocks.  accumulate != 0: += into dW/db. */
size_t rgbnm_gemm_tn_workspace(int M, int No, int Ki);
/* ... or room for `splits` (1 .. 128) slices of fp32 partial sums only: rgbnm_gemm_tn takes any workspace of at least one slice and
 * splits the token axis no further than the workspace allows (a job queued in a group bracket is split at most 256 / its own
 * 128 x 192 output tiles ways: swinv2.py sizes its queued jobs' workspaces by that -- 13 GB of scratch per backward before) */
size_t rgbnm_gemm_tn_workspace_splits(int No, int Ki, int splits);
int rgbnm_gemm_tn(int dtype, const void* dY, int ldy, const void* X, int ldx, float* dW, float* db, int M, int No,
                  int Ki, int perm_heads, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Grouped weight gradients.  Between _begin and _end (same host thread) bf16 rgbnm_gemm_tn calls over the SAME row count are
 * queued -- up to four -- and run as ONE launch at _end (or when the queue fills / the row count changes): with T output tiles
 * in total every job is split 256 / T ways instead of 256 / its own tiles, so fewer fp32 partial sums are written and re-read,
 * and one reduction launch serves all jobs.  dW / db hold nothing until _end returns; every queued call needs ITS OWN workspace
 * (the partial sums live there until _end).  The ViT block backward groups its four dW GEMMs this way internally; this pair is
 * for callers that drive the Linears one by one (swinv2.py: the two dW GEMMs of an MLP).  Reference: autograd computes each
 * Linear's weight gradient as its own GEMM (torch.nn.Linear backward). */
void rgbnm_gemm_tn_group_begin(void);
int rgbnm_gemm_tn_group_end(void* stream);
/* The same with room for up to max_jobs (<= 48) queued GEMMs: a caller that brackets a whole backward pass (swinv2.py: every
 * Linear of a stage has the same row count) gets launches of up to 256 output tiles -- what is queued runs when the next job
 * would pass 256 tiles, when the row count changes, when max_jobs are queued, and at _end.  With no token split left the kernel
 * writes dW / db itself: no partial sums, no reduction.  The operands of a queued call must stay alive and unmodified, and its
 * dW / db hold nothing, until _end returns. */
void rgbnm_gemm_tn_group_begin_n(int max_jobs);
/* Brackets nest by joining: a _begin inside an open bracket (also the ones rgbnm_head_bwd / rgbnm_vit_block_bwd open internally)
 * only counts, its _end neither launches nor closes anything; the jobs run at the OUTERMOST _end.  A _begin on a thread whose queue
 * still holds jobs of a pass that died before its _end starts from an empty queue (those jobs are dropped, never launched).
 * _begin_id names the bracket (id != 0): rgbnm_gemm_tn_group_abort(id), callable from ANY host thread, makes the thread that owns
 * the bracket drop its queue without launching, the next time it touches it -- for callers whose backward nodes run on an autograd
 * worker thread while the pass is found abandoned on another one (swinv2.py).  Reference: none (torch launches every GEMM at once). */
void rgbnm_gemm_tn_group_begin_id(int max_jobs, unsigned long long id);
void rgbnm_gemm_tn_group_abort(unsigned long long id);

/* One nn.Linear in the flat fp32 master buffer and where its shadows go (offsets in elements). */
typedef struct rgbnm_linear_desc {
  long long w_off;     /* master: weight [N,K] fp32                                   */
  long long b_off;     /* master: bias [N] fp32                                       */
  long long ws_off;    /* shadow: [N,K] in dtype (rows de-interleaved if perm_heads)  */
  long long wst_off;   /* shadow: [K,N] in dtype (transpose of the above)             */
  long long bperm_off; /* bias_perm: [N] fp32 de-interleaved bias (only if perm_heads) */
  int N, K;
  int perm_heads;      /* >0 for the qkv Linear: number of heads                      */
  int add_identity;    /* shadows hold W + I (residual Linear y = x W^T + x, plainvit.py:345-347) */
  int ldn;             /* row stride of the [K,N] shadow; 0 = N.  > N pads N for 16-byte rows (a class count that is not a
                          multiple of 8: the [N,K] shadow then has ldn rows too, the extra ones left as the caller zeroed them) */
  int pair;            /* != 0: block-diagonal shadows diag(W, W) [2N,2K] and diag(W^T, W^T) [2K,2N] (off-diagonal blocks left as
                          the caller zeroed them).  x [M,K] read as [M/2,2K] times diag(W,W)^T is y [M,N] read as [M/2,2N]: a
                          96-wide Linear (SwinV2-T stage 1) then runs on the kernels tuned for 192-wide rows            */
  int chain_kind;      /* rgbnm_prep_weights_chain: 0, or which Linear of an encoder block this is: 1 qkv, 2 projection, 3 fc1, 4 fc2 */
  long long chain_off; /* ... and the element offset of that block's image inside the forward / backward chain images */
  int bias_mode;       /* bias_perm[bperm_off ..] also receives (perm_heads == 0): 1 the bias b_off as it is; 2 the SwinV2 qkv bias
                          q_bias | 0 | v_bias (swinv2.py:150-152: k has no bias), q_bias at b_off, v_bias at b2_off, N / 3 each.
                          With `pair` the N values are written twice (the operand of the row-paired GEMM): 2N floats */
  int _pad2;
  long long b2_off;
} rgbnm_linear_desc;

/* master fp32 -> per-step operand shadows for every Linear (descs_dev: device array of ndesc descriptors). */
int rgbnm_prep_weights(int dtype, const rgbnm_linear_desc* descs_dev, int ndesc, const float* master, void* shadow,
                       float* bias_perm, void* stream);
/* The same launch also writes the chain images of the one-launch encoder kernels (rgbnm_vit_chain_fwd / _bwd below; bf16,
 * E = 192, 3 heads) straight from the fp32 masters: for every descriptor with chain_kind != 0 the weight goes to its place in
 * chain_fwd ([N, K] orientation) and chain_bwd ([K, N]) -- either may be NULL -- exactly where rgbnm_chain_gather over the
 * shadows would put it (rgb-no-more_amd/chain.py documents the layout; the kernel carries its arithmetic inverse).
 * skip_chain_shadows != 0: the [N, K] / [K, N] shadows of those descriptors are NOT written (nothing reads them while the
 * one-launch kernels run the encoder).  Before round 6 this was four launches per step (prep, bias gather, two image gathers). */
int rgbnm_prep_weights_chain(int dtype, const rgbnm_linear_desc* descs_dev, int ndesc, const float* master, void* shadow,
                             float* bias_perm, void* chain_fwd, void* chain_bwd, int skip_chain_shadows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (nn.LayerNorm(emb), plainvit.py:513,522,551) and head pooling (:551-552)
 * ------------------------------------------------------------------------------------------- */
int rgbnm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int M, int E, float eps, void* stream);
size_t rgbnm_layernorm_bwd_workspace(int M, int E);
/* dx = [dres +] LN'(dy); dgamma/dbeta fp32.  dres may be NULL. */
int rgbnm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, int M, int E,
                        int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* pooled[b,:] = mean_t LN(x[b,t,:]) */
int rgbnm_head_pool_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* pooled, float* mean,
                        float* rstd, int B, int N, int E, float eps, void* stream);
int rgbnm_head_pool_bwd(int dtype, const void* dpooled, const void* x, const float* gamma, const float* mean,
                        const float* rstd, void* dx, float* dgamma, float* dbeta, int B, int N, int E, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention (MultiHeadAttention.forward, plainvit.py:445-464): qkv [B*N, 3*heads*64] as [q|k|v] column
 * blocks, out [B*N, heads*64] ('b n (h d)'), lse [B*heads*N] fp32.  scale = 1/sqrt(emb_size) (!).
 * ------------------------------------------------------------------------------------------- */
int rgbnm_attention_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int heads, float scale,
                        void* stream);
int rgbnm_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                        int B, int N, int heads, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sub-block reshuffle (patch2rearrange + apply_subblock + collapser + cat, plainvit.py:71-88,50-69,192,
 * 200-216) for patch_size 16: y [B,1,Hb,Wb,8,8], cbcr [B,2,Hb/2,Wb/2,8,8] -> feat [B*(Hb/2)*(Wb/2), 384].
 * conv16: the 16x16 fp32 conversion matrix A (dct_ops.py:180-208).
 * ------------------------------------------------------------------------------------------- */
int rgbnm_subblock_embed(int in_dtype, int out_dtype, const void* y, const void* cbcr, const float* conv16,
                         void* feat, int B, int Hb, int Wb, int transpose_a, void* stream);
/* The same on a batch that RandomMixup_DCT (utils/cls_transforms.py:163-176) has NOT been applied to yet: lam_dev (device, two
 * floats, may be NULL = no mixing) mixes image b with image b - 1 (mod B) as the values are loaded, rounded to in_dtype exactly
 * as rgbnm_mixup(in_dtype -> in_dtype) stores them -- the same bits as rgbnm_mixup followed by rgbnm_subblock_embed without the
 * mixed batch ever existing in memory (two launches and a round trip of the batch less per step). */
int rgbnm_subblock_embed_mix(int in_dtype, int out_dtype, const void* y, const void* cbcr, const float* lam_dev, const float* conv16,
                             void* feat, int B, int Hb, int Wb, int transpose_a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DCT-domain data path: replaces, per batch and on device, what the reference does per sample
 * on the CPU inside DataLoader workers: dequantise+clamp (datasets.py:288-293), RandomResizedCrop_DCT
 * (custom_transforms.py:631-663 -> dct_ops.py crop :584-599, resize :529-580), RandomFlip_DCT (:926-942),
 * RandAugment_dct / _apply_op_dct (:944-1127), ToRange (:436-454).  All random draws are made by the caller.
 * ------------------------------------------------------------------------------------------- */
#define RGBNM_OP_IDENTITY 0
#define RGBNM_OP_AUTOCONTRAST 1    /* autocontrast_dct on Y                 (dct_ops.py:862-887)  */
#define RGBNM_OP_POSTERIZE 2       /* iarg0 = bit offset, iarg1 = table length round(2040/2^b)+1 (:889-914) */
#define RGBNM_OP_SOLARIZEADD 3     /* iarg0 = addition                      (:653-679)            */
#define RGBNM_OP_COLOR 4           /* contrast_dct on CbCr, fmag = factor   (:839-860)            */
#define RGBNM_OP_CONTRAST 5        /* contrast_dct on Y, fmag = factor                            */
#define RGBNM_OP_BRIGHTNESS 6      /* fmag = factor - 1                     (:817-837)            */
#define RGBNM_OP_MIDFREQAUG 7      /* iarg0 = index of the 8x8 fp32 multiplier in `filters` (:710-746) */
#define RGBNM_OP_CUTOUT 8          /* iarg0 = pad (luma, even), iarg1/iarg2 = centre h/w (luma blocks) (:776-815) */
#define RGBNM_OP_TRANSLATEX 9      /* iarg0 = luma block shift int(m - m%2) (:748-774)            */
#define RGBNM_OP_TRANSLATEY 10
#define RGBNM_OP_ROTATE90 11       /* iarg0 = +1 counter-clockwise / -1 clockwise (:99-130)       */
#define RGBNM_OP_AUTOSATURATION 12 /* autocontrast_dct on CbCr jointly                            */
#define RGBNM_OP_GRAYSCALE 13      /* CbCr *= 0                             (custom_transforms.py:1004-1005) */
#define RGBNM_OP_CHROMADROP 14     /* iarg0 != 0 drops Cb, else Cr          (:1011-1015)          */
#define RGBNM_OP_SHARPNESS 15      /* sharpblur_dct, iarg0 = filter index   (dct_ops.py:681-708)  */
#define RGBNM_OP_INVERT 16         /* invert_dct: all coefficients * -1     (dct_ops.py:623-629)  */
#define RGBNM_OP_SOLARIZE 17       /* iarg0 = floor(threshold): blocks with luma DC > threshold negated (:631-651) */
#define RGBNM_OP_FREQENHANCE 18    /* fmag = factor: AC coefficients * factor, rounded (:1015-1035) */
#define RGBNM_OP_EQUALIZE 19       /* equalize_dct: histogram equalisation of the luma DCs (:916-955) */

typedef struct rgbnm_aug_params {
  int crop_i, crop_j, crop_h, crop_w; /* luma blocks; chroma box = luma box / 2 (custom_transforms.py:647-652) */
  int flip;                           /* horizontal flip after the resize */
  int op[2];
  float fmag[2];
  int iarg0[2], iarg1[2], iarg2[2];
} rgbnm_aug_params;

size_t rgbnm_dct_augment_workspace(int B);
/* Yq [B,1,Hy,Wy,8,8], CbCrq [B,2,Hc,Wc,8,8] (NULL: grayscale -> zero chroma), quant [B,3,8,8]: int16 as returned by
 * read_coefficients.  crop_w must be 56, 28 or 14 (resize /2, identity, x2 -> 28x28 luma / 14x14 chroma blocks).
 * params are needed twice: on the device (kernels) and on the host (validated before launch).
 * conv16: A(8,2) 16x16 fp32; filters: [n][64] fp32 multipliers for MIDFREQAUG/SHARPNESS (may be NULL if unused).
 * outY [B,1,28,28,8,8], outC [B,2,14,14,8,8] in out_dtype, after ToRange(-1,1; -1024,1016); out_dtype 2 (RGBNM_DT_I16)
 * stores the int16 coefficients themselves, WITHOUT ToRange (the per-transform classes of custom_transforms.py chain on it).
 * entry_clamp bit 0: clamp before the first op (RandAugment_dct.forward, :1106-1108); bit 1: the input is already
 * de-quantised (unit tables) and must NOT be clamped in front of the crop / resize / flip stage (the reference's per-transform
 * classes pass out-of-range coefficients of a preceding resize through unchanged).  nops in {0,1,2}. */
int rgbnm_dct_augment(const int16_t* Yq, const int16_t* CbCrq, const int16_t* quant, const rgbnm_aug_params* params_dev,
                      const rgbnm_aug_params* params_host, const float* conv16, const float* filters, void* outY,
                      void* outC, int out_dtype, int B, int Hy, int Wy, int Hc, int Wc, int entry_clamp, int nops,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same pipeline for a `size` x `size` output block grid: 28 (ViT pipelines, get_transform('imagenet_dct')) or 32
 * (SwinV2, 'imagenet_dct_swin', datasets.py:370-382; crop sides 16 / 32 / 64).  outY [B,1,size,size,8,8],
 * outC [B,2,size/2,size/2,8,8]. */
size_t rgbnm_dct_augment_workspace_ex(int B, int size);
int rgbnm_dct_augment_ex(const int16_t* Yq, const int16_t* CbCrq, const int16_t* quant, const rgbnm_aug_params* params_dev,
                         const rgbnm_aug_params* params_host, const float* conv16, const float* filters, void* outY,
                         void* outC, int out_dtype, int size, int B, int Hy, int Wy, int Hc, int Wc, int entry_clamp,
                         int nops, void* workspace, size_t workspace_bytes, void* stream);

/* The same pipeline on HOST-CROPPED input (loader.DCTBatchLoader(crop_on_host=True); SURVEY.md 8f f2: "ship only the crop
 * box"): image b's crop box alone, as contiguous int16 blocks [crop_h][crop_w][64] at Ypacked + y_off[b] and
 * [2][crop_h/2][crop_w/2][64] at Cpacked + c_off[b] (element offsets, device arrays of B entries; Cpacked NULL: grayscale).
 * params[b].crop_* still hold the box in the ORIGINAL Hy x Wy grid (validated against it on the host exactly as in
 * rgbnm_dct_augment); the kernels read the box from origin 0 with its own row pitch.  Output bit-identical to
 * rgbnm_dct_augment_ex on the whole grids.  Replaces datasets.py:274-297 + pipeline_utils.py:70-73 (.to(device) of whole
 * coefficient tensors). */
int rgbnm_dct_augment_packed(const int16_t* Ypacked, const int16_t* Cpacked, const long long* y_off, const long long* c_off,
                             const int16_t* quant, const rgbnm_aug_params* params_dev, const rgbnm_aug_params* params_host,
                             const float* conv16, const float* filters, void* outY, void* outC, int out_dtype, int size, int B,
                             int Hy, int Wy, int Hc, int Wc, int entry_clamp, int nops, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Train-step tail
 * ------------------------------------------------------------------------------------------- */
/* CrossEntropyLoss (pipeline_utils.py:535) with soft [B,C] fp32 or hard int64 [B] targets (exactly one non-NULL).
 * loss[0] = mean over rows; dlogits (dtype dl_dtype, may be NULL) = d loss / d logits * grad_scale*B... see .hip:
 * dlogits = (softmax*sum(t) - t) * grad_scale, pass grad_scale = 1/B for the mean reduction. */
int rgbnm_softxent(int dl_dtype, const float* logits, const float* soft_target, const long long* hard_target,
                   float* loss_rows, float* loss, void* dlogits, int B, int C, float grad_scale, void* stream);
/* The same loss as one launch per direction (round 6).  _loss: loss_rows [B], row_stats [2B] (log-sum-exp, target mass per row)
 * and loss[0] = the mean in ONE launch: the workgroup that finishes last sums the rows in a fixed order; `ticket` is one zeroed
 * 32-bit word of caller-owned device memory that the launch leaves zero again (one ticket per stream that may run the call).
 * _grad: dlogits (dl_dtype) = (softmax * sum(t) - t) * grad_scale * gout_dev[0] from the saved row statistics; gout_dev (device,
 * may be NULL = 1) is autograd's output gradient of the loss (GradScaler's scale under fp16 AMP, train.py:159): no host sync and
 * no element-wise launches to scale or convert it. */
int rgbnm_softxent_loss(const float* logits, const float* soft_target, const long long* hard_target, float* loss_rows,
                        float* row_stats, float* loss, unsigned* ticket, int B, int C, void* stream);
int rgbnm_softxent_grad(int dl_dtype, const float* logits, const float* soft_target, const long long* hard_target,
                        const float* row_stats, const float* gout_dev, void* dlogits, int B, int C, float grad_scale, void* stream);
/* The same two launches on the target RandomMixup_DCT would have built (cls_transforms.py:163-176: one-hot labels, rolled by one, mixed
 * with lam): target[b][c] = (labels[b] == c ? lam[0] : 0) + (labels[b-1] == c ? lam[1] : 0) is evaluated where it is used, the
 * dense [B, C] target and the launch that writes it do not exist (cls_transforms.LazyTarget).  Same bits as rgbnm_mixup_target
 * followed by rgbnm_softxent_loss / _grad on its output. */
int rgbnm_softxent_loss_mix(const float* logits, const long long* labels, const float* mix_lam_dev, float* loss_rows,
                            float* row_stats, float* loss, unsigned* ticket, int B, int C, void* stream);
int rgbnm_softxent_grad_mix(int dl_dtype, const float* logits, const long long* labels, const float* mix_lam_dev,
                            const float* row_stats, const float* gout_dev, void* dlogits, int B, int C, float grad_scale,
                            void* stream);
/* RandomMixup_DCT (cls_transforms.py:163-176): out[b] = lam[0]*in[b] + lam[1]*in[b-1]; lam on device. */
int rgbnm_mixup(int in_dtype, int out_dtype, const void* in, void* out, const float* lam_dev, int B,
                long long per_sample, void* stream);
int rgbnm_mixup_target(const long long* labels, float* target, const float* lam_dev, int B, int C, void* stream);
/* clip_grad_norm_(max_norm) + AdamW(weight_decay=0) + WeightDecay (train.py:163-165; custom_optims.py:37-42)
 * over flat fp32 buffers of n elements (n % 256 == 0; every tensor starts on a 256-element boundary);
 * wd_flag_per_256[i] != 0 marks chunks that belong to a decayed tensor; wd_factor = (lr/base_lr)*wd.
 * step = 1-based Adam step.  norm_out (may be NULL) receives the pre-clip global norm. max_norm<=0: no clip. */
size_t rgbnm_clip_adamw_wd_workspace(void);
int rgbnm_clip_adamw_wd_step(float* p, const float* g, float* m, float* v, const unsigned char* wd_flag_per_256,
                             long long n, float lr, float beta1, float beta2, float eps, int step, float wd_factor,
                             float max_norm, float* norm_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Composite stages: one call per autograd node (ViT, plainvit.py:559-611)
 * ------------------------------------------------------------------------------------------- */
typedef struct rgbnm_vit_cfg {
  int dtype, B, N, E, heads;
  float ln_eps, attn_scale;
} rgbnm_vit_cfg;

typedef struct rgbnm_block_params {      /* TransformerEncoderBlock, plainvit.py:493-529 */
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float *bqkv_perm, *bproj, *b1, *b2;
  const void *wqkv, *wqkv_t, *wproj, *wproj_t, *w1, *w1_t, *w2, *w2_t;
} rgbnm_block_params;

typedef struct rgbnm_block_acts {        /* saved for backward; caller-owned */
  void* x_in;   /* [M,E]   block input (residual stream)   */
  void* xn1;    /* [M,E]   LN1 output                      */
  float *mean1, *rstd1;
  void* qkv;    /* [M,3I]  q|k|v                           */
  float* lse;   /* [B*heads*N]                             */
  void* attn;   /* [M,I]   merged heads, pre-projection    */
  void* x_mid;  /* [M,E]   after attention residual        */
  void* xn2;    /* [M,E]                                   */
  float *mean2, *rstd2;
  void* u;      /* [M,4E]  gelu'(fc1 pre-activation)       */
  void* gl;     /* [M,4E]  gelu(u)                         */
  void* x_out;  /* [M,E]                                   */
} rgbnm_block_acts;

typedef struct rgbnm_block_grads {       /* fp32, reference layouts */
  float *dln1_g, *dln1_b, *dln2_g, *dln2_b;
  float *dwqkv, *dbqkv, *dwproj, *dbproj, *dw1, *db1, *dw2, *db2;
} rgbnm_block_grads;

typedef struct rgbnm_block_scratch {     /* backward temporaries, caller-owned, reusable across blocks */
  void* du;      /* [M,4E] */
  void* dxn;     /* [M,E]  */
  void* dx_mid;  /* [M,E]  */
  void* dattn;   /* [M,I]  */
  void* dqkv;    /* [M,3I] */
  void* ws;      /* generic workspace */
  size_t ws_bytes;
} rgbnm_block_scratch;

size_t rgbnm_vit_workspace(const rgbnm_vit_cfg* cfg);                      /* heads of up to 1024 classes */
size_t rgbnm_vit_workspace_ex(const rgbnm_vit_cfg* cfg, int n_classes);    /* any head width (n_classes % 8 == 0) */
int rgbnm_vit_block_fwd(const rgbnm_vit_cfg* cfg, const rgbnm_block_params* p, const rgbnm_block_acts* a,
                        void* stream);
/* Chained forward over consecutive blocks: when rgbnm_vit_ln_chain(cfg) is 1 the epilogue of this block's fc2 can
 * also produce the NEXT block's LN1 (next_a->xn1 / mean1 / rstd1 from next_p->ln1_g / ln1_b; requires
 * next_a->x_in == a->x_out), and flags bit 0 tells a block that its own LN1 was produced that way and must be
 * skipped.  next_p / next_a may be NULL.  rgbnm_vit_block_fwd(cfg, p, a, s) == rgbnm_vit_block_fwd_chain(cfg, p, a, 0,
 * NULL, NULL, s).  (LayerNorm stays reference arithmetic; only the launch and one read of x are saved.) */
int rgbnm_vit_ln_chain(const rgbnm_vit_cfg* cfg);
int rgbnm_vit_block_fwd_chain(const rgbnm_vit_cfg* cfg, const rgbnm_block_params* p, const rgbnm_block_acts* a,
                              int flags, const rgbnm_block_params* next_p, const rgbnm_block_acts* next_a,
                              void* stream);
/* dy = grad wrt x_out, dx = grad wrt x_in (dx may alias dy). */
int rgbnm_vit_block_bwd(const rgbnm_vit_cfg* cfg, const rgbnm_block_params* p, const rgbnm_block_acts* a,
                        const rgbnm_block_grads* g, const rgbnm_block_scratch* s, const void* dy, void* dx,
                        void* stream);

/* ---- The whole encoder forward as ONE launch, one workgroup per image (csrc/vit_chain.hip) --------------------------------
 * Reference: the `depth` TransformerEncoderBlocks of models/plainvit.py:493-529 applied in sequence (:601-611).  bf16, E = 192,
 * 3 heads, 196 tokens (JPEG-Ti); needs rgbnm_gelu_table_init on the device.  The residual stream, LayerNorm outputs, q and the
 * attention output stay in registers, K / V of one head in LDS; every tensor the backward reads (rgbnm_block_acts) is written
 * exactly as rgbnm_vit_block_fwd writes it, so rgbnm_vit_block_bwd runs unchanged behind it.
 * rgbnm_chain_block: one block's parameters and outputs (device pointers); `blocks` is a HOST array of `depth` (<= 12) of them,
 *   copied into the kernel's argument segment (no upload: the call may be captured into a HIP graph).  x_in of block 0 is the
 *   x0 argument; block i's x_out is block i + 1's input.
 * wimg: the block's "chain image" -- its four weight matrices as they lie in LDS (rows permuted, 16-byte chunks swizzled,
 *   consumption order), rgbnm_chain_image_elems() bf16 elements, written per step by rgbnm_chain_gather(src = operand shadows,
 *   idx = constant int32 table built on the host: rgb-no-more_amd/chain.py documents the layout).
 * Returns RGBNM_OK, 1 when the configuration is not eligible (the caller runs the blocks one by one), or a negative error. */
typedef struct rgbnm_chain_block {
  const void* wimg;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *bqkv_perm, *bproj, *b1, *b2;
  void* xn1; float *mean1, *rstd1;
  void* qkv; float* lse; void* attn; void* x_mid; void* xn2; float *mean2, *rstd2;
  void* u; void* gl; void* x_out;
} rgbnm_chain_block;
size_t rgbnm_chain_block_bytes(void);          /* sizeof(rgbnm_chain_block) as the library was built */
long long rgbnm_chain_image_elems(void);       /* bf16 elements of one block's chain image */
int rgbnm_chain_gather(const void* src, const int* idx, void* dst, long long n, void* stream);   /* dst[i] = src[idx[i]], bf16, n % 8 == 0 */
int rgbnm_vit_chain_fwd(const rgbnm_vit_cfg* cfg, const rgbnm_chain_block* blocks, int depth, const void* x0, void* stream);

/* ---- The data path of the whole encoder BACKWARD as ONE launch, one workgroup per image (csrc/vit_chain_bwd.hip) -----------
 * Reference: the backward of the same `depth` blocks as autograd runs it (models/plainvit.py:493-529).  For every block, last to
 * first: du = (dy . W2) * gelu'(u), d(x_mid) = dy + LN2'(du . W1), d(attention output) = d(x_mid) . Wproj, attention backward,
 * dx = d(x_mid) + LN1'(d(qkv) . Wqkv) -- the kernels rgbnm_vit_block_bwd launches one by one, same arithmetic, same bits.  It
 * leaves du / dx_mid / dqkv / dx of EVERY block in the caller's buffers (operands of the weight-gradient GEMMs) and the
 * per-image partial sums of the LayerNorm parameter gradients in part2 / part1 ([B][2][192] each: d(gamma) | d(beta)); the caller
 * then runs rgbnm_vit_block_bwd_dw per block (any order), which launches the four weight-gradient GEMMs and submits the
 * reductions exactly as rgbnm_vit_block_bwd does.  blocks: HOST array of `depth` (<= 12) rgbnm_chain_bwd_block (copied into the
 * kernel's argument segment); blk[i].dy must be blk[i + 1].dx for i < depth - 1.  wimg: the backward chain image (rgb-no-more_amd/chain.py, from the TRANSPOSED operand
 * shadows by rgbnm_chain_gather).  dattn: [B * 196, 192] scratch.  Returns RGBNM_OK, 1 = not eligible, negative = error. */
typedef struct rgbnm_chain_bwd_block {
  const void* wimg;
  const float *ln1_g, *ln2_g;
  const void* x_in; const float *mean1, *rstd1;
  const void* qkv; const float* lse; const void* attn; const void* x_mid; const float *mean2, *rstd2; const void* u;
  const void* dy;
  void* du; void* dx_mid; void* dqkv; void* dx;
  float *part2, *part1;
} rgbnm_chain_bwd_block;
size_t rgbnm_chain_bwd_block_bytes(void);
int rgbnm_vit_chain_bwd(const rgbnm_vit_cfg* cfg, const rgbnm_chain_bwd_block* blocks, int depth, void* dattn, void* stream);
/* The weight / bias / LayerNorm-parameter gradients of one block from what rgbnm_vit_chain_bwd left behind: the dW part of
 * rgbnm_vit_block_bwd (scratch->du / dx_mid / dqkv as filled by the chain kernel; dy = the block's output gradient). */
int rgbnm_vit_block_bwd_dw(const rgbnm_vit_cfg* cfg, const rgbnm_block_acts* a, const rgbnm_block_grads* g,
                           const rgbnm_block_scratch* s, const void* dy, const float* part2, const float* part1, void* stream);
/* The same for n <= 12 blocks in ONE grouped weight-gradient launch (arrays of n pointers; every block with its own workspace
 * scratch[i]->ws): the 4 n GEMMs share the 256 workgroups, so the token axis is split 256 / (21 n) ways instead of 12 -- all twelve
 * blocks of JPEG-Ti: no split, no fp32 partial sums to write and re-read. */
int rgbnm_vit_blocks_bwd_dw(const rgbnm_vit_cfg* cfg, int n, const rgbnm_block_acts* const* a, const rgbnm_block_grads* const* g,
                            const rgbnm_block_scratch* const* s, const void* const* dy, const float* const* part2,
                            const float* const* part1, void* stream);
/* ... and the patch embedding's weight gradient (what rgbnm_patch_embed_bwd computes: pe_dw [E,384] / pe_db [E] = pe_dx0^T . pe_feat
 * over the same token axis; pe_dx0 = the gradient rgbnm_vit_chain_bwd left for block 0's input) as one more job of the same
 * grouped launch -- 252 + 4 = 256 output tiles at JPEG-Ti's B = 256: still no token split.  pe_dx0 NULL = the call above. */
int rgbnm_vit_blocks_bwd_dw_pe(const rgbnm_vit_cfg* cfg, int n, const rgbnm_block_acts* const* a, const rgbnm_block_grads* const* g,
                               const rgbnm_block_scratch* const* s, const void* const* dy, const float* const* part2,
                               const float* const* part1, const void* pe_dx0, const void* pe_feat, float* pe_dw, float* pe_db,
                               void* pe_ws, size_t pe_ws_bytes, void* stream);

/* Table GELU of the bf16 path (csrc/mlp_fused.hip): in bf16 mode the pre-activation is rounded to bf16 before the GELU
 * (models/plainvit.py:487-488 under autocast), so gelu / gelu' are functions of 16 bits.  _init is a SET-UP call (it
 * synchronises `stream`; not capturable; idempotent per device): it builds the table of the library's own GELU arithmetic for
 * all 65536 inputs plus a compact LDS image of it, and reads the window back.  On devices where it has been called and the
 * image fits, the fused FeedForwardBlock forward looks gelu / gelu' up instead of computing them (option gelu_table) -- the
 * same bits as the arithmetic for every finite input except those whose u or u / 2 is a bf16 denormal (|u| < 2.36e-38), which
 * give a denormal of the right sign instead of u / 2.  Never called: the arithmetic runs.
 * _info (test / diagnostics, synchronises): win16 = {valid, A0, P1, N1, image dwords, ...}, full65536 (may be NULL) =
 * gelu(u) | gelu'(u) << 16 per bf16 bit pattern u. */
int rgbnm_gelu_table_init(void* stream);
int rgbnm_gelu_table_info(int* win16, unsigned* full65536);

/* Held gradient reductions.  Every backward entry point of this header (rgbnm_vit_block_bwd, rgbnm_head_bwd,
 * rgbnm_patch_embed_bwd, rgbnm_gemm_tn, rgbnm_layernorm_bwd, ...) finishes with a small reduction of its split partial
 * sums (reference: autograd accumulates straight into .grad, models/plainvit.py runs under torch.autograd).  A caller that
 * reads no gradient before the END of the backward pass -- one GPU, or one all-reduce after the pass -- may bracket the pass:
 *   rgbnm_reduce_hold_begin();  ... backward calls, each with ITS OWN workspace region ...
 *   rgbnm_reduce_hold_end(table, table_host, rgbnm_reduce_hold_table_bytes(), stream);
 * and all reductions issued in between (on this host thread) run as ONE launch at _end, with the same summation order (same
 * bits).  `table` is a device buffer and `table_host` a HOST buffer of the same size, both owned by the caller, allocated
 * together (table_host zero-filled) and freed together, untouched between steps: table_host records what the device table
 * holds, and the job table is uploaded only when it differs from that record (so a graph capture of a steady-state pass
 * contains no copy; a pass whose table changed while the stream is capturing is captured WITH its upload when table_host is
 * page-locked -- the copy node reads table_host at every replay, so it must stay untouched while the graph lives -- and returns
 * RGBNM_EINVAL when it is pageable).  The partial sums live in
 * the workspaces until _end: regions must not be shared between the calls of one bracket.  One bracket per host thread:
 * _begin inside an open bracket returns RGBNM_EINVAL.  rgbnm_reduce_hold_cancel() leaves the mode without running anything
 * (error paths). */
int rgbnm_reduce_hold_begin(void);
int rgbnm_reduce_hold_end(void* table, void* table_host, size_t table_bytes, void* stream);
void rgbnm_reduce_hold_cancel(void);
size_t rgbnm_reduce_hold_table_bytes(void);

/* PatchEmbedding_DCT_Group (plainvit.py:157-218): feat = subblock(y,cbcr); x0 = feat.Wpe^T + b + sincos */
int rgbnm_patch_embed_fwd(const rgbnm_vit_cfg* cfg, int in_dtype, const void* y, const void* cbcr,
                          const float* conv16, const void* wpe, const float* bpe, const float* pos, void* feat,
                          void* x0, int Hb, int Wb, void* stream);
/* ... with the batch mixup applied on the way in (rgbnm_subblock_embed_mix; lam_dev NULL = rgbnm_patch_embed_fwd) */
int rgbnm_patch_embed_fwd_mix(const rgbnm_vit_cfg* cfg, int in_dtype, const void* y, const void* cbcr, const float* lam_dev,
                              const float* conv16, const void* wpe, const float* bpe, const float* pos, void* feat, void* x0,
                              int Hb, int Wb, void* stream);
int rgbnm_patch_embed_bwd(const rgbnm_vit_cfg* cfg, const void* dx0, const void* feat, float* dwpe, float* dbpe,
                          void* ws, size_t ws_bytes, void* stream);

typedef struct rgbnm_head_params {       /* ClassificationHead, plainvit.py:542-557 */
  const float *ln_g, *ln_b, *b1, *b2;
  const void *w1, *w1_t, *w2, *w2_t;
  int n_classes;
  int _pad;
} rgbnm_head_params;
typedef struct rgbnm_head_acts {
  void* x;        /* [M,E] encoder output */
  float *mean, *rstd;
  void* pooled;   /* [B,E] */
  void* h1;       /* [B,E] tanh output */
  float* logits;  /* [B,C] fp32 */
} rgbnm_head_acts;
typedef struct rgbnm_head_grads {
  float *dln_g, *dln_b, *dw1, *db1, *dw2, *db2;
} rgbnm_head_grads;
int rgbnm_head_fwd(const rgbnm_vit_cfg* cfg, const rgbnm_head_params* p, const rgbnm_head_acts* a, void* stream);
/* dlogits [B,C] in cfg->dtype; da, dpooled: [B,E] scratch; dx [M,E] out. */
/* Workspace that lets rgbnm_head_bwd keep its three sets of split partial sums side by side (needed when the call sits inside
 * a rgbnm_reduce_hold_begin / _end bracket; with less -- rgbnm_vit_workspace_ex -- they re-use one region, outside a bracket). */
size_t rgbnm_head_bwd_workspace(const rgbnm_vit_cfg* cfg, int n_classes);
int rgbnm_head_bwd(const rgbnm_vit_cfg* cfg, const rgbnm_head_params* p, const rgbnm_head_acts* a,
                   const rgbnm_head_grads* g, const void* dlogits, void* da, void* dpooled, void* dx, void* ws,
                   size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SwinV2 DCT (models/swinv2.py; BASELINE config 5) - the parts that are not plain Linears (those use rgbnm_gemm_nt/tn).
 * ------------------------------------------------------------------------------------------- */
/* PatchEmbedding_DCT_Group with patch 4 (swinv2.py:505-576, plainvit.py:50-88): every 8x8 block is decomposed,
 * X' = A^T X A with convY = conversion_matrix(4,2) / convC = conversion_matrix(2,4) (8x8 fp32 each), tokens on the
 * 2Hb x 2Wb grid, feat [B, 2Hb*2Wb, 24] = Y 4x4 | Cb 2x2 | Cr 2x2 (einops split '(p1 pdh)(p2 pdw)', coefficient-major). */
int rgbnm_swin_embed(int in_dtype, int out_dtype, const void* y, const void* cbcr, const float* convY, const float* convC,
                     void* feat, int B, int Hb, int Wb, void* stream);
/* LayerNorm of any width E % 4 == 0, E <= 768:  y = [res +] [sample_scale[row / rows_per_sample] *] LN(x)
 * (res-post-norm + DropPath scale, swinv2.py:302-307).  Backward: dx = LN'(scale * dy), dgamma / dbeta fp32. */
int rgbnm_ln_generic_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* res,
                         const float* sample_scale, int rows_per_sample, void* y, float* mean, float* rstd, int M, int E,
                         float eps, void* stream);
size_t rgbnm_ln_generic_bwd_workspace(int M, int E);
int rgbnm_ln_generic_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean,
                         const float* rstd, const float* sample_scale, int rows_per_sample, void* dx, float* dgamma,
                         float* dbeta, int M, int E, int accumulate, void* workspace, size_t workspace_bytes,
                         void* stream);
/* WindowAttention + cyclic shift + window partition / reverse (swinv2.py:143-182, 273-300) on token-major tensors:
 * qkv [B, res*res, 3C] (q | k | v, heads of 32), bias [heads, 64, 64] fp32 (= 16 sigmoid(cpb_mlp(table))[index]),
 * scale [heads] fp32 (= exp(min(logit_scale, ln 100))), shift in {0, 4}: window 8x8, cosine attention, shift mask -100.
 * out [B, res*res, C]; lse [B * nW * heads * 64] saved for backward.  Backward writes dbias (per-wave partials in the
 * workspace, deterministic reduction), one partial d(scale) per (window, head) into dscale_part [B * nW * heads] and -- when
 * dscale is not NULL -- their sum into dscale [heads] (a job of the same batched reduction). */
int rgbnm_window_attention_fwd(int dtype, const void* qkv, const float* bias, const float* scale, void* out, float* lse,
                               int B, int res, int C, int heads, int shift, void* stream);
size_t rgbnm_window_attention_bwd_workspace(int B, int res, int heads);
int rgbnm_window_attention_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* bias,
                               float* dscale /* [heads] or NULL */, const float* scale, const float* lse, void* dqkv,
                               float* dbias, float* dscale_part, int B, int res, int C, int heads, int shift, void* workspace,
                               size_t workspace_bytes, void* stream);
/* Continuous position bias + logit scale of EVERY WindowAttention of the model (swinv2.py:158-168: cpb_mlp on the 15 x 15
 * relative_coords_table, gathered by relative_position_index, 16 sigmoid; exp of the clamped logit_scale) in two launches per
 * direction.  blocks: HOST array (copied into the kernels' arguments; any count, heads <= 24).  Per block: w1 [512,2], b1 [512],
 * w2 [heads,512], ls [heads] fp32 parameters; forward writes bias [heads,64,64] and scale [heads]; backward reads dbias
 * [heads,64,64] and dscale [heads] and writes dw1, db1, dw2, dls.  coords [225,2] fp32; index [4096] int32 (= relative_position_index);
 * inv_index [225,64] int32: the positions that read each table entry, ascending, padded with 4096 (fixed summation order: run-to-run
 * identical bits).  table / dtable: caller-owned fp32 scratch of rgbnm_swin_cpb_table_elems(nblocks) elements; `table` carries the
 * forward's MLP outputs to the backward. */
typedef struct rgbnm_cpb_block {
  const float *w1, *b1, *w2, *ls;
  float *bias, *scale;
  const float *dbias, *dscale;
  float *dw1, *db1, *dw2, *dls;
  int heads, _pad;
} rgbnm_cpb_block;
size_t rgbnm_swin_cpb_table_elems(int nblocks);
int rgbnm_swin_cpb_fwd(const rgbnm_cpb_block* blocks, int nblocks, const float* coords, const int* index, float* table, void* stream);
int rgbnm_swin_cpb_bwd(const rgbnm_cpb_block* blocks, int nblocks, const float* coords, const int* inv_index, const float* table,
                       float* dtable, void* stream);
/* PatchMerging's concat (swinv2.py:357-362): [B, res*res, C] -> [B, (res/2)^2, 4C] (inverse != 0: the reverse copy). */
int rgbnm_merge_gather(int dtype, const void* in, void* out, int B, int res, int C, int inverse, void* stream);
/* mean over tokens [B,N,C] -> [B,C] (backward != 0: [B,C] -> [B,N,C], dy / N). */
int rgbnm_token_mean(int dtype, const void* in, void* out, int B, int N, int C, int backward, void* stream);


/* ---- calibration micro-kernels (measurement only, SURVEY.md section 8d: attainable peaks on the box) ----------------
 * rgbnm_calib_mfma_bf16: `workgroups` x 4 waves each issue iters x 4 independent 32x32x16 bf16 MFMAs (32768 flop each),
 * no memory traffic.  rgbnm_calib_stream: 16 B per lane grid-stride; mode 0 copy, 1 read-only, 2 write-only.
 * `sink` is a 4-byte device word that is never actually written. */
int rgbnm_calib_mfma_bf16(int workgroups, int iters, float* sink, void* stream);
int rgbnm_calib_stream(const void* src, void* dst, size_t bytes, int mode, int workgroups, void* sink, void* stream);
/* One wave issues 16 back-to-back 1 KB stores (mode 0) / loads (mode 1), 8 rows x 128 B at row stride ld bytes:
 * out[(wg * waves + w) * 2 + {0, 1}] = cycles to issue them / cycles until they have all completed. */
int rgbnm_calib_vmem_issue(int mode, int workgroups, int waves, void* buf, size_t wave_bytes, int ld,
                           unsigned long long* out, void* stream);
/* Every workgroup (`waves` waves) re-reads its own L2-resident slice of `slice_bytes` (a multiple of waves * 8 KB) `iters`
 * times: mode 0 plain 16-byte loads, mode 1 LDS-DMA.  Prices L2 -> CU traffic (weight re-streaming of the row-panel kernels). */
int rgbnm_calib_l2(const void* buf, size_t slice_bytes, int iters, int mode, int workgroups, int waves, void* sink,
                   void* stream);
/* What two waves of one SIMD share: wave w of a `waves`-wave workgroup (w and w + 4 share a SIMD) runs role_dev[w] for `iters`
 * rounds -- 0 idle, 1: 8 MFMAs (32x32x16 bf16, independent accumulators), 2: 64 v_fma_f32, 3: 32 v_pk_fma_f32, 4: 16 v_exp_f32 +
 * 16 v_rcp_f32, 5: 32 v_cvt_pk_bf16_f32, 6: 8 MFMAs and 64 v_fma_f32 in ONE stream; out[wg * waves + w] = cycles of the loop. */
int rgbnm_calib_pipes(const int* role_dev, int waves, int iters, int workgroups, unsigned long long* out, float* sink,
                      void* stream);

/* Stand-in for a collective's channels: `workgroups` x 256 threads, each holding `lds_bytes` of LDS, stay resident for `ticks`
 * s_memtime ticks (or until *stop != 0; stop may be NULL), re-reading their 4 KB-multiple slice of buf meanwhile (mode 1) or sleeping
 * (mode 0).  Measurement / test perturber only (tools/cu_steal_probe.py, tests/test_chain_soak.py). */
int rgbnm_calib_occupy(const void* buf, size_t slice_bytes, int workgroups, int lds_bytes, long long ticks, int mode,
                       const int* stop, void* sink, void* stream);
/* the same with a residency log (device memory, 3 x workgroups 64-bit words): [3 wg] = s_memrealtime (100 MHz) at the workgroup's
 * start, [3 wg + 1] at its end, [3 wg + 2] = (XCC id << 32) | HW_ID */
int rgbnm_calib_occupy_log(const void* buf, size_t slice_bytes, int workgroups, int lds_bytes, long long ticks, int mode,
                           const int* stop, void* sink, unsigned long long* log, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RGBNM_H */
