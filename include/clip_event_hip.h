/* clip_event_hip.h -- C ABI of libclip_event_hip.so (gfx950 / MI355X).
 *
 * The reference (limanling/clip-event) has no FFI: its hot path is a Python module API
 * (src/clip-event/model_clip.py) that reaches the device through torch.nn ops.  This
 * library is the device side of the drop-in: one entry point per op call site of that
 * path, each citing the reference lines it replaces.  The Python mirror of the reference
 * API (clip_event_amd.model.CLIP, CriterionContrastive, CriterionAlignment) calls these
 * through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions:
 *   - device pointers are borrowed, never freed or retained; workspaces are caller-provided;
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*); no entry
 *     point synchronises the device or allocates device memory;
 *   - return 0 on success, a negative errno-style code on failure; ce_last_error() returns
 *     the message of the calling thread's last failure;
 *   - "bf16" = raw bfloat16 bits (uint16_t); "f32" = float; row-major, leading dimensions in
 *     ELEMENTS;
 *   - activations are token-major [rows = batch*tokens, width]; master weights / gradients are fp32, GEMM
 *     operands bf16, accumulation fp32; the residual stream is fp32 or IEEE fp16 (ce_tower_desc.stream16).
 */
#ifndef CLIP_EVENT_HIP_H
#define CLIP_EVENT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ce_last_error(void);
int ce_version(void);

/* ---- GEMM epilogues (ce_gemm_nt) ---- */
enum {
    CE_EPI_BF16 = 0,          /* out(bf16) = acc                                            */
    CE_EPI_F32 = 1,           /* out(f32)  = acc                                            */
    CE_EPI_BIAS_BF16 = 2,     /* out(bf16) = acc + bias[n]                                  */
    CE_EPI_BIAS_F32 = 3,      /* out(f32)  = acc + bias[n]                                  */
    CE_EPI_BIAS_RESID_F32 = 4,/* out(f32)  = resid(f32) + acc + bias[n]     (x + proj(..))  */
    CE_EPI_BIAS_GELU = 5,     /* a = acc + bias; out(bf16) = dQuickGELU(a) = s + 1.702 a s (1 - s), s = sigmoid(1.702 a): the
                               * factor the backward needs, from the fp32 a; out2(bf16) = QuickGELU(a) = a s */
    CE_EPI_GELUGRAD_BF16 = 6, /* out(bf16) = acc * aux(bf16), aux = the derivative BIAS_GELU saved; out2 (nullable) is reused as
                               * a float[N] that receives += the column sums of out (bias gradient)      */
    CE_EPI_BIAS_RESID_F16 = 7 /* out(f16)  = resid(f16) + acc + bias[n]: the residual add on an fp16 stream (CE_T_F16;
                               * resid / out point at IEEE half data, ldr / ldo in elements; stores saturate at 65504) */
};

/* Element types of residual-stream operands (ce_layernorm_*_t, ce_token_embed_t, ce_cast_t, ce_tower_desc.stream16).
 * The reference keeps the stream in fp32 (model_clip.py:190-200).  It is read and written four times per block and
 * direction and never enters a matrix unit, so its storage format is a bandwidth choice: IEEE fp16 (11 significand bits;
 * bf16 has 8) leaves the gradient noise floor where the bf16 GEMM operands put it (tests/stream16_emulation.py: median
 * per-parameter error x 1.04 on BASELINE config 1; a bf16 stream: x 2.4).  fp16 stores saturate at +-65504, and a
 * GRADIENT stream holds gradient * scale, a power of two chosen per backward pass from the largest element of the
 * gradient that enters the tower (ce_grad_scale; gradients of a mean loss sit below fp16's normal range). */
#ifndef CE_T_F32
#define CE_T_F32 0
#define CE_T_BF16 1
#define CE_T_F16 2
#endif

/* C[M,N] = A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulate, fused epilogue.
 * Replaces nn.Linear / MHA in_proj,out_proj / Conv2d(k=s=patch) forward and their
 * input-gradient GEMMs (model_clip.py:175-180, :188, :219, :230, :329, :415). */
int ce_gemm_nt(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int epilogue,
               const float* bias, const void* resid, long ldr, void* out, long ldo, void* out2, long ldo2,
               const void* aux, long ldaux, void* stream);

/* out[Nn,Kk] (f32) += P[M,Nn]^T . Q[M,Kk]  (weight gradients; fp32 atomic accumulation, so the
 * caller zeroes `out` once per step).  splits<=0 picks the M split that fills the chip. */
int ce_gemm_tn(const void* P, long ldp, const void* Q, long ldq, int M, int Nn, int Kk, float* out, long ldo,
               int splits, void* stream);
/* same, plus the bias gradient bias_grad[n] (f32 [Nn], nullable) += sum_m P[m,n] */
int ce_gemm_tn_bias(const void* P, long ldp, const void* Q, long ldq, int M, int Nn, int Kk, float* out, long ldo,
                    float* bias_grad, int splits, void* stream);
/* 1..CE_TN_MAX_GROUP weight-gradient problems sharing M in one launch (the four Linear layers of a residual block, or of
 * several consecutive blocks: with about one resident round of tiles the launch needs no M split, and an unsplit tile is
 * added by plain read-modify-write instead of float atomics) */
#define CE_TN_MAX_GROUP 36
int ce_gemm_tn_grouped(int count, const void* const* P, const long* ldp, const void* const* Q, const long* ldq, int M,
                       const int* Nn, const int* Kk, float* const* out, const long* ldo, int splits, void* stream);
/* the same with overwrite != 0: out = P^T Q (whatever out held is discarded; the caller need not zero it).  A step's FIRST
 * contribution to a weight gradient: an unsplit tile then stores its accumulators instead of adding them with float
 * atomics (memory-side, 1.3 TB/s chip-wide), and the step's zero-fill of these tensors (85 % of the gradient buffer) goes away.
 * The outputs of one call must not overlap. */
int ce_gemm_tn_grouped_ex(int count, const void* const* P, const long* ldp, const void* const* Q, const long* ldq, int M,
                          const int* Nn, const int* Kk, float* const* out, const long* ldo, int splits, int overwrite,
                          void* stream);

/* ---- fp8 (OCP e4m3) operand path (gemm_fp8.hip; BASELINE config 5, tensors as convert_weights model_clip.py:554-575) ----
 * q[r,:] (e4m3 bytes) = x[r,:] (bf16) * 2^e_r with the power of two that puts the row's amax into (224, 448],
 * scale[r] = 2^-e_r (1 for an all-zero row); exact scaling, so the bytes are reproducible anywhere; K <= 4096 */
int ce_quant_rows_fp8(const void* x, long ldx, void* q, long ldq, float* scale, int M, int K, void* stream);
/* the same quantisation for a table of matrices in ONE launch (the per-step requantisation of every block weight and
 * its transpose: 288 launches of a few microseconds each otherwise).  jobs live in device memory; tile_start = first
 * 4-row group of the job in the launch's grid; K <= 4096. */
typedef struct ce_quant_job {
    const void* src; void* dst; float* scale;
    long lds_; long ldd;         /* leading dimensions: src in bf16 elements, dst in bytes */
    int rows, cols;
    int group_start, pad_;
} ce_quant_job;
int ce_quant_rows_fp8_multi(const ce_quant_job* jobs_device, int njobs, int total_groups, void* stream);
/* C[m,n] = sa[m] * sb[n] * sum_k A8[m,k] * B8[n,k] (e4m3 operands, one fp32 scale per row of each, fp32 accumulate on
 * v_mfma_scale_f32_16x16x128_f8f6f4) with the fused epilogues of ce_gemm_nt (BF16, BIAS_BF16, BIAS_RESID_F32, BIAS_GELU,
 * GELUGRAD_BF16).  K % 128 == 0; lda/ldb in bytes (= elements). */
int ce_gemm_nt_fp8(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, int M, int N,
                   int K, int epilogue, const float* bias, const void* resid, long ldr, void* out, long ldo, void* out2,
                   long ldo2, const void* aux, long ldaux, void* stream);
/* MX form of the same path (round 3): one E8M0 scale byte per row and 32 contraction values -- the block the scaled MFMA
 * applies natively, so quantisation can be done by whoever PRODUCES the operand, block by block, with no row-wide pass.
 * q [M,K] e4m3 bytes, scale8 [M, K/32] bytes (lds_ = bytes per scale row); the block's amax is mapped into (224, 448];
 * K % 32 == 0 (ce_gemm_nt_mx8: K % 128 == 0, scale rows K/32 bytes apart and 4-byte aligned). */
int ce_quant_mx_fp8(const void* x, long ldx, void* q, long ldq, void* scale8, long lds_, int M, int K, void* stream);
/* job table as ce_quant_rows_fp8_multi; job.scale points at the [rows, cols/32] byte array of the job */
int ce_quant_mx_fp8_multi(const ce_quant_job* jobs_device, int njobs, int total_groups, void* stream);
int ce_gemm_nt_mx8(const void* A8, long lda, const void* sa8, const void* B8, long ldb, const void* sb8, int M, int N, int K,
                   int epilogue, const float* bias, const void* resid, long ldr, void* out, long ldo, void* out2, long ldo2,
                   const void* aux, long ldaux, void* stream);
/* y = LayerNorm(x[rows[r]] or x[r]) over D columns, fp32 statistics (eps inside the sqrt).
 * y is bf16 (out_f32=0: the next GEMM's operand) or fp32 (ln_pre: the residual stream).
 * Writes mean/rstd [M] for the backward.  Replaces LayerNorm.forward, model_clip.py:157-163. */
int ce_layernorm_fwd(const float* x, long ldx, const int* rows, const float* w, const float* b, void* y, long ldy,
                     int out_f32, float* mean, float* rstd, int M, int D, float eps, void* stream);
/* the same with typed operands: x_type CE_T_F32 / CE_T_F16 (the stream), y_type CE_T_BF16 (GEMM operand) / CE_T_F32 /
 * CE_T_F16 (ln_pre writing the stream) */
int ce_layernorm_fwd_t(const void* x, int x_type, long ldx, const int* rows, const float* w, const float* b, void* y,
                       int y_type, long ldy, float* mean, float* rstd, int M, int D, float eps, void* stream);

/* the same plus -- q8 != NULL, y bf16 -- an e4m3 copy of the output with one fp32 scale per row, written from the registers
 * that hold the row: q8[r,:] and qscale[r] equal ce_quant_rows_fp8 of y[r,:] bit for bit (fp8 operand path: the quantisation
 * pass of the GEMM that consumes y disappears) */
int ce_layernorm_fwd_q8(const void* x, int x_type, long ldx, const int* rows, const float* w, const float* b, void* y,
                        int y_type, long ldy, float* mean, float* rstd, int M, int D, float eps, void* q8, long ldq,
                        float* qscale, void* stream);

/* dx_out[dst] = (dx_in ? dx_in[dst] : 0) + dLN(dy[r]); dst = rows ? rows[r] : r; dxb = bf16 copy
 * (nullable); dw/db (f32 [D]) accumulate atomically (caller zeroes once per step); dxsum (f32 [D],
 * nullable) += column sums of dx_out = the bias gradient of the Linear that produced this stream. */
int ce_layernorm_bwd(const void* dy, long lddy, int dy_f32, const float* x, long ldx, const int* rows,
                     const float* mean, const float* rstd, const float* w, const float* dx_in, float* dx_out,
                     long lddx, void* dxb, long lddxb, float* dw, float* db, float* dxsum, int M, int D,
                     void* stream);
/* the same with typed operands: dy CE_T_BF16 / CE_T_F32 / CE_T_F16 (a gradient stream), x CE_T_F32 / CE_T_F16, dx_in and
 * dx_out CE_T_F32 / CE_T_F16.  An fp16 gradient operand holds gradient * *gscale (gscale: DEVICE pointer to the scale of
 * this backward pass, ce_grad_scale; it is read back / stored in those units; may be NULL when no operand is fp16);
 * dxb, dw, db, dxsum are always in true units. */
int ce_layernorm_bwd_t(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx, const int* rows,
                       const float* mean, const float* rstd, const float* w, const void* dx_in, int dxin_type,
                       void* dx_out, int dx_type, long lddx, void* dxb, long lddxb, float* dw, float* db, float* dxsum,
                       const float* gscale, int M, int D, void* stream);

/* the same plus -- q8 != NULL -- the e4m3 copy (+ per-row scales, ce_quant_rows_fp8's rule) of dxb, indexed like dxb */
int ce_layernorm_bwd_q8(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx, const int* rows,
                        const float* mean, const float* rstd, const float* w, const void* dx_in, int dxin_type,
                        void* dx_out, int dx_type, long lddx, void* dxb, long lddxb, float* dw, float* db, float* dxsum,
                        const float* gscale, int M, int D, void* q8, long ldq, float* qscale, void* stream);

/* The same backward WITHOUT atomics on the parameter gradients: the launch's workgroups (ce_layernorm_bwd_blocks(M, D) of them) each
 * write their partial column sums -- d gamma | d beta | (want_dxsum) column sums of dx -- as one row of partials[blocks][3][D], and
 * ce_layernorm_fold adds the rows of any number of such launches into their destinations in ONE launch (dw += ..., db += ...,
 * dxsum += ...).  Why: every workgroup of a launch adds into the same 2-3 rows, float atomics on one address serialise at the memory
 * side, and that phase is 3.7 us of a 23 us launch; the tower backward (48 LayerNorms per ViT-B/32 tower pass) folds once at its end. */
int ce_layernorm_bwd_blocks(int M, int D);
int ce_layernorm_bwd_partials(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx, const int* rows,
                              const float* mean, const float* rstd, const float* w, const void* dx_in, int dxin_type,
                              void* dx_out, int dx_type, long lddx, void* dxb, long lddxb, int want_dxsum,
                              const float* gscale, int M, int D, void* q8, long ldq, float* qscale, float* partials,
                              void* stream);
#define CE_LN_FOLD_MAX 64
typedef struct ce_ln_fold_job {
    const float* partials; /* device: [blocks][3][D] */
    float *dw, *db, *dxsum; /* device destinations (dxsum nullable) */
    int blocks, D;
} ce_ln_fold_job;
int ce_layernorm_fold(const ce_ln_fold_job* jobs /* HOST array; no two jobs of a call may share a destination */, int njobs, void* stream);

/* Self-attention core on the packed in-projection output qkv[B*L, 3*H*64] (bf16; q | k | v column
 * blocks, head h at columns h*64): o[B*L, H*64] = softmax(q k^T / 8 + causal?) v, lse[B*H*L] (f32)
 * saved for the backward.  L <= 128: one workgroup per (sample, head); longer sequences (ViT-B/16, ViT-L/14): tiled
 * over 64-key blocks with an online softmax.  Replaces the core of nn.MultiheadAttention as called at
 * model_clip.py:188 (mask from model_clip.py:377-384).
 * cu_seqlens (int32 [B+1], nullable): variable-length batch, sample b owns rows cu_seqlens[b] ..
 * cu_seqlens[b+1]-1 (at most L of them); NULL = dense, sample b owns rows b*L .. b*L+L-1.  lse stays [B,H,L]. */
int ce_attention_fwd(const void* qkv, long ld, void* o, long ldo, float* lse, const int* cu_seqlens, int B, int L,
                     int H, int causal, void* stream);
/* bias_grad (f32 [3*H*64], nullable) += column sums of dqkv (the in_proj bias gradient) */
int ce_attention_bwd(const void* qkv, long ld, const void* o, long ldo, const void* dout, long lddo,
                     const float* lse, void* dqkv, long lddq, float* bias_grad, const int* cu_seqlens, int B, int L,
                     int H, int causal, void* stream);

/* ---- input side, bookkeeping (embed.hip) ---- */
/* image f32 [B,3,R,R] -> patch rows bf16 [B*(R/p)^2, k_padded], columns ordered (c, py, px) like the
 * flattened conv1.weight; columns >= 3*p*p are zero.  Conv2d(k=s=p), model_clip.py:219,235. */
int ce_im2col(const float* image, void* patches, int B, int resolution, int patch, int k_padded, void* stream);
/* x0[b,t,:] = (t==0 ? class_embedding : patch_out[b*(T-1)+t-1,:]) + positional_embedding[t,:]
 * (model_clip.py:237-242); the backward emits the bf16 patch-row gradient for the conv wgrad. */
int ce_vision_assemble(const float* patch_out, const float* cls, const float* pos, float* x0, int B, int tokens,
                       int D, void* stream);
int ce_vision_assemble_bwd(const float* dx0, void* dpatch, int B, int tokens, int D, void* stream);
/* x0[r,:] = token_embedding[ids[e],:] + positional_embedding[e % tokens,:], e = src_rows ? src_rows[r] : r
 * (model_clip.py:400-403).  src_rows selects the LIVE tokens of a packed batch: under the causal mask
 * (model_clip.py:377-384) the rows after a caption's EOT can influence neither its feature (taken at the EOT,
 * model_clip.py:415) nor any gradient, so the text tower runs on the rows up to each EOT only.  The backward
 * skips exact-zero gradient elements (they change nothing and would serialise on the padding id). */
int ce_token_embed(const int64_t* ids, const int* src_rows, const float* table, const float* pos, float* x0, long rows,
                   int tokens, int D, int vocab, void* stream);
/* the same writing x0 in element type x0_type (CE_T_F32 / CE_T_F16) */
int ce_token_embed_t(const int64_t* ids, const int* src_rows, const float* table, const float* pos, void* x0, int x0_type,
                     long rows, int tokens, int D, int vocab, void* stream);
int ce_token_embed_bwd(const int64_t* ids, const int* src_rows, const float* dx0, float* dtable, long rows, int D,
                       int vocab, void* stream);
/* packed batch: dpos[t,:] += sum_{b : len_b > t} dx0[cu_seqlens[b] + t, :]   (cu_seqlens int32 [n+1]) */
int ce_pos_embed_bwd_packed(const float* dx0, const int* cu_seqlens, float* dpos, int n, int tokens, int D,
                            void* stream);
/* out[i] (+)= sum_b x[b*slab + i], i < n  (positional / class embedding gradients) */
int ce_batch_reduce(const float* x, float* out, int B, long slab, long n, int accumulate, void* stream);
/* out[n] += sum_m x[m,n]  (bias gradients; atomics) */
int ce_colsum_bf16(const void* x, long ld, float* out, int M, int N, void* stream);
/* fp32 master weight [R,C] -> bf16 copy [R,C] (nullable) and bf16 transposed copy [C,R] (nullable) */
int ce_cast_transpose(const float* w, void* w16, long ld16, void* w16t, long ld16t, int R, int C, void* stream);
int ce_cast_bf16(const float* x, void* y, long n, void* stream);
/* y[i] (dst_type) = x[i] (src_type) * mul, element types CE_T_F32 / CE_T_BF16 / CE_T_F16; n a multiple of 4 */
int ce_cast_t(const void* x, int src_type, void* y, int dst_type, float mul, long n, void* stream);
/* the same with the factor in DEVICE memory: y = x * *scale, or x / *scale with divide != 0 */
int ce_cast_scaled(const void* x, int src_type, void* y, int dst_type, const float* scale, int divide, long n, void* stream);
/* *scale = the power of two s with s * max|x| in (target / 2, target] (1 when x is all zero or not finite): the scale of an
 * fp16 gradient stream whose top-of-tower gradient is x (fp32, n elements).  Stores saturate at 65504, so 65504 / target is
 * the growth the gradient may see on its way down the tower (target 64: 1024x; measured x17-34 in the ViT-B/32 text tower).  scratch: CE_GRAD_SCALE_SCRATCH floats. */
#define CE_GRAD_SCALE_SCRATCH 256
int ce_grad_scale(const float* x, long n, float target, float* scratch, float* scale, void* stream);
/* Saturation telemetry of the fp16 streams.  The reference keeps both streams in fp32 (model_clip.py:190-200) and stops on a
 * non-finite loss (engine.py:79-82); an fp16 store that clamps at +-65504 produces neither an inf nor a NaN, so the clamp is
 * made visible instead: register a DEVICE buffer of two unsigned counters (NULL switches it off; one buffer per process = per
 * GPU).  From then on every LayerNorm launch adds to them, sticky until the caller zeroes the buffer:
 *   [0] forward stream: LayerNorm-forward (wave, lane) slots that read a residual-stream element at the fp16 limit -- every
 *       stream row passes through a LayerNorm forward before anything else reads it, so no clamped store escapes;
 *   [1] gradient stream: LayerNorm-backward slots whose output gradient * scale reached the limit (or was not finite)
 *       before the clamping store.
 * Non-zero means the step computed with a clipped activation / gradient: lower CE_GRAD_TARGET, or run the stream in fp32. */
int ce_stream16_set_counters(unsigned int* device_counters);
/* dst[r,c] += src[r,c] for c < cols (rows with different strides: real columns of a column-padded gradient) */
int ce_add_cols(const float* src, long lds, float* dst, long ldd, int rows, int cols, void* stream);
/* gather / scatter whole rows: dst[dst_rows?dst_rows[i]:i] = src[src_rows?src_rows[i]:i], 16-byte granules */
int ce_copy_rows(const void* src, long src_stride_bytes, const int* src_rows, void* dst, long dst_stride_bytes,
                 const int* dst_rows, int n, int row_bytes, void* stream);
/* dst (M rows) = zeros except dst[dst_rows[i]] = src[i], i < n; dst_rows strictly ascending.  One pass instead of a
 * memset + ce_copy_rows (the pruned last block's backward hands the gradients of the B consumed rows to kernels that take
 * the dense [M, width] layout). */
int ce_scatter_rows_zero(const void* src, long src_stride_bytes, void* dst, long dst_stride_bytes, const int* dst_rows, int n,
                         int M, int row_bytes, void* stream);
/* rows[r] = r*tokens + argmax_t ids[r,t]  (EOT gather index, model_clip.py:415; first maximum) */
int ce_eot_rows(const int64_t* ids, int* rows, long n, int tokens, void* stream);

/* ---- image preprocessing (preprocess.hip) ---- */
/* One entry per OUTPUT image (several may share a source: whole image + object patches, dataset_voa.py:195-233).
 * Geometry is torchvision's Resize(n)/CenterCrop(n) on the region (x0,y0,w,h) of an HWC uint8 RGB image; the host
 * fills it (clip_event_amd/preprocess.py).  row0/rows = the region rows the vertical pass reads. */
typedef struct {
    const unsigned char* src;  /* device pointer, HWC uint8 RGB */
    long pitch;                /* bytes per source row */
    int x0, y0, w, h;          /* region = image.crop((x0, y0, x0+w, y0+h)), inside the image */
    int ow, oh;                /* size after Resize(n): shorter side n */
    int left, top;             /* CenterCrop offsets in the resized image */
    int row0, rows;            /* first region row and row count needed by the n cropped output rows */
    long tmp_off;              /* byte offset of this output's [rows][n][3] block in the scratch buffer */
} ce_preproc_desc;
size_t ce_preprocess_table_bytes(int n_out, int n_px, int kmax);
/* out[o] (f32 [3,n,n]) = Normalize(ToTensor(CenterCrop(Resize(region_o, BICUBIC)))), clip.py:62-69, bit-exact with
 * Pillow's resampling.  kmax >= 2*ceil(2*max_scale)+1 taps; table = ce_preprocess_table_bytes(); tmp = sum of
 * rows*n*3 bytes; mean/std = HOST pointers to 3 floats. */
int ce_preprocess(const ce_preproc_desc* descs_device, int n_out, int n_px, int kmax, int max_rows, void* table,
                  void* tmp, float* out, const float* mean, const float* std, void* stream);

/* ---- contrastive head (head.hip) ---- */
int ce_l2norm_fwd(const float* f, long ldf, float* y, long ldy, float* inv_norm, int n, int E, void* stream);
int ce_l2norm_bwd(const float* dy, long lddy, const float* y, long ldy, const float* inv_norm, float* df, long lddf,
                  int n, int E, int accumulate, void* stream);
/* C[m,n] = alpha' * sum_k A[m*sam+k*sak] * B[k*sbk+n*sbn] + beta*C; alpha' = alpha * (alpha_ptr ?
 * (alpha_exp ? exp(*alpha_ptr) : *alpha_ptr) : 1).  fp32; logits and their gradients. */
int ce_sgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M,
             int N, int K, const float* alpha_ptr, float alpha, int alpha_exp, float beta, void* stream);
/* mean cross-entropy over the selected rows (sel = index_pos or NULL): *loss += mean(lse - logit[label]) */
int ce_xent_fwd(const float* logits, long ld, const int64_t* labels, const int64_t* sel, float* row_lse,
                float* loss, int nrows, int C, void* stream);
int ce_xent_bwd(const float* logits, long ld, const int64_t* labels, const int64_t* sel, const float* row_lse,
                const float* grad, float* dlogits, long ldd, int nrows, int C, void* stream);
int ce_dot(const float* a, const float* b, long n, float* out, void* stream);
/* per-instance logits_per_image (model_clip.py:509-521): lpi[b,k] = exp(*logit_scale) <In[b], Tn[b*K+k]> */
int ce_instance_logits(const float* In, const float* Tn, const float* logit_scale, float* lpi, int B, int K, int E,
                       void* stream);
int ce_instance_logits_bwd(const float* dlpi, const float* In, const float* Tn, const float* logit_scale, float* dIn,
                           float* dTn, int B, int K, int E, void* stream);
/* mean-reduced elementwise losses: mode 0 = BCEWithLogits, mode 1 = KLDiv (model_clip.py:626-629) */
int ce_elem_loss_fwd(const float* x, const float* y, long n, int mode, float* loss, void* stream);
int ce_elem_loss_bwd(const float* x, const float* y, long n, int mode, const float* grad, float* dx, void* stream);

/* ---- optimiser (optim.hip): clip_grad_norm_(.,max_norm) + Adam(L2 weight decay), engine.py:89-90 ---- */
int ce_sumsq(const float* g, long n, float* out, void* stream);
/* base[table[2b] .. table[2b+1]) = 0 for b < nchunks (element offsets, multiples of 4, in DEVICE memory): the gradient
 * zero-fill of a step restricted to the tensors that are accumulated into (ce_tower_desc.wgrad_overwrite) */
int ce_zero_segments(float* base, const long* table_device, int nchunks, void* stream);
/* p_bf16 (nullable): bf16 mirror of the updated parameters, same flat layout (the GEMM operand copies) */
int ce_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const float* sumsq, float max_norm,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
/* dst[c][r] = src[r][c] (bf16, dense) for a device table of matrices in one launch; tile_start = running
 * count of 64x64 tiles (jobs sorted by it) */
typedef struct ce_transpose_job {
    const void* src;
    void* dst;
    int rows, cols, tile_start, pad_;
} ce_transpose_job;
int ce_multi_transpose_bf16(const ce_transpose_job* jobs_device, int njobs, int total_tiles, void* stream);
/* clip + Adam (ce_adam_step's arithmetic, element for element) in a form that also leaves the TRANSPOSED bf16 copies of the weight
 * matrices behind: `jobs` lists matrices whose `src` points into the flat bf16 mirror `p_bf16` (same offset in p / g / m / v) and whose
 * `dst` receives W^T ([cols][rows]; rows, cols multiples of 8); one 64 x 64 tile per workgroup, `total_tiles` as for
 * ce_multi_transpose_bf16.  `segments` = [lo, hi) element ranges (multiples of 4, at most 2048 long: one workgroup each) of everything that is not one
 * of those matrices.  Replaces ce_adam_step + the ce_multi_transpose_bf16 pass of the next step (engine.py:87-95). */
int ce_adam_step_tiles(float* p, const float* g, float* m, float* v, void* p_bf16, const ce_transpose_job* jobs_device, int njobs,
                       int total_tiles, const long* segments_device, int nsegments, const float* sumsq, float max_norm, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* ---- transformer tower runner (tower.cpp): the 12x ResidualAttentionBlock loop of
 * Transformer.forward (model_clip.py:171-211) and its backward, all launches issued from C++ ---- */
typedef struct ce_block_params {
    /* fp32 master parameters (reference state-dict tensors, model_clip.py:175-182) */
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const float *b_qkv, *b_out, *b_fc, *b_proj;
    /* bf16 GEMM operands: w_* = [out,in] as stored by the reference, wt_* = [in,out] transposed copy */
    const void *w_qkv, *w_out, *w_fc, *w_proj;
    const void *wt_qkv, *wt_out, *wt_fc, *wt_proj;
    /* fp32 gradients, accumulated (+=) */
    float *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b;
    float *g_b_qkv, *g_b_out, *g_b_fc, *g_b_proj;
    float *g_w_qkv, *g_w_out, *g_w_fc, *g_w_proj;
    /* fp8 path (read only when ce_tower_desc.fp8 != 0): e4m3 copies of w_* ([out,in], scale per out row) and of
     * wt_* ([in,out], scale per in row), see ce_quant_rows_fp8 */
    const void *w8_qkv, *w8_out, *w8_fc, *w8_proj;
    const float *s8_qkv, *s8_out, *s8_fc, *s8_proj;
    const void *wt8_qkv, *wt8_out, *wt8_fc, *wt8_proj;
    const float *st8_qkv, *st8_out, *st8_fc, *st8_proj;
} ce_block_params;

typedef struct ce_tower_desc {
    int layers, width, heads, tokens, causal;
    const ce_block_params* blocks; /* host array [layers] of device pointers */
    int fp8;                       /* bit 0: forward Linear GEMMs on the fp8 path, bit 1: the input-gradient GEMMs too
                                    * (activations / gradients are quantised per row on the fly; weight gradients,
                                    * attention, LayerNorm and the residual stream are unchanged) */
    int stream16;                  /* 0: fp32 residual stream and gradient stream (x0 / x_out / dx / dx_sel are float);
                                    * 1: both in IEEE fp16 (CE_T_F16): x0, x_out, dx point at half data, the stash keeps half
                                    * x_mid / x_out, dx holds gradient * *grad_scale.  dx_sel stays float. */
    int wgrad_overwrite;           /* backward: != 0 = this call makes the step's FIRST contribution to the four weight
                                    * gradients g_w_* of every block it covers: they are written (ce_gemm_tn_grouped_ex), not
                                    * accumulated, and need not have been zeroed.  Bias / LayerNorm gradients always accumulate. */
    const float* grad_scale;       /* backward with stream16: DEVICE pointer to the power of two the fp16 gradient stream of this
                                    * pass is stored multiplied by (ce_grad_scale of the top-of-tower gradient) */
} ce_tower_desc;

/* bytes of activation stash + backward scratch for `batch` samples */
size_t ce_tower_workspace_bytes(const ce_tower_desc* d, int batch);
/* x_out[B*T, width] (f32) = blocks(x0[B*T, width] (f32)); stash kept in `workspace` for the backward;
 * x0 must stay valid until ce_tower_backward has run.
 * sel_rows (int32 [B], strictly ascending, flat row index of the ONE token per sample whose output is consumed: CLS for the
 * image tower, model_clip.py:256; EOT for the text tower, :415) selects the pruned mode: the last block's
 * out-projection and MLP run on those B rows only (the other rows of the last block's output are never read
 * by the reference either) and x_out is [B, width].
 * rows / cu_seqlens: a dense batch has rows = batch*tokens and cu_seqlens NULL; a packed batch (text tower,
 * see ce_token_embed) has rows = cu_seqlens[batch] <= batch*tokens activation rows, sample b owning rows
 * cu_seqlens[b] .. cu_seqlens[b+1]-1; sel_rows then index the packed rows. */
int ce_tower_forward(const ce_tower_desc* d, int batch, int rows, const int* cu_seqlens, const void* x0,
                     void* workspace, void* x_out, const int* sel_rows, void* stream);
/* dx (f32 [B*T, width]) = gradient w.r.t. x0.  Full mode (sel_rows NULL): dx holds the gradient w.r.t. x_out on
 * entry (in place).  Pruned mode: dx_sel (f32 [B, width]) is the gradient w.r.t. the [B, width] output and dx is
 * output only.  Parameter gradients are accumulated into the g_* buffers. */
int ce_tower_backward(const ce_tower_desc* d, int batch, int rows, const int* cu_seqlens, const void* x0,
                      void* workspace, void* dx, const int* sel_rows, const float* dx_sel, void* stream);
/* the same backward for blocks layer_hi .. layer_lo only (both inclusive, top-down); successive calls covering
 * layers-1 .. 0 with the same buffers equal one ce_tower_backward.  After the call that ends at block l the
 * gradients of every block >= l are final, so a data-parallel caller can start reducing them
 * (clip_event_amd/distributed.py) while the lower blocks still run. */
int ce_tower_backward_range(const ce_tower_desc* d, int batch, int rows, const int* cu_seqlens, const void* x0,
                            void* workspace, void* dx, const int* sel_rows, const float* dx_sel, int layer_hi,
                            int layer_lo, void* stream);

/* ---- optimal-transport alignment + region pooling (ot.hip) ---- */
/* dist[b] = trace(C_b T_b): cosine cost between txt[b] (M rows) and img[b] (N rows), IPOT plan
 * (beta, `iters` outer iterations, 1 inner) computed without gradient; pads are uint8 [B,M]/[B,N].
 * Strides in elements: batch stride, row stride (unit column stride).  T [B,N,M] and the inverse row
 * norms are kept for the backward.  model_ot.py:8-83; M,N <= 64. */
int ce_ot_fwd(const float* txt, long tsb, long tsr, const float* img, long isb, long isr,
              const unsigned char* txt_pad, const unsigned char* img_pad, float* dist, float* T, float* txt_inv,
              float* img_inv, int B, int M, int N, int D, float beta, int iters, void* stream);
/* dtxt [B,M,D], dimg [B,N,D] (dense) for upstream grad[b] on dist[b] (gradient through the cost only) */
int ce_ot_bwd(const float* txt, long tsb, long tsr, const float* img, long isb, long isr, const float* T,
              const float* txt_inv, const float* img_inv, const float* grad, float* dtxt, float* dimg, int B, int M,
              int N, int D, void* stream);
/* out[i,:] = mean(grid[img_i, x0:x1, y0:y1, :]); boxes int32 [nbox,5] = {img,x0,y0,x1,y1}
 * (model_clip.py:438-443; strides of the grid view in elements); backward adds into dense dgrid [B,g,g,E] */
int ce_bbox_pool_fwd(const float* grid, long sb, long s0, long s1, const int* boxes, float* out, int nbox, int E,
                     void* stream);
int ce_bbox_pool_bwd(const float* dout, const int* boxes, float* dgrid, int g, int nbox, int E, void* stream);

/* Fused contrastive head for the 'ce' criterion over the batch (model_clip.py:496-521 logits + :633-662 cross entropy)
 * that never forms the [nq, nk] logits matrix (336 MB at N = 4096, K = 5): q [.,E], k [nk,E] are L2-NORMALISED fp32
 * features (ce_l2norm_fwd), query r = row sel[r] of q (sel nullable: row r), its target column labels[sel[r]].
 *   fwd: lse[r] = log sum_c exp(s <q_r, k_c>), *loss += mean_r (lse[r] - s <q_r, k_label>), s = exp(*logit_scale);
 *        workspace = ce_infonce_workspace_bytes(nq) bytes of scratch.
 *   bwd: for the upstream scalar *grad: dq[sel[r],:] += s sum_c G[r,c] k_c, dk[c,:] += s sum_r G[r,c] q_r,
 *        *dlogit_scale += sum G[r,c] s <q_r,k_c>, G = grad/nq (softmax_r - onehot); dq / dk are gradients w.r.t. the
 *        normalised features (feed ce_l2norm_bwd) and must be zero-filled (or hold earlier contributions).
 * fp32 on v_mfma_f32_32x32x2_f32; E a multiple of 128, 128..1024. */
#define CE_INFONCE_MAX_SPLITS 64
size_t ce_infonce_workspace_bytes(int nq);
int ce_infonce_fwd(const float* q, long ldq, const int64_t* sel, int nq, const float* k, long ldk, int nk, int E,
                   const float* logit_scale, const int64_t* labels, float* lse, float* loss, void* workspace, void* stream);
int ce_infonce_bwd(const float* q, long ldq, const int64_t* sel, int nq, const float* k, long ldk, int nk, int E,
                   const float* logit_scale, const int64_t* labels, const float* lse, const float* grad, float* dq, float* dk,
                   float* dlogit_scale, void* stream);

/* Small-batch contrastive head in three launches: feature normalisation, logits_per_image / logits_per_text over the batch
 * (model_clip.py:496-521), CriterionContrastive 'ce' with index_pos (model_clip.py:633-662) and the whole backward down to the raw
 * features, in fp32 -- for the sizes where the head between the towers' forward and backward is pure launch latency (config 2:
 * 256 x 256 logits, 21 launches / 200 us in head.hip's form).  1 <= nI, nT <= 1024, nsel <= nT, E <= 1024 and a multiple of 4.
 *   fi [nI,E], ft [nT,E] raw features; labels_i [nI] target column of image row i; sel [nsel] (nullable: rows 0..nsel-1) the text rows
 *   that carry a loss, labels_t [nT] the target column of every text row (row sel[r] uses labels_t[sel[r]], as ce_xent_fwd does).
 * ce_head_small_fwd fills `workspace` (ce_head_small_workspace_floats floats): normalised features, P = (softmax - onehot) / rows
 * of both directions and, at ce_head_small_scalars_offset, four floats {loss_i, loss_t, sum P_i * lpi, sum P_t * lpt}.
 * ce_head_small_bwd takes the upstream gradients of the two losses as DEVICE scalars (nullable = 0) and writes dfi, dft and
 * *dlogit_scale (nullable). */
size_t ce_head_small_workspace_floats(int nI, int nT, int nsel, int E);
size_t ce_head_small_scalars_offset(int nI, int nT, int nsel, int E);
int ce_head_small_fwd(const float* fi, const float* ft, int nI, int nT, int E, const float* logit_scale, const int64_t* labels_i,
                      const int64_t* labels_t, const int64_t* sel, int nsel, float* workspace, void* stream);
int ce_head_small_bwd(int nI, int nT, int nsel, int E, const float* logit_scale, const float* g_i, const float* g_t,
                      const int64_t* sel, const float* workspace, float* dfi, float* dft, float* dlogit_scale, void* stream);

/* Region / argument InfoNCE of the train_arg branch (model_clip.py:456-488) for every image of the batch in one
 * launch.  region / desc / label: f32 [R,E] rows grouped per image, image g owning rows offsets[g] .. offsets[g+1]-1 (at
 * most 16; max_rows = the largest group, checked on the host); label may be NULL when use_label = 0.  Adds
 *   loss_bbox += sum_g CE(s r^ d^T) [+ CE(s r^ l^T)],  loss_arg += sum_g CE(s d^ r^T) [+ CE(s l^ r^T)] [+ CE(s d^ l^T)]
 * (x^ = x/|x|, s = exp(*logit_scale), targets arange(n), mean over the n rows; the l terms with use_label, the last with
 * role_text).  The backward takes the upstream scalars g_bbox / g_arg (device) and writes dregion / ddesc / dlabel (the
 * caller zero-fills them) and adds the logit_scale gradient. */
int ce_region_nce_fwd(const float* region, const float* desc, const float* label, const int* offsets, int groups,
                      int max_rows, int E, const float* logit_scale, int use_label, int role_text, float* loss_bbox,
                      float* loss_arg, void* stream);
int ce_region_nce_bwd(const float* region, const float* desc, const float* label, const int* offsets, int groups,
                      int max_rows, int E, const float* logit_scale, int use_label, int role_text, const float* g_bbox,
                      const float* g_arg, float* dregion, float* ddesc, float* dlabel, float* dlogit_scale, void* stream);

/* ==================================================================================================================
 * CE_DIAG -- diagnostics: profiler, tuning hooks and lane-map probes.  Used by bench.py's instrumented pass, tools/ and
 * tests/; the product path (clip_event_amd/*.py) calls none of them.
 * ================================================================================================================== */
/* ---- optional profiler: HIP events on the launch stream around every launch, summed per kernel
 * class.  ce_profile_collect fills out[class][4] = {launches, total ms, algorithmic FLOPs, algorithmic
 * bytes} and resets.  Used by bench.py for the `roofline` object; off by default. ---- */
#define CE_PROF_NT_FAMILIES 8
enum {
    CE_PROF_GEMM_NT0 = 0, /* + CE_PROF_NT_FAMILIES * epilogue id (0..7) + kernel family: 0 gemm_nt_kernel (128x128, register
                           * staged), 1 gemm_nt256_kernel<.,.,2> (160x128, 4 waves), 2 gemm_nt256_kernel<.,.,4> (256 columns,
                           * 8 waves), 3 gemm_nt32_kernel (160x256x32), 4 gemm_nt8_kernel (fp8), 5 gemm_nt160lw_kernel (160x256,
                           * loader waves), 6 gemm_nt160p_kernel (its persistent form), 7 gemm_nt_skinny_kernel (M <= 512) -- one class per
                           * rocprofv3 kernel row */
    CE_PROF_GEMM_TN = 64, CE_PROF_ATTN_FWD = 65, CE_PROF_ATTN_BWD = 66, CE_PROF_LN_FWD = 67, CE_PROF_LN_BWD = 68,
    CE_PROF_COLSUM = 69, CE_PROF_OTHER = 70, CE_PROF_GEMM_TN2 = 71 /* gemm_tn2_kernel; CE_PROF_GEMM_TN = gemm_tn3_kernel */,
    CE_PROF_NCLASS = 72
};
void ce_profile_enable(int on);
int ce_profile_collect(double* out, int max_classes);
int ce_profile_num_classes(void);
const char* ce_profile_class_name(int cls);

/* tuning hook (tools/, tests/): force the NT tile variant -- 0 auto, 3..8 rows/32 of the 256-column kernel,
 * 32 = 160x256x32 two-workgroup kernel, 104 = 160x128 four-wave, 160 = three-stage ring, 161 = loader waves (one tile
 * per workgroup), 162 = persistent loader waves (163..165: with 96/128/160-row tiles); 1000..1999 = tile walk of the
 * persistent kernel: 1000 XCD-owned chunks of tiles_m / 8 row panels, 1001 launch-wide (default), 1001 + n chunks of n */
void ce_gemm_nt_tune(int variant);
/* the two-height tile plan of the most recent persistent NT launch (tests): row panels of 160 rows, height of the others in
 * 32-row units; 0, 0 when the launch used one height */
int ce_gemm_nt_last_plan(int* tall_panels, int* short_tm);
/* tuning hook (tools/ only): 0 auto, 4/5/6/8 = tile height (x32 rows) of the 256-column kernel, 105 = 160x128 tile */
void ce_gemm_nt_fp8_tune(int variant);

/* CU budget of the NT GEMM launch policies: every kernel of the family puts ONE 156 KiB workgroup on a CU and sizes its grid to
 * fill the chip exactly once, so a CU held by another stream's kernel (an RCCL channel during the gradient all-reduce,
 * train.py:222-225's DDP) makes the launch wait for a second round.  `cus` < 256 sizes the one-round and the persistent grids
 * for that many CUs and leaves the rest to whoever holds them; 0 restores the default (CE_GEMM_CUS, else 256).  Process-wide;
 * call between steps. */
int ce_gemm_set_cu_budget(int cus);
/* DYNAMIC tile lists in the persistent NT kernel (off by default; CE_NT_DYNAMIC=1 or this call; -1 = back to the environment's
 * choice): a workgroup's first tile is its static one, every further tile comes from a per-launch device counter, fetched by
 * one wave while the current tile is multiplied.  A workgroup the dispatcher could not place -- its CU is held by another
 * stream's kernel, e.g. an RCCL channel during the gradient all-reduce -- then finds the list empty when it starts instead of
 * holding the launch up for a whole static tile list (measured: 1.2-1.4x per launch beside a resident foreign workgroup). */
int ce_gemm_set_dynamic_tiles(int on);

/* "CU hog" (bench.py --cu-hog): `blocks` workgroups that each occupy one CU (96 KiB of LDS, 256 threads) for
 * `microseconds` of wall time and do nothing -- what RCCL's channel kernels take away from the GEMM grids during a gradient
 * all-reduce, so that the 8-GPU contention risk can be sized on one GPU.  Bounded spin: every wave exits when the time is up. */
int ce_cu_hog(int blocks, float microseconds, void* stream);
/* shader clock in MHz that the last hog's first workgroup saw over its lifetime (s_memtime ticks per 100 MHz s_memrealtime
 * tick): what the chip clocked at under the load that ran beside the hog.  Synchronises the device. */
double ce_cu_hog_clock_mhz(void);

/* Debug probes: raw MFMA / transposed-LDS-read lane maps (tests/test_hip_probes.py). */
int ce_probe_mfma(int shape, const void* a_frags, const void* b_frags, float* out, void* stream);
/* one v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3): a_frags / b_frags 64 lanes x 32 bytes, scale_a / scale_b 64 ints (E8M0 in
 * byte 0), out 64 lanes x 4 floats */
int ce_probe_mfma_scale(const void* a_frags, const void* b_frags, const int* scale_a, const int* scale_b, float* out, void* stream);
int ce_probe_tr16(const void* image, int n_elems, const int* byte_off, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIP_EVENT_HIP_H */
