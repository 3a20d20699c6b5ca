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
 *   - activations are token-major [rows = batch*tokens, width]; the fp32 residual stream and
 *     fp32 master weights / gradients stay fp32, GEMM operands are bf16, accumulation fp32.
 */
#ifndef CLIP_EVENT_HIP_H
#define CLIP_EVENT_HIP_H

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
    CE_EPI_BIAS_GELU = 5,     /* out(bf16) = a = acc + bias; out2(bf16) = a*sigmoid(1.702a) */
    CE_EPI_GELUGRAD_BF16 = 6  /* out(bf16) = acc * dQuickGELU(aux(bf16))                    */
};

/* C[M,N] = A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulate, fused epilogue.
 * Replaces nn.Linear / MHA in_proj,out_proj / Conv2d(k=s=patch) forward and their
 * input-gradient GEMMs (model_clip.py:175-180, :188, :219, :230, :329, :415). */
int ce_gemm_nt(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int epilogue,
               const float* bias, const float* resid, long ldr, void* out, long ldo, void* out2, long ldo2,
               const void* aux, long ldaux, void* stream);

/* out[Nn,Kk] (f32) += P[M,Nn]^T . Q[M,Kk]  (weight gradients; fp32 atomic accumulation, so the
 * caller zeroes `out` once per step).  splits<=0 picks the M split that fills the chip. */
int ce_gemm_tn(const void* P, long ldp, const void* Q, long ldq, int M, int Nn, int Kk, float* out, long ldo,
               int splits, void* stream);

/* y = LayerNorm(x[rows[r]] or x[r]) over D columns, fp32 statistics (eps inside the sqrt).
 * y is bf16 (out_f32=0: the next GEMM's operand) or fp32 (ln_pre: the residual stream).
 * Writes mean/rstd [M] for the backward.  Replaces LayerNorm.forward, model_clip.py:157-163. */
int ce_layernorm_fwd(const float* x, long ldx, const int* rows, const float* w, const float* b, void* y, long ldy,
                     int out_f32, float* mean, float* rstd, int M, int D, float eps, void* stream);

/* dx_out[dst] = (dx_in ? dx_in[dst] : 0) + dLN(dy[r]); dst = rows ? rows[r] : r; dxb = bf16 copy
 * (nullable); dw/db (f32 [D]) accumulate atomically (caller zeroes once per step). */
int ce_layernorm_bwd(const void* dy, long lddy, int dy_f32, const float* x, long ldx, const int* rows,
                     const float* mean, const float* rstd, const float* w, const float* dx_in, float* dx_out,
                     long lddx, void* dxb, long lddxb, float* dw, float* db, int M, int D, void* stream);

/* Self-attention core on the packed in-projection output qkv[B*L, 3*H*64] (bf16; q | k | v column
 * blocks, head h at columns h*64): o[B*L, H*64] = softmax(q k^T / 8 + causal?) v, lse[B*H*L] (f32)
 * saved for the backward.  L <= 128.  Replaces the core of nn.MultiheadAttention as called at
 * model_clip.py:188 (mask from model_clip.py:377-384). */
int ce_attention_fwd(const void* qkv, long ld, void* o, long ldo, float* lse, int B, int L, int H, int causal,
                     void* stream);
int ce_attention_bwd(const void* qkv, long ld, const void* o, long ldo, const void* dout, long lddo,
                     const float* lse, void* dqkv, long lddq, int B, int L, int H, int causal, void* stream);

/* Debug probes: raw MFMA / transposed-LDS-read lane maps (tests/test_hip_probes.py). */
int ce_probe_mfma(int shape, const void* a_frags, const void* b_frags, float* out, void* stream);
int ce_probe_tr16(const void* image, int n_elems, const int* byte_off, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIP_EVENT_HIP_H */
