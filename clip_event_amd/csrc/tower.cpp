// Host-side runner for one transformer tower: issues every launch of the residual-block loop
// (Transformer.forward / ResidualAttentionBlock.forward, model_clip.py:171-211) and of its
// backward from C++, on one HIP stream, with a caller-provided workspace.  No allocation, no
// synchronisation, no Python in the loop.
//
// Activation stash per block (rows M = batch*tokens, width d), all needed by the backward:
//   x_in f32 (previous block's output), h1 bf16, qkv bf16 [M,3d], o bf16, lse f32,
//   x_mid f32, h2 bf16, a bf16 [M,4d] (pre-GELU), g bf16 [M,4d], LayerNorm mean/rstd.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(reinterpret_cast<char*>(b)) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct BlockStash {
    float* x_mid;
    float* x_out;  // next block's input
    bf16_t *h1, *qkv, *o, *h2, *a, *g;
    float *mean1, *rstd1, *mean2, *rstd2, *lse;
};

struct Layout {
    static constexpr int MAX_LAYERS = 64;
    BlockStash blk[MAX_LAYERS];
    bf16_t *dxb, *dxb2, *dh, *da, *dqkv, *d_o;
    size_t bytes;
};

void carve(const ce_tower_desc* d, int batch, void* ws, Layout& L) {
    Carver c(ws);
    const size_t M = (size_t)batch * d->tokens, w = d->width;
    for (int l = 0; l < d->layers; ++l) {
        BlockStash& s = L.blk[l];
        s.x_mid = c.take<float>(M * w);
        s.x_out = (l + 1 < d->layers) ? c.take<float>(M * w) : nullptr;  // last block writes the caller's x_out
        s.h1 = c.take<bf16_t>(M * w);
        s.qkv = c.take<bf16_t>(M * 3 * w);
        s.o = c.take<bf16_t>(M * w);
        s.h2 = c.take<bf16_t>(M * w);
        s.a = c.take<bf16_t>(M * 4 * w);
        s.g = c.take<bf16_t>(M * 4 * w);
        s.mean1 = c.take<float>(M);
        s.rstd1 = c.take<float>(M);
        s.mean2 = c.take<float>(M);
        s.rstd2 = c.take<float>(M);
        s.lse = c.take<float>((size_t)batch * d->heads * d->tokens);
    }
    L.dxb = c.take<bf16_t>(M * w);
    L.dxb2 = c.take<bf16_t>(M * w);
    L.dh = c.take<bf16_t>(M * w);
    L.da = c.take<bf16_t>(M * 4 * w);
    L.dqkv = c.take<bf16_t>(M * 3 * w);
    L.d_o = c.take<bf16_t>(M * w);
    L.bytes = (c.off + 255) & ~size_t(255);
}

int check_desc(const ce_tower_desc* d, int batch) {
    CE_CHECK_ARG(d && d->blocks, "tower: null descriptor");
    CE_CHECK_ARG(d->layers > 0 && d->layers <= Layout::MAX_LAYERS, "tower: layers=%d out of range", d->layers);
    CE_CHECK_ARG(d->width == d->heads * 64, "tower: width %d != heads %d * 64", d->width, d->heads);
    CE_CHECK_ARG(d->tokens > 0 && d->tokens <= 128, "tower: tokens=%d unsupported (1..128)", d->tokens);
    CE_CHECK_ARG(batch > 0, "tower: empty batch");
    return 0;
}

#define TRY(call)            \
    do {                     \
        int rc__ = (call);   \
        if (rc__ != 0) return rc__; \
    } while (0)

}  // namespace

extern "C" size_t ce_tower_workspace_bytes(const ce_tower_desc* d, int batch) {
    if (check_desc(d, batch) != 0) return 0;
    Layout L;
    carve(d, batch, nullptr, L);
    return L.bytes;
}

extern "C" int ce_tower_forward(const ce_tower_desc* d, int batch, const float* x0, void* workspace, float* x_out,
                                void* stream) {
    TRY(check_desc(d, batch));
    CE_CHECK_ARG(x0 && workspace && x_out, "ce_tower_forward: null buffer");
    Layout L;
    carve(d, batch, workspace, L);
    const int M = batch * d->tokens, w = d->width;
    const float* x = x0;
    for (int l = 0; l < d->layers; ++l) {
        const ce_block_params& p = d->blocks[l];
        BlockStash& s = L.blk[l];
        float* xo = (l + 1 < d->layers) ? s.x_out : x_out;
        TRY(ce_layernorm_fwd(x, w, nullptr, p.ln1_w, p.ln1_b, s.h1, w, 0, s.mean1, s.rstd1, M, w, 1e-5f, stream));
        TRY(ce_gemm_nt(s.h1, w, p.w_qkv, w, M, 3 * w, w, CE_EPI_BIAS_BF16, p.b_qkv, nullptr, 0, s.qkv, 3 * w, nullptr, 0,
                       nullptr, 0, stream));
        TRY(ce_attention_fwd(s.qkv, 3 * w, s.o, w, s.lse, batch, d->tokens, d->heads, d->causal, stream));
        TRY(ce_gemm_nt(s.o, w, p.w_out, w, M, w, w, CE_EPI_BIAS_RESID_F32, p.b_out, x, w, s.x_mid, w, nullptr, 0, nullptr,
                       0, stream));
        TRY(ce_layernorm_fwd(s.x_mid, w, nullptr, p.ln2_w, p.ln2_b, s.h2, w, 0, s.mean2, s.rstd2, M, w, 1e-5f, stream));
        TRY(ce_gemm_nt(s.h2, w, p.w_fc, w, M, 4 * w, w, CE_EPI_BIAS_GELU, p.b_fc, nullptr, 0, s.a, 4 * w, s.g, 4 * w,
                       nullptr, 0, stream));
        TRY(ce_gemm_nt(s.g, 4 * w, p.w_proj, 4 * w, M, w, 4 * w, CE_EPI_BIAS_RESID_F32, p.b_proj, s.x_mid, w, xo, w,
                       nullptr, 0, nullptr, 0, stream));
        x = xo;
    }
    return 0;
}

extern "C" int ce_tower_backward(const ce_tower_desc* d, int batch, const float* x0, void* workspace, float* dx,
                                 void* stream) {
    TRY(check_desc(d, batch));
    CE_CHECK_ARG(x0 && workspace && dx, "ce_tower_backward: null buffer");
    Layout L;
    carve(d, batch, workspace, L);
    const int M = batch * d->tokens, w = d->width;
    // dxb_a: bf16 gradient at the block output (operand of mlp.c_proj's dgrad/wgrad);
    // dxb_b: bf16 gradient at x_mid (operand of attn.out_proj's dgrad/wgrad).  Both stay alive until the
    // block's four weight gradients run as ONE grouped launch after the last dgrad.
    bf16_t* dxb_a = L.dxb;
    bf16_t* dxb_b = L.dxb2;
    TRY(ce_cast_bf16(dx, dxb_a, (long)M * w, stream));
    for (int l = d->layers - 1; l >= 0; --l) {
        const ce_block_params& p = d->blocks[l];
        BlockStash& s = L.blk[l];
        const float* x_in = (l == 0) ? x0 : L.blk[l - 1].x_out;
        // ---- mlp.c_proj : x_out = x_mid + g Wp^T + bp ----
        TRY(ce_gemm_nt(dxb_a, w, p.wt_proj, w, M, 4 * w, w, CE_EPI_GELUGRAD_BF16, nullptr, nullptr, 0, L.da, 4 * w, nullptr,
                       0, s.a, 4 * w, stream));                                   // da = (dx Wp) * gelu'(a)
        if (l == d->layers - 1) TRY(ce_colsum_bf16(dxb_a, w, p.g_b_proj, M, w, stream));   // lower blocks: fused in ln_1's backward
        // ---- mlp.c_fc : a = h2 Wf^T + bf ----
        TRY(ce_gemm_nt(L.da, 4 * w, p.wt_fc, 4 * w, M, w, 4 * w, CE_EPI_BF16, nullptr, nullptr, 0, L.dh, w, nullptr, 0,
                       nullptr, 0, stream));                                      // dh2 = da Wf
        TRY(ce_colsum_bf16(L.da, 4 * w, p.g_b_fc, M, 4 * w, stream));
        // ---- ln_2 (+ residual); also the column sums of dx = attn.out_proj bias gradient ----
        TRY(ce_layernorm_bwd(L.dh, w, 0, s.x_mid, w, nullptr, s.mean2, s.rstd2, p.ln2_w, dx, dx, w, dxb_b, w, p.g_ln2_w,
                             p.g_ln2_b, p.g_b_out, M, w, stream));
        // ---- attn.out_proj : x_mid = x_in + o Wo^T + bo ----
        TRY(ce_gemm_nt(dxb_b, w, p.wt_out, w, M, w, w, CE_EPI_BF16, nullptr, nullptr, 0, L.d_o, w, nullptr, 0, nullptr, 0,
                       stream));                                                  // d_o = dx Wo
        // ---- attention core ----
        TRY(ce_attention_bwd(s.qkv, 3 * w, s.o, w, L.d_o, w, s.lse, L.dqkv, 3 * w, batch, d->tokens, d->heads, d->causal,
                             stream));
        // ---- attn.in_proj : qkv = h1 Wqkv^T + bqkv ----
        TRY(ce_gemm_nt(L.dqkv, 3 * w, p.wt_qkv, 3 * w, M, w, 3 * w, CE_EPI_BF16, nullptr, nullptr, 0, L.dh, w, nullptr, 0,
                       nullptr, 0, stream));                                      // dh1 = dqkv Wqkv
        TRY(ce_colsum_bf16(L.dqkv, 3 * w, p.g_b_qkv, M, 3 * w, stream));
        // ---- the four weight gradients of this block, one launch ----
        {
            const void* P[4] = {dxb_a, L.da, dxb_b, L.dqkv};
            const long ldp[4] = {w, 4L * w, w, 3L * w};
            const void* Q[4] = {s.g, s.h2, s.o, s.h1};
            const long ldq[4] = {4L * w, w, w, w};
            const int Nn[4] = {w, 4 * w, w, 3 * w};
            const int Kk[4] = {4 * w, w, w, w};
            float* out[4] = {p.g_w_proj, p.g_w_fc, p.g_w_out, p.g_w_qkv};
            const long ldo[4] = {4L * w, w, w, w};
            TRY(ce_gemm_tn_grouped(4, P, ldp, Q, ldq, M, Nn, Kk, out, ldo, 0, stream));
        }
        // ---- ln_1 (+ residual); column sums of dx = previous block's mlp.c_proj bias gradient ----
        TRY(ce_layernorm_bwd(L.dh, w, 0, x_in, w, nullptr, s.mean1, s.rstd1, p.ln1_w, dx, dx, w, dxb_a, w, p.g_ln1_w,
                             p.g_ln1_b, (l > 0) ? d->blocks[l - 1].g_b_proj : nullptr, M, w, stream));
    }
    return 0;
}
