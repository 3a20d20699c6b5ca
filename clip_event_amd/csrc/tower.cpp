// Host-side runner for one transformer tower: issues every launch of the residual-block loop
// (Transformer.forward / ResidualAttentionBlock.forward, model_clip.py:171-211) and of its
// backward from C++, on one HIP stream, with a caller-provided workspace.  No allocation, no
// synchronisation, no Python in the loop.
//
// Activation stash per block (rows M = batch*tokens, width d), all needed by the backward:
//   x_in (previous block's output; stream type: f32, or f16 with ce_tower_desc.stream16), h1 bf16, qkv bf16 [M,3d], o bf16,
//   lse f32, x_mid (stream type), h2 bf16, a bf16 [M,4d] (QuickGELU' of the c_fc output), g bf16 [M,4d] (QuickGELU of it), LayerNorm mean/rstd.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <map>
#include <mutex>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

// Weight gradients are off the critical path of the backward chain (nothing downstream of a block reads them), so the
// backward DEFERS them: the four problems of a block are queued and several consecutive blocks' queues go out as ONE
// grouped launch (ce_gemm_tn_grouped): several rounds of unsplit tiles, so each round's atomic epilogue (memory-side
// float atomics run at 1.3 TB/s chip-wide: a 10-20 % tail of a one-round launch) overlaps the next round's contraction.  The gradient operands (bf16
// dY buffers) of the queued blocks stay alive in a ring of WG_SETS buffer sets.
constexpr int WG_MAX_BLOCKS = 8;            // blocks per grouped launch (x 4 problems <= CE_TN_MAX_GROUP)
constexpr int WG_SETS = WG_MAX_BLOCKS + 1;  // a block writes its own set and the next block's dxb
// LayerNorm-backward partial sums: a ring of LNP_RING buffer sets (two per block); ce_layernorm_fold runs whenever the ring is
// full and at the end of a backward call, so a tower workspace carries 8 sets, not one per layer (ViT-L: 24) -- they only live
// between a LayerNorm backward and the next fold on the same stream.
constexpr int LNP_RING = 8;

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
    void* x_mid;   // residual stream after the attention half (stream element type)
    void* x_out;   // next block's input
    bf16_t *h1, *qkv, *o, *h2, *a, *g;
    float *mean1, *rstd1, *mean2, *rstd2, *lse;
};

struct Layout {
    static constexpr int MAX_LAYERS = 64;
    BlockStash blk[MAX_LAYERS];
    bf16_t *dxb[WG_SETS], *dxb2[WG_SETS], *da[WG_SETS], *dqkv[WG_SETS];   // ring of gradient-operand sets (block l uses
                                                                           // set l % WG_SETS): alive until the queued
                                                                           // weight gradients of the block have launched
    bf16_t *dh, *d_o;
    uint8_t* q8;      // fp8 path: the quantised A operand of the GEMM being issued (rows x 4 width bytes) ...
    float* q8s;       // ... and its per-row scales
    // compact [batch, .] buffers of the pruned last block (xs_in / xs_mid / dxs_mid hold stream-typed data)
    float *xs_in, *xs_mid, *dxs_mid, *means, *rstds;
    bf16_t *os, *h2s, *as, *gs, *dxbs, *dxb2s, *das, *dhs, *dos;
    float* lnp[LNP_RING][2];        // backward: per-workgroup partial sums of the two LayerNorm backwards of a block ([blocks][3][width], ce_layernorm_fold), ring slot l % LNP_RING
    size_t bytes;
};

// `rows` = activation rows the stash is laid out for: batch*tokens, or fewer for a packed (variable-length) batch
void carve(const ce_tower_desc* d, int batch, size_t rows, void* ws, Layout& L) {
    Carver c(ws);
    const size_t M = rows, w = d->width;
    const size_t esz = d->stream16 ? 2 : 4;                               // bytes per residual-stream element
    for (int l = 0; l < d->layers; ++l) {
        BlockStash& s = L.blk[l];
        s.x_mid = c.take<char>(M * w * esz);
        s.x_out = (l + 1 < d->layers) ? c.take<char>(M * w * esz) : nullptr;  // last block writes the caller's x_out
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
    for (int q = 0; q < WG_SETS; ++q) {
        L.dxb[q] = c.take<bf16_t>(M * w);
        L.dxb2[q] = c.take<bf16_t>(M * w);
        L.da[q] = c.take<bf16_t>(M * 4 * w);
        L.dqkv[q] = c.take<bf16_t>(M * 3 * w);
    }
    L.dh = c.take<bf16_t>(M * w);
    L.d_o = c.take<bf16_t>(M * w);
    L.q8 = d->fp8 ? c.take<uint8_t>(M * 4 * w) : nullptr;
    L.q8s = d->fp8 ? c.take<float>(M) : nullptr;
    const size_t Bn = (size_t)batch;
    L.xs_in = c.take<float>(Bn * w); L.xs_mid = c.take<float>(Bn * w); L.dxs_mid = c.take<float>(Bn * w);
    L.means = c.take<float>(Bn); L.rstds = c.take<float>(Bn);
    L.os = c.take<bf16_t>(Bn * w); L.h2s = c.take<bf16_t>(Bn * w); L.as = c.take<bf16_t>(Bn * 4 * w); L.gs = c.take<bf16_t>(Bn * 4 * w);
    L.dxbs = c.take<bf16_t>(Bn * w); L.dxb2s = c.take<bf16_t>(Bn * w); L.das = c.take<bf16_t>(Bn * 4 * w);
    L.dhs = c.take<bf16_t>(Bn * w); L.dos = c.take<bf16_t>(Bn * w);
    const size_t lnp_floats = (size_t)ce_layernorm_bwd_blocks((int)M, (int)w) * 3 * w;
    for (int l = 0; l < LNP_RING; ++l) {
        L.lnp[l][0] = l < d->layers ? c.take<float>(lnp_floats) : nullptr;
        L.lnp[l][1] = l < d->layers ? c.take<float>(lnp_floats) : nullptr;
    }
    L.bytes = (c.off + 255) & ~size_t(255);
}

int check_rows(const ce_tower_desc* d, int batch, int rows, const int* cu) {
    CE_CHECK_ARG(rows > 0 && (long)rows <= (long)batch * d->tokens, "tower: rows=%d outside 1..batch*tokens", rows);
    CE_CHECK_ARG(cu || rows == batch * d->tokens, "tower: a dense batch has batch*tokens rows (got %d)", rows);
    return 0;
}

int check_desc(const ce_tower_desc* d, int batch) {
    CE_CHECK_ARG(d && d->blocks, "tower: null descriptor");
    CE_CHECK_ARG(d->layers > 0 && d->layers <= Layout::MAX_LAYERS, "tower: layers=%d out of range", d->layers);
    CE_CHECK_ARG(d->width == d->heads * 64, "tower: width %d != heads %d * 64", d->width, d->heads);
    CE_CHECK_ARG(d->tokens > 0 && d->tokens <= 4096, "tower: tokens=%d unsupported (1..4096)", d->tokens);
    CE_CHECK_ARG(batch > 0, "tower: empty batch");
    CE_CHECK_ARG(d->stream16 == 0 || d->stream16 == 1, "tower: stream16=%d (0: fp32 residual stream, 1: fp16)", d->stream16);
    return 0;
}

#define TRY(call)            \
    do {                     \
        int rc__ = (call);   \
        if (rc__ != 0) return rc__; \
    } while (0)

// One Linear-layer GEMM of the block, C = A . W^T (+ epilogue): bf16 (ce_gemm_nt), or -- `use8` -- A quantised per row to
// e4m3 into the layout's scratch and multiplied with the e4m3 copy of the weight (ce_gemm_nt_fp8).  Shapes the fp8
// kernel does not take (K not a multiple of 128, K > 4096) stay on bf16.
// `q8_of` = the bf16 matrix whose e4m3 copy currently sits in the layout's scratch (written by a LayerNorm that had the rows
// in registers, ce_layernorm_*_q8): its quantisation pass is skipped.
int linear(bool use8, const Layout& L, const void*& q8_of, const void* A, long lda, const void* W, const void* W8, const float* S8, int M,
           int N, int K, int epi, const float* bias, const void* resid, long ldr, void* out, long ldo, void* out2,
           long ldo2, const void* aux, long ldaux, void* stream) {
    if (use8 && W8 && S8 && K % 128 == 0 && K <= 4096) {
        if (q8_of != A || lda != K) TRY(ce_quant_rows_fp8(A, lda, L.q8, K, L.q8s, M, K, stream));
        q8_of = nullptr;                         // the scratch is free for the next producer once this GEMM is enqueued
        return ce_gemm_nt_fp8(L.q8, K, L.q8s, W8, K, S8, M, N, K, epi, bias, resid, ldr, out, ldo, out2, ldo2, aux, ldaux,
                              stream);
    }
    return ce_gemm_nt(A, lda, W, K, M, N, K, epi, bias, resid, ldr, out, ldo, out2, ldo2, aux, ldaux, stream);
}

}  // namespace

extern "C" size_t ce_tower_workspace_bytes(const ce_tower_desc* d, int batch) {
    if (check_desc(d, batch) != 0) return 0;
    Layout L;
    carve(d, batch, (size_t)batch * d->tokens, nullptr, L);
    return L.bytes;
}

extern "C" int ce_tower_forward(const ce_tower_desc* d, int batch, int rows, const int* cu_seqlens, const void* x0,
                                void* workspace, void* x_out, const int* sel_rows, void* stream) {
    TRY(check_desc(d, batch));
    TRY(check_rows(d, batch, rows, cu_seqlens));
    CE_CHECK_ARG(x0 && workspace && x_out, "ce_tower_forward: null buffer");
    Layout L;
    carve(d, batch, rows, workspace, L);
    const int M = rows, w = d->width;
    const bool f8 = (d->fp8 & 1) != 0;
    const int ST = d->stream16 ? CE_T_F16 : CE_T_F32;                         // residual-stream element type
    const int EPI_RESID = d->stream16 ? CE_EPI_BIAS_RESID_F16 : CE_EPI_BIAS_RESID_F32;
    const long esz = d->stream16 ? 2 : 4;
    const void* x = x0;
    const void* q8_of = nullptr;
    // fp8 path: a LayerNorm whose output feeds an e4m3 GEMM writes the e4m3 copy + row scales itself (width % 128 == 0, <= 4096)
    const bool lnq = f8 && w % 128 == 0 && w <= 4096;
    for (int l = 0; l < d->layers; ++l) {
        const ce_block_params& p = d->blocks[l];
        BlockStash& s = L.blk[l];
        void* xo = (l + 1 < d->layers) ? s.x_out : x_out;
        TRY(ce_layernorm_fwd_q8(x, ST, w, nullptr, p.ln1_w, p.ln1_b, s.h1, CE_T_BF16, w, s.mean1, s.rstd1, M, w, 1e-5f,
                                lnq ? L.q8 : nullptr, w, L.q8s, stream));
        if (lnq) q8_of = s.h1;
        TRY(linear(f8, L, q8_of, s.h1, w, p.w_qkv, p.w8_qkv, p.s8_qkv, M, 3 * w, w, CE_EPI_BIAS_BF16, p.b_qkv, nullptr, 0, s.qkv, 3 * w, nullptr, 0,
                       nullptr, 0, stream));
        TRY(ce_attention_fwd(s.qkv, 3 * w, s.o, w, s.lse, cu_seqlens, batch, d->tokens, d->heads, d->causal, stream));
        if (sel_rows && l + 1 == d->layers) {
            // pruned last block: only the selected token of each sample feeds the head, so the out-projection
            // and the MLP run on `batch` rows (75 % of this block's GEMM work is never needed)
            const int Bn = batch;
            TRY(ce_copy_rows(s.o, w * 2L, sel_rows, L.os, w * 2L, nullptr, Bn, w * 2, stream));
            TRY(ce_copy_rows(x, w * esz, sel_rows, L.xs_in, w * esz, nullptr, Bn, (int)(w * esz), stream));
            TRY(linear(f8, L, q8_of, L.os, w, p.w_out, p.w8_out, p.s8_out, Bn, w, w, EPI_RESID, p.b_out, L.xs_in, w, L.xs_mid, w, nullptr,
                           0, nullptr, 0, stream));
            TRY(ce_layernorm_fwd_t(L.xs_mid, ST, w, nullptr, p.ln2_w, p.ln2_b, L.h2s, CE_T_BF16, w, L.means, L.rstds, Bn, w, 1e-5f, stream));
            TRY(linear(f8, L, q8_of, L.h2s, w, p.w_fc, p.w8_fc, p.s8_fc, Bn, 4 * w, w, CE_EPI_BIAS_GELU, p.b_fc, nullptr, 0, L.as, 4 * w, L.gs, 4 * w,
                           nullptr, 0, stream));
            TRY(linear(f8, L, q8_of, L.gs, 4 * w, p.w_proj, p.w8_proj, p.s8_proj, Bn, w, 4 * w, EPI_RESID, p.b_proj, L.xs_mid, w, x_out, w,
                           nullptr, 0, nullptr, 0, stream));
            break;
        }
        TRY(linear(f8, L, q8_of, s.o, w, p.w_out, p.w8_out, p.s8_out, M, w, w, EPI_RESID, p.b_out, x, w, s.x_mid, w, nullptr, 0, nullptr,
                       0, stream));
        TRY(ce_layernorm_fwd_q8(s.x_mid, ST, w, nullptr, p.ln2_w, p.ln2_b, s.h2, CE_T_BF16, w, s.mean2, s.rstd2, M, w, 1e-5f,
                                lnq ? L.q8 : nullptr, w, L.q8s, stream));
        if (lnq) q8_of = s.h2;
        TRY(linear(f8, L, q8_of, s.h2, w, p.w_fc, p.w8_fc, p.s8_fc, M, 4 * w, w, CE_EPI_BIAS_GELU, p.b_fc, nullptr, 0, s.a, 4 * w, s.g, 4 * w,
                       nullptr, 0, stream));
        TRY(linear(f8, L, q8_of, s.g, 4 * w, p.w_proj, p.w8_proj, p.s8_proj, M, w, 4 * w, EPI_RESID, p.b_proj, s.x_mid, w, xo, w,
                       nullptr, 0, nullptr, 0, stream));
        x = xo;
    }
    return 0;
}

extern "C" int ce_tower_backward(const ce_tower_desc* d, int batch, int rows, const int* cu_seqlens, const void* x0,
                                 void* workspace, void* dx, const int* sel_rows, const float* dx_sel, void* stream) {
    return ce_tower_backward_range(d, batch, rows, cu_seqlens, x0, workspace, dx, sel_rows, dx_sel,
                                   d ? d->layers - 1 : 0, 0, stream);
}

extern "C" int ce_tower_backward_range(const ce_tower_desc* d, int batch, int rows, const int* cu_seqlens,
                                       const void* x0, void* workspace, void* dx, const int* sel_rows,
                                       const float* dx_sel, int layer_hi, int layer_lo, void* stream) {
    TRY(check_desc(d, batch));
    TRY(check_rows(d, batch, rows, cu_seqlens));
    CE_CHECK_ARG(x0 && workspace && dx, "ce_tower_backward: null buffer");
    CE_CHECK_ARG(layer_lo >= 0 && layer_lo <= layer_hi && layer_hi < d->layers, "ce_tower_backward_range: layers %d..%d outside 0..%d",
                 layer_hi, layer_lo, d->layers - 1);
    Layout L;
    carve(d, batch, rows, workspace, L);
    const int M = rows, w = d->width;
    const bool b8 = (d->fp8 & 2) != 0;
    // Residual stream x (stash, x0) and gradient stream dx in element type ST; an fp16 gradient stream holds gradient * *GS
    // (ce_layernorm_bwd_t reads and writes it in those units, everything else -- dxb copies, parameter gradients -- is in
    // true units).  dx_sel and the compact buffers of the pruned block's first LayerNorm stay fp32.
    const int ST = d->stream16 ? CE_T_F16 : CE_T_F32;
    const void* q8_of = nullptr;
    const bool lnq = b8 && w % 128 == 0 && w <= 4096;              // fp8 input-gradient GEMMs: LayerNorm backward writes the e4m3 copy of dxb
    const float* GS = d->stream16 ? d->grad_scale : nullptr;        // device scalar (ce_grad_scale), set by the caller per pass
    CE_CHECK_ARG(!d->stream16 || GS, "ce_tower_backward: stream16 needs ce_tower_desc.grad_scale (device pointer)");
    const long esz = d->stream16 ? 2 : 4;
    // Per block l (buffer set q = l % WG_SETS): dxb_a = bf16 gradient at the block output (operand of mlp.c_proj's
    // dgrad/wgrad, written by block l+1's ln_1 backward), dxb_b = bf16 gradient at x_mid (attn.out_proj), da, dqkv.
    // They stay alive until the block's four queued weight gradients have gone out in a grouped launch (same
    // stream, so a later block may reuse a set as soon as that launch is ENQUEUED; the queue never holds more than
    // WG_MAX_BLOCKS blocks and a block only writes sets l and l-1).
    const int last = d->layers - 1;
    // The blocks' LayerNorm backwards write per-workgroup partial sums of d gamma / d beta / the projection bias gradient they
    // carry instead of adding them atomically (every workgroup of a launch adds into the same rows: 3.7 us of a 23 us launch);
    // one ce_layernorm_fold at the end of the call adds them into the gradients.  CE_LN_FOLD=0: the atomic form.
    static const int ln_fold = getenv("CE_LN_FOLD") ? atoi(getenv("CE_LN_FOLD")) : 1;
    ce_ln_fold_job fold_jobs[2 * LNP_RING];
    int n_fold = 0;
    int fold_rc = 0;
    const int ln_blocks = ce_layernorm_bwd_blocks(M, w);
    auto ln_bwd = [&](const void* dy, const void* xin, const float* mean, const float* rstd, const float* gamma, void* dxb, float* dgw, float* dgb,
                      float* dxs, void* q8, float* partial) -> int {
        if (!ln_fold)
            return ce_layernorm_bwd_q8(dy, CE_T_BF16, w, xin, ST, w, nullptr, mean, rstd, gamma, dx, ST, dx, ST, w, dxb, w, dgw, dgb, dxs, GS, M, w, q8, w,
                                       L.q8s, stream);
        fold_jobs[n_fold++] = ce_ln_fold_job{partial, dgw, dgb, dxs, ln_blocks, w};
        const int rc = ce_layernorm_bwd_partials(dy, CE_T_BF16, w, xin, ST, w, nullptr, mean, rstd, gamma, dx, ST, dx, ST, w, dxb, w, dxs ? 1 : 0, GS, M, w, q8, w,
                                                 L.q8s, partial, stream);
        if (rc == 0 && n_fold == 2 * LNP_RING) {               // ring full: fold now, the slots are free again (same stream)
            fold_rc = ce_layernorm_fold(fold_jobs, n_fold, stream);
            n_fold = 0;
            return fold_rc;
        }
        return rc;
    };
    // Where to cut the block sequence into grouped launches.  A launch of T unsplit 256x256 tiles takes ceil(T / 256)
    // rounds of the 256 CUs and every round costs a full tile time, so the cuts are chosen (small dynamic programme over
    // the blocks of this call) to minimise the total number of rounds: ViT-B/32 image tower, 108 tiles per block, 11
    // blocks below the pruned one: 7 blocks (756 tiles, 2.95 rounds) + 4 blocks and the pruned block's in_proj (459
    // tiles, 1.8 rounds) = 5 rounds; fixed groups of 5 took 6.  Shapes the 256x256 kernel does not take are counted in
    // 128x128 tiles over the 512 two-per-CU slots.  CE_WGRAD_GROUP = n forces groups of n blocks.
    static const int force_group = getenv("CE_WGRAD_GROUP") ? atoi(getenv("CE_WGRAD_GROUP")) : 0;
    bool cut_after[Layout::MAX_LAYERS] = {};
    auto plan_cuts = [&](int hi, int lo, long extra_tiles) {       // blocks hi .. lo (top-down) are about to be queued
        const int n = hi - lo + 1;
        if (n <= 0) return;
        const bool big = (w % 256 == 0) && M >= 2048;
        const long t = big ? 12L * (w / 256) * (w / 256) : 12L * ((w + 127) / 128) * ((w + 127) / 128);
        const long slots = big ? 256 : 512;
        long cost[Layout::MAX_LAYERS + 1];
        int take[Layout::MAX_LAYERS + 1];
        cost[0] = 0;
        for (int k = 1; k <= n; ++k) {                               // k blocks, counted from the BOTTOM of the range
            cost[k] = -1;
            for (int g = 1; g <= WG_MAX_BLOCKS && g <= k; ++g) {     // the topmost group of those k has g blocks
                if (force_group >= 1 && g != force_group && g != k) continue;
                const long tiles = g * t + (k == n ? extra_tiles : 0);   // the group that starts the range inherits the queue
                const long c = cost[k - g] + (tiles + slots - 1) / slots * 1000 + 1;   // rounds first, then fewer launches
                if (cost[k] < 0 || c < cost[k]) { cost[k] = c; take[k] = g; }
            }
        }
        int l = hi;
        for (int k = n; k > 0; k -= take[k]) {
            l -= take[k];
            cut_after[l + 1] = true;                                  // flush once block l+1 has been queued
        }
    };
    struct Pending {
        const void* P[CE_TN_MAX_GROUP]; long ldp[CE_TN_MAX_GROUP];
        const void* Q[CE_TN_MAX_GROUP]; long ldq[CE_TN_MAX_GROUP];
        int Nn[CE_TN_MAX_GROUP], Kk[CE_TN_MAX_GROUP];
        float* out[CE_TN_MAX_GROUP]; long ldo[CE_TN_MAX_GROUP];
        int count = 0, blocks = 0;
    } pend;
    auto queue = [&](const void* P, long ldp, const void* Q, long ldq, int Nn, int Kk, float* out, long ldo) {
        const int i = pend.count++;
        pend.P[i] = P; pend.ldp[i] = ldp; pend.Q[i] = Q; pend.ldq[i] = ldq;
        pend.Nn[i] = Nn; pend.Kk[i] = Kk; pend.out[i] = out; pend.ldo[i] = ldo;
    };
    auto flush = [&]() -> int {
        if (pend.count == 0) return 0;
        const int rc = ce_gemm_tn_grouped_ex(pend.count, pend.P, pend.ldp, pend.Q, pend.ldq, M, pend.Nn, pend.Kk, pend.out,
                                             pend.ldo, 0, d->wgrad_overwrite, stream);
        pend.count = 0;
        pend.blocks = 0;
        return rc;
    };
    int top = layer_hi;
    if (layer_hi < last) {
        // continuation of an earlier call: dx and the bf16 copy in set (layer_hi & 1) were left by block layer_hi+1
    } else if (sel_rows) {
        // ---- pruned last block (see ce_tower_forward): compact rows through the MLP and the out-projection ----
        CE_CHECK_ARG(dx_sel, "ce_tower_backward: pruned mode needs dx_sel");
        const int l = top, Bn = batch, q = l % WG_SETS;
        const ce_block_params& p = d->blocks[l];
        BlockStash& s = L.blk[l];
        const void* x_in = (l == 0) ? x0 : L.blk[l - 1].x_out;
        TRY(ce_cast_bf16(dx_sel, L.dxbs, (long)Bn * w, stream));
        TRY(linear(b8, L, q8_of, L.dxbs, w, p.wt_proj, p.wt8_proj, p.st8_proj, Bn, 4 * w, w, CE_EPI_GELUGRAD_BF16, nullptr, nullptr, 0, L.das, 4 * w, nullptr,
                       0, L.as, 4 * w, stream));
        TRY(ce_colsum_bf16(L.dxbs, w, p.g_b_proj, Bn, w, stream));
        TRY(linear(b8, L, q8_of, L.das, 4 * w, p.wt_fc, p.wt8_fc, p.st8_fc, Bn, w, 4 * w, CE_EPI_BF16, nullptr, nullptr, 0, L.dhs, w, nullptr, 0,
                       nullptr, 0, stream));
        TRY(ce_colsum_bf16(L.das, 4 * w, p.g_b_fc, Bn, 4 * w, stream));
        TRY(ce_layernorm_bwd_t(L.dhs, CE_T_BF16, w, L.xs_mid, ST, w, nullptr, L.means, L.rstds, p.ln2_w, dx_sel, CE_T_F32, L.dxs_mid,
                               ST, w, L.dxb2s, w, p.g_ln2_w, p.g_ln2_b, p.g_b_out, GS, Bn, w, stream));
        TRY(linear(b8, L, q8_of, L.dxb2s, w, p.wt_out, p.wt8_out, p.st8_out, Bn, w, w, CE_EPI_BF16, nullptr, nullptr, 0, L.dos, w, nullptr, 0, nullptr, 0,
                       stream));
        // attention sees dO only on the selected rows
        TRY(ce_scatter_rows_zero(L.dos, w * 2L, L.d_o, w * 2L, sel_rows, Bn, M, w * 2, stream));
        TRY(ce_attention_bwd(s.qkv, 3 * w, s.o, w, L.d_o, w, s.lse, L.dqkv[q], 3 * w, p.g_b_qkv, cu_seqlens, batch, d->tokens, d->heads,
                             d->causal, stream));
        {
            const void* P[3] = {L.dxbs, L.das, L.dxb2s};
            const long ldp[3] = {w, 4L * w, w};
            const void* Q[3] = {L.gs, L.h2s, L.os};
            const long ldq[3] = {4L * w, w, w};
            const int Nn[3] = {w, 4 * w, w};
            const int Kk[3] = {4 * w, w, w};
            float* out[3] = {p.g_w_proj, p.g_w_fc, p.g_w_out};
            const long ldo[3] = {4L * w, w, w};
            TRY(ce_gemm_tn_grouped_ex(3, P, ldp, Q, ldq, Bn, Nn, Kk, out, ldo, 0, d->wgrad_overwrite, stream));
        }
        queue(L.dqkv[q], 3L * w, s.h1, w, 3 * w, w, p.g_w_qkv, w);     // goes out with the next block(s)' gradients
        TRY(linear(b8, L, q8_of, L.dqkv[q], 3 * w, p.wt_qkv, p.wt8_qkv, p.st8_qkv, M, w, 3 * w, CE_EPI_BF16, nullptr, nullptr, 0, L.dh, w, nullptr, 0,
                       nullptr, 0, stream));
        // residual path: dx = scatter(dx at x_mid of the selected rows), then + ln_1 backward
        TRY(ce_scatter_rows_zero(L.dxs_mid, w * esz, dx, w * esz, sel_rows, Bn, M, (int)(w * esz), stream));
        TRY(ce_layernorm_bwd_t(L.dh, CE_T_BF16, w, x_in, ST, w, nullptr, s.mean1, s.rstd1, p.ln1_w, dx, ST, dx, ST, w,
                               L.dxb[(l + WG_SETS - 1) % WG_SETS], w, p.g_ln1_w, p.g_ln1_b,
                               (l > 0) ? d->blocks[l - 1].g_b_proj : nullptr, GS, M, w, stream));
        top = l - 1;
    } else {
        if (d->stream16) TRY(ce_cast_scaled(dx, ST, L.dxb[top % WG_SETS], CE_T_BF16, GS, 1, (long)M * w, stream));
        else TRY(ce_cast_bf16(reinterpret_cast<const float*>(dx), L.dxb[top % WG_SETS], (long)M * w, stream));
    }
    plan_cuts(top, layer_lo, pend.count > 0 ? ((w % 256 == 0 && M >= 2048) ? 3L * (w / 256) * (w / 256)
                                                                              : 3L * ((w + 127) / 128) * ((w + 127) / 128)) : 0);
    for (int l = top; l >= layer_lo; --l) {
        const ce_block_params& p = d->blocks[l];
        BlockStash& s = L.blk[l];
        const int q = l % WG_SETS;
        bf16_t *dxb_a = L.dxb[q], *dxb_b = L.dxb2[q], *da = L.da[q], *dqkv = L.dqkv[q];
        const void* x_in = (l == 0) ? x0 : L.blk[l - 1].x_out;
        // ---- mlp.c_proj : x_out = x_mid + g Wp^T + bp ----
        TRY(linear(b8, L, q8_of, dxb_a, w, p.wt_proj, p.wt8_proj, p.st8_proj, M, 4 * w, w, CE_EPI_GELUGRAD_BF16, nullptr, nullptr, 0, da, 4 * w, p.g_b_fc,
                       4 * w, s.a, 4 * w, stream));                               // da = (dx Wp) * a (the saved gelu'); g_b_fc += colsum(da)
        if (l == last) TRY(ce_colsum_bf16(dxb_a, w, p.g_b_proj, M, w, stream));   // lower blocks: fused in ln_1's backward
        // ---- mlp.c_fc : a = h2 Wf^T + bf ----
        TRY(linear(b8, L, q8_of, da, 4 * w, p.wt_fc, p.wt8_fc, p.st8_fc, M, w, 4 * w, CE_EPI_BF16, nullptr, nullptr, 0, L.dh, w, nullptr, 0,
                       nullptr, 0, stream));                                      // dh2 = da Wf
        // ---- ln_2 (+ residual); also the column sums of dx = attn.out_proj bias gradient ----
        TRY(ln_bwd(L.dh, s.x_mid, s.mean2, s.rstd2, p.ln2_w, dxb_b, p.g_ln2_w, p.g_ln2_b, p.g_b_out, lnq ? L.q8 : nullptr, L.lnp[l % LNP_RING][1]));
        if (lnq) q8_of = dxb_b;
        // ---- attn.out_proj : x_mid = x_in + o Wo^T + bo ----
        TRY(linear(b8, L, q8_of, dxb_b, w, p.wt_out, p.wt8_out, p.st8_out, M, w, w, CE_EPI_BF16, nullptr, nullptr, 0, L.d_o, w, nullptr, 0, nullptr, 0,
                       stream));                                                  // d_o = dx Wo
        // ---- attention core ----
        TRY(ce_attention_bwd(s.qkv, 3 * w, s.o, w, L.d_o, w, s.lse, dqkv, 3 * w, p.g_b_qkv, cu_seqlens, batch, d->tokens, d->heads,
                             d->causal, stream));
        // ---- the four weight gradients of this block: queued, launched with the neighbouring blocks' ----
        queue(dxb_a, w, s.g, 4L * w, w, 4 * w, p.g_w_proj, 4L * w);
        queue(da, 4L * w, s.h2, w, 4 * w, w, p.g_w_fc, w);
        queue(dxb_b, w, s.o, w, w, w, p.g_w_out, w);
        queue(dqkv, 3L * w, s.h1, w, 3 * w, w, p.g_w_qkv, w);
        ++pend.blocks;
        if (cut_after[l] || pend.count + 4 > CE_TN_MAX_GROUP) TRY(flush());
        // ---- attn.in_proj : qkv = h1 Wqkv^T + bqkv ----
        TRY(linear(b8, L, q8_of, dqkv, 3 * w, p.wt_qkv, p.wt8_qkv, p.st8_qkv, M, w, 3 * w, CE_EPI_BF16, nullptr, nullptr, 0, L.dh, w, nullptr, 0,
                       nullptr, 0, stream));                                      // dh1 = dqkv Wqkv
        // ---- ln_1 (+ residual); column sums of dx = previous block's mlp.c_proj bias gradient.  It writes the
        // next block's dxb (set l-1) ----
        TRY(ln_bwd(L.dh, x_in, s.mean1, s.rstd1, p.ln1_w, L.dxb[(l + WG_SETS - 1) % WG_SETS], p.g_ln1_w, p.g_ln1_b,
                   (l > 0) ? d->blocks[l - 1].g_b_proj : nullptr, (lnq && l > layer_lo) ? L.q8 : nullptr, L.lnp[l % LNP_RING][0]));
        if (lnq && l > layer_lo) q8_of = L.dxb[(l + WG_SETS - 1) % WG_SETS];     // consumed by the next block's GELU' GEMM in THIS call
    }
    // the caller hands the gradients of blocks >= layer_lo to the all-reduce as soon as this returns: nothing stays queued
    TRY(flush());
    if (n_fold > 0) TRY(ce_layernorm_fold(fold_jobs, n_fold, stream));
    return 0;
}

