// Small-batch contrastive head in THREE launches (gfx950).
//
// Between the towers' forward and their backward nothing else can run, and at BASELINE config 2's size (256 x 256 logits) the
// head of head.hip is 21 launches of 3-25 us each -- 200 us of a 12.4 ms step spent on 0.1 % of its FLOPs
// (profiles/r04_timeline_two_stream_one_step_v2.txt).  The same arithmetic -- feature normalisation, logits_per_image /
// logits_per_text over the batch (model_clip.py:496-521), CriterionContrastive 'ce' with index_pos (model_clip.py:633-662)
// and the whole backward down to the raw features -- in fp32, as there:
//   hs_norm_kernel    In = fi / |fi|, Tn = ft / |ft| (one wave per row); workgroup 0 zeroes the scalars and builds the
//                     inverse of index_pos
//   hs_logits_kernel  a workgroup owns 8 rows of logits_per_image (all nI rows) or of the SELECTED rows of logits_per_text:
//                     s q.k against every key, row log-sum-exp, the loss (mean over the rows), P = (softmax - onehot) / rows
//                     (the gradient of the loss w.r.t. the logits for a unit upstream gradient) and sum P * logits (d logit_scale)
//   hs_grad_kernel    a workgroup owns 8 rows of dIn = s G Tn or dTn = s G^T In with G = g_i P_i + g_t scatter(P_t)^T, the
//                     l2-normalisation backward in its epilogue (the whole row is in the workgroup), d logit_scale
// nI, nT, nsel <= 1024, E <= 1024 and a multiple of 4; beyond that engine.py uses head.hip or the fused infonce.hip.
#include <stdlib.h>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

constexpr int HS_MAX = 1024;       // largest nI / nT / nsel / E

struct HsLayout {
    size_t in, tn, inv_i, inv_t, p_i, p_t, scal, inv_sel, floats;
};
__host__ __device__ inline HsLayout hs_layout(int nI, int nT, int nsel, int E) {
    HsLayout L;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~size_t(3); return at; };
    L.in = take((size_t)nI * E);
    L.tn = take((size_t)nT * E);
    L.inv_i = take(nI);
    L.inv_t = take(nT);
    L.p_i = take((size_t)nI * nT);
    L.p_t = take((size_t)nsel * nI);
    L.scal = take(4);               // loss_i, loss_t, sum P_i * lpi, sum P_t * lpt
    L.inv_sel = take(nT);           // int: position r of text row j in index_pos, or -1
    L.floats = o;
    return L;
}

__global__ __launch_bounds__(256) void hs_norm_kernel(const float* __restrict__ fi, const float* __restrict__ ft, int nI, int nT,
                                                       int E, const int64_t* __restrict__ sel, int nsel, float* __restrict__ ws) {
    const HsLayout L = hs_layout(nI, nT, nsel, E);
    if (blockIdx.x == 0) {          // block-uniform: the scalars and the inverse of index_pos (one workgroup: no race)
        if (threadIdx.x < 4) ws[L.scal + threadIdx.x] = 0.f;
        int* inv = reinterpret_cast<int*>(ws + L.inv_sel);
        for (int j = threadIdx.x; j < nT; j += 256) inv[j] = sel ? -1 : (j < nsel ? j : -1);
        __syncthreads();
        if (sel)
            for (int r = threadIdx.x; r < nsel; r += 256) {
                const long j = sel[r];
                if (j >= 0 && j < nT) inv[j] = r;
            }
    }
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nI + nT) return;
    const float* src = r < nI ? fi + (long)r * E : ft + (long)(r - nI) * E;
    float* dst = r < nI ? ws + L.in + (long)r * E : ws + L.tn + (long)(r - nI) * E;
    float s = 0.f;
    for (int c = lane * 4; c < E; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
        s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float inv = 1.0f / sqrtf(wave_sum(s));          // no eps, as the reference (model_clip.py:497-498)
    for (int c = lane * 4; c < E; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c) * inv;
    if (lane == 0) (r < nI ? ws[L.inv_i + r] : ws[L.inv_t + (r - nI)]) = inv;
}

// sum over the 256 threads of a workgroup for HS_RB values at once (LDS scratch red[4][HS_RB]); every thread gets the sums
template <int HS_RB>
__device__ __forceinline__ void block_sum8(float (&v)[HS_RB], float* red, int lane, int wave) {
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) v[r] = wave_sum(v[r]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int r = 0; r < HS_RB; ++r) red[wave * HS_RB + r] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) v[r] = (red[r] + red[HS_RB + r]) + (red[2 * HS_RB + r] + red[3 * HS_RB + r]);
}

template <int HS_RB>       // rows per workgroup: 8, 4 or 2 -- the launcher takes the largest that still gives every CU a workgroup
__global__ __launch_bounds__(256) void hs_logits_kernel(int nI, int nT, int nsel, int E, const float* __restrict__ logit_scale,
                                                         const int64_t* __restrict__ labels_i, const int64_t* __restrict__ labels_t,
                                                         const int64_t* __restrict__ sel, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float hs_lds[];
    const HsLayout L = hs_layout(nI, nT, nsel, E);
    const int nblk_i = (nI + HS_RB - 1) / HS_RB;
    const bool img = (int)blockIdx.x < nblk_i;                       // block-uniform: image rows, or selected text rows
    const int r0 = (img ? blockIdx.x : blockIdx.x - nblk_i) * HS_RB;
    const int nrows = img ? nI : nsel, nkeys = img ? nT : nI;
    const float* Q = ws + (img ? L.in : L.tn);
    const float* Kmat = ws + (img ? L.tn : L.in);
    float* P = ws + (img ? L.p_i : L.p_t);
    const int64_t* labels = img ? labels_i : labels_t;
    const float s = __expf(*logit_scale);
    float* sq = hs_lds;                                             // [HS_RB][E] query rows
    float* sS = sq + HS_RB * E;                                     // [HS_RB][nkeys] logits
    float* red = sS + HS_RB * nkeys;                                // [4][HS_RB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid * 4; idx < HS_RB * E; idx += 1024) {
        const int rr = idx / E, c = idx - rr * E;
        const int r = r0 + rr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) {
            const long qrow = img ? r : (sel ? sel[r] : r);
            v = *reinterpret_cast<const f32x4*>(Q + qrow * E + c);
        }
        *reinterpret_cast<f32x4*>(sq + idx) = v;
    }
    __syncthreads();
    for (int j = tid; j < nkeys; j += 256) {
        const float* krow = Kmat + (long)j * E;
        float acc[HS_RB];
#pragma unroll
        for (int r = 0; r < HS_RB; ++r) acc[r] = 0.f;
        // eight 16-byte loads of the key row in flight per step (E % 4 == 0: the tail below takes what is left of a 32-column step)
        int c = 0;
        for (; c + 32 <= E; c += 32) {
            f32x4 k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k[u] = *reinterpret_cast<const f32x4*>(krow + c + 4 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < HS_RB; ++r) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(sq + r * E + c + 4 * u);      // broadcast read
                    acc[r] += (q[0] * k[u][0] + q[1] * k[u][1]) + (q[2] * k[u][2] + q[3] * k[u][3]);
                }
        }
        for (; c < E; c += 4) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(krow + c);
#pragma unroll
            for (int r = 0; r < HS_RB; ++r) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(sq + r * E + c);
                acc[r] += (q[0] * k[0] + q[1] * k[1]) + (q[2] * k[2] + q[3] * k[3]);
            }
        }
#pragma unroll
        for (int r = 0; r < HS_RB; ++r) sS[r * nkeys + j] = s * acc[r];
    }
    __syncthreads();
    // row-wise log-sum-exp over the keys (every thread takes a strided share of every row)
    float mx[HS_RB], sm[HS_RB];
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) {
        mx[r] = -INFINITY;
        for (int j = tid; j < nkeys; j += 256) mx[r] = fmaxf(mx[r], sS[r * nkeys + j]);
        mx[r] = wave_max(mx[r]);
    }
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int r = 0; r < HS_RB; ++r) red[wave * HS_RB + r] = mx[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) {
        mx[r] = fmaxf(fmaxf(red[r], red[HS_RB + r]), fmaxf(red[2 * HS_RB + r], red[3 * HS_RB + r]));
        sm[r] = 0.f;
        for (int j = tid; j < nkeys; j += 256) sm[r] += __expf(sS[r * nkeys + j] - mx[r]);
    }
    block_sum8<HS_RB>(sm, red, lane, wave);
    const float invn = 1.0f / (float)nrows;
    float loss = 0.f, dls = 0.f;
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) {
        if (r0 + r >= nrows) break;                                 // block-uniform
        const float lse = mx[r] + __logf(sm[r]);
        const long lab = labels[img ? (long)(r0 + r) : (sel ? sel[r0 + r] : (long)(r0 + r))];    // labels_per_text has one entry per TEXT row
        for (int j = tid; j < nkeys; j += 256) {
            const float lp = sS[r * nkeys + j];
            const float p = (__expf(lp - lse) - (j == lab ? 1.0f : 0.0f)) * invn;
            P[(long)(r0 + r) * nkeys + j] = p;
            dls += p * lp;
            if (j == lab) loss += (lse - lp) * invn;
        }
    }
    loss = wave_sum(loss);
    dls = wave_sum(dls);
    if (lane == 0) {
        atomicAdd(ws + L.scal + (img ? 0 : 1), loss);
        atomicAdd(ws + L.scal + (img ? 2 : 3), dls);
    }
}

template <int HS_RB>
__global__ __launch_bounds__(256) void hs_grad_kernel(int nI, int nT, int nsel, int E, const float* __restrict__ logit_scale,
                                                       const float* __restrict__ g_i, const float* __restrict__ g_t,
                                                       const int64_t* __restrict__ sel, const float* __restrict__ ws,
                                                       float* __restrict__ dfi, float* __restrict__ dft, float* __restrict__ dls_out) {
    extern __shared__ __attribute__((aligned(16))) float hs_lds[];
    const HsLayout L = hs_layout(nI, nT, nsel, E);
    const int nblk_i = (nI + HS_RB - 1) / HS_RB;
    const bool img = (int)blockIdx.x < nblk_i;
    const int r0 = (img ? blockIdx.x : blockIdx.x - nblk_i) * HS_RB;
    const int nrows = img ? nI : nT, nother = img ? nT : nI;
    const float gi = g_i ? *g_i : 0.f, gt = g_t ? *g_t : 0.f;
    const float s = __expf(*logit_scale);
    const float* P_i = ws + L.p_i;
    const float* P_t = ws + L.p_t;
    const int* inv_sel = reinterpret_cast<const int*>(ws + L.inv_sel);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && tid == 0 && dls_out) *dls_out = gi * ws[L.scal + 2] + gt * ws[L.scal + 3];
    // coefficient block, transposed: C[k][rr] = G[row r0 + rr][k] (image rows: k = text row j, plus -- behind them -- the nsel
    // selected text rows' P_t column block; text rows: k = image row i, both terms folded into one coefficient)
    float* C = hs_lds;                                              // [(nother + (img ? nsel : 0))][HS_RB]
    float* red = C + ((((size_t)(nother + (img ? nsel : 0)) * HS_RB) + 3) & ~size_t(3));       // 16-byte aligned: the slices behind it are read as float4
    if (img) {
        for (int idx = tid; idx < nT * HS_RB; idx += 256) {
            const int rr = idx / nT, j = idx - rr * nT;             // P_i rows are contiguous over j
            C[j * HS_RB + rr] = (r0 + rr < nI) ? gi * P_i[(long)(r0 + rr) * nT + j] : 0.f;
        }
        for (int idx = tid; idx < nsel * HS_RB; idx += 256) {
            const int r = idx / HS_RB, rr = idx - r * HS_RB;        // P_t[r][i0 .. i0 + 7]: 8 consecutive floats
            C[(nT + r) * HS_RB + rr] = (r0 + rr < nI) ? gt * P_t[(long)r * nI + r0 + rr] : 0.f;
        }
    } else {
        for (int idx = tid; idx < nI * HS_RB; idx += 256) {
            const int i = idx / HS_RB, jj = idx - i * HS_RB;        // P_i[i][j0 .. j0 + 7]: 8 consecutive floats
            const int j = r0 + jj;
            float c = 0.f;
            if (j < nT) {
                c = gi * P_i[(long)i * nT + j];
                const int r = inv_sel[j];
                if (r >= 0) c += gt * P_t[(long)r * nI + i];
            }
            C[i * HS_RB + jj] = c;
        }
    }
    __syncthreads();
    const float* other = ws + (img ? L.tn : L.in);                  // the rows the coefficients multiply
    const float* self = ws + (img ? L.in : L.tn);
    const float* inv_self = ws + (img ? L.inv_i : L.inv_t);
    float* out = img ? dfi : dft;
    // thread t owns columns 4 (t % ng) .. + 3 and the contraction slice t / ng of KS (ng = E / 4 column groups, KS = 256 / ng
    // rounded down to a power of two: E = 512 -> two slices); the slices are summed through LDS below.  Four rows of the other
    // matrix are requested per step so that their latencies overlap.
    const int ng = E / 4;
    int KS = 1;
    while (KS * 2 * ng <= 256) KS *= 2;
    const int grp = tid % ng, ks = tid / ng;
    const int c0 = grp * 4;
    const bool col_ok = tid < ng * KS;
    f32x4 acc[HS_RB];
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto sweep = [&](const float* Cc, int n, auto row_of) __attribute__((always_inline)) {
        const int per = (n + KS - 1) / KS, lo = ks * per, hi = min(n, lo + per);
        int k = lo;
        for (; k + 4 <= hi; k += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(other + row_of(k + u) * E + c0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* cp = Cc + (k + u) * HS_RB;                     // HS_RB consecutive floats: one or two wide broadcast reads
#pragma unroll
                for (int r = 0; r < HS_RB; ++r) acc[r] += v[u] * cp[r];
            }
        }
        for (; k < hi; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(other + row_of(k) * E + c0);
            const float* cp = Cc + k * HS_RB;
#pragma unroll
            for (int r = 0; r < HS_RB; ++r) acc[r] += v * cp[r];
        }
    };
    if (col_ok) {
        sweep(C, nother, [](int k) { return (long)k; });
        if (img) sweep(C + (size_t)nT * HS_RB, nsel, [&](int r) { return sel ? (long)sel[r] : (long)r; });
    }
    if (KS > 1) {                                                   // block-uniform: sum the contraction slices (slice 0 keeps the result)
        float* part = red + 4 * HS_RB;                              // [KS - 1][HS_RB][E] behind the reduction scratch
        __syncthreads();
        if (col_ok && ks > 0)
#pragma unroll
            for (int r = 0; r < HS_RB; ++r) *reinterpret_cast<f32x4*>(part + ((size_t)(ks - 1) * HS_RB + r) * E + c0) = acc[r];
        __syncthreads();
        if (col_ok && ks == 0)
            for (int q = 1; q < KS; ++q)
#pragma unroll
                for (int r = 0; r < HS_RB; ++r) acc[r] += *reinterpret_cast<const f32x4*>(part + ((size_t)(q - 1) * HS_RB + r) * E + c0);
    }
    const bool owner = col_ok && ks == 0;
    // l2-normalisation backward: df = inv * (dn - n <n, dn>), dn = s * acc
    float dot[HS_RB];
    f32x4 nv[HS_RB];
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) {
        nv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        dot[r] = 0.f;
        acc[r] *= s;
        if (owner && r0 + r < nrows) {
            nv[r] = *reinterpret_cast<const f32x4*>(self + (long)(r0 + r) * E + c0);
            dot[r] = (nv[r][0] * acc[r][0] + nv[r][1] * acc[r][1]) + (nv[r][2] * acc[r][2] + nv[r][3] * acc[r][3]);
        }
    }
    block_sum8<HS_RB>(dot, red, lane, wave);
    if (owner) {
#pragma unroll
        for (int r = 0; r < HS_RB; ++r) {
            if (r0 + r >= nrows) break;
            const float inv = inv_self[r0 + r];
            *reinterpret_cast<f32x4*>(out + (long)(r0 + r) * E + c0) = (acc[r] - nv[r] * dot[r]) * inv;
        }
    }
}

}  // namespace

extern "C" size_t ce_head_small_workspace_floats(int nI, int nT, int nsel, int E) {
    if (nI <= 0 || nT <= 0 || nsel <= 0 || E <= 0) return 0;
    return hs_layout(nI, nT, nsel, E).floats;
}

extern "C" size_t ce_head_small_scalars_offset(int nI, int nT, int nsel, int E) {
    if (nI <= 0 || nT <= 0 || nsel <= 0 || E <= 0) return 0;
    return hs_layout(nI, nT, nsel, E).scal;
}

static void hs_attributes() {
    static bool done = false;
    if (done) return;
    hipFuncSetAttribute(reinterpret_cast<const void*>(hs_logits_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hs_logits_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hs_logits_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hs_grad_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hs_grad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hs_grad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    done = true;
}
// rows per workgroup: the largest of 8 / 4 / 2 that still makes `want` workgroups out of `rows` rows (CE_HEAD_SMALL_RB forces one)
static int hs_rows_per_wg(int rows, int want) {
    static const int forced = getenv("CE_HEAD_SMALL_RB") ? atoi(getenv("CE_HEAD_SMALL_RB")) : 0;
    if (forced == 8 || forced == 4 || forced == 2) return forced;
    for (int rb = 8; rb > 2; rb >>= 1)
        if ((rows + rb - 1) / rb >= want) return rb;
    return 2;
}

static int hs_check(int nI, int nT, int nsel, int E) {
    CE_CHECK_ARG(nI > 0 && nT > 0 && nsel > 0 && nI <= HS_MAX && nT <= HS_MAX && nsel <= nT,
                 "ce_head_small: need 1 <= nI, nT <= %d and nsel <= nT (nI=%d nT=%d nsel=%d)", HS_MAX, nI, nT, nsel);
    CE_CHECK_ARG(E > 0 && E <= HS_MAX && E % 4 == 0, "ce_head_small: E=%d must be a multiple of 4, <= %d", E, HS_MAX);
    return 0;
}

extern "C" int ce_head_small_fwd(const float* fi, const float* ft, int nI, int nT, int E, const float* logit_scale,
                                 const int64_t* labels_i, const int64_t* labels_t, const int64_t* sel, int nsel,
                                 float* workspace, void* stream) {
    if (hs_check(nI, nT, nsel, E) != 0) return -22;
    CE_CHECK_ARG(fi && ft && logit_scale && labels_i && labels_t && workspace, "ce_head_small_fwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hs_norm_kernel, dim3(ce_div_up(nI + nT, 4)), dim3(256), 0, s, fi, ft, nI, nT, E, sel, nsel, workspace);
    const int nkeys = nI > nT ? nI : nT;
    hs_attributes();
    // (bound by its per-thread key-row reads, not by rows per workgroup: 38.9 us at 8 rows, 45.7 at 2 for 256 + 256 rows)
    const int rb = hs_rows_per_wg(nI + nsel, 48);
    const size_t lds = ((size_t)rb * E + (size_t)rb * nkeys + 4 * rb) * sizeof(float);
    const dim3 grid(ce_div_up(nI, rb) + ce_div_up(nsel, rb));
    if (rb == 8) hipLaunchKernelGGL(hs_logits_kernel<8>, grid, dim3(256), lds, s, nI, nT, nsel, E, logit_scale, labels_i, labels_t, sel, workspace);
    else if (rb == 4) hipLaunchKernelGGL(hs_logits_kernel<4>, grid, dim3(256), lds, s, nI, nT, nsel, E, logit_scale, labels_i, labels_t, sel, workspace);
    else hipLaunchKernelGGL(hs_logits_kernel<2>, grid, dim3(256), lds, s, nI, nT, nsel, E, logit_scale, labels_i, labels_t, sel, workspace);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_head_small_bwd(int nI, int nT, int nsel, int E, const float* logit_scale, const float* g_i, const float* g_t,
                                 const int64_t* sel, const float* workspace, float* dfi, float* dft, float* dlogit_scale,
                                 void* stream) {
    if (hs_check(nI, nT, nsel, E) != 0) return -22;
    CE_CHECK_ARG(logit_scale && workspace && dfi && dft, "ce_head_small_bwd: null argument");
    hs_attributes();
    const int nmax = nI > nT ? nI : nT;
    int KS = 1;
    while (KS * 2 * (E / 4) <= 256) KS *= 2;
    // (41.7 us at 8 rows per workgroup, 30 at 2 for 256 + 256 rows: more workgroups, shorter accumulator chains)
    const int rb = hs_rows_per_wg(nI + nT, 192);
    const size_t lds = ((size_t)(nmax + nsel) * rb + 4 + 4 * rb + (size_t)(KS - 1) * rb * E) * sizeof(float);
    CE_CHECK_ARG(lds <= 128 * 1024, "ce_head_small_bwd: %zu bytes of LDS for nI=%d nT=%d nsel=%d E=%d", lds, nI, nT, nsel, E);
    const dim3 grid(ce_div_up(nI, rb) + ce_div_up(nT, rb));
    hipStream_t st = (hipStream_t)stream;
    if (rb == 8) hipLaunchKernelGGL(hs_grad_kernel<8>, grid, dim3(256), lds, st, nI, nT, nsel, E, logit_scale, g_i, g_t, sel, workspace, dfi, dft, dlogit_scale);
    else if (rb == 4) hipLaunchKernelGGL(hs_grad_kernel<4>, grid, dim3(256), lds, st, nI, nT, nsel, E, logit_scale, g_i, g_t, sel, workspace, dfi, dft, dlogit_scale);
    else hipLaunchKernelGGL(hs_grad_kernel<2>, grid, dim3(256), lds, st, nI, nT, nsel, E, logit_scale, g_i, g_t, sel, workspace, dfi, dft, dlogit_scale);
    CE_LAUNCH_CHECK();
    return 0;
}
