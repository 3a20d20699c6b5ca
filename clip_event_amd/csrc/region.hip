// Region / argument InfoNCE of the train_arg branch (model_clip.py:456-488), every image of the batch in ONE launch.
//
// Per image with n usable boxes the reference forms, from the pooled region features r [n,E], the role-description
// features d [n,E] and (train_arg = desc_type*) the role-label features l [n,E]:
//     loss_per_bbox += CE(s r^ d^T, arange n) [+ CE(s r^ l^T, arange n)]
//     loss_per_arg  += CE(s d^ r^T, arange n) [+ CE(s l^ r^T, arange n)] [+ CE(s d^ l^T, arange n)]   (desc_type_text)
// (x^ = x / |x|, s = exp(logit_scale), CE = mean cross-entropy over the n rows).  The n x n problems are tiny and
// ragged, so one wave takes one image: ce_region_nce_fwd reads the three feature matrices through per-image offsets
// and adds the two losses; ce_region_nce_bwd recomputes the n x n matrices and writes the feature gradients (through
// the normalisation) and the logit_scale gradient.  Launch count is independent of the batch size; fp32 throughout.
#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

constexpr int RG_MAX = 16;     // boxes per image

__device__ __forceinline__ float row_dot(const float* __restrict__ a, const float* __restrict__ b, int E, int lane) {
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += a[c] * b[c];
    return wave_sum(s);
}

struct RegionArgs {
    const float* r; const float* d; const float* l;       // [R,E] each (l nullable)
    const int* offsets;                                    // [G+1] first row of every image's group
    const float* logit_scale;
    int G, E, use_label, role_text;
};

// S[i][j] = s <a_i, b_j> inva_i invb_j into LDS
__device__ __forceinline__ void sim_matrix(const float* a, const float* b, const float* inva, const float* invb, int n, int E,
                                           float s, float (*S)[RG_MAX], int lane) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const float v = row_dot(a + (long)i * E, b + (long)j * E, E, lane);
            if (lane == 0) S[i][j] = s * v * inva[i] * invb[j];
        }
}

// mean over rows i of (logsumexp_j S[i][j] - S[i][i]); `cols`: the same on the transpose
__device__ __forceinline__ float ce_diag(float (*S)[RG_MAX], int n, bool cols, int lane) {
    float v = 0.f;
    if (lane < n) {
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, cols ? S[j][lane] : S[lane][j]);
        float sum = 0.f;
        for (int j = 0; j < n; ++j) sum += __expf((cols ? S[j][lane] : S[lane][j]) - mx);
        v = (mx + __logf(sum) - S[lane][lane]) / (float)n;
    }
    return wave_sum(v);
}

__global__ __launch_bounds__(64) void region_nce_fwd_kernel(RegionArgs p, float* __restrict__ loss_bbox,
                                                            float* __restrict__ loss_arg) {
    __shared__ float S[RG_MAX][RG_MAX];
    __shared__ float inv[3][RG_MAX];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int r0 = p.offsets[g], n = p.offsets[g + 1] - r0;
    if (n <= 0) return;
    const int E = p.E;
    const float s = __expf(*p.logit_scale);
    const float* R = p.r + (long)r0 * E;
    const float* D = p.d + (long)r0 * E;
    const float* Lb = p.l ? p.l + (long)r0 * E : nullptr;
    for (int i = 0; i < n; ++i) {
        const float a = row_dot(R + (long)i * E, R + (long)i * E, E, lane);
        const float b = row_dot(D + (long)i * E, D + (long)i * E, E, lane);
        const float c = Lb ? row_dot(Lb + (long)i * E, Lb + (long)i * E, E, lane) : 1.f;
        if (lane == 0) { inv[0][i] = 1.0f / sqrtf(a); inv[1][i] = 1.0f / sqrtf(b); inv[2][i] = 1.0f / sqrtf(c); }
    }
    __syncthreads();
    float lb = 0.f, la = 0.f;
    sim_matrix(R, D, inv[0], inv[1], n, E, s, S, lane);
    __syncthreads();
    lb += ce_diag(S, n, false, lane);
    la += ce_diag(S, n, true, lane);
    __syncthreads();
    if (p.use_label && Lb) {
        sim_matrix(R, Lb, inv[0], inv[2], n, E, s, S, lane);
        __syncthreads();
        lb += ce_diag(S, n, false, lane);
        la += ce_diag(S, n, true, lane);
        __syncthreads();
        if (p.role_text) {
            sim_matrix(D, Lb, inv[1], inv[2], n, E, s, S, lane);
            __syncthreads();
            la += ce_diag(S, n, false, lane);
        }
    }
    if (lane == 0) {
        atomicAdd(loss_bbox, lb);
        atomicAdd(loss_arg, la);
    }
}

// dS[i][j] = wr (softmax_row(S)[i][j] - [i==j]) / n + wc (softmax_col(S)[i][j] - [i==j]) / n, in place; returns sum dS*S
__device__ __forceinline__ float ce_diag_grad(float (*S)[RG_MAX], float (*dS)[RG_MAX], int n, float wr, float wc, int lane) {
    if (lane < n) {        // lane = row i: row softmax
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, S[lane][j]);
        float sum = 0.f;
        for (int j = 0; j < n; ++j) sum += __expf(S[lane][j] - mx);
        for (int j = 0; j < n; ++j) dS[lane][j] = wr * (__expf(S[lane][j] - mx) / sum - (j == lane ? 1.f : 0.f)) / (float)n;
    }
    __syncthreads();
    if (lane < n && wc != 0.f) {   // lane = column j: column softmax
        float mx = -INFINITY;
        for (int i = 0; i < n; ++i) mx = fmaxf(mx, S[i][lane]);
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += __expf(S[i][lane] - mx);
        for (int i = 0; i < n; ++i) dS[i][lane] += wc * (__expf(S[i][lane] - mx) / sum - (i == lane ? 1.f : 0.f)) / (float)n;
    }
    __syncthreads();
    float t = 0.f;
    for (int idx = lane; idx < n * n; idx += 64) t += dS[idx / n][idx % n] * S[idx / n][idx % n];
    return wave_sum(t);
}

// grad_a[i][e] += s inva_i sum_j dS[i][j] invb_j b_j[e]   (gradient w.r.t. the NORMALISED a_i; transpose = by columns)
__device__ __forceinline__ void accum_rows(float* __restrict__ ga, const float* __restrict__ b, const float* invb,
                                           float (*dS)[RG_MAX], bool transpose, int n, int E, float s, int lane) {
    for (int i = 0; i < n; ++i)
        for (int c = lane; c < E; c += 64) {
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += (transpose ? dS[j][i] : dS[i][j]) * invb[j] * b[(long)j * E + c];
            ga[(long)i * E + c] += s * acc;
        }
}

// in place: g_i <- (g_i - x^_i <g_i, x^_i>) inv_i   (x^ = x inv)
__device__ __forceinline__ void through_norm(float* __restrict__ gx, const float* __restrict__ x, const float* inv, int n, int E,
                                             int lane) {
    for (int i = 0; i < n; ++i) {
        const float iv = inv[i];
        const float t = row_dot(gx + (long)i * E, x + (long)i * E, E, lane) * iv;      // <g, x^>
        for (int c = lane; c < E; c += 64) gx[(long)i * E + c] = (gx[(long)i * E + c] - x[(long)i * E + c] * iv * t) * iv;
    }
}

__global__ __launch_bounds__(64) void region_nce_bwd_kernel(RegionArgs p, const float* __restrict__ g_bbox,
                                                            const float* __restrict__ g_arg, float* __restrict__ dr,
                                                            float* __restrict__ dd, float* __restrict__ dl,
                                                            float* __restrict__ dls) {
    __shared__ float S[RG_MAX][RG_MAX];
    __shared__ float dS[RG_MAX][RG_MAX];
    __shared__ float inv[3][RG_MAX];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int r0 = p.offsets[g], n = p.offsets[g + 1] - r0;
    if (n <= 0) return;
    const int E = p.E;
    const float s = __expf(*p.logit_scale);
    const float gb = *g_bbox, ga = *g_arg;
    const float* R = p.r + (long)r0 * E;
    const float* D = p.d + (long)r0 * E;
    const float* Lb = p.l ? p.l + (long)r0 * E : nullptr;
    float* dR = dr + (long)r0 * E;
    float* dD = dd + (long)r0 * E;
    float* dL = dl ? dl + (long)r0 * E : nullptr;
    for (int i = 0; i < n; ++i) {
        const float a = row_dot(R + (long)i * E, R + (long)i * E, E, lane);
        const float b = row_dot(D + (long)i * E, D + (long)i * E, E, lane);
        const float c = Lb ? row_dot(Lb + (long)i * E, Lb + (long)i * E, E, lane) : 1.f;
        if (lane == 0) { inv[0][i] = 1.0f / sqrtf(a); inv[1][i] = 1.0f / sqrtf(b); inv[2][i] = 1.0f / sqrtf(c); }
    }
    // the gradient buffers of this image's rows start at zero (the caller zero-fills): accumulate w.r.t. x^ first
    __syncthreads();
    float dscale = 0.f;
    sim_matrix(R, D, inv[0], inv[1], n, E, s, S, lane);
    __syncthreads();
    dscale += ce_diag_grad(S, dS, n, gb, ga, lane);
    accum_rows(dR, D, inv[1], dS, false, n, E, s, lane);
    accum_rows(dD, R, inv[0], dS, true, n, E, s, lane);
    __syncthreads();
    if (p.use_label && Lb) {
        sim_matrix(R, Lb, inv[0], inv[2], n, E, s, S, lane);
        __syncthreads();
        dscale += ce_diag_grad(S, dS, n, gb, ga, lane);
        accum_rows(dR, Lb, inv[2], dS, false, n, E, s, lane);
        accum_rows(dL, R, inv[0], dS, true, n, E, s, lane);
        __syncthreads();
        if (p.role_text) {
            sim_matrix(D, Lb, inv[1], inv[2], n, E, s, S, lane);
            __syncthreads();
            dscale += ce_diag_grad(S, dS, n, ga, 0.f, lane);
            accum_rows(dD, Lb, inv[2], dS, false, n, E, s, lane);
            accum_rows(dL, D, inv[1], dS, true, n, E, s, lane);
            __syncthreads();
        }
    }
    // the projection's dot products read gradient columns other lanes have just written
    __syncthreads();
    through_norm(dR, R, inv[0], n, E, lane);
    through_norm(dD, D, inv[1], n, E, lane);
    if (dL && p.use_label) through_norm(dL, Lb, inv[2], n, E, lane);
    if (lane == 0) atomicAdd(dls, dscale);
}

int check(const RegionArgs& a, const char* what) {
    CE_CHECK_ARG(a.r && a.d && a.offsets && a.logit_scale, "%s: null buffer", what);
    CE_CHECK_ARG(a.G > 0 && a.E > 0, "%s: empty problem", what);
    CE_CHECK_ARG(!a.use_label || a.l, "%s: train_arg = desc_type* needs the label features", what);
    return 0;
}

}  // namespace

extern "C" int ce_region_nce_fwd(const float* region, const float* desc, const float* label, const int* offsets, int groups,
                                 int max_rows, int E, const float* logit_scale, int use_label, int role_text, float* loss_bbox,
                                 float* loss_arg, void* stream) {
    RegionArgs a{region, desc, label, offsets, logit_scale, groups, E, use_label, role_text};
    if (int rc = check(a, "ce_region_nce_fwd")) return rc;
    CE_CHECK_ARG(max_rows >= 1 && max_rows <= RG_MAX, "ce_region_nce_fwd: %d boxes in one image (at most %d)", max_rows, RG_MAX);
    CE_CHECK_ARG(loss_bbox && loss_arg, "ce_region_nce_fwd: null output");
    hipLaunchKernelGGL(region_nce_fwd_kernel, dim3(groups), dim3(64), 0, (hipStream_t)stream, a, loss_bbox, loss_arg);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_region_nce_bwd(const float* region, const float* desc, const float* label, const int* offsets, int groups,
                                 int max_rows, int E, const float* logit_scale, int use_label, int role_text,
                                 const float* g_bbox, const float* g_arg, float* dregion, float* ddesc, float* dlabel,
                                 float* dlogit_scale, void* stream) {
    RegionArgs a{region, desc, label, offsets, logit_scale, groups, E, use_label, role_text};
    if (int rc = check(a, "ce_region_nce_bwd")) return rc;
    CE_CHECK_ARG(max_rows >= 1 && max_rows <= RG_MAX, "ce_region_nce_bwd: %d boxes in one image (at most %d)", max_rows, RG_MAX);
    CE_CHECK_ARG(g_bbox && g_arg && dregion && ddesc && dlogit_scale && (!use_label || dlabel), "ce_region_nce_bwd: null buffer");
    hipLaunchKernelGGL(region_nce_bwd_kernel, dim3(groups), dim3(64), 0, (hipStream_t)stream, a, g_bbox, g_arg, dregion, ddesc,
                       dlabel, dlogit_scale);
    CE_LAUNCH_CHECK();
    return 0;
}
