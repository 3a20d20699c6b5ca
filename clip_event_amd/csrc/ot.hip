// Optimal-transport graph-alignment loss and the region (bbox) pooling of the argument branch.
//
//   ce_ot_fwd / ce_ot_bwd      cosine cost + IPOT(beta=0.5, 50 outer x 1 inner) + trace(C T)
//                              reference: model_ot.py:8-83 (cost_matrix_cosine, ipot, trace,
//                              optimal_transport_dist), called from CriterionAlignment
//                              (model_clip.py:679-715).  T is computed without gradient (the
//                              reference runs ipot under no_grad and detaches T), so the backward
//                              only flows through the cost: dC[m][n] = g * T[n][m].
//   ce_bbox_pool_fwd / _bwd    mean of grid features over an integer patch box
//                              (model_clip.py:438-443, utils_image.py:28-32)
//
// The reference spends ~8 tiny launches per IPOT iteration (400 per step); here one workgroup
// per sample keeps C, A, T, sigma, delta in LDS for all 50 iterations: latency-bound work
// becomes one launch.  M, N <= 64 entities / objects per sample.
#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

constexpr int OT_MAX = 64;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ot_fwd_kernel(const float* __restrict__ x, long xsb, long xsr,
                                                     const float* __restrict__ y, long ysb, long ysr,
                                                     const unsigned char* __restrict__ xpad,
                                                     const unsigned char* __restrict__ ypad, float* __restrict__ dist,
                                                     float* __restrict__ Tout, float* __restrict__ xinv_out,
                                                     float* __restrict__ yinv_out, int M, int N, int D, float beta,
                                                     int iters, float eps) {
    __shared__ float C[OT_MAX * OT_MAX];   // [m][n]
    __shared__ float A[OT_MAX * OT_MAX];   // [n][m]
    __shared__ float T[OT_MAX * OT_MAX];   // [n][m]
    __shared__ float sigma[OT_MAX], delta[OT_MAX], xinv[OT_MAX], yinv[OT_MAX];
    __shared__ unsigned char xp[OT_MAX], yp[OT_MAX];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (long)b * xsb;
    const float* yb = y + (long)b * ysb;
    // 1. row norms (F.normalize: x / max(||x||, eps)), one wave per row
    for (int r = wave; r < M + N; r += 4) {
        const float* row = r < M ? xb + (long)r * xsr : yb + (long)(r - M) * ysr;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += row[d] * row[d];
        s = wave_sum(s);
        const float inv = 1.0f / fmaxf(sqrtf(s), eps);
        if (lane == 0) {
            if (r < M) xinv[r] = inv; else yinv[r - M] = inv;
        }
    }
    if (tid < M) xp[tid] = xpad[(long)b * M + tid];
    if (tid < N) yp[tid] = ypad[(long)b * N + tid];
    __syncthreads();
    float xlen = 0.f, ylen = 0.f;
    for (int m = 0; m < M; ++m) xlen += xp[m] ? 0.f : 1.f;
    for (int n = 0; n < N; ++n) ylen += yp[n] ? 0.f : 1.f;
    // 2. cosine cost, padded pairs -> 0 (model_ot.py:14-18, :74-75); A = exp(-C^T/beta), T = 1, both 0 on pads
    for (int idx = tid; idx < M * N; idx += 256) {
        const int m = idx / N, n = idx - m * N;
        const float* xr = xb + (long)m * xsr;
        const float* yr = yb + (long)n * ysr;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += xr[d] * yr[d];
        const bool jp = xp[m] || yp[n];
        const float c = jp ? 0.f : 1.0f - s * xinv[m] * yinv[n];
        C[m * N + n] = c;
        A[n * M + m] = jp ? 0.f : __expf(-c / beta);
        T[n * M + m] = jp ? 0.f : 1.f;
    }
    if (tid < M) sigma[tid] = xp[tid] ? 0.f : 1.0f / xlen;
    __syncthreads();
    // 3. IPOT (model_ot.py:55-61): Q = A*T; delta = 1/(y_len Q sigma + y_mask); sigma = 1/(x_len delta Q + x_mask)
    for (int it = 0; it < iters; ++it) {
        if (tid < N) {
            float s = 0.f;
            for (int m = 0; m < M; ++m) s += A[tid * M + m] * T[tid * M + m] * sigma[m];
            delta[tid] = 1.0f / (ylen * s + (yp[tid] ? 1e4f : 0.f));
        }
        __syncthreads();
        if (tid < M) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += delta[n] * (A[n * M + tid] * T[n * M + tid]);
            sigma[tid] = 1.0f / (xlen * s + (xp[tid] ? 1e4f : 0.f));
        }
        __syncthreads();
        for (int idx = tid; idx < M * N; idx += 256) {
            const int n = idx / M, m = idx - n * M;
            T[idx] = delta[n] * (A[idx] * T[idx]) * sigma[m];
        }
        __syncthreads();
    }
    // final mask (model_ot.py:62) and distance = trace(C @ T) = sum_m sum_n C[m][n] T[n][m]
    float part = 0.f;
    for (int idx = tid; idx < M * N; idx += 256) {
        const int n = idx / M, m = idx - n * M;
        const float t = (xp[m] || yp[n]) ? 0.f : T[idx];
        Tout[(long)b * M * N + idx] = t;
        part += C[m * N + n] * t;
    }
    const float total = block_sum(part, red);
    if (tid == 0) dist[b] = total;
    if (tid < M) xinv_out[(long)b * M + tid] = xinv[tid];
    if (tid < N) yinv_out[(long)b * N + tid] = yinv[tid];
}

// dx_m = P_m( -sum_n g T[n][m] yhat_n ),  P_m(v) = xinv (v - xhat <v, xhat>)  (xinv v when the norm was clamped)
__global__ __launch_bounds__(256) void ot_bwd_kernel(const float* __restrict__ x, long xsb, long xsr,
                                                     const float* __restrict__ y, long ysb, long ysr,
                                                     const float* __restrict__ Tin, const float* __restrict__ xinv_in,
                                                     const float* __restrict__ yinv_in, const float* __restrict__ g,
                                                     float* __restrict__ dx, float* __restrict__ dy, int M, int N, int D,
                                                     float eps) {
    __shared__ float T[OT_MAX * OT_MAX];
    __shared__ float xinv[OT_MAX], yinv[OT_MAX];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gb = g[b];
    for (int idx = tid; idx < M * N; idx += 256) T[idx] = Tin[(long)b * M * N + idx] * gb;
    if (tid < M) xinv[tid] = xinv_in[(long)b * M + tid];
    if (tid < N) yinv[tid] = yinv_in[(long)b * N + tid];
    __syncthreads();
    const float* xb = x + (long)b * xsb;
    const float* yb = y + (long)b * ysb;
    const float clamp_inv = 1.0f / eps;
    for (int side = 0; side < 2; ++side) {
        const int R = side == 0 ? M : N, Cn = side == 0 ? N : M;
        for (int r = 0; r < R; ++r) {
            const float* self = side == 0 ? xb + (long)r * xsr : yb + (long)r * ysr;
            const float sinv = side == 0 ? xinv[r] : yinv[r];
            float v[4], sh[4];       // up to D = 1024 columns per 256 threads
            float part = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = tid + 256 * k;
                v[k] = 0.f;
                sh[k] = 0.f;
                if (d < D) {
                    float acc = 0.f;
                    for (int c = 0; c < Cn; ++c) {
                        const float t = side == 0 ? T[c * M + r] : T[r * M + c];
                        const float* other = side == 0 ? yb + (long)c * ysr : xb + (long)c * xsr;
                        const float oinv = side == 0 ? yinv[c] : xinv[c];
                        acc += t * other[d] * oinv;
                    }
                    v[k] = -acc;
                    sh[k] = self[d] * sinv;
                    part += v[k] * sh[k];
                }
            }
            const float s = block_sum(part, red);
            const bool clamped = sinv >= clamp_inv;   // ||row|| <= eps: denominator was the constant eps
            float* out = side == 0 ? dx + ((long)b * M + r) * D : dy + ((long)b * N + r) * D;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = tid + 256 * k;
                if (d < D) out[d] = clamped ? sinv * v[k] : sinv * (v[k] - sh[k] * s);
            }
        }
    }
}

// boxes[i] = {image, x0, y0, x1, y1}; out[i,:] = mean over grid[image, x0:x1, y0:y1, :] (first grid axis is
// indexed by the x range: the reference's quirk, model_clip.py:439)
__global__ void bbox_pool_fwd_kernel(const float* __restrict__ grid, long sb, long s0, long s1, const int* __restrict__ boxes,
                                     float* __restrict__ out, int nbox, int E) {
    const int i = blockIdx.x;
    const int img = boxes[i * 5], x0 = boxes[i * 5 + 1], y0 = boxes[i * 5 + 2], x1 = boxes[i * 5 + 3], y1 = boxes[i * 5 + 4];
    const float cnt = (float)((x1 - x0) * (y1 - y0));
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float s = 0.f;
        for (int a = x0; a < x1; ++a)
            for (int c = y0; c < y1; ++c) s += grid[(long)img * sb + (long)a * s0 + (long)c * s1 + e];
        out[(long)i * E + e] = s / cnt;      // empty box -> 0/0 = NaN, like torch.mean of an empty slice
    }
}

__global__ void bbox_pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ boxes, float* __restrict__ dgrid,
                                     int g, int nbox, int E) {
    const int i = blockIdx.x;
    const int img = boxes[i * 5], x0 = boxes[i * 5 + 1], y0 = boxes[i * 5 + 2], x1 = boxes[i * 5 + 3], y1 = boxes[i * 5 + 4];
    const float inv = 1.0f / (float)((x1 - x0) * (y1 - y0));
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const float v = dout[(long)i * E + e] * inv;
        for (int a = x0; a < x1; ++a)
            for (int c = y0; c < y1; ++c) atomicAdd(dgrid + (((long)img * g + a) * g + c) * E + e, v);
    }
}

}  // namespace

extern "C" int ce_ot_fwd(const float* txt, long tsb, long tsr, const float* img, long isb, long isr,
                         const unsigned char* txt_pad, const unsigned char* img_pad, float* dist, float* T,
                         float* txt_inv, float* img_inv, int B, int M, int N, int D, float beta, int iters,
                         void* stream) {
    CE_CHECK_ARG(B > 0 && M > 0 && N > 0 && D > 0, "ce_ot_fwd: empty problem");
    CE_CHECK_ARG(M <= OT_MAX && N <= OT_MAX, "ce_ot_fwd: at most %d entities/objects per sample (M=%d N=%d)", OT_MAX, M, N);
    hipLaunchKernelGGL(ot_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, txt, tsb, tsr, img, isb, isr, txt_pad,
                       img_pad, dist, T, txt_inv, img_inv, M, N, D, beta, iters, 1e-5f);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_ot_bwd(const float* txt, long tsb, long tsr, const float* img, long isb, long isr, const float* T,
                         const float* txt_inv, const float* img_inv, const float* grad, float* dtxt, float* dimg, int B,
                         int M, int N, int D, void* stream) {
    CE_CHECK_ARG(B > 0 && M > 0 && N > 0 && D > 0 && D <= 1024, "ce_ot_bwd: bad shape (D <= 1024)");
    CE_CHECK_ARG(M <= OT_MAX && N <= OT_MAX, "ce_ot_bwd: at most %d entities/objects per sample", OT_MAX);
    hipLaunchKernelGGL(ot_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, txt, tsb, tsr, img, isb, isr, T, txt_inv,
                       img_inv, grad, dtxt, dimg, M, N, D, 1e-5f);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_bbox_pool_fwd(const float* grid, long sb, long s0, long s1, const int* boxes, float* out, int nbox,
                                int E, void* stream) {
    CE_CHECK_ARG(nbox > 0 && E > 0, "ce_bbox_pool_fwd: empty");
    hipLaunchKernelGGL(bbox_pool_fwd_kernel, dim3(nbox), dim3(256), 0, (hipStream_t)stream, grid, sb, s0, s1, boxes, out,
                       nbox, E);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_bbox_pool_bwd(const float* dout, const int* boxes, float* dgrid, int g, int nbox, int E,
                                void* stream) {
    CE_CHECK_ARG(nbox > 0 && E > 0 && g > 0, "ce_bbox_pool_bwd: empty");
    hipLaunchKernelGGL(bbox_pool_bwd_kernel, dim3(nbox), dim3(256), 0, (hipStream_t)stream, dout, boxes, dgrid, g, nbox, E);
    CE_LAUNCH_CHECK();
    return 0;
}
