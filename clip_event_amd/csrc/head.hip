// Contrastive head: feature L2-normalisation, fp32 logits, cross-entropy (gfx950).
//
// Replaces model_clip.py:496-521 (normalise, exp(logit_scale), logits_per_text / logits_per_image
// over batch or per instance) and CriterionContrastive (model_clip.py:620-662) with their
// autograd.  The head is <0.1 % of the step's FLOPs; it is kept in fp32 (the reference's
// precision) so the logits the drop-in API returns match the reference tightly.
//
//   ce_l2norm_fwd / _bwd   f / ||f||  (no eps, as the reference)
//   ce_sgemm               C = alpha * op(A) op(B) (+ beta*C), arbitrary strides, fp32 LDS-tiled
//   ce_xent_fwd / _bwd     row-wise cross entropy on selected rows (index_pos), mean reduction
//   ce_dot                 out += sum a*b   (logit_scale gradient)
#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

// one wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ f, long ldf, float* __restrict__ y,
                                                         long ldy, float* __restrict__ inv_norm, int n, int E) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) {
        const float v = f[(long)r * ldf + c];
        s += v * v;
    }
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int c = lane; c < E; c += 64) y[(long)r * ldy + c] = f[(long)r * ldf + c] * inv;
    if (lane == 0) inv_norm[r] = inv;
}

// df = (dy - y <dy, y>) * inv_norm   (+= when accumulate)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, long lddy,
                                                         const float* __restrict__ y, long ldy,
                                                         const float* __restrict__ inv_norm, float* __restrict__ df,
                                                         long lddf, int n, int E, int accumulate) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += dy[(long)r * lddy + c] * y[(long)r * ldy + c];
    s = wave_sum(s);
    const float inv = inv_norm[r];
    for (int c = lane; c < E; c += 64) {
        float v = (dy[(long)r * lddy + c] - y[(long)r * ldy + c] * s) * inv;
        if (accumulate) v += df[(long)r * lddf + c];
        df[(long)r * lddf + c] = v;
    }
}

// C[m,n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + beta * C[m,n]; TxT tile, (T/16)x(T/16) per thread.
// T = 64 for big problems, T = 32 when a 64-tile grid would leave most of the chip idle (B = 256 logits).
template <int T>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long sam, long sak,
                                                    const float* __restrict__ B, long sbk, long sbn,
                                                    float* __restrict__ C, long ldc, int M, int N, int K,
                                                    const float* __restrict__ alpha_ptr, float alpha, int alpha_exp,
                                                    float beta) {
    constexpr int R = T / 16;
    __shared__ float sA[16][T + 1];
    __shared__ float sB[16][T + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    float acc[R][R] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int idx = threadIdx.x; idx < T * 16; idx += 256) {
            // map so that the fastest-varying thread index follows the unit-stride axis of each operand
            int kk, mm;
            if (sak == 1) { kk = idx & 15; mm = idx >> 4; } else { mm = idx % T; kk = idx / T; }
            const int m = m0 + mm, k = k0 + kk;
            sA[kk][mm] = (m < M && k < K) ? A[(long)m * sam + (long)k * sak] : 0.f;
            int kb, nn;
            if (sbk == 1) { kb = idx & 15; nn = idx >> 4; } else { nn = idx % T; kb = idx / T; }
            const int n = n0 + nn, k2 = k0 + kb;
            sB[kb][nn] = (n < N && k2 < K) ? B[(long)k2 * sbk + (long)n * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[R], b[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                a[i] = sA[kk][ty + 16 * i];
                b[i] = sB[kk][tx + 16 * i];
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
    float al = alpha;
    if (alpha_ptr) al *= alpha_exp ? __expf(*alpha_ptr) : *alpha_ptr;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int n = n0 + tx + 16 * j;
            if (n >= N) continue;
            float v = al * acc[i][j];
            if (beta != 0.f) v += beta * C[(long)m * ldc + n];
            C[(long)m * ldc + n] = v;
        }
    }
}

// per selected row r (src row = sel ? sel[r] : r): lse, loss_r = lse - logit[label]; loss_sum += loss_r / nrows
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ logits, long ld,
                                                       const long* __restrict__ labels, const long* __restrict__ sel,
                                                       float* __restrict__ row_lse, float* __restrict__ loss, int nrows,
                                                       int C) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const long src = sel ? sel[r] : (long)r;
    const float* row = logits + src * ld;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(row[c] - mx);
    s = wave_sum(s);
    const float l = mx + __logf(s);
    if (lane == 0) {
        row_lse[r] = l;
        const long y = labels[src];
        atomicAdd(loss, (l - row[y]) / (float)nrows);
    }
}

// dlogits[src, c] = g * (exp(logit - lse) - [c == label]) / nrows ; rows not selected are left untouched
// (caller zero-fills dlogits)
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logits, long ld,
                                                       const long* __restrict__ labels, const long* __restrict__ sel,
                                                       const float* __restrict__ row_lse, const float* __restrict__ gptr,
                                                       float* __restrict__ dlogits, long ldd, int nrows, int C) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const long src = sel ? sel[r] : (long)r;
    const float g = *gptr / (float)nrows;
    const float l = row_lse[r];
    const long y = labels[src];
    for (int c = lane; c < C; c += 64) {
        float p = __expf(logits[src * ld + c] - l);
        if (c == y) p -= 1.0f;
        dlogits[src * ldd + c] = g * p;
    }
}

__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                  float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}


// per-instance logits (model_clip.py:509-521): lpi[b,k] = s * <In[b], Tn[b*K+k]>, one wave per (b,k)
__global__ __launch_bounds__(256) void instance_logits_kernel(const float* __restrict__ In, const float* __restrict__ Tn,
                                                              const float* __restrict__ ls, float* __restrict__ lpi,
                                                              int B, int K, int E) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * K) return;
    const int b = r / K;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += In[(long)b * E + c] * Tn[(long)r * E + c];
    s = wave_sum(s);
    if (lane == 0) lpi[r] = __expf(*ls) * s;
}

// dIn[b] += s sum_k d[b,k] Tn[b*K+k] ; dTn[b*K+k] += s d[b,k] In[b]   (one wave per image)
__global__ __launch_bounds__(256) void instance_logits_bwd_kernel(const float* __restrict__ d, const float* __restrict__ In,
                                                                  const float* __restrict__ Tn, const float* __restrict__ ls,
                                                                  float* __restrict__ dIn, float* __restrict__ dTn, int B,
                                                                  int K, int E) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float sc = __expf(*ls);
    for (int c = lane; c < E; c += 64) {
        const float iv = In[(long)b * E + c];
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const float dk = sc * d[b * K + k];
            acc += dk * Tn[((long)b * K + k) * E + c];
            dTn[((long)b * K + k) * E + c] += dk * iv;
        }
        dIn[(long)b * E + c] += acc;
    }
}

// elementwise losses with 'mean' reduction over all n elements (nn.BCEWithLogitsLoss / nn.KLDivLoss,
// model_clip.py:626-629).  mode 0: bce-with-logits, mode 1: kl-div (input = log-probabilities).
__global__ __launch_bounds__(256) void elem_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                            int mode, float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float xv = x[i], yv = y[i];
        if (mode == 0) s += fmaxf(xv, 0.f) - xv * yv + log1pf(__expf(-fabsf(xv)));
        else s += (yv > 0.f) ? yv * (__logf(yv) - xv) : 0.f;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) / (float)n);
}

__global__ void elem_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n, int mode,
                                     const float* __restrict__ g, float* __restrict__ dx) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n) return;
    const float gv = *g / (float)n;
    dx[i] = (mode == 0) ? gv * (1.0f / (1.0f + __expf(-x[i])) - y[i]) : -gv * y[i];
}

}  // namespace

extern "C" int ce_l2norm_fwd(const float* f, long ldf, float* y, long ldy, float* inv_norm, int n, int E,
                             void* stream) {
    CE_CHECK_ARG(n > 0 && E > 0, "ce_l2norm_fwd: empty");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ce_div_up(n, 4)), dim3(256), 0, (hipStream_t)stream, f, ldf, y, ldy,
                       inv_norm, n, E);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_l2norm_bwd(const float* dy, long lddy, const float* y, long ldy, const float* inv_norm, float* df,
                             long lddf, int n, int E, int accumulate, void* stream) {
    CE_CHECK_ARG(n > 0 && E > 0, "ce_l2norm_bwd: empty");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ce_div_up(n, 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy,
                       inv_norm, df, lddf, n, E, accumulate);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_sgemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                        int M, int N, int K, const float* alpha_ptr, float alpha, int alpha_exp, float beta,
                        void* stream) {
    CE_CHECK_ARG(M > 0 && N > 0 && K > 0, "ce_sgemm: empty");
    if ((long)ce_div_up(N, 64) * ce_div_up(M, 64) >= 256)
        hipLaunchKernelGGL(sgemm_kernel<64>, dim3(ce_div_up(N, 64), ce_div_up(M, 64)), dim3(256), 0, (hipStream_t)stream, A,
                           sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha_ptr, alpha, alpha_exp, beta);
    else if ((long)ce_div_up(N, 32) * ce_div_up(M, 32) >= 256)
        hipLaunchKernelGGL(sgemm_kernel<32>, dim3(ce_div_up(N, 32), ce_div_up(M, 32)), dim3(256), 0, (hipStream_t)stream, A,
                           sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha_ptr, alpha, alpha_exp, beta);
    else
        hipLaunchKernelGGL(sgemm_kernel<16>, dim3(ce_div_up(N, 16), ce_div_up(M, 16)), dim3(256), 0, (hipStream_t)stream, A,
                           sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha_ptr, alpha, alpha_exp, beta);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_xent_fwd(const float* logits, long ld, const int64_t* labels, const int64_t* sel, float* row_lse,
                           float* loss, int nrows, int C, void* stream) {
    CE_CHECK_ARG(nrows > 0 && C > 0, "ce_xent_fwd: empty");
    hipLaunchKernelGGL(xent_fwd_kernel, dim3(ce_div_up(nrows, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       (const long*)labels, (const long*)sel, row_lse, loss, nrows, C);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_xent_bwd(const float* logits, long ld, const int64_t* labels, const int64_t* sel,
                           const float* row_lse, const float* grad, float* dlogits, long ldd, int nrows, int C,
                           void* stream) {
    CE_CHECK_ARG(nrows > 0 && C > 0, "ce_xent_bwd: empty");
    hipLaunchKernelGGL(xent_bwd_kernel, dim3(ce_div_up(nrows, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       (const long*)labels, (const long*)sel, row_lse, grad, dlogits, ldd, nrows, C);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_dot(const float* a, const float* b, long n, float* out, void* stream) {
    CE_CHECK_ARG(n > 0, "ce_dot: empty");
    long blocks = (n + 1023) / 1024;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(dot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_instance_logits(const float* In, const float* Tn, const float* logit_scale, float* lpi, int B, int K,
                                  int E, void* stream) {
    CE_CHECK_ARG(B > 0 && K > 0 && E > 0, "ce_instance_logits: empty");
    hipLaunchKernelGGL(instance_logits_kernel, dim3(ce_div_up(B * K, 4)), dim3(256), 0, (hipStream_t)stream, In, Tn,
                       logit_scale, lpi, B, K, E);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_instance_logits_bwd(const float* dlpi, const float* In, const float* Tn, const float* logit_scale,
                                      float* dIn, float* dTn, int B, int K, int E, void* stream) {
    CE_CHECK_ARG(B > 0 && K > 0 && E > 0, "ce_instance_logits_bwd: empty");
    hipLaunchKernelGGL(instance_logits_bwd_kernel, dim3(ce_div_up(B, 4)), dim3(256), 0, (hipStream_t)stream, dlpi, In, Tn,
                       logit_scale, dIn, dTn, B, K, E);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_elem_loss_fwd(const float* x, const float* y, long n, int mode, float* loss, void* stream) {
    CE_CHECK_ARG(n > 0 && (mode == 0 || mode == 1), "ce_elem_loss_fwd: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(elem_loss_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, mode, loss);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_elem_loss_bwd(const float* x, const float* y, long n, int mode, const float* grad, float* dx,
                                void* stream) {
    CE_CHECK_ARG(n > 0 && (mode == 0 || mode == 1), "ce_elem_loss_bwd: bad arguments");
    hipLaunchKernelGGL(elem_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n,
                       mode, grad, dx);
    CE_LAUNCH_CHECK();
    return 0;
}
