// LayerNorm forward / backward, fp32 statistics, one wave64 per row (gfx950).
//
// Replaces `LayerNorm(nn.LayerNorm)` as used by ln_pre / ln_1 / ln_2 / ln_post / ln_final
// (model_clip.py:157-163, :176, :182, :225, :229, :327) and its autograd.  HBM-bound:
// a row lives in registers (float4 per lane per 256-column chunk), sums by wave shuffles.
// The forward writes the bf16 GEMM operand (or the fp32 residual stream for ln_pre) and the
// per-row mean / rstd; the backward fuses the residual-gradient add, emits the fp32 gradient
// stream plus its bf16 copy (operand of the next dgrad/wgrad GEMMs) and reduces dgamma/dbeta
// per workgroup before one atomic per column.
#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

template <int IT, bool OUT_F32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ rows,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     void* __restrict__ y, long ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    const long src = rows ? (long)rows[r] : (long)r;
    const float* xr = x + src * ldx;
    f32x4 v[IT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = i * 256 + lane * 4;
        v[i] = (c < D) ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            f32x4 d = v[i] - mu;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        mean[r] = mu;
        rstd[r] = rs;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            f32x4 g = *reinterpret_cast<const f32x4*>(w + c);
            f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
            f32x4 o = (v[i] - mu) * rs * g + bb;
            if constexpr (OUT_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (long)r * ldy + c) = o;
            } else {
                u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + (long)r * ldy + c) = pk;
            }
        }
    }
}

// dy: bf16 (DY_F32=false) or fp32.  dst row = rows ? rows[r] : r for x / dx (scatter form used
// when only the CLS / EOT row of each sample went through the LayerNorm).
template <int IT, bool DY_F32>
__global__ __launch_bounds__(1024) void ln_bwd_kernel(const void* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                     long ldx, const int* __restrict__ rows,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ w, const float* __restrict__ dx_in,
                                                     float* __restrict__ dx_out, long lddx, bf16_t* __restrict__ dxb,
                                                     long lddxb, float* __restrict__ dw, float* __restrict__ db,
                                                     float* __restrict__ dxsum, int M, int D) {
    // 16 waves per workgroup, one row per wave at a time; at most 256 workgroups so that the per-column
    // atomics (dgamma / dbeta / dx column sums) see little same-address contention
    extern __shared__ __attribute__((aligned(16))) float red[];   // [16][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 aw[IT], ab[IT], ax[IT], g[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        aw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ax[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = i * 256 + lane * 4;
        g[i] = (c < D) ? *reinterpret_cast<const f32x4*>(w + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invD = 1.0f / (float)D;
    for (int r = blockIdx.x * 16 + wave; r < M; r += gridDim.x * 16) {
        const long dst = rows ? (long)rows[r] : (long)r;
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[IT], gy[IT], din[IT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {      // every load of the row is issued before the reductions (one memory latency per row)
            const int c = i * 256 + lane * 4;
            din[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < D && dx_in) din[i] = *reinterpret_cast<const f32x4*>(dx_in + dst * lddx + c);
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < D) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(x + dst * ldx + c);
                f32x4 d;
                if constexpr (DY_F32) {
                    d = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + (long)r * lddy + c);
                } else {
                    u32x2 pk = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + (long)r * lddy + c);
                    d = f32x4{bf_lo(pk[0]), bf_hi(pk[0]), bf_lo(pk[1]), bf_hi(pk[1])};
                }
                xh[i] = (xv - mu) * rs;
                gy[i] = d * g[i];
                aw[i] += d * xh[i];
                ab[i] += d;
                s1 += (gy[i][0] + gy[i][1]) + (gy[i][2] + gy[i][3]);
                f32x4 t = gy[i] * xh[i];
                s2 += (t[0] + t[1]) + (t[2] + t[3]);
            } else {
                xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                gy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < D) {
                f32x4 o = (gy[i] - c1 - xh[i] * c2) * rs + din[i];
                *reinterpret_cast<f32x4*>(dx_out + dst * lddx + c) = o;
                ax[i] += o;
                if (dxb) {
                    u32x2 pk = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
                    *reinterpret_cast<u32x2*>(dxb + dst * lddxb + c) = pk;
                }
            }
        }
    }
    // cross-wave reduce of the dgamma / dbeta partials, then one atomic per column per block
#pragma unroll
    for (int pass = 0; pass < (dxsum ? 3 : 2); ++pass) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < D) *reinterpret_cast<f32x4*>(&red[wave * D + c]) = pass == 0 ? aw[i] : (pass == 1 ? ab[i] : ax[i]);
        }
        __syncthreads();
        float* dstp = pass == 0 ? dw : (pass == 1 ? db : dxsum);
        for (int c = threadIdx.x; c < D; c += 1024) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < 16; ++wv) t += red[wv * D + c];
            atomicAdd(dstp + c, t);
        }
    }
}

}  // namespace

#define LN_DISPATCH(D, CALL)                                     \
    do {                                                         \
        if ((D) <= 256) { CALL(1); }                             \
        else if ((D) <= 512) { CALL(2); }                        \
        else if ((D) <= 768) { CALL(3); }                        \
        else if ((D) <= 1024) { CALL(4); }                       \
        else { CALL(8); }                                        \
    } while (0)

extern "C" int ce_layernorm_fwd(const float* x, long ldx, const int* rows, const float* w, const float* b, void* y,
                                long ldy, int out_f32, float* mean, float* rstd, int M, int D, float eps,
                                void* stream) {
    CE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "ce_layernorm_fwd: need 0<D<=2048, D%%4==0 (D=%d M=%d)", D, M);
    CE_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0, "ce_layernorm_fwd: leading dimensions must be multiples of 4");
    dim3 grid(ce_div_up(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    CeProfScope prof(CE_PROF_LN_FWD, 8.0 * M * D, (4.0 + (out_f32 ? 4.0 : 2.0)) * M * D, s);
#define CALL(IT)                                                                                                   \
    if (out_f32)                                                                                                   \
        hipLaunchKernelGGL((ln_fwd_kernel<IT, true>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps); \
    else                                                                                                           \
        hipLaunchKernelGGL((ln_fwd_kernel<IT, false>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps)
    LN_DISPATCH(D, CALL);
#undef CALL
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_layernorm_bwd(const void* dy, long lddy, int dy_f32, const float* x, long ldx, const int* rows,
                                const float* mean, const float* rstd, const float* w, const float* dx_in,
                                float* dx_out, long lddx, void* dxb, long lddxb, float* dw, float* db, float* dxsum, int M,
                                int D, void* stream) {
    CE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "ce_layernorm_bwd: need 0<D<=2048, D%%4==0 (D=%d M=%d)", D, M);
    CE_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && lddxb % 4 == 0, "ce_layernorm_bwd: leading dimensions must be multiples of 4");
    int blocks = ce_div_up(M, 16);
    if (blocks > 256) blocks = 256;
    dim3 grid(blocks), block(1024);
    const size_t lds = 16 * (size_t)D * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    CeProfScope prof(CE_PROF_LN_BWD, 16.0 * M * D, ((dy_f32 ? 4.0 : 2.0) + 4.0 + (dx_in ? 4.0 : 0.0) + 4.0 + (dxb ? 2.0 : 0.0)) * M * D, s);
#define CALL(IT)                                                                                                    \
    if (dy_f32)                                                                                                     \
        hipLaunchKernelGGL((ln_bwd_kernel<IT, true>), grid, block, lds, s, dy, lddy, x, ldx, rows, mean, rstd, w, dx_in, \
                           dx_out, lddx, (bf16_t*)dxb, lddxb, dw, db, dxsum, M, D);                                        \
    else                                                                                                            \
        hipLaunchKernelGGL((ln_bwd_kernel<IT, false>), grid, block, lds, s, dy, lddy, x, ldx, rows, mean, rstd, w, dx_in, \
                           dx_out, lddx, (bf16_t*)dxb, lddxb, dw, db, dxsum, M, D)
    LN_DISPATCH(D, CALL);
#undef CALL
    CE_LAUNCH_CHECK();
    return 0;
}
