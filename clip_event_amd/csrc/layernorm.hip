// LayerNorm forward / backward, fp32 statistics, one wave64 per row (gfx950).
//
// Replaces `LayerNorm(nn.LayerNorm)` as used by ln_pre / ln_1 / ln_2 / ln_post / ln_final
// (model_clip.py:157-163, :176, :182, :225, :229, :327) and its autograd.  HBM-bound:
// a row lives in registers (4 elements per lane per 256-column chunk), sums by wave shuffles.
// The forward writes the bf16 GEMM operand (or the residual stream for ln_pre) and the
// per-row mean / rstd; the backward fuses the residual-gradient add, emits the gradient
// stream plus its bf16 copy (operand of the next dgrad/wgrad GEMMs) and reduces dgamma/dbeta
// per workgroup before one atomic per column.
//
// Stream operands (x, dx_in, dx_out, and dy where it IS the gradient stream) carry an element type
// (CE_T_F32 / CE_T_F16, common.hpp): the type is a kernel argument, wave-uniform, so the same code serves
// the fp32 and the fp16 residual stream.  An fp16 GRADIENT stream holds gradient * gscale.
#include <stdlib.h>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

// Element types are TEMPLATE parameters: with the type as a run-time argument every load sat behind a (uniform) branch
// and was waited for before the next one was issued -- one memory latency per 16 bytes instead of one per row (measured:
// LayerNorm backward 1.48 -> 1.85 ms/step on the fp32 stream).
//
// Row descriptors: every row access of these kernels is a bounds-checked buffer operation on a descriptor of ONE row
// (base = the row, range = D elements), so a lane past D needs no branch (its load returns 0, its store is dropped) and the
// kernels are straight-line code.  Why that matters: the compiler closes a divergent `if (c < D) { load / store }` with
// `s_waitcnt vmcnt(0)` and can no longer count the stores in flight, so (a) the three 8-byte loads of a D = 768 row went out
// one memory latency after the other, (b) gamma / beta, loaded inside the row loop, cost three exposed cache latencies per
// row, (c) every wait for a prefetched row also waited for the previous row's stores to be acknowledged.
// The row index must be wave-uniform (readfirstlane'd by the caller).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* base, long row, long ld, int esz, int D) {
    return make_rsrc(reinterpret_cast<const char*>(base) + row * ld * esz, (uint32_t)(D * esz));
}
#ifndef CE_LN_NT
#define CE_LN_NT 0      // 2 = nt policy on the backward's row loads and stores (every byte is touched once per launch): measured no better
#endif
template <int T> struct Raw4 { typedef u32x2 type; };      // unconverted 4-element group of a typed operand: what a prefetch keeps in registers
template <> struct Raw4<CE_T_F32> { typedef f32x4 type; };
template <int T, int AUX = 0>
__device__ __forceinline__ typename Raw4<T>::type ld4_row(__amdgpu_buffer_rsrc_t r, int c) {
    if constexpr (T == CE_T_F32) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, c * 4, 0, AUX));
    else return __builtin_amdgcn_raw_buffer_load_b64(r, c * 2, 0, AUX);
}
template <int T>
__device__ __forceinline__ f32x4 cvt4(typename Raw4<T>::type raw) {
    if constexpr (T == CE_T_F32) return raw;
    else if constexpr (T == CE_T_F16) return f16x4_to_f32(raw);
    else return f32x4{bf_lo(raw[0]), bf_hi(raw[0]), bf_lo(raw[1]), bf_hi(raw[1])};
}
template <int T, int AUX = 0>
__device__ __forceinline__ void st4_row(__amdgpu_buffer_rsrc_t r, int c, f32x4 v) {
    if constexpr (T == CE_T_F32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, c * 4, 0, AUX);
    else if constexpr (T == CE_T_F16) __builtin_amdgcn_raw_buffer_store_b64(f32_to_f16x4_sat(v), r, c * 2, 0, AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}, r, c * 2, 0, AUX);
}
// one float from lane 0 (the only lane inside a 4-byte descriptor), no branch
__device__ __forceinline__ void st1_lane0(float* dst, float v, int lane) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), make_rsrc(dst, 4u), lane * 4, 0, 0);
}

// Per-row e4m3 copy of a row the wave holds in registers as bf16 pairs (the fp8 operand path, ce_quant_rows_fp8's rule:
// the power of two that maps the row's amax into (224, 448]; exact scaling, so the bytes equal that kernel's): the row
// quantisation pass of the GEMM that consumes this output disappears.  pk[i] = this lane's 4 bf16 values of chunk i.
template <int IT>
__device__ __forceinline__ void quant_row_from_regs(const u32x2 (&pk)[IT], int lane, int D, uint8_t* __restrict__ qrow,
                                                    float* __restrict__ qscale) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i)
        if (i * 256 + lane * 4 < D)
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(bf_lo(pk[i][0])), fabsf(bf_hi(pk[i][0])))),
                         fmaxf(fabsf(bf_lo(pk[i][1])), fabsf(bf_hi(pk[i][1]))));
    amax = wave_max_dpp(amax);
    const uint32_t ab = __float_as_uint(amax);
    const bool live = amax >= 7.8886090522101181e-31f;          // 2^-100
    const int e = 9 - ((int)((ab >> 23) & 0xff) - 126) - (((ab & 0x7fffffu) > 0x600000u) ? 1 : 0);
    const float inv = live ? __uint_as_float((uint32_t)(e + 127) << 23) : 1.0f;
    st1_lane0(qscale, live ? __uint_as_float((uint32_t)(127 - e) << 23) : 1.0f, lane);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qrow, (uint32_t)D);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(pk[i][0]) * inv, bf_hi(pk[i][0]) * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(pk[i][1]) * inv, bf_hi(pk[i][1]) * inv, w, true);
        __builtin_amdgcn_raw_buffer_store_b32(w, rq, i * 256 + lane * 4, 0, 0);
    }
}

// RPW rows per wave, all their loads issued before the first reduction.  A launch of 12,800 rows takes 12-13 us whether a row is
// 1.5 KB (fp16 stream) or 3 KB (fp32), and whether its workgroups make one resident round (RPW = 2) or two (RPW = 1, default):
// neither bytes nor rounds bound it (tools/bench_hbm.py).
template <int IT, int XT, int YT, int RPW>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, long ldx, const int* __restrict__ rows,
                                                     const float* __restrict__ w, const float* __restrict__ b,
                                                     void* __restrict__ y, long ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int D, float eps,
                                                     uint8_t* __restrict__ q8, long ldq, float* __restrict__ qscale,
                                                     unsigned int* __restrict__ sat) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: the row descriptors are built from it
    const int rbase = (blockIdx.x * 4 + wave) * RPW;
    if (rbase >= M) return;
    // straight-line code on row descriptors (row_rsrc above); gamma / beta are requested with the row and consumed after the
    // two reductions
    f32x4 v[RPW][IT], g[IT], bb[IT];
    float s[RPW];
    float am = 0.f;                                        // fp16 stream: largest |element| this lane read (saturation telemetry)
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = min(rbase + k, M - 1);               // a wave's surplus row repeats the last one (never stored)
        const int src = __builtin_amdgcn_readfirstlane(rows ? rows[r] : r);
        const __amdgpu_buffer_rsrc_t rx = row_rsrc(x, src, ldx, XT == CE_T_F32 ? 4 : 2, D);
        s[k] = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            v[k][i] = cvt4<XT>(ld4_row<XT>(rx, i * 256 + lane * 4));      // 0 past D
            s[k] += (v[k][i][0] + v[k][i][1]) + (v[k][i][2] + v[k][i][3]);
            if constexpr (XT == CE_T_F16) am = absmax4(am, v[k][i]);
        }
    }
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (uint32_t)(D * 4)), rb = make_rsrc(b, (uint32_t)(D * 4));
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            g[i] = ld4_row<CE_T_F32>(rw, i * 256 + lane * 4);
            bb[i] = ld4_row<CE_T_F32>(rb, i * 256 + lane * 4);
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = rbase + k;
        if (r >= M) break;                                  // wave-uniform
        const float mu = wave_sum_dpp(s[k]) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            f32x4 d = v[k][i] - mu;
            d = (i * 256 + lane * 4 < D) ? d : f32x4{0.f, 0.f, 0.f, 0.f};
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rs = rsqrtf(wave_sum_dpp(q) / (float)D + eps);
        st1_lane0(mean + r, mu, lane);
        st1_lane0(rstd + r, rs, lane);
        const __amdgpu_buffer_rsrc_t ry = row_rsrc(y, r, ldy, YT == CE_T_F32 ? 4 : 2, D);
        u32x2 pk[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const f32x4 o = (v[k][i] - mu) * rs * g[i] + bb[i];        // past D: gamma = beta = 0 -> 0
            pk[i] = u32x2{0u, 0u};
            if constexpr (YT == CE_T_BF16) pk[i] = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            st4_row<YT>(ry, i * 256 + lane * 4, o);
        }
        if constexpr (YT == CE_T_BF16) {
            if (q8) quant_row_from_regs<IT>(pk, lane, D, q8 + (long)r * ldq, qscale + r);       // wave-uniform
        }
    }
    // an element AT the fp16 limit was clamped by the store that wrote it (or sits exactly on the edge): count it.  The branch
    // comes last, behind every store of the kernel, so the straight-line row code above keeps its counted waits.
    if constexpr (XT == CE_T_F16) {
        if (sat && am >= CE_F16_LIMIT) atomicAdd(sat, 1u);
    }
}

// dy: bf16 / fp32 / fp16-scaled stream.  dst row = rows ? rows[r] : r for x / dx (scatter form used
// when only the CLS / EOT row of each sample went through the LayerNorm).
// NW waves per workgroup (16 up to D = 512, 8 up to 1024, 4 beyond: a row in flight + a row being reduced must fit the
// registers: 16 waves spilled at D = 768), one row per
// wave at a time; at most 256 workgroups so that the per-column atomics (dgamma / dbeta / dx column sums) see little
// same-address contention.
// PF: the NEXT row's loads (kept raw: 18 registers at D = 768 on the fp16 stream, 30 on the fp32 one) are issued before the
// current row is reduced.  Without it every wave of the chip loads, then every wave computes -- the rows of a launch are
// consumed in 3-4 chip-wide rounds with the memory system idle during each round's arithmetic.
template <int IT, int NW, int DYT, int XT, int DIT, int DOT, bool PF>
#ifndef CE_LN_BWD_MINB
#define CE_LN_BWD_MINB 1
#endif
__global__ __launch_bounds__(64 * NW, CE_LN_BWD_MINB) void ln_bwd_kernel(const void* __restrict__ dy, long lddy, const void* __restrict__ x,
                                                         long ldx, const int* __restrict__ rows,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ w, const void* __restrict__ dx_in,
                                                         void* __restrict__ dx_out, long lddx, bf16_t* __restrict__ dxb,
                                                         long lddxb, float* __restrict__ dw, float* __restrict__ db,
                                                         float* __restrict__ dxsum, const float* __restrict__ gscale_ptr,
                                                         int M, int D, uint8_t* __restrict__ q8, long ldq,
                                                         float* __restrict__ qscale, float* __restrict__ partial,
                                                         unsigned int* __restrict__ sat) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][D] + strip [3][D]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR: the row descriptors are built from it
    f32x4 aw[IT], ab[IT], ax[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        aw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ax[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invD = 1.0f / (float)D;
    // a scaled fp16 stream holds gradient * gscale (a power of two kept in device memory, ce_grad_scale): read back in
    // true units, stored scaled
    const float gscale = gscale_ptr ? *gscale_ptr : 1.0f;
    const float dy_mul = DYT == CE_T_F16 ? 1.0f / gscale : 1.0f;
    const float din_mul = DIT == CE_T_F16 ? 1.0f / gscale : 1.0f;
    const float out_mul = DOT == CE_T_F16 ? gscale : 1.0f;
    float om = 0.f;                                               // fp16 gradient stream: largest |value| handed to the clamping store
    // rows in flight: PFD rows' raw loads (18 registers each at D = 768 on the fp16 stream) are outstanding while a row is reduced.
    // One is enough: 1, 2 and 3 measure the same 23.2-23.5 us per launch (tools/diag/ab_ln_occ.sh) -- with the loads no longer
    // serialised the row loop is bound by its ~350 vector instructions per row (two waves per SIMD), not by rows in flight.
#ifndef CE_LN_PFD
#define CE_LN_PFD 1
#endif
    constexpr int PFD = !PF ? 1 : ((NW >= 16 && XT == CE_T_F32 && DYT == CE_T_F32) ? 1 : CE_LN_PFD);   // (all-fp32 rows at 16 waves: two would spill)
    struct Pending {
        typename Raw4<XT>::type xr[IT];
        typename Raw4<DYT>::type dr[IT];
        typename Raw4<DIT>::type ir[IT];
        float mu, rs;
        int dst;
    };
    Pending pend[PFD];
    // gamma sits in registers for the whole launch (inside the row loop its three 16-byte loads were each waited for on
    // their own: three exposed cache latencies per row)
    f32x4 gw[IT];
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (uint32_t)(D * 4));
#pragma unroll
        for (int i = 0; i < IT; ++i) gw[i] = ld4_row<CE_T_F32>(rw, i * 256 + lane * 4);
    }
    constexpr int XB = XT == CE_T_F32 ? 4 : 2, DYB = DYT == CE_T_F32 ? 4 : 2, DIB = DIT == CE_T_F32 ? 4 : 2, DOB = DOT == CE_T_F32 ? 4 : 2;
    auto fetch = [&](Pending& P, int r) __attribute__((always_inline)) {      // every load of a row, issued back to back (r wave-uniform)
        P.dst = __builtin_amdgcn_readfirstlane(rows ? rows[r] : r);
        P.mu = mean[r];
        P.rs = rstd[r];
        const __amdgpu_buffer_rsrc_t rx = row_rsrc(x, P.dst, ldx, XB, D), rdy = row_rsrc(dy, r, lddy, DYB, D);
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            P.xr[i] = ld4_row<XT, CE_LN_NT>(rx, i * 256 + lane * 4);
            P.dr[i] = ld4_row<DYT, CE_LN_NT>(rdy, i * 256 + lane * 4);
        }
        if (dx_in) {                                               // wave-uniform
            const __amdgpu_buffer_rsrc_t rin = row_rsrc(dx_in, P.dst, lddx, DIB, D);
#pragma unroll
            for (int i = 0; i < IT; ++i) P.ir[i] = ld4_row<DIT, CE_LN_NT>(rin, i * 256 + lane * 4);
        }
    };
    const int stride = gridDim.x * NW;
    // one row: its pending loads move into working registers, the slot is refilled with row `next` (if any) BEFORE this row's
    // reductions and stores
    auto consume = [&](Pending& P, int next) __attribute__((always_inline)) {
        const int cdst = P.dst;
        const float cmu = P.mu, crs = P.rs;
        f32x4 xh[IT], gy[IT], din[IT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const f32x4 xv = cvt4<XT>(P.xr[i]);                        // past D every operand reads as 0
            f32x4 d = cvt4<DYT>(P.dr[i]);
            if constexpr (DYT == CE_T_F16) d *= dy_mul;
            din[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (dx_in) {
                din[i] = cvt4<DIT>(P.ir[i]);
                if constexpr (DIT == CE_T_F16) din[i] *= din_mul;
            }
            xh[i] = (xv - cmu) * crs;
            gy[i] = d * gw[i];
            aw[i] += d * xh[i];
            ab[i] += d;
            s1 += (gy[i][0] + gy[i][1]) + (gy[i][2] + gy[i][3]);
            f32x4 t = gy[i] * xh[i];
            s2 += (t[0] + t[1]) + (t[2] + t[3]);
        }
        if constexpr (PF) {
            if (next < M) fetch(P, next);
        }
        const float c1 = wave_sum_dpp(s1) * invD, c2 = wave_sum_dpp(s2) * invD;
        u32x2 pkq[IT];
        const __amdgpu_buffer_rsrc_t ro = row_rsrc(dx_out, cdst, lddx, DOB, D);
        const __amdgpu_buffer_rsrc_t rb = row_rsrc(dxb ? (const void*)dxb : (const void*)dx_out, cdst, lddxb, 2, dxb ? D : 0);   // no dxb: an empty range drops the stores
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = i * 256 + lane * 4;
            f32x4 o = (gy[i] - c1 - xh[i] * c2) * crs + din[i];
            o = (c < D) ? o : f32x4{0.f, 0.f, 0.f, 0.f};              // (xh is -mu * rstd past D)
            ax[i] += o;
            if constexpr (DOT == CE_T_F16) {
                const f32x4 os = o * out_mul;
                om = absmax4(om, os);
                st4_row<DOT, CE_LN_NT>(ro, c, os);
            } else {
                st4_row<DOT, CE_LN_NT>(ro, c, o);
            }
            pkq[i] = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            __builtin_amdgcn_raw_buffer_store_b64(pkq[i], rb, c * 2, 0, CE_LN_NT);
        }
        if (q8) quant_row_from_regs<IT>(pkq, lane, D, q8 + (long)cdst * ldq, qscale + cdst);      // e4m3 copy of the dxb row (wave-uniform)
        if constexpr (!PF) {
            if (next < M) fetch(P, next);
        }
    };
    const int r0 = blockIdx.x * NW + wave;
#pragma unroll
    for (int k = 0; k < PFD; ++k)
        if (r0 + k * stride < M) fetch(pend[k], r0 + k * stride);
    for (int r = r0; r < M; r += PFD * stride) {
#pragma unroll
        for (int k = 0; k < PFD; ++k)
            if (r + k * stride < M) consume(pend[k], r + (k + PFD) * stride);
    }
    // saturation telemetry: a gradient that reached the fp16 limit (or is not finite) was clamped by its store.  !(om < limit) is
    // also true for a NaN.
    if constexpr (DOT == CE_T_F16) {
        if (sat && !(om < CE_F16_LIMIT)) atomicAdd(sat + 1, 1u);
    }
    // dgamma / dbeta (/ dx column sums): per pass the waves park their partials in red[NW][D] and thread c folds column c into
    // strip[pass][c]; then every column gets ONE global atomic per workgroup, each workgroup starting at a different 64-column
    // segment of the [passes x D] list.  Every workgroup of the launch adds into the same 2-3 rows and float atomics on one
    // address serialise at the memory side: walked in lockstep from column 0, pass after pass (round 3), only the segments of
    // one pass were busy at a time and the phase took 3.7 us of a 24 us launch (-DCE_DIAG_LN_NO_ATOMICS).
    const int P = dxsum ? 3 : 2;
    float* strip = red + NW * D;                                   // [3][D]
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (pass >= P) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < D) *reinterpret_cast<f32x4*>(&red[wave * D + c]) = pass == 0 ? aw[i] : (pass == 1 ? ab[i] : ax[i]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 64 * NW) {
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) t += red[wv * D + c];
            strip[pass * D + c] = t;
        }
    }
    __syncthreads();
    const int total = P * D;
    if (partial) {
        // no atomics: this workgroup's sums as row blockIdx.x of partial[blocks][3][D]; ce_layernorm_fold adds the rows later (all 256
        // workgroups adding into the same 2-3 rows of the parameter gradient serialise on the float-atomic units: 3.7 us of a 23 us launch)
        float* pp = partial + (size_t)blockIdx.x * 3 * D;
        for (int j = threadIdx.x; j < total; j += 64 * NW) pp[j] = strip[j];
        return;
    }
    const int start = (int)((blockIdx.x * 64u) % (unsigned)total);
    for (int j = threadIdx.x; j < total; j += 64 * NW) {
        int e = j + start;
        if (e >= total) e -= total;
        const int pass = e >= 2 * D ? 2 : (e >= D ? 1 : 0);
        float* dstp = pass == 0 ? dw : (pass == 1 ? db : dxsum);
#ifndef CE_DIAG_LN_NO_ATOMICS
        atomicAdd(dstp + (e - pass * D), strip[e]);
#else
        if (strip[e] == 123.456f) dstp[e - pass * D] = strip[e];
#endif
    }
}

struct LnFoldJobs {
    ce_ln_fold_job job[CE_LN_FOLD_MAX];
};
// dst[pass][c] += sum over the workgroups' rows of partial[b][pass][c]: one thread per (pass, column), rows read coalesced
__global__ __launch_bounds__(256) void ln_fold_kernel(LnFoldJobs jobs) {
    const ce_ln_fold_job& jb = jobs.job[blockIdx.y];
    const int D = jb.D, P = jb.dxsum ? 3 : 2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P * D) return;
    float t = 0.f;
    const float* pp = jb.partials + j;
#pragma unroll 8
    for (int b = 0; b < jb.blocks; ++b) t += pp[(size_t)b * 3 * D];
    const int pass = j / D, c = j - pass * D;
    float* dst = pass == 0 ? jb.dw : (pass == 1 ? jb.db : jb.dxsum);
    dst[c] += t;
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

extern "C" int ce_layernorm_fwd_t(const void* x, int x_type, long ldx, const int* rows, const float* w, const float* b,
                                  void* y, int y_type, long ldy, float* mean, float* rstd, int M, int D, float eps,
                                  void* stream) {
    return ce_layernorm_fwd_q8(x, x_type, ldx, rows, w, b, y, y_type, ldy, mean, rstd, M, D, eps, nullptr, 0, nullptr, stream);
}

extern "C" int ce_layernorm_fwd_q8(const void* x, int x_type, long ldx, const int* rows, const float* w, const float* b,
                                   void* y, int y_type, long ldy, float* mean, float* rstd, int M, int D, float eps,
                                   void* q8v, long ldq, float* qscale, void* stream) {
    uint8_t* q8 = reinterpret_cast<uint8_t*>(q8v);
    CE_CHECK_ARG(!q8 || (y_type == CE_T_BF16 && qscale && ldq >= D && ldq % 4 == 0), "ce_layernorm_fwd_q8: the e4m3 copy needs a bf16 output, scales and ldq >= D");
    CE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "ce_layernorm_fwd: need 0<D<=2048, D%%4==0 (D=%d M=%d)", D, M);
    CE_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0, "ce_layernorm_fwd: leading dimensions must be multiples of 4");
    static const int rpw_env = getenv("CE_LN_FWD_RPW") ? atoi(getenv("CE_LN_FWD_RPW")) : 0;
    const int rpw = D > 1024 ? 1 : (rpw_env == 2 ? 2 : 1);     // CE_LN_FWD_RPW=2: two rows per wave (one resident round at 12,800 rows): measured equal, 12.6 vs 12.7 us
    dim3 grid(ce_div_up(M, 4 * rpw)), block(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned int* sat = ce_sat_counters();
    CeProfScope prof(CE_PROF_LN_FWD, 8.0 * M * D, (double)(ce_type_bytes(x_type) + ce_type_bytes(y_type)) * M * D, s);
    const int combo = x_type * 4 + y_type;      // the combinations the path uses; anything else is an argument error
#define CALL(IT)                                                                                                              \
    switch (combo) {                                                                                                          \
        case CE_T_F32 * 4 + CE_T_BF16: if (rpw == 2 && IT <= 4) hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F32, CE_T_BF16, (IT <= 4 ? 2 : 1)>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); else hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F32, CE_T_BF16, 1>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); break; \
        case CE_T_F32 * 4 + CE_T_F32: if (rpw == 2 && IT <= 4) hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F32, CE_T_F32, (IT <= 4 ? 2 : 1)>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); else hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F32, CE_T_F32, 1>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); break;   \
        case CE_T_F32 * 4 + CE_T_F16: if (rpw == 2 && IT <= 4) hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F32, CE_T_F16, (IT <= 4 ? 2 : 1)>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); else hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F32, CE_T_F16, 1>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); break;   \
        case CE_T_F16 * 4 + CE_T_BF16: if (rpw == 2 && IT <= 4) hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F16, CE_T_BF16, (IT <= 4 ? 2 : 1)>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); else hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F16, CE_T_BF16, 1>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); break; \
        case CE_T_F16 * 4 + CE_T_F32: if (rpw == 2 && IT <= 4) hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F16, CE_T_F32, (IT <= 4 ? 2 : 1)>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); else hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F16, CE_T_F32, 1>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); break;   \
        case CE_T_F16 * 4 + CE_T_F16: if (rpw == 2 && IT <= 4) hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F16, CE_T_F16, (IT <= 4 ? 2 : 1)>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); else hipLaunchKernelGGL((ln_fwd_kernel<IT, CE_T_F16, CE_T_F16, 1>), grid, block, 0, s, x, ldx, rows, w, b, y, ldy, mean, rstd, M, D, eps, q8, ldq, qscale, sat); break;   \
        default: CE_CHECK_ARG(false, "ce_layernorm_fwd: element types x=%d y=%d are not built (x: f32 / f16, y: bf16 / f32 / f16)", x_type, y_type); \
    }
    LN_DISPATCH(D, CALL);
#undef CALL
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_layernorm_fwd(const float* x, long ldx, const int* rows, const float* w, const float* b, void* y,
                                long ldy, int out_f32, float* mean, float* rstd, int M, int D, float eps,
                                void* stream) {
    return ce_layernorm_fwd_t(x, CE_T_F32, ldx, rows, w, b, y, out_f32 ? CE_T_F32 : CE_T_BF16, ldy, mean, rstd, M, D, eps,
                              stream);
}

extern "C" int ce_layernorm_bwd_t(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx,
                                  const int* rows, const float* mean, const float* rstd, const float* w,
                                  const void* dx_in, int dxin_type, void* dx_out, int dx_type, long lddx, void* dxb,
                                  long lddxb, float* dw, float* db, float* dxsum, const float* gscale, int M, int D,
                                  void* stream) {
    return ce_layernorm_bwd_q8(dy, dy_type, lddy, x, x_type, ldx, rows, mean, rstd, w, dx_in, dxin_type, dx_out, dx_type, lddx, dxb,
                               lddxb, dw, db, dxsum, gscale, M, D, nullptr, 0, nullptr, stream);
}

static int ln_bwd_blocks(int M, int D) {
    const int nw = D <= 512 ? 16 : (D <= 1024 ? 8 : 4);   // = the NW of the instantiation LN_DISPATCH picks (IT <= 2: 16, 3-4: 8, 8: 4)
    int blocks = ce_div_up(M, nw);
    static const int cap = getenv("CE_LN_BWD_BLOCKS") ? atoi(getenv("CE_LN_BWD_BLOCKS")) : 256;
    return blocks > cap ? cap : blocks;
}
extern "C" int ce_layernorm_bwd_blocks(int M, int D) { return ln_bwd_blocks(M, D); }

static int ln_bwd_launch(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx,
                         const int* rows, const float* mean, const float* rstd, const float* w,
                         const void* dx_in, int dxin_type, void* dx_out, int dx_type, long lddx, void* dxb,
                         long lddxb, float* dw, float* db, float* dxsum, const float* gscale, int M, int D,
                         void* q8v, long ldq, float* qscale, float* partial, void* stream);

extern "C" int ce_layernorm_bwd_q8(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx,
                                   const int* rows, const float* mean, const float* rstd, const float* w,
                                   const void* dx_in, int dxin_type, void* dx_out, int dx_type, long lddx, void* dxb,
                                   long lddxb, float* dw, float* db, float* dxsum, const float* gscale, int M, int D,
                                   void* q8v, long ldq, float* qscale, void* stream) {
    return ln_bwd_launch(dy, dy_type, lddy, x, x_type, ldx, rows, mean, rstd, w, dx_in, dxin_type, dx_out, dx_type, lddx, dxb, lddxb, dw, db,
                         dxsum, gscale, M, D, q8v, ldq, qscale, nullptr, stream);
}

extern "C" int ce_layernorm_bwd_partials(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx,
                                         const int* rows, const float* mean, const float* rstd, const float* w,
                                         const void* dx_in, int dxin_type, void* dx_out, int dx_type, long lddx, void* dxb,
                                         long lddxb, int want_dxsum, const float* gscale, int M, int D, void* q8v, long ldq,
                                         float* qscale, float* partials, void* stream) {
    CE_CHECK_ARG(partials, "ce_layernorm_bwd_partials: null partials buffer");
    // (dw / db / dxsum only select the passes here: the kernel writes partials and returns before it would touch them)
    float* tag = partials;
    return ln_bwd_launch(dy, dy_type, lddy, x, x_type, ldx, rows, mean, rstd, w, dx_in, dxin_type, dx_out, dx_type, lddx, dxb, lddxb, tag, tag,
                         want_dxsum ? tag : nullptr, gscale, M, D, q8v, ldq, qscale, partials, stream);
}

extern "C" int ce_layernorm_fold(const ce_ln_fold_job* jobs, int njobs, void* stream) {
    CE_CHECK_ARG(jobs && njobs > 0, "ce_layernorm_fold: no jobs");
    for (int base = 0; base < njobs; base += CE_LN_FOLD_MAX) {
        LnFoldJobs pack;
        const int n = njobs - base < CE_LN_FOLD_MAX ? njobs - base : CE_LN_FOLD_MAX;
        int maxd = 0;
        for (int i = 0; i < n; ++i) {
            pack.job[i] = jobs[base + i];
            CE_CHECK_ARG(pack.job[i].partials && pack.job[i].dw && pack.job[i].db && pack.job[i].blocks > 0 && pack.job[i].D > 0,
                         "ce_layernorm_fold: job %d incomplete", base + i);
            if (pack.job[i].D > maxd) maxd = pack.job[i].D;
        }
        hipLaunchKernelGGL(ln_fold_kernel, dim3(ce_div_up(3 * maxd, 256), n), dim3(256), 0, (hipStream_t)stream, pack);
    }
    CE_LAUNCH_CHECK();
    return 0;
}

static int ln_bwd_launch(const void* dy, int dy_type, long lddy, const void* x, int x_type, long ldx,
                         const int* rows, const float* mean, const float* rstd, const float* w,
                         const void* dx_in, int dxin_type, void* dx_out, int dx_type, long lddx, void* dxb,
                         long lddxb, float* dw, float* db, float* dxsum, const float* gscale, int M, int D,
                         void* q8v, long ldq, float* qscale, float* partial, void* stream) {
    uint8_t* q8 = reinterpret_cast<uint8_t*>(q8v);
    CE_CHECK_ARG(!q8 || (dxb && qscale && ldq >= D && ldq % 4 == 0), "ce_layernorm_bwd_q8: the e4m3 copy is a copy of dxb (needs dxb, scales, ldq >= D)");
    CE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "ce_layernorm_bwd: need 0<D<=2048, D%%4==0 (D=%d M=%d)", D, M);
    CE_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && lddxb % 4 == 0, "ce_layernorm_bwd: leading dimensions must be multiples of 4");
    if (!dx_in) dxin_type = CE_T_F32;
    CE_CHECK_ARG(gscale || (dy_type != CE_T_F16 && dxin_type != CE_T_F16 && dx_type != CE_T_F16),
                 "ce_layernorm_bwd: an fp16 gradient operand needs the device scale (ce_grad_scale)");
    const int nw = D <= 512 ? 16 : (D <= 1024 ? 8 : 4);   // = the NW of the instantiation LN_DISPATCH picks (IT <= 2: 16, 3-4: 8, 8: 4)
    const int blocks = ln_bwd_blocks(M, D);
    dim3 grid(blocks), block(64 * nw);
    const size_t lds = (size_t)(nw + 3) * D * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    unsigned int* sat = ce_sat_counters();
    CeProfScope prof(CE_PROF_LN_BWD, 16.0 * M * D,
                     (double)(ce_type_bytes(dy_type) + ce_type_bytes(x_type) + (dx_in ? ce_type_bytes(dxin_type) : 0) +
                              ce_type_bytes(dx_type) + (dxb ? 2 : 0)) * M * D, s);
    // (dy, x, dx_in, dx_out) combinations of the path: the block LayerNorms on either stream format, the compact pruned
    // block (fp32 dx_sel in, stream out), ln_post / ln_final (stream x, fp32 out), ln_pre (gradient stream as dy)
    const int combo = ((dy_type * 4 + x_type) * 4 + dxin_type) * 4 + dx_type;
#define LNB(DYT, XT, DIT, DOT) (((DYT * 4 + XT) * 4 + DIT) * 4 + DOT)
#define LAUNCH(IT, DYT, XT, DIT, DOT)                                                                                         \
    hipLaunchKernelGGL((ln_bwd_kernel<IT, (IT <= 2 ? 16 : (IT <= 4 ? 8 : 4)), DYT, XT, DIT, DOT, true>), grid, block, lds, s, dy, lddy, \
                       x, ldx, rows, mean, rstd, w, dx_in, dx_out, lddx, (bf16_t*)dxb, lddxb, dw, db, dxsum, gscale, M, D, q8, ldq, qscale, partial, sat)
#define CALL(IT)                                                                                                              \
    switch (combo) {                                                                                                          \
        case LNB(CE_T_BF16, CE_T_F32, CE_T_F32, CE_T_F32): LAUNCH(IT, CE_T_BF16, CE_T_F32, CE_T_F32, CE_T_F32); break;        \
        case LNB(CE_T_F32, CE_T_F32, CE_T_F32, CE_T_F32): LAUNCH(IT, CE_T_F32, CE_T_F32, CE_T_F32, CE_T_F32); break;          \
        case LNB(CE_T_BF16, CE_T_F16, CE_T_F16, CE_T_F16): LAUNCH(IT, CE_T_BF16, CE_T_F16, CE_T_F16, CE_T_F16); break;        \
        case LNB(CE_T_BF16, CE_T_F16, CE_T_F32, CE_T_F16): LAUNCH(IT, CE_T_BF16, CE_T_F16, CE_T_F32, CE_T_F16); break;        \
        case LNB(CE_T_BF16, CE_T_F16, CE_T_F32, CE_T_F32): LAUNCH(IT, CE_T_BF16, CE_T_F16, CE_T_F32, CE_T_F32); break;        \
        case LNB(CE_T_BF16, CE_T_F16, CE_T_F16, CE_T_F32): LAUNCH(IT, CE_T_BF16, CE_T_F16, CE_T_F16, CE_T_F32); break;        \
        case LNB(CE_T_F16, CE_T_F32, CE_T_F32, CE_T_F32): LAUNCH(IT, CE_T_F16, CE_T_F32, CE_T_F32, CE_T_F32); break;          \
        case LNB(CE_T_F16, CE_T_F16, CE_T_F32, CE_T_F32): LAUNCH(IT, CE_T_F16, CE_T_F16, CE_T_F32, CE_T_F32); break;          \
        default: CE_CHECK_ARG(false, "ce_layernorm_bwd: element types dy=%d x=%d dx_in=%d dx_out=%d are not built", dy_type, \
                              x_type, dxin_type, dx_type);                                                                    \
    }
    LN_DISPATCH(D, CALL);
#undef CALL
#undef LAUNCH
#undef LNB
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_layernorm_bwd(const void* dy, long lddy, int dy_f32, const float* x, long ldx, const int* rows,
                                const float* mean, const float* rstd, const float* w, const float* dx_in,
                                float* dx_out, long lddx, void* dxb, long lddxb, float* dw, float* db, float* dxsum, int M,
                                int D, void* stream) {
    return ce_layernorm_bwd_t(dy, dy_f32 ? CE_T_F32 : CE_T_BF16, lddy, x, CE_T_F32, ldx, rows, mean, rstd, w, dx_in, CE_T_F32,
                              dx_out, CE_T_F32, lddx, dxb, lddxb, dw, db, dxsum, nullptr, M, D, stream);
}
