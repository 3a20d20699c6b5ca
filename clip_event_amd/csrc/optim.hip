// Fused optimiser step over the flat fp32 parameter / gradient buffers (gfx950, HBM-bound).
//
// Replaces engine.py:89-90: torch.nn.utils.clip_grad_norm_(params, 1) followed by
// torch.optim.Adam(lr, weight_decay).step() (L2 weight decay, not AdamW; engine.py:141-146).
// Two launches per step instead of ~900: a sum-of-squares reduction into a device scalar,
// then one Adam pass that reads the scalar (no host sync) and applies the clip coefficient
// max_norm / (norm + 1e-6) (clamped to 1) on the fly.
#include <stdlib.h>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

// CE_ADAM_NT (compile time): cache policy of the p / m / v streams (each element is touched once per step): 0 plain,
// 1 nt stores, 2 (default) nt loads and stores -- tools/bench_hbm.py adam: 914 / 920 / 861 us (4.97 / 4.93 / 5.27 TB/s)
#ifndef CE_ADAM_NT
#define CE_ADAM_NT 2
#endif
#if CE_ADAM_NT >= 1
#define CE_ADAM_ST(val, ptr) __builtin_nontemporal_store(val, ptr)
#else
#define CE_ADAM_ST(val, ptr) (*(ptr) = (val))
#endif
#if CE_ADAM_NT >= 2
#define CE_ADAM_LD(ptr) __builtin_nontemporal_load(ptr)
#else
#define CE_ADAM_LD(ptr) (*(ptr))
#endif

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    const long stride = gridDim.x * 1024L;
    for (long i = blockIdx.x * 1024L + threadIdx.x * 4; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        } else {
            for (long k = i; k < n; ++k) s += g[k] * g[k];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// One element of clip + Adam.  Every kernel of this file updates through this function, with the contraction of a * b + c into
// fused multiply-adds spelled out, so that the flat kernel, the tile kernel and the segment kernel give the same bits.
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float coef, float wd, float b1, float b2,
                                          float lr_bc1, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
    g = g * coef;
    if (wd != 0.f) g = __builtin_fmaf(p, wd, g);
    m = __builtin_fmaf(m, b1, g * (1.0f - b1));
    v = __builtin_fmaf(v, b2, (g * g) * (1.0f - b2));
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = __builtin_fmaf(-lr_bc1, m / denom, p);
}
__device__ __forceinline__ void adam_elem4(f32x4& p, f32x4 g, f32x4& m, f32x4& v, float coef, float wd, float b1, float b2,
                                           float lr_bc1, float bc2_sqrt, float eps) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_elem(pe, g[e], me, ve, coef, wd, b1, b2, lr_bc1, bc2_sqrt, eps);
        p[e] = pe; m[e] = me; v[e] = ve;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, bf16_t* __restrict__ p16, long n,
                                                   const float* __restrict__ sumsq, float max_norm, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    float coef = 1.0f;
    if (sumsq) {
        const float norm = sqrtf(*sumsq);
        coef = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    // two 16-byte chunks per lane and iteration, all eight loads issued before the first use: the pass is pure streaming
    // (16 B read + 14 B written per parameter) and wants as many bytes in flight as the registers allow
    const long stride = gridDim.x * 2048L;
    for (long i0 = blockIdx.x * 2048L + threadIdx.x * 4; i0 < n; i0 += stride) {
        const long i1 = i0 + 1024;
        if (i1 + 3 < n) {
            f32x4 pv[2], gv[2], mv[2], vv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long i = u ? i1 : i0;
                pv[u] = CE_ADAM_LD(reinterpret_cast<f32x4*>(p + i));
                gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i));
                mv[u] = CE_ADAM_LD(reinterpret_cast<f32x4*>(m + i));
                vv[u] = CE_ADAM_LD(reinterpret_cast<f32x4*>(v + i));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long i = u ? i1 : i0;
                adam_elem4(pv[u], gv[u], mv[u], vv[u], coef, wd, b1, b2, lr / bc1, bc2_sqrt, eps);
                CE_ADAM_ST(pv[u], reinterpret_cast<f32x4*>(p + i));
                if (p16) {
                    u32x2 pk = {pack_bf2(pv[u][0], pv[u][1]), pack_bf2(pv[u][2], pv[u][3])};
                    *reinterpret_cast<u32x2*>(p16 + i) = pk;
                }
                CE_ADAM_ST(mv[u], reinterpret_cast<f32x4*>(m + i));
                CE_ADAM_ST(vv[u], reinterpret_cast<f32x4*>(v + i));
            }
        } else {
            for (int u = 0; u < 2; ++u) {
                const long ib = u ? i1 : i0;
                for (long k = ib; k < n && k < ib + 4; ++k) {
                    float pk = p[k], mk = m[k], vk = v[k];
                    adam_elem(pk, g[k], mk, vk, coef, wd, b1, b2, lr / bc1, bc2_sqrt, eps);
                    p[k] = pk;
                    m[k] = mk;
                    v[k] = vk;
                    if (p16) p16[k] = f2bf(pk);
                }
            }
        }
    }
}

// base[table[2b] .. table[2b+1]) = 0 for chunk b (element offsets, multiples of 4): the step's gradient zero-fill minus the
// tensors whose first gradient contribution is a store (ce_gemm_tn_grouped_ex overwrite)
__global__ __launch_bounds__(256) void zero_segments_kernel(float* __restrict__ base, const long* __restrict__ table) {
    const long lo = table[2 * blockIdx.x], hi = table[2 * blockIdx.x + 1];
    for (long i = lo + threadIdx.x * 4; i < hi; i += 1024)
        __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(base + i));
}

// dst[c][r] = src[r][c] for a table of bf16 matrices, one launch: the transposed operand copies of every
// weight (input-gradient GEMMs read W^T) are rebuilt from the bf16 mirror the Adam kernel wrote.
__global__ __launch_bounds__(256) void multi_transpose_kernel(const ce_transpose_job* __restrict__ jobs, int njobs) {
    // 64 x 64 tile through LDS; global accesses are 16 bytes per lane on both sides when rows/cols are multiples
    // of 8 (every weight here), 2 bytes per lane otherwise.  Row stride 66 elements = 33 dwords: the 8 rows a lane
    // gathers for one 16-byte transposed store sit in 8 different banks.
    __shared__ bf16_t tile[64][66];
    int j = 0;
    const int b = blockIdx.x;
    while (j + 1 < njobs && b >= jobs[j + 1].tile_start) ++j;      // block-uniform
    const ce_transpose_job job = jobs[j];
    const int t = b - job.tile_start;
    const int tiles_c = (job.cols + 63) / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(job.src);
    bf16_t* dst = reinterpret_cast<bf16_t*>(job.dst);
    const bool wide = (job.rows % 8 == 0) && (job.cols % 8 == 0) &&
                      ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) % 16 == 0);
    if (wide) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = threadIdx.x + k * 256;                 // 64 rows x 8 chunks
            const int r = idx >> 3, ch = idx & 7;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r0 + r < job.rows && c0 + ch * 8 < job.cols)
                v = *reinterpret_cast<const u32x4*>(src + (long)(r0 + r) * job.cols + c0 + ch * 8);
            uint32_t* trow = reinterpret_cast<uint32_t*>(&tile[r][ch * 8]);
            trow[0] = v[0]; trow[1] = v[1]; trow[2] = v[2]; trow[3] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = threadIdx.x + k * 256;                 // 64 output rows (source columns) x 8 chunks
            const int c = idx >> 3, ch = idx & 7;
            if (c0 + c < job.cols && r0 + ch * 8 < job.rows) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    w[e] = (uint32_t)tile[ch * 8 + 2 * e][c] | ((uint32_t)tile[ch * 8 + 2 * e + 1][c] << 16);
                u32x4 v = {w[0], w[1], w[2], w[3]};
                *reinterpret_cast<u32x4*>(dst + (long)(c0 + c) * job.rows + r0 + ch * 8) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;        // 64 x 4
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        tile[ty + 4 * k][tx] = (r < job.rows && c < job.cols) ? src[(long)r * job.cols + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < job.rows && c < job.cols) dst[(long)c * job.rows + r] = tile[tx][ty + 4 * k];
    }
}

// clip + Adam over a table of bf16-mirrored MATRICES, one 64 x 64 tile per workgroup: the update of `adam_kernel` element for
// element, and with it BOTH operand copies of the new weights -- the row-major bf16 mirror and, through an LDS transpose of the
// tile, the W^T copy the input-gradient GEMMs read (`job.dst`, [cols][rows]) -- so that no separate transpose pass re-reads the
// mirror (`multi_transpose_kernel`: 0.17 GB read + a launch beside the next forward's first kernels).  `job.src` points into the
// flat mirror `p16`: its offset there is the matrix's offset in every flat buffer.  rows, cols multiples of 8.
__global__ __launch_bounds__(256) void adam_tiles_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ p16,
                                                         const ce_transpose_job* __restrict__ jobs, int njobs,
                                                         const float* __restrict__ sumsq, float max_norm, float lr, float b1, float b2,
                                                         float eps, float wd, float bc1, float bc2_sqrt) {
    __shared__ bf16_t tile[64][66];
    float coef = 1.0f;
    if (sumsq) coef = fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    int j = 0;
    const int b = blockIdx.x;
    while (j + 1 < njobs && b >= jobs[j + 1].tile_start) ++j;      // block-uniform
    const ce_transpose_job job = jobs[j];
    const int t = b - job.tile_start;
    const int tiles_c = (job.cols + 63) / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const long off = reinterpret_cast<const bf16_t*>(job.src) - p16;
    bf16_t* dst = reinterpret_cast<bf16_t*>(job.dst);
    // 64 rows x 16 four-element chunks; all sixteen loads of a thread in flight before the first use
    f32x4 pv[4], gv[4], mv[4], vv[4];
    bool ok[4];
    long at[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x + k * 256;
        const int r = idx >> 4, ch = idx & 15;
        ok[k] = r0 + r < job.rows && c0 + ch * 4 < job.cols;
        at[k] = off + (long)(r0 + r) * job.cols + c0 + ch * 4;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pv[k] = z; gv[k] = z; mv[k] = z; vv[k] = z;
        if (ok[k]) {
            pv[k] = CE_ADAM_LD(reinterpret_cast<f32x4*>(p + at[k]));
            gv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + at[k]));
            mv[k] = CE_ADAM_LD(reinterpret_cast<f32x4*>(m + at[k]));
            vv[k] = CE_ADAM_LD(reinterpret_cast<f32x4*>(v + at[k]));
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x + k * 256;
        const int r = idx >> 4, ch = idx & 15;
        adam_elem4(pv[k], gv[k], mv[k], vv[k], coef, wd, b1, b2, lr / bc1, bc2_sqrt, eps);
        const u32x2 pk = {pack_bf2(pv[k][0], pv[k][1]), pack_bf2(pv[k][2], pv[k][3])};
        uint32_t* trow = reinterpret_cast<uint32_t*>(&tile[r][ch * 4]);
        trow[0] = pk[0]; trow[1] = pk[1];
        if (ok[k]) {
            CE_ADAM_ST(pv[k], reinterpret_cast<f32x4*>(p + at[k]));
            __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(p16 + at[k]));       // (nt on the mirror and on W^T: 1034 -> 986 us for
                                                                                          //  the ViT-B/32 step in a loop of its own)
            CE_ADAM_ST(mv[k], reinterpret_cast<f32x4*>(m + at[k]));
            CE_ADAM_ST(vv[k], reinterpret_cast<f32x4*>(v + at[k]));
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = threadIdx.x + k * 256;                 // 64 output rows (source columns) x 8 chunks of 8 source rows
        const int c = idx >> 3, ch = idx & 7;
        if (c0 + c < job.cols && r0 + ch * 8 < job.rows) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = (uint32_t)tile[ch * 8 + 2 * e][c] | ((uint32_t)tile[ch * 8 + 2 * e + 1][c] << 16);
            __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4*>(dst + (long)(c0 + c) * job.rows + r0 + ch * 8));
        }
    }
}

// the same update over a table of [lo, hi) chunks of the flat buffers (everything that is not one of the matrices above): one
// workgroup per chunk of at most 2048 elements, both 16-byte pieces of a thread requested before the first use (as adam_kernel)
__global__ __launch_bounds__(256) void adam_segments_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, bf16_t* __restrict__ p16, const long* __restrict__ table,
                                                            const float* __restrict__ sumsq, float max_norm, float lr, float b1, float b2,
                                                            float eps, float wd, float bc1, float bc2_sqrt) {
    float coef = 1.0f;
    if (sumsq) coef = fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    const long lo = table[2 * blockIdx.x], hi = table[2 * blockIdx.x + 1];
    f32x4 pv[2], gv[2], mv[2], vv[2];
    long at[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        at[u] = lo + threadIdx.x * 4 + u * 1024;
        ok[u] = at[u] + 3 < hi;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pv[u] = z; gv[u] = z; mv[u] = z; vv[u] = z;
        if (ok[u]) {
            pv[u] = CE_ADAM_LD(reinterpret_cast<f32x4*>(p + at[u]));
            gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + at[u]));
            mv[u] = CE_ADAM_LD(reinterpret_cast<f32x4*>(m + at[u]));
            vv[u] = CE_ADAM_LD(reinterpret_cast<f32x4*>(v + at[u]));
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!ok[u]) continue;
        adam_elem4(pv[u], gv[u], mv[u], vv[u], coef, wd, b1, b2, lr / bc1, bc2_sqrt, eps);
        CE_ADAM_ST(pv[u], reinterpret_cast<f32x4*>(p + at[u]));
        if (p16) *reinterpret_cast<u32x2*>(p16 + at[u]) = u32x2{pack_bf2(pv[u][0], pv[u][1]), pack_bf2(pv[u][2], pv[u][3])};
        CE_ADAM_ST(mv[u], reinterpret_cast<f32x4*>(m + at[u]));
        CE_ADAM_ST(vv[u], reinterpret_cast<f32x4*>(v + at[u]));
    }
}

}  // namespace

extern "C" int ce_multi_transpose_bf16(const ce_transpose_job* jobs_device, int njobs, int total_tiles, void* stream) {
    CE_CHECK_ARG(jobs_device && njobs > 0 && total_tiles > 0, "ce_multi_transpose_bf16: empty");
    hipLaunchKernelGGL(multi_transpose_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_zero_segments(float* base, const long* table_device, int nchunks, void* stream) {
    CE_CHECK_ARG(base && table_device && nchunks > 0, "ce_zero_segments: empty");
    hipLaunchKernelGGL(zero_segments_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, base, table_device);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_sumsq(const float* g, long n, float* out, void* stream) {
    CE_CHECK_ARG(n > 0, "ce_sumsq: empty");
    long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const float* sumsq,
                            float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                            void* stream) {
    CE_CHECK_ARG(n > 0 && step >= 1, "ce_adam_step: need n>0 and step>=1");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    long blocks = (n + 2047) / 2048;
    static const long cap = getenv("CE_ADAM_BLOCKS") ? atol(getenv("CE_ADAM_BLOCKS")) : 4096;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, sumsq,
                       max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_adam_step_tiles(float* p, const float* g, float* m, float* v, void* p_bf16, const ce_transpose_job* jobs_device,
                                  int njobs, int total_tiles, const long* segments_device, int nsegments, const float* sumsq,
                                  float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                  void* stream) {
    CE_CHECK_ARG(p && g && m && v && p_bf16 && step >= 1, "ce_adam_step_tiles: null buffer or step < 1");
    CE_CHECK_ARG((jobs_device && njobs > 0 && total_tiles > 0) || (segments_device && nsegments > 0), "ce_adam_step_tiles: nothing to update");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    if (njobs > 0)
        hipLaunchKernelGGL(adam_tiles_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                           jobs_device, njobs, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    if (nsegments > 0)
        hipLaunchKernelGGL(adam_segments_kernel, dim3((unsigned)nsegments), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                           segments_device, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    CE_LAUNCH_CHECK();
    return 0;
}
