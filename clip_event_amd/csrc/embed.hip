// Input-side and bookkeeping kernels of the two towers (gfx950): all HBM-bound, vectorised
// 16-byte accesses, no inter-workgroup communication except final float atomics.
//
//   ce_im2col            image fp32 NCHW -> bf16 patch rows      (Conv2d k=s=patch, model_clip.py:219,235)
//   ce_vision_assemble   [cls | patches] + positional embedding   (model_clip.py:237-242)
//   ce_vision_assemble_bwd
//   ce_token_embed       token_embedding[ids] + positional        (model_clip.py:400-403)
//   ce_token_embed_bwd   scatter-add into the embedding gradient
//   ce_batch_reduce      sum over the batch axis (positional / class embedding gradients)
//   ce_colsum_bf16       bias gradients (column sums of a bf16 activation gradient)
//   ce_cast_transpose    fp32 master weight -> bf16 copy + bf16 transposed copy
//   ce_cast_bf16         fp32 -> bf16
//   ce_eot_rows          argmax token id per row -> flat row index (model_clip.py:415)
#include <limits.h>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

// one thread = 8 consecutive output columns of one patch row
__global__ void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int R, int ps, int grid,
                              int Kp) {
    const long total = (long)B * grid * grid * (Kp / 8);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % (Kp / 8));
        const long row = idx / (Kp / 8);
        const int gx = (int)(row % grid), gy = (int)((row / grid) % grid), b = (int)(row / ((long)grid * grid));
        float v[8];
        if (ps % 8 == 0 && R % 4 == 0) {
            // patch sizes 16 / 32: the 8 columns are 8 consecutive pixels of one image row -> two 16-byte loads
            const int col = cg * 8;
            const int c = col / (ps * ps), py = (col / ps) % ps, px = col % ps;
            const float* src = img + (((long)b * 3 + c) * R + gy * ps + py) * R + gx * ps + px;
            const f32x4 lo = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            const f32x4 hi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 4));
            u32x4 pk = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
            *reinterpret_cast<u32x4*>(out + row * Kp + cg * 8) = pk;
            continue;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = cg * 8 + e;
            float val = 0.f;
            if (col < 3 * ps * ps) {
                const int c = col / (ps * ps), py = (col / ps) % ps, px = col % ps;
                val = img[(((long)b * 3 + c) * R + gy * ps + py) * R + gx * ps + px];
            }
            v[e] = val;
        }
        u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(out + row * Kp + cg * 8) = pk;
    }
}

// x0[b, t, :] = (t == 0 ? cls : patch[b*P + t-1, :]) + pos[t, :]      (fp32, 4 columns per thread)
__global__ void vision_assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                       const float* __restrict__ pos, float* __restrict__ x0, int B, int Ltok, int D) {
    const long total = (long)B * Ltok * (D / 4);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (D / 4)) * 4;
        const long row = idx / (D / 4);
        const int t = (int)(row % Ltok);
        const long b = row / Ltok;
        f32x4 v = (t == 0) ? *reinterpret_cast<const f32x4*>(cls + c)
                           : *reinterpret_cast<const f32x4*>(patch + (b * (Ltok - 1) + t - 1) * D + c);
        v += *reinterpret_cast<const f32x4*>(pos + (long)t * D + c);
        *reinterpret_cast<f32x4*>(x0 + row * D + c) = v;
    }
}

// dpatch (bf16) [B*P, D] = dx0[b, 1+p, :]
__global__ void vision_assemble_bwd_kernel(const float* __restrict__ dx0, bf16_t* __restrict__ dpatch, int B, int Ltok,
                                           int D) {
    const long total = (long)B * (Ltok - 1) * (D / 4);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (D / 4)) * 4;
        const long row = idx / (D / 4);
        const long b = row / (Ltok - 1);
        const int p = (int)(row % (Ltok - 1));
        f32x4 v = *reinterpret_cast<const f32x4*>(dx0 + (b * Ltok + p + 1) * D + c);
        u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(dpatch + row * D + c) = pk;
    }
}

// `src` (nullable): packed batch, output row r is element src[r] of the flattened [n, Ltok] id matrix
__global__ void token_embed_kernel(const long* __restrict__ ids, const int* __restrict__ src,
                                   const float* __restrict__ table, const float* __restrict__ pos,
                                   void* __restrict__ x0, int x0_type, long rows, int Ltok, int D, int vocab) {
    const long total = rows * (D / 4);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (D / 4)) * 4;
        const long row = idx / (D / 4);
        const long e = src ? (long)src[row] : row;
        long id = ids[e];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // ids are validated on the host; clamp keeps the access in bounds
        f32x4 v = *reinterpret_cast<const f32x4*>(table + id * D + c);
        v += *reinterpret_cast<const f32x4*>(pos + (long)(e % Ltok) * D + c);
        store4_t(x0, row * D + c, x0_type, v);
    }
}

// dtable[ids[row], :] += dx0[row, :]   one wave-instruction = 256 contiguous bytes of one row
// Exact zeros are skipped: under the causal mask the rows after a caption's EOT receive no gradient at all, and
// they all carry the padding id 0 -- without the skip they serialise thousands of same-address atomics.
__global__ void token_embed_bwd_kernel(const long* __restrict__ ids, const int* __restrict__ src,
                                       const float* __restrict__ dx0, float* __restrict__ dtable, long rows, int D,
                                       int vocab) {
    const long total = rows * D;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / D;
        const int c = (int)(idx % D);
        const float v = dx0[idx];
        if (v == 0.f) continue;
        long id = ids[src ? (long)src[row] : row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        atomicAdd(dtable + id * D + c, v);
    }
}

// Positional-embedding gradient of a packed batch: dpos[t, :] += sum over the samples b longer than t of
// dx0[cu[b] + t, :].  grid = (Ltok, SPLIT): each block sums every SPLIT-th sample, D/4 threads x 4 columns.
__global__ void pos_embed_bwd_packed_kernel(const float* __restrict__ dx0, const int* __restrict__ cu,
                                            float* __restrict__ dpos, int n, int D) {
    const int t = blockIdx.x;
    const int c = threadIdx.x * 4;
    if (c >= D) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const int r0 = cu[b], len = cu[b + 1] - r0;         // block-uniform
        if (t < len) acc += *reinterpret_cast<const f32x4*>(dx0 + (long)(r0 + t) * D + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (acc[e] != 0.f) atomicAdd(dpos + (long)t * D + c + e, acc[e]);
}

// dst[r, c] += src[r, c], c < cols (row strides differ: real columns of a column-padded gradient)
__global__ void add_cols_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, int rows,
                                int cols) {
    const long total = (long)rows * cols;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols;
        const int c = (int)(idx - r * cols);
        dst[r * ldd + c] += src[r * lds + c];
    }
}

// out[t, c] (+)= sum_b x[b, t, c]  over `B` slabs of `slab` floats, rows [row0, row0+nrows) of each slab
__global__ void batch_reduce_kernel(const float* __restrict__ x, float* __restrict__ out, int B, long slab, long n,
                                    int accumulate) {
    const long i4 = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 4 <= B; b += 4) {
        f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (long)b * slab + i4);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (long)(b + 1) * slab + i4);
        f32x4 a2 = *reinterpret_cast<const f32x4*>(x + (long)(b + 2) * slab + i4);
        f32x4 a3 = *reinterpret_cast<const f32x4*>(x + (long)(b + 3) * slab + i4);
        acc += (a0 + a1) + (a2 + a3);
    }
    for (; b < B; ++b) acc += *reinterpret_cast<const f32x4*>(x + (long)b * slab + i4);
    if (accumulate) acc += *reinterpret_cast<const f32x4*>(out + i4);
    *reinterpret_cast<f32x4*>(out + i4) = acc;
}

// out[n] += sum_m x[m, n].  Block = 32 column groups (8 bf16 = 16 B per lane) x 8 row groups over a
// 64-row slab; 4 independent 16-byte loads in flight per thread; LDS reduce over the row groups,
// then one atomic per column per block.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, float* __restrict__ out,
                                                          int M, int N) {
    __shared__ float red[8][257];
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + cg * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        const int r0 = blockIdx.y * 64 + rg;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + (h * 4 + u) * 8;
                v[u] = (r < M) ? *reinterpret_cast<const u32x4*>(x + (long)r * ld + c) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += bf_lo(v[u][e]);
                    acc[2 * e + 1] += bf_hi(v[u][e]);
                }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cg * 8 + e] = acc[e];
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
        atomicAdd(out + col, t);
    }
}

// w[R, C] fp32 -> w16[R, C] bf16 and w16t[C, R] bf16 through a 32x33 LDS tile
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, bf16_t* __restrict__ w16,
                                                             long ld16, bf16_t* __restrict__ w16t, long ld16t, int R,
                                                             int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = (r < R && c < C) ? w[(long)r * C + c] : 0.f;
        tile[ty + 8 * k][tx] = v;
        if (w16 && r < R && c < C) w16[(long)r * ld16 + c] = f2bf(v);
    }
    __syncthreads();
    if (w16t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < R && c < C) w16t[(long)c * ld16t + r] = f2bf(tile[tx][ty + 8 * k]);
        }
    }
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    const long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(y + i) = pk;
    } else {
        for (long k = i; k < n; ++k) y[k] = f2bf(x[k]);
    }
}

// y = x * mul between the stream element types (fp32 / bf16 / fp16 with saturation); n % 4 == 0
__global__ void cast_t_kernel(const void* __restrict__ x, int src_type, void* __restrict__ y, int dst_type, float mul, long n) {
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4)
        store4_t(y, i, dst_type, load4_t(x, i, src_type) * mul);
}

// y = x * s or x / s with s in device memory (the gradient stream's scale)
__global__ void cast_scaled_kernel(const void* __restrict__ x, int src_type, void* __restrict__ y, int dst_type,
                                   const float* __restrict__ scale, int divide, long n) {
    const float s = *scale;
    const float mul = divide ? 1.0f / s : s;
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4)
        store4_t(y, i, dst_type, load4_t(x, i, src_type) * mul);
}

// scale of an fp16 gradient stream: the power of two that puts max|x| at `target` (stores saturate at 65504, so the
// headroom between target and 65504 is what the gradient may grow by on its way down the tower)
__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    float m = 0.f;
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i + 3 < n; i += (long)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n & ~3L) + threadIdx.x]));
    m = wave_max(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ void grad_scale_final_kernel(const float* __restrict__ part, int nparts, float target, float* __restrict__ scale) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) m = fmaxf(m, part[i]);
    m = wave_max(m);
    if (threadIdx.x == 0) {
        float s = 1.0f;
        if (m > 0.f && m < INFINITY) {
            int e;
            frexpf(target / m, &e);                      // target / m = f 2^e, f in [0.5, 1): 2^(e-1) <= target / m
            e = e - 1 < -24 ? -24 : (e - 1 > 60 ? 60 : e - 1);
            s = ldexpf(1.0f, e);
        }
        *scale = s;
    }
}

// rows[r] = r*Ltok + argmax_t ids[r, t]  (first maximum, like torch.argmax on these rows).  One wave per row, lane-strided
// loads and a butterfly on (value, index): one memory latency per row (a thread per row walked 77 dependent loads:
// 35 us at the head of the text path, ahead of the packing's host read-back).
__global__ __launch_bounds__(256) void eot_rows_kernel(const long* __restrict__ ids, int* __restrict__ rows, long n, int Ltok) {
    const int lane = threadIdx.x & 63;
    const long r = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    long best = LONG_MIN;
    int arg = INT_MAX;
    for (int t = lane; t < Ltok; t += 64) {
        const long v = ids[r * Ltok + t];
        if (v > best) {
            best = v;
            arg = t;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (ob > best || (ob == best && oa < arg)) {
            best = ob;
            arg = oa;
        }
    }
    if (lane == 0) rows[r] = (int)(r * Ltok + arg);
}

// dst[dst_rows ? dst_rows[i] : i, :] = src[src_rows ? src_rows[i] : i, :]   (row_bytes multiple of 16)
__global__ void copy_rows_kernel(const char* __restrict__ src, long src_stride, const int* __restrict__ src_rows,
                                 char* __restrict__ dst, long dst_stride, const int* __restrict__ dst_rows, int n,
                                 int chunks) {
    const long total = (long)n * chunks;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / chunks), c = (int)(idx - (long)i * chunks);
        const long sr = src_rows ? src_rows[i] : i, dr = dst_rows ? dst_rows[i] : i;
        *reinterpret_cast<u32x4*>(dst + dr * dst_stride + c * 16L) =
            *reinterpret_cast<const u32x4*>(src + sr * src_stride + c * 16L);
    }
}

// dst[r, :] = (r == dst_rows[i] for some i) ? src[i, :] : 0 for every r < M; dst_rows ascending (binary search per row)
__global__ void scatter_rows_zero_kernel(const char* __restrict__ src, long src_stride, char* __restrict__ dst, long dst_stride,
                                         const int* __restrict__ dst_rows, int n, int M, int chunks) {
    const long total = (long)M * chunks;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / chunks), c = (int)(idx - (long)r * chunks);
        int lo = 0, hi = n - 1;                       // first i with dst_rows[i] >= r
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (dst_rows[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        u32x4 v = {0u, 0u, 0u, 0u};
        if (dst_rows[lo] == r) v = *reinterpret_cast<const u32x4*>(src + (long)lo * src_stride + c * 16L);
        *reinterpret_cast<u32x4*>(dst + (long)r * dst_stride + c * 16L) = v;
    }
}

int grid_for(long threads, int block = 256, int cap = 8192) {
    long g = (threads + block - 1) / block;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int ce_im2col(const float* image, void* patches, int B, int resolution, int patch, int k_padded,
                         void* stream) {
    CE_CHECK_ARG(B > 0 && patch > 0 && resolution % patch == 0, "ce_im2col: resolution %d not a multiple of patch %d", resolution, patch);
    CE_CHECK_ARG(k_padded % 8 == 0 && k_padded >= 3 * patch * patch, "ce_im2col: k_padded must be a multiple of 8 >= 3*patch^2");
    const int grid = resolution / patch;
    const long threads = (long)B * grid * grid * (k_padded / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(threads)), dim3(256), 0, (hipStream_t)stream, image,
                       (bf16_t*)patches, B, resolution, patch, grid, k_padded);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_vision_assemble(const float* patch_out, const float* cls, const float* pos, float* x0, int B,
                                  int tokens, int D, void* stream) {
    CE_CHECK_ARG(B > 0 && tokens > 1 && D % 4 == 0, "ce_vision_assemble: bad shape");
    hipLaunchKernelGGL(vision_assemble_kernel, dim3(grid_for((long)B * tokens * (D / 4))), dim3(256), 0,
                       (hipStream_t)stream, patch_out, cls, pos, x0, B, tokens, D);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_vision_assemble_bwd(const float* dx0, void* dpatch, int B, int tokens, int D, void* stream) {
    CE_CHECK_ARG(B > 0 && tokens > 1 && D % 4 == 0, "ce_vision_assemble_bwd: bad shape");
    hipLaunchKernelGGL(vision_assemble_bwd_kernel, dim3(grid_for((long)B * (tokens - 1) * (D / 4))), dim3(256), 0,
                       (hipStream_t)stream, dx0, (bf16_t*)dpatch, B, tokens, D);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_token_embed_t(const int64_t* ids, const int* src_rows, const float* table, const float* pos, void* x0,
                                int x0_type, long rows, int tokens, int D, int vocab, void* stream) {
    CE_CHECK_ARG(rows > 0 && tokens > 0 && D % 4 == 0 && vocab > 0, "ce_token_embed: bad shape");
    CE_CHECK_ARG(x0_type == CE_T_F32 || x0_type == CE_T_F16, "ce_token_embed: x0 element type %d", x0_type);
    hipLaunchKernelGGL(token_embed_kernel, dim3(grid_for(rows * (D / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const long*)ids, src_rows, table, pos, x0, x0_type, rows, tokens, D, vocab);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_token_embed(const int64_t* ids, const int* src_rows, const float* table, const float* pos, float* x0,
                              long rows, int tokens, int D, int vocab, void* stream) {
    return ce_token_embed_t(ids, src_rows, table, pos, x0, CE_T_F32, rows, tokens, D, vocab, stream);
}

extern "C" int ce_token_embed_bwd(const int64_t* ids, const int* src_rows, const float* dx0, float* dtable, long rows,
                                  int D, int vocab, void* stream) {
    CE_CHECK_ARG(rows > 0 && D > 0 && vocab > 0, "ce_token_embed_bwd: bad shape");
    hipLaunchKernelGGL(token_embed_bwd_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream,
                       (const long*)ids, src_rows, dx0, dtable, rows, D, vocab);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_pos_embed_bwd_packed(const float* dx0, const int* cu_seqlens, float* dpos, int n, int tokens, int D,
                                       void* stream) {
    CE_CHECK_ARG(n > 0 && tokens > 0 && D > 0 && D % 4 == 0 && D <= 4096 && cu_seqlens, "ce_pos_embed_bwd_packed: bad shape");
    const int split = n < 8 ? n : 8;
    const int threads = ((D / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(pos_embed_bwd_packed_kernel, dim3(tokens, split), dim3(threads), 0, (hipStream_t)stream, dx0,
                       cu_seqlens, dpos, n, D);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_batch_reduce(const float* x, float* out, int B, long slab, long n, int accumulate, void* stream) {
    CE_CHECK_ARG(B > 0 && n > 0 && n % 4 == 0 && slab % 4 == 0 && n <= slab, "ce_batch_reduce: bad shape");
    hipLaunchKernelGGL(batch_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       out, B, slab, n, accumulate);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_colsum_bf16(const void* x, long ld, float* out, int M, int N, void* stream) {
    CE_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "ce_colsum_bf16: N and ld must be multiples of 8");
    const int gx = ce_div_up(N, 256);
    const int gy = ce_div_up(M, 64);
    CeProfScope prof(CE_PROF_COLSUM, (double)M * N, 2.0 * M * N, (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, out, M,
                       N);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_cast_transpose(const float* w, void* w16, long ld16, void* w16t, long ld16t, int R, int C,
                                 void* stream) {
    CE_CHECK_ARG(R > 0 && C > 0, "ce_cast_transpose: bad shape");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(ce_div_up(C, 32), ce_div_up(R, 32)), dim3(256), 0, (hipStream_t)stream,
                       w, (bf16_t*)w16, ld16, (bf16_t*)w16t, ld16t, R, C);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_add_cols(const float* src, long lds, float* dst, long ldd, int rows, int cols, void* stream) {
    CE_CHECK_ARG(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "ce_add_cols: bad shape");
    hipLaunchKernelGGL(add_cols_kernel, dim3(grid_for((long)rows * cols)), dim3(256), 0, (hipStream_t)stream, src, lds, dst,
                       ldd, rows, cols);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_cast_t(const void* x, int src_type, void* y, int dst_type, float mul, long n, void* stream) {
    CE_CHECK_ARG(n > 0 && n % 4 == 0, "ce_cast_t: n must be a positive multiple of 4 (n=%ld)", n);
    CE_CHECK_ARG(src_type >= CE_T_F32 && src_type <= CE_T_F16 && dst_type >= CE_T_F32 && dst_type <= CE_T_F16,
                 "ce_cast_t: element types %d -> %d", src_type, dst_type);
    hipLaunchKernelGGL(cast_t_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, src_type, y, dst_type, mul, n);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_cast_scaled(const void* x, int src_type, void* y, int dst_type, const float* scale, int divide, long n,
                              void* stream) {
    CE_CHECK_ARG(n > 0 && n % 4 == 0 && scale, "ce_cast_scaled: n must be a positive multiple of 4 (n=%ld), scale a device pointer", n);
    CE_CHECK_ARG(src_type >= CE_T_F32 && src_type <= CE_T_F16 && dst_type >= CE_T_F32 && dst_type <= CE_T_F16,
                 "ce_cast_scaled: element types %d -> %d", src_type, dst_type);
    hipLaunchKernelGGL(cast_scaled_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, src_type, y, dst_type,
                       scale, divide, n);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_grad_scale(const float* x, long n, float target, float* scratch, float* scale, void* stream) {
    CE_CHECK_ARG(x && n > 0 && scratch && scale && target > 0.f, "ce_grad_scale: bad arguments");
    int blocks = (int)((n / 4 + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > CE_GRAD_SCALE_SCRATCH ? CE_GRAD_SCALE_SCRATCH : blocks);
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
    hipLaunchKernelGGL(grad_scale_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, blocks, target, scale);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_cast_bf16(const float* x, void* y, long n, void* stream) {
    CE_CHECK_ARG(n > 0, "ce_cast_bf16: empty");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (bf16_t*)y, n);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_scatter_rows_zero(const void* src, long src_stride_bytes, void* dst, long dst_stride_bytes,
                                    const int* dst_rows, int n, int M, int row_bytes, void* stream) {
    CE_CHECK_ARG(n > 0 && M >= n && row_bytes > 0 && row_bytes % 16 == 0 && src_stride_bytes % 16 == 0 && dst_stride_bytes % 16 == 0 &&
                 dst_rows, "ce_scatter_rows_zero: rows must be multiples of 16 bytes, 0 < n <= M");
    const int chunks = row_bytes / 16;
    hipLaunchKernelGGL(scatter_rows_zero_kernel, dim3(grid_for((long)M * chunks)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, src_stride_bytes, (char*)dst, dst_stride_bytes, dst_rows, n, M, chunks);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_eot_rows(const int64_t* ids, int* rows, long n, int tokens, void* stream) {
    CE_CHECK_ARG(n > 0 && tokens > 0, "ce_eot_rows: bad shape");
    hipLaunchKernelGGL(eot_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const long*)ids, rows, n, tokens);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_copy_rows(const void* src, long src_stride_bytes, const int* src_rows, void* dst,
                            long dst_stride_bytes, const int* dst_rows, int n, int row_bytes, void* stream) {
    CE_CHECK_ARG(n > 0 && row_bytes > 0 && row_bytes % 16 == 0 && src_stride_bytes % 16 == 0 && dst_stride_bytes % 16 == 0,
                 "ce_copy_rows: rows must be multiples of 16 bytes");
    const int chunks = row_bytes / 16;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((long)n * chunks)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, src_stride_bytes, src_rows, (char*)dst, dst_stride_bytes, dst_rows, n, chunks);
    CE_LAUNCH_CHECK();
    return 0;
}

// ---- CE_DIAG: a "CU hog" (sizing the contention between the GEMM grids and RCCL's kernels on one GPU) ----
// `blocks` workgroups of 256 threads that each hold 96 KiB of LDS (so no two share a CU, and a CU that hosts one cannot take a
// 156 KiB GEMM workgroup) and spin on the 100 MHz wall clock for `microseconds`: what a ring all-reduce's channels look
// like to the rest of the chip, minus the memory traffic.  Every wave leaves the loop once the time is up (bounded spin).
namespace {
__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks, unsigned int* sink, int mode) {
    extern __shared__ unsigned int hog_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned int n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (mode == 0) __builtin_amdgcn_s_sleep(32);
        else if (mode == 1) __builtin_amdgcn_s_sleep(127);
        ++n;                                                      // mode 2: busy spin
    }
    hog_lds[threadIdx.x & 255] = n;
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) {       // shader-clock ticks and 100 MHz ticks this workgroup lived for
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        reinterpret_cast<unsigned long long*>(sink)[0] = c1 - c0;
        reinterpret_cast<unsigned long long*>(sink)[1] = r1 - t0;
        sink[4] = hog_lds[(threadIdx.x + 1) & 255];           // keeps the LDS allocation and the loop alive
    }
}
}  // namespace

static unsigned int* g_hog_sink = nullptr;
// shader clock (MHz) the last hog's first workgroup saw over its lifetime: s_memtime ticks per 100 MHz s_memrealtime tick
// (synchronises the device)
extern "C" double ce_cu_hog_clock_mhz(void) {
    if (!g_hog_sink) return 0.0;
    unsigned long long h[2] = {0, 0};
    hipDeviceSynchronize();
    hipMemcpy(h, g_hog_sink, sizeof(h), hipMemcpyDeviceToHost);
    return h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
}

extern "C" int ce_cu_hog(int blocks, float microseconds, void* stream) {
    CE_CHECK_ARG(blocks > 0 && blocks <= 256 && microseconds > 0.f && microseconds <= 1.0e6f, "ce_cu_hog: blocks 1..256, time up to 1 s");
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    const unsigned long long ticks = (unsigned long long)(microseconds * 100.0f);      // s_memrealtime counts at 100 MHz
    // CE_HOG_LDS / CE_HOG_THREADS (diagnostics): footprint of one hog workgroup (default 96 KiB, 256 threads)
    static const int hog_lds = getenv("CE_HOG_LDS") ? atoi(getenv("CE_HOG_LDS")) : 96 * 1024;
    static const int hog_threads = getenv("CE_HOG_THREADS") ? atoi(getenv("CE_HOG_THREADS")) : 256;
    static const int hog_mode = getenv("CE_HOG_MODE") ? atoi(getenv("CE_HOG_MODE")) : 0;     // 0 s_sleep 32, 1 s_sleep 127, 2 busy spin
    static const int hog_chop = getenv("CE_HOG_CHOP") ? atoi(getenv("CE_HOG_CHOP")) : 1;     // the time as this many back-to-back launches
    if (!g_hog_sink) hipMalloc(&g_hog_sink, 64);
    for (int c = 0; c < hog_chop; ++c)
        hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(hog_threads), hog_lds < 1024 ? 1024 : hog_lds, (hipStream_t)stream,
                           ticks / hog_chop, g_hog_sink, hog_mode);
    CE_LAUNCH_CHECK();
    return 0;
}
