// On-device image preprocessing: the reference's
//   Compose([Resize(n, BICUBIC), CenterCrop(n), convert("RGB"), ToTensor(), Normalize(mean, std)])   clip.py:62-69
// and the object-patch variant (image.crop(bbox) first, dataset_voa.py:222-233), for a batch of uint8 HWC images of
// different sizes, bit for bit: the resampling is Pillow's (Resample.c): separable, horizontal pass first, 8-bit
// intermediate, taps computed in double, normalised, rounded to 22-bit fixed point.  HBM-bound byte work:
//   coeff kernel : one thread per (output, axis, index) -> taps + bounds of the n cropped columns / rows
//   pass 1       : thread per (output, needed source row, column, channel) -> uint8 [rows][n][3]
//   pass 2       : thread per (output, row, column) -> vertical taps, /255, (x - mean) / std -> fp32 [3][n][n]
#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// Resample.c bicubic_filter, a = -0.5.  No FMA contraction: the taps must round like Pillow's (plain x86-64 doubles).
__device__ double bicubic(double x) {
#pragma clang fp contract(off)
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
    return 0.0;
}

// taps[o][axis][i][k], bounds[o][axis][i] = {first source index, tap count} for cropped output index i
__global__ void preproc_coeff_kernel(const ce_preproc_desc* __restrict__ descs, int n_out, int n_px, int kmax,
                                     int* __restrict__ taps, int* __restrict__ bounds) {
#pragma clang fp contract(off)
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)n_out * 2 * n_px) return;
    const int i = (int)(idx % n_px), axis = (int)((idx / n_px) & 1), o = (int)(idx / (2L * n_px));
    const ce_preproc_desc d = descs[o];
    const int in_size = axis ? d.h : d.w, out_size = axis ? d.oh : d.ow, first = axis ? d.top : d.left;
    int* k = taps + idx * kmax;
    int* b = bounds + idx * 2;
    const int xx = first + i;
    if (in_size == out_size) {          // Pillow skips the pass: identity
        k[0] = 1 << PRECISION_BITS;
        b[0] = xx; b[1] = 1;
        return;
    }
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > kmax) xmax = kmax;       // host sized kmax from the largest scale: never taken
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
        double w = bicubic((x + xmin - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        k[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    b[0] = xmin; b[1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[o][r][x][c] = horizontal resample of ROI row (row0 + r), cropped columns, r < rows
__global__ void preproc_h_kernel(const ce_preproc_desc* __restrict__ descs, int n_px, int kmax,
                                 const int* __restrict__ taps, const int* __restrict__ bounds,
                                 unsigned char* __restrict__ tmp) {
    const int o = blockIdx.y;
    const ce_preproc_desc d = descs[o];
    const long total = (long)d.rows * n_px * 3;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % 3), x = (int)((idx / 3) % n_px), r = (int)(idx / (3L * n_px));
        const long t = ((long)o * 2 + 0) * n_px + x;
        const int x0 = bounds[t * 2], cnt = bounds[t * 2 + 1];
        const int* k = taps + t * kmax;
        const unsigned char* src = d.src + (long)(d.y0 + d.row0 + r) * d.pitch + (long)(d.x0 + x0) * 3 + c;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int j = 0; j < cnt; ++j) acc += (int)src[j * 3] * k[j];
        tmp[d.tmp_off + idx] = clip8(acc);
    }
}

__global__ void preproc_v_kernel(const ce_preproc_desc* __restrict__ descs, int n_px, int kmax,
                                 const int* __restrict__ taps, const int* __restrict__ bounds,
                                 const unsigned char* __restrict__ tmp, float* __restrict__ out, float m0, float m1,
                                 float m2, float s0, float s1, float s2) {
    const int o = blockIdx.y;
    const ce_preproc_desc d = descs[o];
    const long total = (long)n_px * n_px;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % n_px), y = (int)(idx / n_px);
        const long t = ((long)o * 2 + 1) * n_px + y;
        const int y0 = bounds[t * 2], cnt = bounds[t * 2 + 1];
        const int* k = taps + t * kmax;
        const unsigned char* src = tmp + d.tmp_off + ((long)(y0 - d.row0) * n_px + x) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int j = 0; j < cnt; ++j) {
            const unsigned char* p = src + (long)j * n_px * 3;
            a0 += (int)p[0] * k[j];
            a1 += (int)p[1] * k[j];
            a2 += (int)p[2] * k[j];
        }
        // ToTensor (/255) and Normalize ((x - mean) / std), both correctly rounded fp32 like torch's div / sub / div
        float* dst = out + (long)o * 3 * total + idx;
        dst[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a0), 255.0f), m0), s0);
        dst[total] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a1), 255.0f), m1), s1);
        dst[2 * total] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a2), 255.0f), m2), s2);
    }
}

}  // namespace

extern "C" size_t ce_preprocess_table_bytes(int n_out, int n_px, int kmax) {
    if (n_out <= 0 || n_px <= 0 || kmax <= 0) return 0;
    return ((size_t)n_out * 2 * n_px * kmax + (size_t)n_out * 2 * n_px * 2) * sizeof(int);
}

extern "C" int ce_preprocess(const ce_preproc_desc* descs_device, int n_out, int n_px, int kmax, int max_rows,
                             void* table, void* tmp, float* out, const float* mean, const float* std, void* stream) {
    CE_CHECK_ARG(descs_device && table && tmp && out && mean && std, "ce_preprocess: null buffer");
    CE_CHECK_ARG(n_out > 0 && n_px > 0 && kmax > 0 && kmax <= 4096 && max_rows > 0, "ce_preprocess: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    int* taps = reinterpret_cast<int*>(table);
    int* bounds = taps + (size_t)n_out * 2 * n_px * kmax;
    const long nc = (long)n_out * 2 * n_px;
    hipLaunchKernelGGL(preproc_coeff_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, descs_device, n_out, n_px,
                       kmax, taps, bounds);
    long bx = ((long)max_rows * n_px * 3 + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(preproc_h_kernel, dim3((unsigned)bx, n_out), dim3(256), 0, s, descs_device, n_px, kmax, taps, bounds,
                       reinterpret_cast<unsigned char*>(tmp));
    long bv = ((long)n_px * n_px + 255) / 256;
    hipLaunchKernelGGL(preproc_v_kernel, dim3((unsigned)bv, n_out), dim3(256), 0, s, descs_device, n_px, kmax, taps, bounds,
                       reinterpret_cast<const unsigned char*>(tmp), out, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    CE_LAUNCH_CHECK();
    return 0;
}
