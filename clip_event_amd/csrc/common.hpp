// Shared device/host helpers for the clip-event gfx950 kernels.
// gfx950 (CDNA4, wave64) only: no CUDA paths, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CE_WAVE 64

typedef uint16_t bf16_t;  // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---- error reporting (C ABI returns 0 / negative code, message via ce_last_error) ----
void ce_set_error(const char* fmt, ...);
#define CE_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ce_set_error(__VA_ARGS__);          \
            return -22; /* -EINVAL */           \
        }                                       \
    } while (0)
#define CE_LAUNCH_CHECK()                                                        \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            ce_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                         hipGetErrorString(e__));                                \
            return -5; /* -EIO */                                                \
        }                                                                        \
    } while (0)

// ---- bf16 <-> f32 ----
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// round-to-nearest-even via the hardware convert (keeps NaN a NaN; v_cvt_pk_bf16_f32 at -O3)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t*>(&b);
}

// ---- wave64 reductions (all 64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- buffer resources (bounds-checked: out-of-range loads return 0, stores are dropped) ----
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// d/dx [x * sigmoid(1.702 x)] = s + 1.702 x s (1 - s)
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// ---- optional event profiler (runtime.cpp); a no-op unless ce_profile_enable(1) ----
int ce_prof_begin(int cls, double flops, double bytes, hipStream_t s);
void ce_prof_end(int idx, hipStream_t s);
struct CeProfScope {
    int idx;
    hipStream_t s;
    CeProfScope(int cls, double flops, double bytes, hipStream_t st) : idx(ce_prof_begin(cls, flops, bytes, st)), s(st) {}
    ~CeProfScope() { ce_prof_end(idx, s); }
};

static inline int ce_div_up(long a, long b) { return (int)((a + b - 1) / b); }
