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

// ---- residual-stream element types (ce_tower_desc.stream16, ce_layernorm_*_t): fp32, or IEEE fp16 ----
// The residual stream and its gradient are read and written four times per block and direction and never multiplied by a
// matrix unit, so their format is a storage choice.  fp16 keeps 11 significand bits (bf16: 8): on BASELINE config 1 the
// per-parameter gradient error against the fp32 oracle grows by a median 4 % over the fp32-stream build (tests/
// stream16_emulation.py; a bf16 stream doubles it).  Range: stores saturate at +-65504; the GRADIENT stream is stored
// multiplied by a power of two (ce_tower_desc.grad_scale, 2^16) because gradients of a mean loss sit far below fp16's
// normal range.
#ifndef CE_T_F32         // same values in include/clip_event_hip.h
#define CE_T_F32 0
#define CE_T_BF16 1
#define CE_T_F16 2
#endif
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ f32x4 f16x4_to_f32(u32x2 raw) {
    return __builtin_convertvector(__builtin_bit_cast(f16x4, raw), f32x4);
}
__device__ __forceinline__ u32x2 f32_to_f16x4_sat(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
    return __builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4));
}
// 4 consecutive elements at element index i (a multiple of 4) of a buffer of element type `type` (wave-uniform)
__device__ __forceinline__ f32x4 load4_t(const void* p, long i, int type) {
    if (type == CE_T_F32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
    const u32x2 raw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(p) + i);
    if (type == CE_T_F16) return f16x4_to_f32(raw);
    return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u), __uint_as_float(raw[1] << 16),
                 __uint_as_float(raw[1] & 0xffff0000u)};
}
__device__ __forceinline__ void store4_t(void* p, long i, int type, f32x4 v) {
    if (type == CE_T_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + i) = v;
    } else if (type == CE_T_F16) {
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p) + i) = f32_to_f16x4_sat(v);
    } else {
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p) + i) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    }
}
__host__ __device__ static inline int ce_type_bytes(int type) { return type == CE_T_F32 ? 4 : 2; }

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

// ---- DPP row (16-lane) reductions: VALU only, no LDS crossbar (a __shfl_xor is a ds_bpermute_b32 + wait) ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {   // every lane active (EXEC all ones) at the call sites
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// Whole-wave sum / max on DPP alone, result broadcast through an SGPR: four butterfly steps inside each 16-lane row (xor 1, xor 2,
// half mirror, mirror: every lane then holds its row's value), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3,
// lane 63 read back.  Six dependent VALU instructions instead of six ds_bpermute_b32 round trips through the LDS crossbar
// (~100 cycles each): the two reductions of a LayerNorm row were ~1.4 k cycles of pure latency.  EXEC must be all ones.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_rows(float x, float identity) {      // rows outside ROW_MASK receive `identity`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0x0B1>(v);                 // quad_perm [1,0,3,2]
    v += dpp_mov<0x04E>(v);                 // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);                 // row_half_mirror
    v += dpp_mov<0x140>(v);                 // row_mirror
    v += dpp_rows<0x142, 0xa>(v, 0.f);      // row_bcast:15 -> rows 1, 3
    v += dpp_rows<0x143, 0xc>(v, 0.f);      // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {      // for values >= 0 (|x| maxima): 0 is the identity
    v = fmaxf(v, dpp_mov<0x0B1>(v));
    v = fmaxf(v, dpp_mov<0x04E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    v = fmaxf(v, dpp_rows<0x142, 0xa>(v, 0.f));
    v = fmaxf(v, dpp_rows<0x143, 0xc>(v, 0.f));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Column sums of a 16 x 16 register block held one row per lane of a 16-lane DPP row (v[k] = element k of lane
// li's row): transpose-reduce butterfly over the pairings row_mirror, row_half_mirror, quad xor 1, quad xor 2 --
// each stage halves the values a lane keeps and adds the partner's copy (15 DPP adds instead of 64).  Lane li
// returns sum_lanes v[k] for k = row16_colsum_index(li).
__device__ __forceinline__ int row16_colsum_index(int li) {
    return (li & 8) | (li & 4) | ((li & 1) << 1) | ((li & 2) >> 1);
}
__device__ __forceinline__ float row16_colsum(const float (&v)[16], int li) {
    const bool s1 = li & 8, s2 = li & 4, s3 = li & 1, s4 = li & 2;
    float a[8], b[4], c[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (s1 ? v[k + 8] : v[k]) + dpp_mov<0x140>(s1 ? v[k] : v[k + 8]);   // row_mirror
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = (s2 ? a[k + 4] : a[k]) + dpp_mov<0x141>(s2 ? a[k] : a[k + 4]);   // row_half_mirror
#pragma unroll
    for (int k = 0; k < 2; ++k) c[k] = (s3 ? b[k + 2] : b[k]) + dpp_mov<0x0B1>(s3 ? b[k] : b[k + 2]);   // quad_perm [1,0,3,2]
    return (s4 ? c[1] : c[0]) + dpp_mov<0x04E>(s4 ? c[0] : c[1]);                                       // quad_perm [2,3,0,1]
}

// ---- buffer resources (bounds-checked: out-of-range loads return 0, stores are dropped) ----
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// v_rcp_f32 (1 ulp) instead of an IEEE division: `/` without fast-math is a ten-instruction sequence, and in the GEMM
// epilogues that apply these to every output element it was most of the epilogue (12.5k of a K = 768 tile's 35k cycles,
// tools/diag/make_nt160p_stamps.py); the results are rounded to bf16 afterwards.
__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
// d/dx [x * sigmoid(1.702 x)] = s + 1.702 x s (1 - s)
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}
// both at once (one sigmoid): g = QuickGELU(x), d = QuickGELU'(x)
// (8 VALU instructions per element: exp(-1.702 x) as ONE multiply + v_exp_f32 -- the constant is -1.702 log2(e) -- and
//  d = s + 1.702 g (1 - s) re-using g; the straightforward form was 10, in epilogues whose cost is their instruction count)
__device__ __forceinline__ void quick_gelu_both(float x, float& g, float& d) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
    g = x * s;
    d = __builtin_fmaf(1.702f * g, 1.0f - s, s);
}

// ---- fp16-stream saturation counters (runtime.cpp: the buffer registered with ce_stream16_set_counters, or null) ----
// [0] forward residual stream: (wave, lane) slots of a LayerNorm forward that READ an element at the fp16 limit (every fp16
//     store of the stream clamps to +-65504 and every stream row is read by a LayerNorm forward before anything else uses it);
// [1] gradient stream: slots of a LayerNorm backward whose output, before the clamp, was at or beyond the limit (or not finite).
unsigned int* ce_sat_counters();
#define CE_F16_LIMIT 65504.0f
__device__ __forceinline__ float absmax4(float m, f32x4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// ---- optional event profiler (runtime.cpp); a no-op unless ce_profile_enable(1) ----
int ce_prof_begin(int cls, double flops, double bytes, hipStream_t s);
void ce_prof_end(int idx, hipStream_t s);
void ce_prof_retag(int idx, int cls);
struct CeProfScope {
    int idx;
    hipStream_t s;
    CeProfScope(int cls, double flops, double bytes, hipStream_t st) : idx(ce_prof_begin(cls, flops, bytes, st)), s(st) {}
    void retag(int cls) { ce_prof_retag(idx, cls); }      // the kernel family is known only after the tile choice
    ~CeProfScope() { ce_prof_end(idx, s); }
};

static inline int ce_div_up(long a, long b) { return (int)((a + b - 1) / b); }
