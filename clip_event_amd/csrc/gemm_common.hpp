// Pieces shared by the NT GEMM kernels (bf16: gemm.hip, fp8: gemm_fp8.hip): argument block, XCD-aware tile order and
// the fused epilogues.  Included inside each file's anonymous namespace.
#pragma once
#include "common.hpp"
#include "../../include/clip_event_hip.h"

struct NTArgs {
    const bf16_t* A; long lda;
    const bf16_t* B; long ldb;
    int M, N, K;
    const float* bias;
    const float* resid; long ldr;
    void* out; long ldo;
    bf16_t* out2; long ldo2;
    const bf16_t* aux; long ldaux;
    int tiles_m, tiles_n;
    int tile_strip = 0;          // persistent kernel: tiles are walked in column strips of this many tiles (0: row-major)
    int tile_chunk = 0;          // persistent kernel: > 0 = XCD-owned walk in chunks of this many row panels (persist_walk)
    unsigned int* tile_queue = nullptr;   // persistent kernel: != null = DYNAMIC tile list -- a workgroup's first tile is its static one,
                                          // every further tile is grid + atomicAdd(*tile_queue, 1) (zeroed by the launcher)
    int tall_panels = 0;         // persistent kernel, two tile heights (template TS > 0): row panels 0 .. tall_panels-1 are 32 TM rows
                                 // tall, the rest 32 TS rows; tiles_m counts both kinds
    const float* sa = nullptr;   // fp8 path: per-row dequantisation scales of A (M) ...
    const float* sb = nullptr;   // ... and of B (N); the epilogue multiplies the accumulator by sa[m] * sb[n]
    const uint8_t* sa8 = nullptr;   // fp8 path, MX form: E8M0 block scales of A [M, K/32] ...
    const uint8_t* sb8 = nullptr;   // ... and of B [N, K/32] (one per 32 contraction values), applied by the MFMA itself
};

constexpr bool epi_has_bias(int epi) {
    return epi == CE_EPI_BIAS_BF16 || epi == CE_EPI_BIAS_RESID_F32 || epi == CE_EPI_BIAS_GELU || epi == CE_EPI_BIAS_F32 ||
           epi == CE_EPI_BIAS_RESID_F16;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous
    // range of tiles so neighbouring tiles (same A row panel) hit the same L2.  Bijective
    // for any nwg.  Speed only, never correctness.
    int xcd = bid & 7, local = bid >> 3;
    int q = nwg >> 3, r = nwg & 7;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
}

// tile index -> (tm, tn) in column strips of p.tile_strip tile columns, row-major inside a strip: the 32 consecutive
// tiles an XCD holds in one round of the persistent kernel are then 8 row panels x 4 column panels (12 A/B panels through
// its L2) instead of 2.7 x 12 (15 panels, the 12 of B re-read every round).  Measured: no gain in the step (CE_NT_STRIP, off).
__device__ __forceinline__ void strip_tile_coords(const NTArgs& p, int tile, int& tm, int& tn) {
    const int gw = p.tile_strip;
    if (gw <= 0 || gw >= p.tiles_n) {
        tm = tile / p.tiles_n;
        tn = tile - tm * p.tiles_n;
        return;
    }
    const int per = gw * p.tiles_m;
    const int strip = tile / per;
    const int within = tile - strip * per;
    const int w = min(gw, p.tiles_n - strip * gw);      // the last strip may be narrower
    tm = within / w;
    tn = strip * gw + within - tm * w;
}

// Tile list of one persistent workgroup: tiles first + t * step, t < count, in the order `persist_coords` decodes.
// Default walk (tile_chunk == 0): the launch-wide order xb, xb + G, ... -- in every round an XCD's 32 workgroups hold 32
// CONSECUTIVE tiles (2.7 row panels x all column panels at N = 4d), a different set of row panels each round, so every XCD
// streams the whole weight matrix through its 4 MiB L2 once per round (r02 PMC: 6.8x the algorithmic reads at
// 12800 x 3072 x 768).  XCD-owned walk (tile_chunk = R > 0): the tile list is ordered in chunks of R row panels, column
// panel outer / row panel inner inside a chunk, and XCD x (blockIdx & 7: speed only, never correctness) owns the x-th
// eighth of that list for the whole launch.  Its R row panels of A (R = tiles_m / 8: 2.5 MB at 12800 x 768) stay in its
// L2 while the weight panels stream past once: fetch = A + 8 W instead of A + 8 W x rounds.
struct PersistWalk { int first, step, count; };
__device__ __forceinline__ PersistWalk persist_walk(const NTArgs& p, int total) {
    const int G = gridDim.x;
    if (p.tile_chunk > 0 && (G & 7) == 0) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = G >> 3;
        const int q = total >> 3, r = total & 7;                      // G <= total: every workgroup gets >= 1 tile
        const int start = xcd * q + min(xcd, r), cnt = q + (xcd < r ? 1 : 0);
        return {start + local, per, (cnt - local + per - 1) / per};
    }
    const int xb = xcd_remap(blockIdx.x, G);
    return {xb, G, (total - xb + G - 1) / G};
}
__device__ __forceinline__ void persist_coords(const NTArgs& p, int tile, int& tm, int& tn) {
    const int R = p.tile_chunk;
    if (R <= 0) { strip_tile_coords(p, tile, tm, tn); return; }
    const int chunk_tiles = R * p.tiles_n;
    const int c = tile / chunk_tiles;
    const int within = tile - c * chunk_tiles;
    const int rc = min(R, p.tiles_m - c * R);                         // the last chunk may hold fewer row panels
    tn = within / rc;
    tm = c * R + within - tn * rc;
}

// fused epilogue for one lane's 4 consecutive output columns n..n+3 of row m
template <int EPI>
__device__ __forceinline__ void nt_epilogue(const NTArgs& p, int m, int n, f32x4 v) {
    if constexpr (epi_has_bias(EPI)) {
        v += *reinterpret_cast<const f32x4*>(p.bias + n);
    }
    if constexpr (EPI == CE_EPI_BF16 || EPI == CE_EPI_BIAS_BF16) {
        u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.ldo + n) = o;
    } else if constexpr (EPI == CE_EPI_F32 || EPI == CE_EPI_BIAS_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (long)m * p.ldo + n) = v;
    } else if constexpr (EPI == CE_EPI_BIAS_RESID_F32) {
        v += *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (long)m * p.ldo + n) = v;
    } else if constexpr (EPI == CE_EPI_BIAS_RESID_F16) {      // fp16 residual stream in and out (saturating)
        v += f16x4_to_f32(*reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(p.resid) + (long)m * p.ldr + n));
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (long)m * p.ldo + n) = f32_to_f16x4_sat(v);
    } else if constexpr (EPI == CE_EPI_BIAS_GELU) {
        // out = QuickGELU'(a) (bf16, kept for the backward), out2 = QuickGELU(a) (bf16), a = acc + bias
        float gv[4], dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) quick_gelu_both(v[e], gv[e], dv[e]);
        u32x2 o = {pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.ldo + n) = o;
        u32x2 g = {pack_bf2(gv[0], gv[1]), pack_bf2(gv[2], gv[3])};
        *reinterpret_cast<u32x2*>(p.out2 + (long)m * p.ldo2 + n) = g;
    } else if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        // out = acc * aux, aux = the bf16 QuickGELU'(a) the forward epilogue saved
        u32x2 a = *reinterpret_cast<const u32x2*>(p.aux + (long)m * p.ldaux + n);
        u32x2 o = {pack_bf2(v[0] * bf_lo(a[0]), v[1] * bf_hi(a[0])), pack_bf2(v[2] * bf_lo(a[1]), v[3] * bf_hi(a[1]))};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.ldo + n) = o;
    }
}

// Fused bias gradient of the GELUGRAD epilogue: every lane holds partial column sums of its 8 columns
// (n_local .. n_local+7 of the workgroup's 256-column tile); the 8 row-phases of a wave and the two waves stacked
// along M share columns.  Each lane parks its 8 partial sums in a [16][256] LDS strip (plain 16-byte stores, row =
// 8*wm + row-phase), then thread c adds column c's 16 entries and issues the workgroup's ONE global atomic for
// that column: 4 atomic wave-instructions per workgroup instead of 64, no cross-lane shuffles.
__device__ __forceinline__ void wg_colsum_flush(char* smem, float* __restrict__ colsum, int n0, int N, int n_local,
                                                int srow, const f32x4& cs0, const f32x4& cs1, int bn = 256) {
    constexpr int SROW = 264;                              // floats: 256 + 8 pad (rows shift by 8 banks)
    float* strip = reinterpret_cast<float*>(smem);
    __syncthreads();                                       // every wave is done with its epilogue slice
    *reinterpret_cast<f32x4*>(strip + srow * SROW + n_local) = cs0;
    *reinterpret_cast<f32x4*>(strip + srow * SROW + n_local + 4) = cs1;
    __syncthreads();
    for (int i = threadIdx.x; i < bn; i += blockDim.x) {      // bn = columns of the workgroup's tile
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += strip[r * SROW + i];
        if (n0 + i < N && t != 0.f) atomicAdd(colsum + n0 + i, t);
    }
}

// 16-byte epilogue store.  CE_EPI_ST_AUX (compile time) is the cache policy of the OUTPUT stores: 2 = nt (default), 0 =
// plain, 16 = sc1 (written through and dropped from the XCD's L2).  The outputs of a launch are 1.5-6x its operand bytes
// and are never re-read by it; plain stores park them in the 4 MiB L2, where they evict the operand panels the other
// tiles of the XCD are about to re-read.  Step A/B (gpurun_out/stpol, both walks): BIAS_RESID_F32 1.85 -> 1.65 ms,
// BIAS_GELU 1.37 -> 1.29, plain bf16 2.21 -> 2.12, qkv 0.92 -> 0.89; step 13.85 -> 13.64 ms.  sc1 is as good for the bf16
// outputs but makes the fp32 residual stream, which the next kernel re-reads, 30 % slower (2.42 ms).
#ifndef CE_EPI_ST_AUX
#define CE_EPI_ST_AUX 2
#endif
__device__ __forceinline__ void epi_store16(void* base, long byte_off, u32x4 v) {
#if CE_EPI_ST_AUX == 0
    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + byte_off) = v;
#elif CE_EPI_ST_AUX == 2
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + byte_off));
#else      // diagnostic builds only: 32-bit buffer offsets (outputs < 4 GiB)
    __builtin_amdgcn_raw_buffer_store_b128(v, make_rsrc(base, 0xffffffffu), (uint32_t)byte_off, 0, CE_EPI_ST_AUX);
#endif
}
__device__ __forceinline__ void epi_store16(void* base, long byte_off, f32x4 v) {
    epi_store16(base, byte_off, __builtin_bit_cast(u32x4, v));
}

// ---- epilogue I/O of the loader-wave kernels: BRANCH-FREE, through bounds-checked buffer descriptors -------------------
// Every global operand of a tile's epilogue (outputs, the fp16 / fp32 residual tile, the bf16 derivative tile) is reached through
// a descriptor whose base is the tile's element (m0, n0) and whose range ends with the last valid row, so rows past M need no
// branch (their loads return 0, their stores are dropped) and an access is ONE v_add (lane offset + the slot's uniform row
// offset) instead of a 64-bit multiply-add chain.  Why it matters beyond the instruction count: a load inside a divergent
// branch (`if (m < M) x = load`) leaves its result "pending" on the path around the branch, so the compiler re-waits
// `s_waitcnt vmcnt(0)` in front of EVERY later use -- and stores count in vmcnt too, so each 8-row slot of the epilogue then
// waited for the previous slot's store to be acknowledged (round 3: the bias vector's two loads cost the persistent kernel
// 14 us of 74 at 12800 x 3072 x 768).  Offsets are 32-bit and 0x80000000 marks a column past N: the host takes these
// kernels only when every operand spans less than 2 GiB.
struct EpiBuf {
    __amdgpu_buffer_rsrc_t r;
    uint32_t v;                  // this lane's byte offset inside the tile (row = its row in the first 8-row slot)
    uint32_t slot;               // bytes per 8 rows
};
__device__ __forceinline__ EpiBuf epi_buf(const void* base, long ld, int esz, int M, int N, int m0, int n0, int row, int col, bool col_ok) {
    EpiBuf b;
    b.r = make_rsrc(reinterpret_cast<const char*>(base) + ((long)m0 * ld + n0) * esz, (uint32_t)((((long)(M - m0) - 1) * ld + (N - n0)) * esz));
    b.v = col_ok ? (uint32_t)(((long)row * ld + col) * esz) : 0x80000000u;
    b.slot = (uint32_t)(8 * ld * esz);
    return b;
}
__device__ __forceinline__ u32x4 epi_bload16(const EpiBuf& b, int slot, int byte = 0) {
    return __builtin_amdgcn_raw_buffer_load_b128(b.r, b.v + (uint32_t)slot * b.slot + (uint32_t)byte, 0, 0);
}
__device__ __forceinline__ void epi_bstore16(const EpiBuf& b, int slot, u32x4 v, int byte = 0) {
    uint32_t off = b.v + (uint32_t)slot * b.slot + (uint32_t)byte;
#ifdef CE_DIAG_EPI_ALIAS      // timing ablation (tools/diag/epi_ablate.sh): stores land in one 1 MiB window per tile base
    off = (off & 0xfffffu) | (off & 0x80000000u);
#endif
    __builtin_amdgcn_raw_buffer_store_b128(v, b.r, off, 0, CE_EPI_ST_AUX);
}
constexpr int epi_out_bytes(int epi) {
    return (epi == CE_EPI_F32 || epi == CE_EPI_BIAS_F32 || epi == CE_EPI_BIAS_RESID_F32) ? 4 : 2;
}
// 8 consecutive output columns of one row (bias already added); eo / eo2 = the outputs, er = the fp32 residual (BIAS_RESID_F32)
template <int EPI>
__device__ __forceinline__ void nt_epilogue8b(const EpiBuf& eo, const EpiBuf& eo2, const EpiBuf& er, int slot, f32x4 v0, f32x4 v1) {
    if constexpr (EPI == CE_EPI_BF16 || EPI == CE_EPI_BIAS_BF16) {
        u32x4 o = {pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]), pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3])};
        epi_bstore16(eo, slot, o);
    } else if constexpr (EPI == CE_EPI_F32 || EPI == CE_EPI_BIAS_F32) {
        epi_bstore16(eo, slot, __builtin_bit_cast(u32x4, v0));
        epi_bstore16(eo, slot, __builtin_bit_cast(u32x4, v1), 16);
    } else if constexpr (EPI == CE_EPI_BIAS_RESID_F32) {
        v0 += __builtin_bit_cast(f32x4, epi_bload16(er, slot));
        v1 += __builtin_bit_cast(f32x4, epi_bload16(er, slot, 16));
        epi_bstore16(eo, slot, __builtin_bit_cast(u32x4, v0));
        epi_bstore16(eo, slot, __builtin_bit_cast(u32x4, v1), 16);
    } else if constexpr (EPI == CE_EPI_BIAS_GELU) {
        float gv[8], dv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            quick_gelu_both(v0[e], gv[e], dv[e]);
            quick_gelu_both(v1[e], gv[4 + e], dv[4 + e]);
        }
        u32x4 o = {pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3]), pack_bf2(dv[4], dv[5]), pack_bf2(dv[6], dv[7])};
        u32x4 g = {pack_bf2(gv[0], gv[1]), pack_bf2(gv[2], gv[3]), pack_bf2(gv[4], gv[5]), pack_bf2(gv[6], gv[7])};
        epi_bstore16(eo, slot, o);
        epi_bstore16(eo2, slot, g);
    }
}

// fused epilogue for 8 consecutive output columns n..n+7 of row m (bias already added)
template <int EPI>
__device__ __forceinline__ void nt_epilogue8(const NTArgs& p, int m, int n, f32x4 v0, f32x4 v1, f32x4& cs0, f32x4& cs1) {
    if constexpr (EPI == CE_EPI_BF16 || EPI == CE_EPI_BIAS_BF16) {
        u32x4 o = {pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]), pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3])};
        epi_store16(p.out, ((long)m * p.ldo + n) * 2, o);
    } else if constexpr (EPI == CE_EPI_F32 || EPI == CE_EPI_BIAS_F32) {
        const long ob = ((long)m * p.ldo + n) * 4;
        epi_store16(p.out, ob, v0);
        epi_store16(p.out, ob + 16, v1);
    } else if constexpr (EPI == CE_EPI_BIAS_RESID_F32) {
        const float* r = p.resid + (long)m * p.ldr + n;
        v0 += *reinterpret_cast<const f32x4*>(r);
        v1 += *reinterpret_cast<const f32x4*>(r + 4);
        const long ob = ((long)m * p.ldo + n) * 4;
        epi_store16(p.out, ob, v0);
        epi_store16(p.out, ob + 16, v1);
    } else if constexpr (EPI == CE_EPI_BIAS_RESID_F16) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.resid) + (long)m * p.ldr + n);
        v0 += f16x4_to_f32(u32x2{r[0], r[1]});
        v1 += f16x4_to_f32(u32x2{r[2], r[3]});
        const u32x2 h0 = f32_to_f16x4_sat(v0), h1 = f32_to_f16x4_sat(v1);
        epi_store16(p.out, ((long)m * p.ldo + n) * 2, u32x4{h0[0], h0[1], h1[0], h1[1]});
    } else if constexpr (EPI == CE_EPI_BIAS_GELU) {
        float gv[8], dv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            quick_gelu_both(v0[e], gv[e], dv[e]);
            quick_gelu_both(v1[e], gv[4 + e], dv[4 + e]);
        }
        u32x4 o = {pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3]), pack_bf2(dv[4], dv[5]), pack_bf2(dv[6], dv[7])};
        epi_store16(p.out, ((long)m * p.ldo + n) * 2, o);
        u32x4 g = {pack_bf2(gv[0], gv[1]), pack_bf2(gv[2], gv[3]), pack_bf2(gv[4], gv[5]), pack_bf2(gv[6], gv[7])};
        epi_store16(p.out2, ((long)m * p.ldo2 + n) * 2, g);
    } else if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        u32x4 a = *reinterpret_cast<const u32x4*>(p.aux + (long)m * p.ldaux + n);
        f32x4 r0 = {v0[0] * bf_lo(a[0]), v0[1] * bf_hi(a[0]), v0[2] * bf_lo(a[1]), v0[3] * bf_hi(a[1])};
        f32x4 r1 = {v1[0] * bf_lo(a[2]), v1[1] * bf_hi(a[2]), v1[2] * bf_lo(a[3]), v1[3] * bf_hi(a[3])};
        cs0 += r0;      // column sums of the result = bias gradient of the Linear whose activation derivative aux is
        cs1 += r1;
        u32x4 o = {pack_bf2(r0[0], r0[1]), pack_bf2(r0[2], r0[3]), pack_bf2(r1[0], r1[1]), pack_bf2(r1[2], r1[3])};
        epi_store16(p.out, ((long)m * p.ldo + n) * 2, o);
    }
}

