// bf16 MFMA GEMMs for the CLIP-Event towers (gfx950 / CDNA4, wave64).
//
//   ce_gemm_nt : C[M,N]   = A[M,K] . B[N,K]^T  (+ fused epilogues)   forward + dgrad
//   ce_gemm_tn : O[Nn,Kk] += P[M,Nn]^T . Q[M,Kk]   (fp32 atomics)     wgrad
//
// Replaces the nn.Linear / nn.MultiheadAttention in/out-projection / Conv2d(stride=kernel)
// GEMMs of the reference (model_clip.py:175-180, :219, :230, :329) and their autograd.
//
// NT tile: 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16
// tiles.  The weight rows feed the MFMA A operand and the activation rows the B operand,
// so a lane's 4 accumulator registers are 4 consecutive output columns n of one output
// row m (8-byte bf16 / 16-byte fp32 epilogue accesses).  Operands are staged global ->
// VGPR -> LDS one K-tile ahead (double-buffered LDS, one barrier per K-tile) with the
// 16-byte-chunk XOR swizzle chunk ^= row&7 that makes the ds_read_b128 fragment reads
// bank-conflict free on 128-byte rows.
//
// TN tile: 128(n) x 128(k) outputs, contraction over 64-row m tiles, 4 waves (2x2), each
// 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles.  Both operands are contraction-major in HBM,
// so fragments come from ds_read_b64_tr_b16 (hardware transposed LDS read) on row-major
// [m][128] LDS tiles padded to a 320-byte row stride (conflict-free for the 4-row x 64-byte
// footprint of a half-wave).  Output columns sit on the lane (col = lane&31), so every
// accumulator register is two 128-byte row segments: the full-rate shape for
// global_atomic_add_f32.  The M range is split across workgroups to fill the chip.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <type_traits>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

#ifndef CE_DIAG_TN3
#define CE_DIAG_TN3 0
#endif

namespace {

// ------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------
constexpr int NT_BM = 128, NT_BN = 128, NT_BK = 64;
constexpr int NT_STAGE_BYTES = (NT_BM + NT_BN) * NT_BK * 2;  // 32 KiB
constexpr int NT_LDS_BYTES = 2 * NT_STAGE_BYTES;             // 64 KiB

#include "gemm_common.hpp"

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * NT_BM, n0 = tn * NT_BN;

    const int rowsA = min(p.M - m0, NT_BM), rowsB = min(p.N - n0, NT_BN);
    __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + (long)m0 * p.lda, (uint32_t)((long)rowsA * p.lda * 2));
    __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B + (long)n0 * p.ldb, (uint32_t)((long)rowsB * p.ldb * 2));

    // staging map: 16-byte chunk q = tid + 256*i -> row = q>>3, chunk c = q&7
    const int s_c = tid & 7;
    const int s_row = tid >> 3;  // + 32*i
    const uint32_t gA0 = (uint32_t)(s_row * p.lda * 2 + s_c * 16);
    const uint32_t gB0 = (uint32_t)(s_row * p.ldb * 2 + s_c * 16);
    const uint32_t gAstep = (uint32_t)(32 * p.lda * 2), gBstep = (uint32_t)(32 * p.ldb * 2);
    const int lds_w = s_row * 128 + ((s_c ^ (s_row & 7)) << 4);  // + 4096*i (32 rows * 128 B), swizzle unchanged

    u32x4 ra[4], rb[4];
    auto issue_loads = [&](int kt) {
        const uint32_t kb = (uint32_t)(kt * NT_BK * 2);
        const bool kvalid = (kt * NT_BK + s_c * 8) < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            ra[i] = kvalid ? __builtin_amdgcn_raw_buffer_load_b128(rA, gA0 + i * gAstep + kb, 0, 0) : z;
            rb[i] = kvalid ? __builtin_amdgcn_raw_buffer_load_b128(rB, gB0 + i * gBstep + kb, 0, 0) : z;
        }
    };
    auto write_stage = [&](int stage) {
        char* sa = smem + stage * NT_STAGE_BYTES;
        char* sb = sa + NT_BM * NT_BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(sa + lds_w + i * 4096) = ra[i];
            *reinterpret_cast<u32x4*>(sb + lds_w + i * 4096) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + NT_BK - 1) / NT_BK;
    issue_loads(0);
    write_stage(0);
    __syncthreads();

    // fragment read map: row = base + (lane&15), k-chunk = ks*4 + (lane>>4); swizzle = row&7 = lane&7
    const int f_row = lane & 15, f_kc = lane >> 4, f_sw = lane & 7;
    const int fa_base = (wm * 64 + f_row) * 128;
    const int fb_base = NT_BM * NT_BK * 2 + (wn * 64 + f_row) * 128;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue_loads(kt + 1);
        const char* st = smem + cur * NT_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + f_kc) ^ f_sw) << 4;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *reinterpret_cast<const bf16x8*>(st + fa_base + t * 2048 + coff);
                wf[t] = *reinterpret_cast<const bf16x8*>(st + fb_base + t * 2048 + coff);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
        if (kt + 1 < nk) write_stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m][n..n+3], m = lane&15 within the 16x16 tile ----
    const int em = m0 + wm * 64 + (lane & 15);
    const int en = n0 + wn * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = em + mt * 16;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = en + nt * 16;
            if (n >= p.N) continue;
            nt_epilogue<EPI>(p, m, n, acc[mt][nt]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// SKINNY NT kernel for the few-row GEMMs (M <= 512: the pruned last block's 256 rows, the feature projections): the
// tile grid of such a problem is 12-48 tiles for 256 CUs and each tile's K loop is a chain of exposed latencies, so the
// parallelism has to come from K.  A workgroup owns a 64 x 64 output tile; its 8 waves each take an eighth of K, with
// both operands read from global memory straight into MFMA fragments (K-contiguous rows: a lane's 8 contraction values
// are one 16-byte load; no LDS staging, next k-step's fragments in flight during the MFMAs), and the eight partial tiles
// are summed through LDS before the fused epilogue.  No atomics, no scratch, one launch.
// ------------------------------------------------------------------------------------------
constexpr int SK_LDS_BYTES = 8 * 64 * 272;
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_skinny_kernel(NTArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const int kw = p.K >> 3;                              // this wave's K slice (a multiple of 32)
    const int nks = kw >> 5;
    const int f_row = lane & 15, f_k = (lane >> 4) * 8;
    const bf16_t* pa[4];
    const bf16_t* pb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {                        // rows past the edge are clamped: their products are never stored
        pa[t] = p.A + (long)min(m0 + t * 16 + f_row, p.M - 1) * p.lda + wave * kw + f_k;
        pb[t] = p.B + (long)min(n0 + t * 16 + f_row, p.N - 1) * p.ldb + wave * kw + f_k;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4], wf[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        af[0][t] = *reinterpret_cast<const bf16x8*>(pa[t]);
        wf[0][t] = *reinterpret_cast<const bf16x8*>(pb[t]);
    }
    for (int ks = 0; ks < nks; ks += 2) {
        if (ks + 1 < nks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[1][t] = *reinterpret_cast<const bf16x8*>(pa[t] + (ks + 1) * 32);
                wf[1][t] = *reinterpret_cast<const bf16x8*>(pb[t] + (ks + 1) * 32);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][nt], af[0][t], acc[t][nt], 0, 0, 0);
        if (ks + 1 < nks) {
            if (ks + 2 < nks) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    af[0][t] = *reinterpret_cast<const bf16x8*>(pa[t] + (ks + 2) * 32);
                    wf[0][t] = *reinterpret_cast<const bf16x8*>(pb[t] + (ks + 2) * 32);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][nt], af[1][t], acc[t][nt], 0, 0, 0);
        }
    }
    // partial tiles -> LDS ([wave][64 rows][64 cols] fp32, 272-byte rows); lane holds row t*16 + (lane&15), cols nt*16 + 4*(lane>>4) ..
    constexpr int EROW = 272;
    char* mine = smem + wave * (64 * EROW);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<f32x4*>(mine + (t * 16 + (lane & 15)) * EROW + (nt * 16 + 4 * (lane >> 4)) * 4) = acc[t][nt];
    __syncthreads();
    const int e_r = tid >> 3, e_c = (tid & 7) * 8;        // 512 threads x 8 columns = the 64 x 64 tile
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        v0 += *reinterpret_cast<const f32x4*>(smem + w * (64 * EROW) + e_r * EROW + e_c * 4);
        v1 += *reinterpret_cast<const f32x4*>(smem + w * (64 * EROW) + e_r * EROW + e_c * 4 + 16);
    }
    const int m = m0 + e_r, gn = n0 + e_c;
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
    if (m < p.M && gn < p.N) {
        if constexpr (epi_has_bias(EPI)) {
            v0 += *reinterpret_cast<const f32x4*>(p.bias + gn);
            v1 += *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
        }
        nt_epilogue8<EPI>(p, m, gn, v0, v1, cs0, cs1);
    }
    if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        if (p.out2) {        // column sums over this wave's 8 rows (lanes with equal lane&7 share columns), then one atomic per lane
            float* colsum = reinterpret_cast<float*>(p.out2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    cs0[e] += __shfl_xor(cs0[e], o, 64);
                    cs1[e] += __shfl_xor(cs1[e], o, 64);
                }
            }
            const int e = lane >> 3;
            const float v = e == 0 ? cs0[0] : e == 1 ? cs0[1] : e == 2 ? cs0[2] : e == 3 ? cs0[3]
                          : e == 4 ? cs1[0] : e == 5 ? cs1[1] : e == 6 ? cs1[2] : cs1[3];
            if (gn + e < p.N) atomicAdd(colsum + gn + e, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// NT kernel, 256x256x64 tile, 8 waves (2 M x 4 N, 128x64 per wave), operands staged straight
// into LDS by global_load_lds (LDS-DMA, 16 B per lane): no VGPR round trip and no ds_write
// issue cost, which is what bounds the 128^2 register-staged kernel.  The DMA destination is
// lane-linear, so the XOR swizzle (chunk ^= row&7) is applied to the per-lane SOURCE address
// and again on the fragment reads.  One barrier per K-tile, next tile in flight during the MFMAs.
// ------------------------------------------------------------------------------------------
// BM = 32*TM rows (TM = m-tiles of 16 per wave): TM=8 -> 256x256, TM=5 -> 160x256.  The smaller tile
// exists for wave quantisation: N = width GEMMs have only 2-3 tile columns, and 160-row tiles give
// 240-248 workgroups for the 256 CUs where 256-row tiles give 150.
constexpr int N2_BN = 256, N2_BK = 64;
constexpr int N2_BTILE_BYTES = N2_BN * N2_BK * 2;           // 32 KiB
constexpr int N2H_LDS_BYTES = 2 * (160 + 128) * 64 * 2;     // 72 KiB: 160x128 tile, two stages (>= 4 epilogue slices)
constexpr int N2_LDS_BYTES = 8 * 64 * 272;                 // 136 KiB: 2 stages (<=128 KiB) or 8 epilogue slices of 17 KiB

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// LDS-DMA through inline asm: hipcc does not track asm memory operations, so it cannot serialise the DMA
// against the fragment reads of the OTHER stage (it did, with the builtin, once the kernel grew an outer
// loop: an s_waitcnt vmcnt(0) in front of every m-tile's first ds_read).  The kernel waits for the DMA
// itself (vmcnt(0) ahead of the barrier that publishes a stage).  M0 (the LDS destination) is saved and
// restored inside the statement; the leading s_nop covers SGPR-write -> VMEM-read wait states.
__device__ __forceinline__ u32x4 make_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    return r;
}
__device__ __forceinline__ void dma16_bounds(u32x4 rsrc, uint32_t lds_dst, uint32_t voff) {
    uint32_t keep;
    asm volatile(
        "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_dst), "s"(rsrc)
        : "memory");
}


// WN = waves along N (64 columns each): 4 -> 8 waves, 256-column tile, one workgroup per CU;
//                                      2 -> 4 waves, 128-column tile, 72 KiB LDS: TWO workgroups per CU, so one's
//                                      prologue / epilogue overlaps the other's K loop, and N = width GEMMs get
//                                      480-496 tiles for the 512 slots.
template <int EPI, int TM, int WN>
__global__ __launch_bounds__(128 * WN, 2) void gemm_nt256_kernel(NTArgs p) {
    constexpr int NW = 2 * WN;                                  // waves per workgroup
    constexpr int N2_BN = 64 * WN;
    constexpr int N2_BTILE_BYTES = N2_BN * N2_BK * 2;
    constexpr int N2_BM = 32 * TM;
    constexpr int N2_TILE_BYTES = N2_BM * N2_BK * 2;            // A tile
    constexpr int N2_STAGE_BYTES = N2_TILE_BYTES + N2_BTILE_BYTES;
    constexpr int A_INSTR = N2_BM / 8;                          // 1-KiB DMA instructions per A tile
    constexpr int A_PER_WAVE = (A_INSTR + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * N2_BM, n0 = tn * N2_BN;

    // staging: one wave-instruction = 8 rows x 128 B; wave w, instruction i covers rows (w*4+i)*8 .. +7.
    // lane l lands at LDS (row r = l>>3, position pos = l&7) and therefore fetches chunk pos ^ (r&7).
    const int s_r = lane >> 3, s_pos = lane & 7;
    const int s_chunk = s_pos ^ s_r;
    const bf16_t* gA[A_PER_WAVE];
    const bf16_t* gB[4];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {    // A instruction index = wave + NW*i (wave-uniform bound check below)
        const int row = (wave + NW * i) * 8 + s_r;
        const int ra = min(m0 + row, p.M - 1);                                  // clamp: rows past the edge are never stored
        gA[i] = p.A + (long)ra * p.lda + s_chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + s_r;
        const int rb = min(n0 + row, p.N - 1);
        gB[i] = p.B + (long)rb * p.ldb + s_chunk * 8;
    }
    auto stage = [&](int st, int kt) {
        char* sa = smem + st * N2_STAGE_BYTES;
        char* sb = sa + N2_TILE_BYTES + wave * 4096;
        const int koff = kt * N2_BK;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i)
            if (wave + NW * i < A_INSTR)
                __builtin_amdgcn_global_load_lds((gptr_t*)(gA[i] + koff), (lptr_t*)(sa + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(gB[i] + koff), (lptr_t*)(sb + i * 1024), 16, 0, 0);
    };

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue operand prefetch.  The fp32 residual tile (BIAS_RESID_F32: 32 B per output row-slot and lane) and the
    // bf16 pre-activation tile (GELUGRAD_BF16: 16 B) used to be read inside the epilogue, serial with the K loop; they
    // are now loaded into registers during the LAST NPF K iterations (a few loads per iteration, issued ahead of that
    // iteration's LDS-DMA so they are the older entries of the in-order vmcnt queue) and are resident when the
    // accumulators come out of the LDS transpose.  Row-slot (mh, it) = output row m0 + wm*16*TM + mh*64 + it*8 + e_r.
    constexpr bool PF_RESID = EPI == CE_EPI_BIAS_RESID_F32 && TM <= 5;
    constexpr bool PF_AUX = EPI == CE_EPI_GELUGRAD_BF16 && TM <= 6;
    constexpr bool PF = PF_RESID || PF_AUX;
    constexpr int SLOTS = ((TM + 3) / 4 - 1) * 8 + ((TM % 4 == 0) ? 8 : (TM % 4) * 2);
    constexpr int NPF = 4;
    constexpr int PER = (SLOTS + NPF - 1) / NPF;
    const int e_r = lane >> 3, e_c = (lane & 7) * 8;          // epilogue map: row e_r (+8 per slot), 8 columns
    const int gn = n0 + wn * 64 + e_c;
    f32x4 rp[PF_RESID ? SLOTS : 1][2];
    u32x4 ap[PF_AUX ? SLOTS : 1];
    auto prefetch_group = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int slot = g * PER + q;
            if (slot < SLOTS) {
                const int m = m0 + wm * (TM * 16) + (slot >> 3) * 64 + (slot & 7) * 8 + e_r;
                const bool ok = m < p.M && gn < p.N;
                if constexpr (PF_RESID) {
                    const float* r = p.resid + (long)m * p.ldr + gn;
                    rp[slot][0] = ok ? *reinterpret_cast<const f32x4*>(r) : f32x4{0.f, 0.f, 0.f, 0.f};
                    rp[slot][1] = ok ? *reinterpret_cast<const f32x4*>(r + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (PF_AUX) {
                    ap[slot] = ok ? *reinterpret_cast<const u32x4*>(p.aux + (long)m * p.ldaux + gn) : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
    };

    const int nk = p.K / N2_BK;
    stage(0, 0);
    __syncthreads();     // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier

    const int f_row = lane & 15, f_kc = lane >> 4, f_sw = lane & 7;
    const int fa_base = (wm * (TM * 16) + f_row) * 128;
    const int fb_base = N2_TILE_BYTES + (wn * 64 + f_row) * 128;

    auto k_iter = [&](int kt) __attribute__((always_inline)) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* st = smem + cur * N2_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + f_kc) ^ f_sw) << 4;
            bf16x8 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(st + fb_base + t * 2048 + coff);
#pragma unroll
            for (int mh = 0; mh * 4 < TM; ++mh) {
                bf16x8 af[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (mh * 4 + t < TM)
                        af[t] = *reinterpret_cast<const bf16x8*>(st + fa_base + (mh * 4 + t) * 2048 + coff);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (mh * 4 + t < TM) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mh * 4 + t][nt] =
                                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[t], acc[mh * 4 + t][nt], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    };
    if constexpr (!PF) {
        for (int kt = 0; kt < nk; ++kt) k_iter(kt);
    } else if (nk >= NPF) {
        for (int kt = 0; kt < nk - NPF; ++kt) k_iter(kt);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            prefetch_group(j);
            k_iter(nk - NPF + j);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NPF; ++j) prefetch_group(j);
        for (int kt = 0; kt < nk; ++kt) k_iter(kt);
    }

    // ---- epilogue through LDS: each wave transposes its 128x64 fp32 sub-tile in two 64x64 halves
    // inside its own 17 KiB slice (row stride 272 B), then every instruction reads/writes global
    // memory as 8 rows x 32 B..256 B contiguous: 16 B per lane, whole cache lines per row.
    // (the loop's last barrier guarantees every wave is done with the operand stages)
    constexpr int EROW = 272;
    char* ebuf = smem + wave * (64 * EROW);
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (epi_has_bias(EPI)) {
        if (gn < p.N) {
            bias0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
            bias1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
        }
    }
#pragma unroll
    for (int mh = 0; mh * 4 < TM; ++mh) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (mh * 4 + t < TM) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    *reinterpret_cast<f32x4*>(ebuf + (t * 16 + (lane & 15)) * EROW + (nt * 16 + 4 * (lane >> 4)) * 4) =
                        acc[mh * 4 + t][nt];
            }
        const int gm0 = m0 + wm * (TM * 16) + mh * 64 + e_r;
        constexpr int ITS_FULL = 8;
        const int its = (TM - mh * 4 >= 4) ? ITS_FULL : (TM - mh * 4) * 2;   // 8 rows per iteration
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it >= its) break;
            const int m = gm0 + it * 8;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4 + 16);
            if (m < p.M && gn < p.N) {
                v0 += bias0;
                v1 += bias1;
                if constexpr (PF_RESID) {
                    const int slot = mh * 8 + it;
                    v0 += rp[slot][0];
                    v1 += rp[slot][1];
                    float* o = reinterpret_cast<float*>(p.out) + (long)m * p.ldo + gn;
                    *reinterpret_cast<f32x4*>(o) = v0;
                    *reinterpret_cast<f32x4*>(o + 4) = v1;
                } else if constexpr (PF_AUX) {
                    const u32x4 a = ap[mh * 8 + it];
                    f32x4 r0 = {v0[0] * bf_lo(a[0]), v0[1] * bf_hi(a[0]), v0[2] * bf_lo(a[1]), v0[3] * bf_hi(a[1])};
                    f32x4 r1 = {v1[0] * bf_lo(a[2]), v1[1] * bf_hi(a[2]), v1[2] * bf_lo(a[3]), v1[3] * bf_hi(a[3])};
                    cs0 += r0;
                    cs1 += r1;
                    u32x4 o = {pack_bf2(r0[0], r0[1]), pack_bf2(r0[2], r0[3]), pack_bf2(r1[0], r1[1]), pack_bf2(r1[2], r1[3])};
                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.ldo + gn) = o;
                } else {
                    nt_epilogue8<EPI>(p, m, gn, v0, v1, cs0, cs1);
                }
            }
        }
    }
    if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        if (p.out2) wg_colsum_flush(smem, reinterpret_cast<float*>(p.out2), n0, p.N, gn - n0, wm * 8 + e_r, cs0, cs1, N2_BN);   // block-uniform
    }
}

// ------------------------------------------------------------------------------------------
// NT kernel, 160x256x64 tile, 8 waves, THREE-stage LDS ring (156 KiB), prefetch distance 2, hand-counted
// LDS-DMA waits.  For the single-round GEMMs (N = width: one tile per CU, nothing else on the CU to hide
// latency) the two-stage loop is DMA-latency bound: the tile issued at the top of an iteration must have landed
// by its end.  Here every DMA gets two iterations to land and the wait before the barrier leaves the newest
// stage in flight (vmcnt(7) / vmcnt(6): waves 0-3 issue 7 DMA instructions per stage, waves 4-7 issue 6).
// ------------------------------------------------------------------------------------------
constexpr int N4_BM = 160, N4_BN = 256, N4_BK = 64, N4_TM = 5;
constexpr int N4_A_BYTES = N4_BM * N4_BK * 2;                // 20 KiB
constexpr int N4_STAGE_BYTES = (N4_BM + N4_BN) * N4_BK * 2;   // 52 KiB
constexpr int N4_LDS_BYTES = 3 * N4_STAGE_BYTES;             // 156 KiB (>= 8 epilogue slices of 17 KiB)
constexpr int N4P_LDS_BYTES = N4_LDS_BYTES + 64;             // persistent kernel: + the two tile ids of its dynamic tile list
#ifndef CE_N4_LOADERS
#define CE_N4_LOADERS 4
#endif
constexpr int N4_LOADERS = CE_N4_LOADERS;                    // loader waves of gemm_nt160lw_kernel (divides 20, 16, 12 and 32: 4 or 2)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt160_kernel(NTArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * N4_BM, n0 = tn * N4_BN;
    const int rowsA = min(p.M - m0, N4_BM), rowsB = min(p.N - n0, N4_BN);
    const u32x4 rA = make_rsrc_words(p.A + (long)m0 * p.lda, (uint32_t)((long)rowsA * p.lda * 2));
    const u32x4 rB = make_rsrc_words(p.B + (long)n0 * p.ldb, (uint32_t)((long)rowsB * p.ldb * 2));

    // DMA map: one instruction = 8 rows x 128 B; lane -> row l>>3, LDS position l&7, source chunk = pos ^ (row&7)
    const int s_row = lane >> 3;
    const int s_chunk = (lane & 7) ^ s_row;
    uint32_t vA[3], vB[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) vA[i] = (uint32_t)(((wave + 8 * i) * 8 + s_row) * p.lda * 2 + s_chunk * 16);   // A instr = wave + 8 i (< 20)
#pragma unroll
    for (int i = 0; i < 4; ++i) vB[i] = (uint32_t)(((wave * 4 + i) * 8 + s_row) * p.ldb * 2 + s_chunk * 16);   // B instr = 4 wave + i
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem;
    auto stage = [&](int st, int kt) {
        const uint32_t base = lds0 + st * N4_STAGE_BYTES;
        const uint32_t kb = (uint32_t)(kt * N4_BK * 2);
        dma16_bounds(rA, base + wave * 1024, vA[0] + kb);
        dma16_bounds(rA, base + (wave + 8) * 1024, vA[1] + kb);
        if (wave < 4) dma16_bounds(rA, base + (wave + 16) * 1024, vA[2] + kb);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16_bounds(rB, base + N4_A_BYTES + (wave * 4 + i) * 1024, vB[i] + kb);
    };
    auto wait_prev = [&](bool newest_in_flight) {      // all but this wave's newest stage have landed
        if (newest_in_flight) {
            if (wave < 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    f32x4 acc[N4_TM][4];
#pragma unroll
    for (int i = 0; i < N4_TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int f_row = lane & 15, f_kc = lane >> 4, f_sw = lane & 7;
    const int fa_base = (wm * (N4_TM * 16) + f_row) * 128;
    const int fb_base = N4_A_BYTES + (wn * 64 + f_row) * 128;

    const int nk = p.K / N4_BK;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    wait_prev(nk > 1);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nxt2 = cur == 0 ? 2 : cur - 1;           // (cur + 2) % 3
        if (kt + 2 < nk) stage(nxt2, kt + 2);
        const char* st = smem + cur * N4_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + f_kc) ^ f_sw) << 4;
            bf16x8 wf[4], af[N4_TM];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(st + fb_base + t * 2048 + coff);
#pragma unroll
            for (int t = 0; t < N4_TM; ++t) af[t] = *reinterpret_cast<const bf16x8*>(st + fa_base + t * 2048 + coff);
#pragma unroll
            for (int t = 0; t < N4_TM; ++t)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[t], acc[t][nt], 0, 0, 0);
        }
        wait_prev(kt + 2 < nk);                            // stage kt+1 landed; kt+2 may still be in flight
        __syncthreads();
        cur = cur == 2 ? 0 : cur + 1;
    }

    // ---- epilogue through LDS (ring memory is free after the last barrier): per wave 64-row x 64-col fp32 slices
    constexpr int EROW = 272;
    char* ebuf = smem + wave * (64 * EROW);
    const int e_r = lane >> 3, e_c = (lane & 7) * 8;
    const int gn = n0 + wn * 64 + e_c;
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (epi_has_bias(EPI)) {
        if (gn < p.N) {
            bias0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
            bias1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
        }
    }
#pragma unroll
    for (int mh = 0; mh * 4 < N4_TM; ++mh) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (mh * 4 + t < N4_TM) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    *reinterpret_cast<f32x4*>(ebuf + (t * 16 + (lane & 15)) * EROW + (nt * 16 + 4 * (lane >> 4)) * 4) =
                        acc[mh * 4 + t][nt];
            }
        const int gm0 = m0 + wm * (N4_TM * 16) + mh * 64 + e_r;
        const int its = (N4_TM - mh * 4 >= 4) ? 8 : (N4_TM - mh * 4) * 2;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it >= its) break;
            const int m = gm0 + it * 8;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4 + 16);
            if (m < p.M && gn < p.N) {
                v0 += bias0;
                v1 += bias1;
                nt_epilogue8<EPI>(p, m, gn, v0, v1, cs0, cs1);
            }
        }
    }
    if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        if (p.out2) {
            float* colsum = reinterpret_cast<float*>(p.out2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    cs0[e] += __shfl_xor(cs0[e], o, 64);
                    cs1[e] += __shfl_xor(cs1[e], o, 64);
                }
            }
            const int e = lane >> 3;      // lane l adds element l>>3 of its 8 column sums: the wave's 64 columns in one instruction
            const float v = e == 0 ? cs0[0] : e == 1 ? cs0[1] : e == 2 ? cs0[2] : e == 3 ? cs0[3]
                          : e == 4 ? cs1[0] : e == 5 ? cs1[1] : e == 6 ? cs1[2] : cs1[3];
            if (gn + e < p.N) atomicAdd(colsum + gn + e, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same 160x256x64 tile and three-stage ring with DEDICATED LOADER WAVES: waves 8-11 issue every LDS-DMA
// instruction (13 each per stage) and wait for them; waves 0-7 only read fragments and issue MFMAs.  In the kernels
// above each wave spends as long issuing its share of the stage (the CU's one vector-memory path takes 16 cycles per
// 1 KiB instruction, and all waves queue on it together) as on its 40 MFMAs, and issues no MFMA meanwhile
// (tools/diag/nt256_stamps.hip); here that queue stalls only waves that have nothing else to do.
// ------------------------------------------------------------------------------------------
// One stage (128 bytes of the contraction per row: 64 bf16 or 128 e4m3 values) of a compute wave's 16 TM x 64 block.
// F8: both operands e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (per-row scales are applied in the epilogue):
// one instruction (32 cycles) per accumulator block instead of two bf16 ones -- the same stage cadence at twice the contraction
// depth.  A lane's 32 values are the 16-byte chunks (g, 4 + g) of its row = the instruction's native k order
// (tools/diag/probe_mfma_scale.py), the operands' rows are staged, swizzled and read exactly as in bf16.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
template <int TM, int F8>
__device__ __forceinline__ void nt160_stage_mma(const char* st, int fa_base, int fb_base, int f_kc, int f_sw, f32x4 (&acc)[TM][4]) {
    if constexpr (F8) {
        constexpr int ONE = 0x7f7f7f7f;
        const int c0 = (f_kc ^ f_sw) << 4, c1 = ((4 + f_kc) ^ f_sw) << 4;
        i32x8_t wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(st + fb_base + t * 2048 + c0);
            const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(st + fb_base + t * 2048 + c1);
            wf[t] = i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        // A fragments two at a time (8 registers each: all TM of them next to the 32 of the weights spilled at TM = 5)
        auto load_a = [&](int t) __attribute__((always_inline)) -> i32x8_t {
            const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(st + fa_base + t * 2048 + c0);
            const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(st + fa_base + t * 2048 + c1);
            return i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        i32x8_t a_cur = load_a(0);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            i32x8_t a_nxt = a_cur;
            if (t + 1 < TM) a_nxt = load_a(t + 1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[t][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], a_cur, acc[t][nt], 0, 0, 0, ONE, 0, ONE);
            __builtin_amdgcn_sched_barrier(0);           // keep the scheduler from hoisting every fragment read to the top
            a_cur = a_nxt;
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + f_kc) ^ f_sw) << 4;
            bf16x8 wf[4], af[TM];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(st + fb_base + t * 2048 + coff);
#pragma unroll
            for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const bf16x8*>(st + fa_base + t * 2048 + coff);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[t], acc[t][nt], 0, 0, 0);
        }
    }
}

template <int EPI, int TM, int F8 = 0>
__global__ __launch_bounds__(64 * (8 + N4_LOADERS), 3) void gemm_nt160lw_kernel(NTArgs p) {
    constexpr int BM = 32 * TM;                                   // 160 / 128 / 96 rows
    constexpr int A_BYTES = BM * N4_BK * 2;
    constexpr int STAGE_BYTES = (BM + N4_BN) * N4_BK * 2;
    constexpr int A_INSTR = BM / 8;                               // 20 / 16 / 12 (N4_LOADERS divides each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * N4_BN;
    constexpr int ES = F8 ? 1 : 2;                               // bytes per operand element; a stage is 128 bytes of every row
    const int rowsA = min(p.M - m0, BM), rowsB = min(p.N - n0, N4_BN);
    const u32x4 rA = make_rsrc_words(reinterpret_cast<const char*>(p.A) + (long)m0 * p.lda * ES, (uint32_t)((long)rowsA * p.lda * ES));
    const u32x4 rB = make_rsrc_words(reinterpret_cast<const char*>(p.B) + (long)n0 * p.ldb * ES, (uint32_t)((long)rowsB * p.ldb * ES));

    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem;
    const int nk = p.K * ES / 128;
    if (wave >= 8) {
        // ---- loader waves: all LDS-DMA instructions of a stage (4 TM A + 32 B, 8 rows x 128 B each), TM + 8 per wave.
        // lane -> row l>>3, LDS position l&7, source chunk = pos ^ (row&7)
        const int lw = wave - 8;
        const int s_row = lane >> 3;
        const int s_chunk = (lane & 7) ^ s_row;
        const uint32_t vA0 = (uint32_t)((lw * 8 + s_row) * p.lda * ES + s_chunk * 16);     // A instr = lw + 4 i (i < TM)
        const uint32_t vB0 = (uint32_t)((lw * 8 + s_row) * p.ldb * ES + s_chunk * 16);     // B instr = lw + 4 i (i < 8)
        const uint32_t stepA = (uint32_t)(8 * N4_LOADERS * p.lda * ES), stepB = (uint32_t)(8 * N4_LOADERS * p.ldb * ES);
        auto stage = [&](int st, int kt) {
            const uint32_t base = lds0 + st * STAGE_BYTES + lw * 1024;
            const uint32_t kb = (uint32_t)(kt * 128);
#pragma unroll
            for (int i = 0; i < A_INSTR / N4_LOADERS; ++i) dma16_bounds(rA, base + i * (1024 * N4_LOADERS), vA0 + i * stepA + kb);
#pragma unroll
            for (int i = 0; i < 32 / N4_LOADERS; ++i) dma16_bounds(rB, base + A_BYTES + i * (1024 * N4_LOADERS), vB0 + i * stepB + kb);
        };
        stage(0, 0);
        if (nk > 1) {
            stage(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_INSTR + 32) / N4_LOADERS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int nxt2 = cur == 0 ? 2 : cur - 1;           // (cur + 2) % 3: read in iteration kt-1, free since its barrier
            if (kt + 2 < nk) {
                stage(nxt2, kt + 2);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A_INSTR + 32) / N4_LOADERS) : "memory");   // stage kt+1 landed; kt+2 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            cur = cur == 2 ? 0 : cur + 1;
        }
        return;
    }

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int f_row = lane & 15, f_kc = lane >> 4, f_sw = lane & 7;
    const int fa_base = (wm * (TM * 16) + f_row) * 128;
    const int fb_base = A_BYTES + (wn * 64 + f_row) * 128;

    // Epilogue operand prefetch (see gemm_nt160p_kernel): the fp16 residual tile / the bf16 derivative tile, the first NPQ
    // 8-row slots requested under the last K iteration, the others as soon as the first four accumulator blocks sit in LDS
    constexpr bool PF_AUX = EPI == CE_EPI_GELUGRAD_BF16 || EPI == CE_EPI_BIAS_RESID_F16;
    constexpr int NPQ = 2;
    constexpr int NSLOT = 2 * TM;
    const int e_r = lane >> 3, e_c = (lane & 7) * 8;
    const int gn = n0 + wn * 64 + e_c;
    const bool col_ok = gn < p.N;
    const int gnc = col_ok ? gn : 0;
    // epilogue I/O through per-tile buffer descriptors, branch-free (EpiBuf, gemm_common.hpp); the bias / column scales on a clamped column
    const int e_row = wm * (TM * 16) + e_r, e_col = wn * 64 + e_c;
    constexpr int OB = epi_out_bytes(EPI);
    const EpiBuf eo = epi_buf(p.out, p.ldo, OB, p.M, p.N, m0, n0, e_row, e_col, col_ok);
    EpiBuf eo2 = eo, er = eo, ea = eo;
    if constexpr (EPI == CE_EPI_BIAS_GELU) eo2 = epi_buf(p.out2, p.ldo2, 2, p.M, p.N, m0, n0, e_row, e_col, col_ok);
    if constexpr (EPI == CE_EPI_BIAS_RESID_F32) er = epi_buf(p.resid, p.ldr, 4, p.M, p.N, m0, n0, e_row, e_col, col_ok);
    if constexpr (EPI == CE_EPI_GELUGRAD_BF16) ea = epi_buf(p.aux, p.ldaux, 2, p.M, p.N, m0, n0, e_row, e_col, col_ok);
    if constexpr (EPI == CE_EPI_BIAS_RESID_F16) ea = epi_buf(p.resid, p.ldr, 2, p.M, p.N, m0, n0, e_row, e_col, col_ok);
    u32x4 ap[PF_AUX ? NSLOT : 1];
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 sb0 = {1.f, 1.f, 1.f, 1.f}, sb1 = {1.f, 1.f, 1.f, 1.f};          // F8: per-column dequantisation scales
    auto load_cols = [&]() __attribute__((always_inline)) {
        if constexpr (epi_has_bias(EPI)) {
            bias0 = *reinterpret_cast<const f32x4*>(p.bias + gnc);
            bias1 = *reinterpret_cast<const f32x4*>(p.bias + gnc + 4);
        }
        if constexpr (F8) {
            sb0 = *reinterpret_cast<const f32x4*>(p.sb + gnc);
            sb1 = *reinterpret_cast<const f32x4*>(p.sb + gnc + 4);
        }
    };
    constexpr bool COLS_EARLY = !PF_AUX && !F8;     // the bias under the last K iteration where 8 registers are to spare

    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == nk - 1) {
            if constexpr (COLS_EARLY) load_cols();
            if constexpr (PF_AUX) {
#pragma unroll
                for (int q = 0; q < NPQ && q < NSLOT; ++q) ap[q] = epi_bload16(ea, q);
            }
        }
        nt160_stage_mma<TM, F8>(smem + cur * STAGE_BYTES, fa_base, fb_base, f_kc, f_sw, acc);
        __syncthreads();                                   // the loaders arrive once stage kt+1 has landed
        cur = cur == 2 ? 0 : cur + 1;
    }

    // ---- epilogue through LDS (ring memory is free after the last barrier): per wave 64-row x 64-col fp32 slices
    constexpr int EROW = 272;
    char* ebuf = smem + wave * (64 * EROW);
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!COLS_EARLY) load_cols();
#pragma unroll
    for (int mh = 0; mh * 4 < TM; ++mh) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (mh * 4 + t < TM) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    *reinterpret_cast<f32x4*>(ebuf + (t * 16 + (lane & 15)) * EROW + (nt * 16 + 4 * (lane >> 4)) * 4) =
                        acc[mh * 4 + t][nt];
            }
        if constexpr (PF_AUX) {      // the blocks just parked free their registers: request every remaining slot
            if (mh == 0) {
#pragma unroll
                for (int q = NPQ; q < NSLOT; ++q) ap[q] = epi_bload16(ea, q);
            }
        }
        const int its = (TM - mh * 4 >= 4) ? 8 : (TM - mh * 4) * 2;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it >= its) break;
            const int es = mh * 8 + it;                     // 8-row slot of this wave's row block
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4 + 16);
            u32x4 a = {0u, 0u, 0u, 0u};
            if constexpr (PF_AUX) {
                a = ap[es];
            }
            if constexpr (F8) {
                const float sa = p.sa[min(m0 + e_row + es * 8, p.M - 1)];
                v0 = v0 * sa * sb0;
                v1 = v1 * sa * sb1;
            }
            v0 += bias0;
            v1 += bias1;
            if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
                // rows past M: the derivative tile reads as 0 there, so they add nothing to the column sums
                f32x4 r0 = {v0[0] * bf_lo(a[0]), v0[1] * bf_hi(a[0]), v0[2] * bf_lo(a[1]), v0[3] * bf_hi(a[1])};
                f32x4 r1 = {v1[0] * bf_lo(a[2]), v1[1] * bf_hi(a[2]), v1[2] * bf_lo(a[3]), v1[3] * bf_hi(a[3])};
                cs0 += r0;
                cs1 += r1;
                u32x4 o = {pack_bf2(r0[0], r0[1]), pack_bf2(r0[2], r0[3]), pack_bf2(r1[0], r1[1]), pack_bf2(r1[2], r1[3])};
                epi_bstore16(eo, es, o);
            } else if constexpr (EPI == CE_EPI_BIAS_RESID_F16) {
                v0 += f16x4_to_f32(u32x2{a[0], a[1]});
                v1 += f16x4_to_f32(u32x2{a[2], a[3]});
                const u32x2 h0 = f32_to_f16x4_sat(v0), h1 = f32_to_f16x4_sat(v1);
                epi_bstore16(eo, es, u32x4{h0[0], h0[1], h1[0], h1[1]});
            } else {
                nt_epilogue8b<EPI>(eo, eo2, er, es, v0, v1);
            }
        }
    }
    if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        if (p.out2) {
            float* colsum = reinterpret_cast<float*>(p.out2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    cs0[e] += __shfl_xor(cs0[e], o, 64);
                    cs1[e] += __shfl_xor(cs1[e], o, 64);
                }
            }
            const int e = lane >> 3;      // lane l adds element l>>3 of its 8 column sums: the wave's 64 columns in one instruction
            const float v = e == 0 ? cs0[0] : e == 1 ? cs0[1] : e == 2 ? cs0[2] : e == 3 ? cs0[3]
                          : e == 4 ? cs1[0] : e == 5 ? cs1[1] : e == 6 ? cs1[2] : cs1[3];
            if (gn + e < p.N) atomicAdd(colsum + gn + e, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// PERSISTENT loader-wave kernel for launches of several rounds of tiles (N = 3d, 4d): one workgroup per CU walks its
// tiles (xb, xb + grid, ...); the ring, the barrier protocol and the stage count per loader never stop at a tile edge, so
// the loaders fetch the next tile's first two stages while the compute waves are in the epilogue of the current one, and
// that epilogue's stores drain under the next tile's MFMAs.  The epilogue transposes 16-row chunks through the ring slot
// of the tile's LAST stage (free until the barrier that opens the next tile's first iteration; one extra barrier per
// tile makes sure every wave has finished reading it).  Needs K >= 128.
// ------------------------------------------------------------------------------------------
// TS > 0: TWO TILE HEIGHTS in one launch -- row panels 0 .. p.tall_panels-1 are 32 TM rows tall, the others 32 TS (< TM); the ring
// keeps the tall layout and the loaders their instruction count (a short tile's missing rows are out of its descriptor's range:
// no memory traffic), the compute waves run a body specialised for the tile's height.  The launcher picks the split so that
// every workgroup's list costs the same (launch_nt): 960 tiles of 160 rows = 3.75 rounds become 3 rounds of 160 + 1 of 128.
template <int EPI, int TM, int F8 = 0, int TS = 0>
__global__ __launch_bounds__(64 * (8 + N4_LOADERS), 3) void gemm_nt160p_kernel(NTArgs p) {
    constexpr int BM = 32 * TM;
    constexpr int A_BYTES = BM * N4_BK * 2;
    constexpr int STAGE_BYTES = (BM + N4_BN) * N4_BK * 2;
    constexpr int A_INSTR = BM / 8;
    constexpr int PER = (A_INSTR + 32) / N4_LOADERS;              // LDS-DMA instructions per loader wave and stage
    // Epilogue operand (the bf16 derivative tile of GELUGRAD, the fp16 residual tile of BIAS_RESID_F16: 16 B per lane and 8-row
    // slot): the first NPQ slots are requested under the last K iteration; the rest go out in groups of four as the epilogue
    // parks a 16-row accumulator block in LDS -- the 16 registers that block frees hold the four slots -- so every slot is in
    // flight at least one block ahead of its use and the register peak stays at the accumulators + NPQ slots.  (Round 2 held
    // the first 6 of a tile's 10 slots and loaded the rest on use: 7 VGPRs spilled at the 168-VGPR budget; 2 held slots
    // without the early requests cost the GELUGRAD class 6 %.)
    constexpr bool PF_AUX = EPI == CE_EPI_GELUGRAD_BF16 || EPI == CE_EPI_BIAS_RESID_F16;
    // NPQ (per tile height, below): slots requested under the last K iteration (GELUGRAD also carries 8 column-sum registers: 2
    // spilled 3 VGPRs at TM = 5; the e4m3 form's fragments are twice as wide: none)
    // first row and height of row panel tm
    auto panel_m0 = [&](int tm) __attribute__((always_inline)) -> int {
        if constexpr (TS > 0) return tm < p.tall_panels ? tm * BM : p.tall_panels * BM + (tm - p.tall_panels) * (32 * TS);
        else return tm * BM;
    };
    auto panel_rows = [&](int tm) __attribute__((always_inline)) -> int {
        if constexpr (TS > 0) return tm < p.tall_panels ? BM : 32 * TS;
        else return BM;
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.tiles_m * p.tiles_n;
    const PersistWalk walk = persist_walk(p, total);              // this workgroup's tiles: first + t * step (grid <= tiles)
    constexpr int ES = F8 ? 1 : 2;                                // bytes per operand element; a stage is 128 bytes of every row
    const int nk = p.K * ES / 128;                                // >= 2
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem;
    // DYNAMIC tile list (p.tile_queue; the launcher sets it only for nk >= 3 and the launch-wide walk): the first tile is the
    // static one, every further tile is fetched from a device counter by compute wave 0 while the current tile runs.  A workgroup
    // that the dispatcher could not place (its CU held by another stream's kernel: an RCCL channel) then finds the
    // list empty when it finally starts, instead of holding the launch up for a whole static list.
    const bool dyn = p.tile_queue != nullptr;
    // end of this workgroup's list: a static list has walk.count tiles (the XCD-owned walk's lists end inside the tile range);
    // a dynamic one runs until the counter passes the last tile
    const int list_end = dyn ? total : walk.first + walk.count * walk.step;
    volatile int* tq = reinterpret_cast<volatile int*>(smem + 3 * N4_STAGE_BYTES);      // [2]: id of tile t in tq[t & 1] (behind the ring)

    if (wave >= 8) {
        // ---- loader waves: instruction j = lw + N4_LOADERS * i of a stage (8 rows x 128 B each; A_INSTR of A, then 32 of B)
        const int lw = wave - 8;
        const int s_row = lane >> 3;
        const int s_chunk = (lane & 7) ^ s_row;
        const uint32_t vA0 = (uint32_t)((lw * 8 + s_row) * p.lda * ES + s_chunk * 16);
        const uint32_t vB0 = (uint32_t)((lw * 8 + s_row) * p.ldb * ES + s_chunk * 16);
        const uint32_t stepA = (uint32_t)(8 * N4_LOADERS * p.lda * ES), stepB = (uint32_t)(8 * N4_LOADERS * p.ldb * ES);
        auto desc_tile = [&](int tile, u32x4& rA, u32x4& rB) {
            int tm, tn;
            persist_coords(p, tile, tm, tn);
            const int m0 = panel_m0(tm), n0 = tn * N4_BN;
            rA = make_rsrc_words(reinterpret_cast<const char*>(p.A) + (long)m0 * p.lda * ES, (uint32_t)((long)min(p.M - m0, panel_rows(tm)) * p.lda * ES));
            rB = make_rsrc_words(reinterpret_cast<const char*>(p.B) + (long)n0 * p.ldb * ES, (uint32_t)((long)min(p.N - n0, N4_BN) * p.ldb * ES));
        };
        auto issue = [&](const u32x4& rA, const u32x4& rB, int slot, int kt) {
            const uint32_t base = lds0 + slot * STAGE_BYTES + lw * 1024;
            const uint32_t kb = (uint32_t)(kt * 128);
#pragma unroll
            for (int i = 0; i < A_INSTR / N4_LOADERS; ++i) dma16_bounds(rA, base + i * (1024 * N4_LOADERS), vA0 + i * stepA + kb);
#pragma unroll
            for (int i = 0; i < 32 / N4_LOADERS; ++i) dma16_bounds(rB, base + A_BYTES + i * (1024 * N4_LOADERS), vB0 + i * stepB + kb);
        };
        u32x4 rA, rB, rA2, rB2;
        desc_tile(walk.first, rA, rB);
        rA2 = rA; rB2 = rB;
        issue(rA, rB, 0, 0);
        issue(rA, rB, 1, 1);
        int slot = 0;                                             // ring slot of the stage the compute waves multiply next
        int cur = walk.first;
        for (int t = 0; cur < list_end; ++t) {
            // the next tile: static list (first + (t + 1) step), or -- dynamic -- the id compute wave 0 fetched during this tile's
            // first K iteration and parked in tq[(t + 1) & 1] (published by the barrier of iteration 1; nk >= 3)
            int nxt = dyn ? total : cur + walk.step;
            bool has_next = !dyn && nxt < list_end;
            if (has_next) desc_tile(nxt, rA2, rB2);
            for (int kt = 0; kt < nk; ++kt) {
                // this wave's part of stage (t, kt) has landed; the following stage (issued already) may be in flight
                if (kt + 1 < nk || has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                  // publishes the stage; the previous one has been read
                if (dyn && kt == nk - 2) {
                    nxt = __builtin_amdgcn_readfirstlane(tq[(t + 1) & 1]);
                    has_next = nxt < total;
                    if (has_next) desc_tile(nxt, rA2, rB2);
                }
                const int s2 = slot == 0 ? 2 : slot - 1;          // its slot takes the stage two ahead
                if (kt + 2 < nk) issue(rA, rB, s2, kt + 2);
                else if (has_next) issue(rA2, rB2, s2, kt + 2 - nk);
                slot = slot == 2 ? 0 : slot + 1;
            }
            __syncthreads();                                      // the tile's last stage has been read: its slot is epilogue scratch
            rA = rA2; rB = rB2;
            cur = nxt;
        }
        return;
    }

    const int wm = wave >> 2, wn = wave & 3;
    const int f_row = lane & 15, f_kc = lane >> 4, f_sw = lane & 7;
    const int fb_base = A_BYTES + (wn * 64 + f_row) * 128;
    constexpr int EROW = 272;
    const int e_r = lane >> 3, e_c = (lane & 7) * 8;
    int slot = 0;
    int cur = walk.first;
    int t = 0;
    // one tile of TMA x 32 rows (TMA = TM, or TS for a short tile): the body is specialised for the height
    auto tile_body = [&](auto tma_c, int tm, int tn) __attribute__((always_inline)) {
        constexpr int TMA = decltype(tma_c)::value;
        constexpr int NPQ = F8 ? 0 : ((EPI == CE_EPI_GELUGRAD_BF16 && TMA >= 5) ? 1 : 2);
        constexpr int NSLOT = 2 * TMA;
        const int fa_base = (wm * (TMA * 16) + f_row) * 128;
        const int m0 = panel_m0(tm), n0 = tn * N4_BN;
        f32x4 acc[TMA][4];
#pragma unroll
        for (int i = 0; i < TMA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int last = slot;
        auto k_iter = [&]() __attribute__((always_inline)) {
            __syncthreads();
            const char* st = smem + slot * STAGE_BYTES;
            last = slot;
            slot = slot == 2 ? 0 : slot + 1;
            nt160_stage_mma<TMA, F8>(st, fa_base, fb_base, f_kc, f_sw, acc);
        };
        // dynamic list: wave 0 requests the next tile id now (one returning atomic from lane 0; the other lanes' offsets are
        // out of the descriptor's range) and parks it in LDS after the first K iteration, where its latency has passed
        int fetched = 0;
        if (dyn && wave == 0) {
            const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.tile_queue, 4u);
            fetched = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rq, lane == 0 ? 0 : 64, 0, 0);
        }
        k_iter();
        if (dyn && wave == 0 && lane == 0) tq[(t + 1) & 1] = (int)gridDim.x + fetched;
        for (int kt = 1; kt + 1 < nk; ++kt) k_iter();
        // Epilogue I/O through per-tile buffer descriptors, branch-free (EpiBuf, gemm_common.hpp); the bias / column scales on a
        // clamped column.
        const int gn = n0 + wn * 64 + e_c;
        const bool col_ok = gn < p.N;
        const int gnc = col_ok ? gn : 0;
        const int e_row = wm * (TMA * 16) + e_r, e_col = wn * 64 + e_c;
        constexpr int OB = epi_out_bytes(EPI);
        const EpiBuf eo = epi_buf(p.out, p.ldo, OB, p.M, p.N, m0, n0, e_row, e_col, col_ok);
        EpiBuf eo2 = eo, er = eo, ea = eo;
        if constexpr (EPI == CE_EPI_BIAS_GELU) eo2 = epi_buf(p.out2, p.ldo2, 2, p.M, p.N, m0, n0, e_row, e_col, col_ok);
        if constexpr (EPI == CE_EPI_BIAS_RESID_F32) er = epi_buf(p.resid, p.ldr, 4, p.M, p.N, m0, n0, e_row, e_col, col_ok);
        if constexpr (EPI == CE_EPI_GELUGRAD_BF16) ea = epi_buf(p.aux, p.ldaux, 2, p.M, p.N, m0, n0, e_row, e_col, col_ok);
        if constexpr (EPI == CE_EPI_BIAS_RESID_F16) ea = epi_buf(p.resid, p.ldr, 2, p.M, p.N, m0, n0, e_row, e_col, col_ok);
        u32x4 ap[PF_AUX ? NSLOT : 1];
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 sb0 = {1.f, 1.f, 1.f, 1.f}, sb1 = {1.f, 1.f, 1.f, 1.f};      // F8: per-column dequantisation scales
        auto load_cols = [&]() __attribute__((always_inline)) {
#ifdef CE_DIAG_NO_BIAS_LOAD
            return;
#endif
            if constexpr (epi_has_bias(EPI)) {
                bias0 = *reinterpret_cast<const f32x4*>(p.bias + gnc);
                bias1 = *reinterpret_cast<const f32x4*>(p.bias + gnc + 4);
            }
            if constexpr (F8) {
                sb0 = *reinterpret_cast<const f32x4*>(p.sb + gnc);
                sb1 = *reinterpret_cast<const f32x4*>(p.sb + gnc + 4);
            }
        };
#ifdef CE_DIAG_BIAS_LATE
        constexpr bool COLS_EARLY = false;
#else
        constexpr bool COLS_EARLY = !PF_AUX && !F8;     // the bias under the last K iteration where 8 registers are to spare
#endif
        if constexpr (COLS_EARLY) load_cols();
        if constexpr (PF_AUX) {
#pragma unroll
            for (int q = 0; q < NPQ && q < NSLOT; ++q) ap[q] = epi_bload16(ea, q);
        }
        k_iter();
        if constexpr (!COLS_EARLY) load_cols();
        __syncthreads();

        // ---- epilogue: 16 rows x 64 columns at a time through this wave's 4.25 KiB of the free slot
        char* ebuf = smem + last * STAGE_BYTES + wave * (16 * EROW);
        f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TMA; ++i) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(ebuf + (lane & 15) * EROW + (nt * 16 + 4 * (lane >> 4)) * 4) = acc[i][nt];
            if constexpr (PF_AUX) {      // the block just parked frees 16 registers: request the next four slots into them
                constexpr int GRP = F8 ? 2 : 4;      // (e4m3 form: two -- its wider fragments leave no room for four)
#pragma unroll
                for (int q = NPQ + GRP * i; q < NPQ + GRP * i + GRP; ++q)
                    if (q < NSLOT) ap[q] = epi_bload16(ea, q);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int es = i * 2 + it;                      // 8-row slot of this wave's row block
                f32x4 v0 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4 + 16);
                u32x4 a = {0u, 0u, 0u, 0u};
                if constexpr (PF_AUX) {
                    a = ap[es];
                }
                if constexpr (F8) {
                    const float sa = p.sa[min(m0 + e_row + es * 8, p.M - 1)];
                    v0 = v0 * sa * sb0;
                    v1 = v1 * sa * sb1;
                }
                v0 += bias0;
                v1 += bias1;
                if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
                    // rows past M: the derivative tile reads as 0 there, so they add nothing to the column sums
                    f32x4 r0 = {v0[0] * bf_lo(a[0]), v0[1] * bf_hi(a[0]), v0[2] * bf_lo(a[1]), v0[3] * bf_hi(a[1])};
                    f32x4 r1 = {v1[0] * bf_lo(a[2]), v1[1] * bf_hi(a[2]), v1[2] * bf_lo(a[3]), v1[3] * bf_hi(a[3])};
                    cs0 += r0;
                    cs1 += r1;
                    u32x4 o = {pack_bf2(r0[0], r0[1]), pack_bf2(r0[2], r0[3]), pack_bf2(r1[0], r1[1]), pack_bf2(r1[2], r1[3])};
                    epi_bstore16(eo, es, o);
                } else if constexpr (EPI == CE_EPI_BIAS_RESID_F16) {
                    v0 += f16x4_to_f32(u32x2{a[0], a[1]});
                    v1 += f16x4_to_f32(u32x2{a[2], a[3]});
                    const u32x2 h0 = f32_to_f16x4_sat(v0), h1 = f32_to_f16x4_sat(v1);
                    epi_bstore16(eo, es, u32x4{h0[0], h0[1], h1[0], h1[1]});
                } else {
                    nt_epilogue8b<EPI>(eo, eo2, er, es, v0, v1);
                }
            }
        }
        if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
            if (p.out2) {
                float* colsum = reinterpret_cast<float*>(p.out2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) {
                        cs0[e] += __shfl_xor(cs0[e], o, 64);
                        cs1[e] += __shfl_xor(cs1[e], o, 64);
                    }
                }
                // after the reduction every lane holds the sums of its 8 columns; lane l adds element l>>3 of them, so the
                // wave's 64 columns go out in ONE atomic instruction (eight 8-lane instructions per wave and tile cost
                // ~3 us per tile at the chip's float-atomic rate)
                const int e = lane >> 3;
                const float v = e == 0 ? cs0[0] : e == 1 ? cs0[1] : e == 2 ? cs0[2] : e == 3 ? cs0[3]
                              : e == 4 ? cs1[0] : e == 5 ? cs1[1] : e == 6 ? cs1[2] : cs1[3];
                if (gn + e < p.N) atomicAdd(colsum + gn + e, v);
            }
        }
    };
    for (; cur < list_end; ++t) {
        int tm, tn;
        persist_coords(p, cur, tm, tn);
        if constexpr (TS > 0) {
            if (tm < p.tall_panels) tile_body(std::integral_constant<int, TM>{}, tm, tn);
            else tile_body(std::integral_constant<int, TS>{}, tm, tn);
        } else {
            tile_body(std::integral_constant<int, TM>{}, tm, tn);
        }
        // next tile: the static list, or the id wave 0 parked during this tile (every wave has passed a barrier since)
        cur = dyn ? __builtin_amdgcn_readfirstlane(tq[(t + 1) & 1]) : cur + walk.step;
    }
}

// ------------------------------------------------------------------------------------------
// NT kernel, 160x256x32 tile, 8 waves, SMALL footprint: 2 x 26 KiB LDS stages and <= 128 VGPRs, so two
// (even three) workgroups share a CU and one workgroup's epilogue / prologue overlaps another's MFMA loop.
// Used for the GEMMs that need several rounds of tiles (N = 3d, 4d); per-tile prologue+epilogue is ~40 % of
// a K = 768 tile when a CU runs one workgroup at a time.  64-byte LDS rows, chunk swizzle
// pos = chunk ^ 2*((row>>3)&1) (conflict-free for ds_read_b128, found by exhaustive search); operands arrive
// by bounds-checked LDS-DMA (inline asm, counted by hand); the epilogue reuses the stage memory.
// ------------------------------------------------------------------------------------------
constexpr int N3_BM = 160, N3_BN = 256, N3_BK = 32, N3_TM = 5;
constexpr int N3_A_BYTES = N3_BM * N3_BK * 2;               // 10 KiB
constexpr int N3_STAGE_BYTES = (N3_BM + N3_BN) * N3_BK * 2;  // 26 KiB
constexpr int N3_LDS_BYTES = 2 * N3_STAGE_BYTES;            // 52 KiB (>= 8 epilogue slices of 4352 B)

template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_nt32_kernel(NTArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * N3_BM, n0 = tn * N3_BN;
    const int rowsA = min(p.M - m0, N3_BM), rowsB = min(p.N - n0, N3_BN);
    const u32x4 rA = make_rsrc_words(p.A + (long)m0 * p.lda, (uint32_t)((long)rowsA * p.lda * 2));
    const u32x4 rB = make_rsrc_words(p.B + (long)n0 * p.ldb, (uint32_t)((long)rowsB * p.ldb * 2));

    // DMA map: one instruction = 16 rows x 64 B; lane -> row l>>2, LDS position l&3, source chunk = pos ^ 2*((row>>3)&1)
    const int s_row = lane >> 2;
    const int s_chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    const uint32_t vA0 = (uint32_t)((wave * 16 + s_row) * p.lda * 2 + s_chunk * 16);          // A instr = wave (and wave+8 if < 10)
    const uint32_t vA1 = (uint32_t)(((wave + 8) * 16 + s_row) * p.lda * 2 + s_chunk * 16);
    const uint32_t vB0 = (uint32_t)((wave * 16 + s_row) * p.ldb * 2 + s_chunk * 16);          // B instr = wave, wave+8
    const uint32_t vB1 = (uint32_t)(((wave + 8) * 16 + s_row) * p.ldb * 2 + s_chunk * 16);
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem;
    auto stage = [&](int st, int kt) {
        const uint32_t base = lds0 + st * N3_STAGE_BYTES;
        const uint32_t kb = (uint32_t)(kt * N3_BK * 2);
        dma16_bounds(rA, base + wave * 1024, vA0 + kb);
        if (wave < 2) dma16_bounds(rA, base + (wave + 8) * 1024, vA1 + kb);
        dma16_bounds(rB, base + N3_A_BYTES + wave * 1024, vB0 + kb);
        dma16_bounds(rB, base + N3_A_BYTES + (wave + 8) * 1024, vB1 + kb);
    };

    f32x4 acc[N3_TM][4];
#pragma unroll
    for (int i = 0; i < N3_TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment map: row = base + (lane&15), chunk = lane>>4, position = chunk ^ 2*((lane>>3)&1)
    const int f_pos = (lane >> 4) ^ (((lane >> 3) & 1) << 1);
    const int fa_base = (wm * (N3_TM * 16) + (lane & 15)) * 64 + f_pos * 16;
    const int fb_base = N3_A_BYTES + (wn * 64 + (lane & 15)) * 64 + f_pos * 16;

    const int nk = p.K / N3_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* st = smem + cur * N3_STAGE_BYTES;
        bf16x8 wf[4], af[N3_TM];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(st + fb_base + t * 1024);
#pragma unroll
        for (int t = 0; t < N3_TM; ++t) af[t] = *reinterpret_cast<const bf16x8*>(st + fa_base + t * 1024);
#pragma unroll
        for (int t = 0; t < N3_TM; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[t], acc[t][nt], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage landed (this wave's DMA)
        __syncthreads();
    }

    // ---- epilogue through LDS (stage memory is free after the last barrier): per wave 16-row x 64-col fp32 slices
    constexpr int EROW = 272;
    char* ebuf = smem + wave * (16 * EROW);
    const int e_r = lane >> 3, e_c = (lane & 7) * 8;
    const int gn = n0 + wn * 64 + e_c;
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (epi_has_bias(EPI)) {
        if (gn < p.N) {
            bias0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
            bias1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
        }
    }
#pragma unroll
    for (int t = 0; t < N3_TM; ++t) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<f32x4*>(ebuf + (lane & 15) * EROW + (nt * 16 + 4 * (lane >> 4)) * 4) = acc[t][nt];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = m0 + wm * (N3_TM * 16) + t * 16 + it * 8 + e_r;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(ebuf + (it * 8 + e_r) * EROW + e_c * 4 + 16);
            if (m < p.M && gn < p.N) {
                v0 += bias0;
                v1 += bias1;
                nt_epilogue8<EPI>(p, m, gn, v0, v1, cs0, cs1);
            }
        }
    }
    if constexpr (EPI == CE_EPI_GELUGRAD_BF16) {
        if (p.out2) wg_colsum_flush(smem, reinterpret_cast<float*>(p.out2), n0, p.N, gn - n0, wm * 8 + e_r, cs0, cs1);   // block-uniform
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel (wgrad)
// ------------------------------------------------------------------------------------------
constexpr int TN_BN = 128, TN_BK = 128, TN_BM = 64;
constexpr int TN_ROW = 320;                              // bytes: 256 + 64 pad
constexpr int TN_TILE_BYTES = TN_BM * TN_ROW;            // 20480
constexpr int TN_STAGE_BYTES = 2 * TN_TILE_BYTES;        // 40960
constexpr int TN_LDS_BYTES = 2 * TN_STAGE_BYTES;         // 81920

struct TNArgs {
    const bf16_t* P; long ldp;
    const bf16_t* Q; long ldq;
    float* out; long ldo;
    int M, Nn, Kk;
    int tiles_n, tiles_k, splits, m_per_split;  // m_per_split multiple of 64
};

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int byte_off) {
    // two ds_read_b64_tr_b16: rows r..r+3 and r+4..r+7 of a [m][cols] image -> 8 contraction values
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + byte_off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + byte_off + 4 * TN_ROW));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;

    int bid = blockIdx.x;
    const int split = bid % p.splits;
    bid /= p.splits;
    const int tk = bid % p.tiles_k, tn = bid / p.tiles_k;
    const int n0 = tn * TN_BN, k0 = tk * TN_BK;
    const int ms = split * p.m_per_split;
    const int me = min(p.M, ms + p.m_per_split);
    if (ms >= me) return;  // uniform per block
    const int rows = me - ms;

    __amdgpu_buffer_rsrc_t rP = make_rsrc(p.P + (long)ms * p.ldp, (uint32_t)((long)rows * p.ldp * 2));
    __amdgpu_buffer_rsrc_t rQ = make_rsrc(p.Q + (long)ms * p.ldq, (uint32_t)((long)rows * p.ldq * 2));

    // staging: chunk q = tid + 256*i -> row = q>>4 (0..63), c = q&15 (8 columns each)
    const int s_c = tid & 15, s_row = tid >> 4;  // + 16*i
    const bool pvalid = (n0 + s_c * 8) < p.Nn, qvalid = (k0 + s_c * 8) < p.Kk;
    const uint32_t gP0 = (uint32_t)(s_row * p.ldp * 2 + (n0 + s_c * 8) * 2);
    const uint32_t gQ0 = (uint32_t)(s_row * p.ldq * 2 + (k0 + s_c * 8) * 2);
    const uint32_t gPstep = (uint32_t)(16 * p.ldp * 2), gQstep = (uint32_t)(16 * p.ldq * 2);
    const int lds_w = s_row * TN_ROW + s_c * 16;  // + i*16*TN_ROW

    u32x4 rp[4], rq[4];
    auto issue_loads = [&](int mt) {
        const uint32_t mbP = (uint32_t)((long)mt * TN_BM * p.ldp * 2);
        const uint32_t mbQ = (uint32_t)((long)mt * TN_BM * p.ldq * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            rp[i] = pvalid ? __builtin_amdgcn_raw_buffer_load_b128(rP, gP0 + i * gPstep + mbP, 0, 0) : z;
            rq[i] = qvalid ? __builtin_amdgcn_raw_buffer_load_b128(rQ, gQ0 + i * gQstep + mbQ, 0, 0) : z;
        }
    };
    auto write_stage = [&](int stage) {
        char* sp = smem + stage * TN_STAGE_BYTES;
        char* sq = sp + TN_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(sp + lds_w + i * 16 * TN_ROW) = rp[i];
            *reinterpret_cast<u32x4*>(sq + lds_w + i * 16 * TN_ROW) = rq[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read map (32x32x16 operand): 16-lane group g = lane>>4 reads the 4-row block
    // rows 8*(g>>1) + {0..3} (then +4), columns 16*(g&1) + {0..15}; lane 4q+p of the group
    // supplies row q, columns 4p..4p+3 and receives column (lane&15).
    const int g = lane >> 4, li = lane & 15;
    const int t_off = (8 * (g >> 1) + (li >> 2)) * TN_ROW + (16 * (g & 1) + 4 * (li & 3)) * 2;
    const int tp_base = t_off + (wn * 64) * 2;                    // in P tile
    const int tq_base = TN_TILE_BYTES + t_off + (wk * 64) * 2;    // in Q tile

    const int nmt = (rows + TN_BM - 1) / TN_BM;
    issue_loads(0);
    write_stage(0);
    __syncthreads();
    for (int mt = 0; mt < nmt; ++mt) {
        const int cur = mt & 1;
        if (mt + 1 < nmt) issue_loads(mt + 1);
        const char* st = smem + cur * TN_STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // 16 contraction rows per step
            bf16x8 pf[2], qf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pf[t] = tr_frag(st, tp_base + s * 16 * TN_ROW + t * 64);
                qf[t] = tr_frag(st, tq_base + s * 16 * TN_ROW + t * 64);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[nt], qf[kt], acc[nt][kt], 0, 0, 0);
        }
        if (mt + 1 < nmt) write_stage(cur ^ 1);
        __syncthreads();
    }

    // epilogue: D[row n][col k]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ek = k0 + wk * 64 + (lane & 31);
    const int en = n0 + wn * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int k = ek + kt * 32;
            if (k >= p.Kk) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = en + nt * 32 + (r & 3) + 8 * (r >> 2);
                if (n < p.Nn) atomicAdd(p.out + (long)n * p.ldo + k, acc[nt][kt][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------
// TN kernel v2: same tile and MFMA decomposition, operands staged by LDS-DMA
// (buffer_load ... lds, bounds-checked so contraction rows past M arrive as zeros) into
// unpadded 256-byte rows; the bank-conflict fix for the transposed reads moves into an XOR
// swizzle of the 16-byte chunk index with (row&3)<<2, applied to the DMA source address and to
// the read address.  Optionally fuses the bias gradient: the waves of the k-tile-0 column
// accumulate sum_m P[m][n] with one extra MFMA against an all-ones operand.
// ------------------------------------------------------------------------------------------
constexpr int T2_TILE_BYTES = TN_BM * 256;               // 16 KiB per operand
constexpr int T2_STAGE_BYTES = 2 * T2_TILE_BYTES;        // 32 KiB
constexpr int T2_LDS_BYTES = 2 * T2_STAGE_BYTES;         // 64 KiB

__device__ __forceinline__ bf16x8 tr_frag2(const char* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// up to 4 weight-gradient problems that share the contraction length M (one residual block's four
// Linear layers) in ONE launch: the tile lists are concatenated, so the launch has enough tiles to fill the
// chip with few (usually 1-2) M splits -> 4x fewer atomic bytes and 4x longer-lived workgroups than four
// separate launches.
struct TNGroup {
    int count, splits, m_per_split, M, depth;
    int overwrite = 0;     // 256x256 kernels, splits == 1: out = product (plain stores) instead of out += product (float atomics)
    int tile_end[CE_TN_MAX_GROUP];
    TNArgs prob[CE_TN_MAX_GROUP];
};

__global__ __launch_bounds__(256, 2) void gemm_tn2_kernel(TNGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;

    // grid = (tiles of all problems) x splits; split is the fastest index so the workgroups that run together
    // sweep the SAME m-range of different tiles and share their P / Q panels in L2.  (A stream-K style split of
    // (tile, m-tile) units balances the grid perfectly but de-synchronises the m-ranges; measured 40 % slower.)
    const int m_tiles = (grp.M + TN_BM - 1) / TN_BM;
    const int mps = grp.m_per_split / TN_BM;             // m-tiles per split
    // lane-constant pieces of the DMA and transposed-read maps
    const int s_r = lane >> 4;
    const int s_chunk = (lane & 15) ^ (s_r << 2);
    const int g = lane >> 4, li = lane & 15;
    const int t_row = 8 * (g >> 1) + (li >> 2);
    const int t_sw = (li >> 2) << 2;
    const int t_cb = 2 * (g & 1) + ((li & 3) >> 1);
    int offP[2], offQ[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        offP[t] = t_row * 256 + (((wn * 8 + t * 4 + t_cb) ^ t_sw) << 4) + (li & 1) * 8;
        offQ[t] = T2_TILE_BYTES + t_row * 256 + (((wk * 8 + t * 4 + t_cb) ^ t_sw) << 4) + (li & 1) * 8;
    }

    {
        // XCD-aware order: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous range of
        // the tile-major list so the tiles_k tiles that read the same P panel (and one tn-range of Q) sit behind
        // ONE L2.  rocprofv3 FETCH_SIZE of the grouped launch: 881 MB/launch without this (314 MB algorithmic).
        const int lb = xcd_remap(blockIdx.x, gridDim.x);
        int tile = lb / grp.splits;
        const int mt0 = (lb - tile * grp.splits) * mps;
        const int mt1 = min(m_tiles, mt0 + mps);
        if (mt0 >= mt1) return;  // uniform per block
        int pi = 0;
        while (pi + 1 < grp.count && tile >= grp.tile_end[pi]) ++pi;   // block-uniform
        if (pi > 0) tile -= grp.tile_end[pi - 1];
        const TNArgs& p = grp.prob[pi];
        // tiles of a problem are enumerated in 4x4 blocks (tn x tk): a contiguous range (one XCD's share) then
        // touches ~(rows + cols) operand panels per block instead of 1 + cols, whichever operand is the wide one
        int tn, tk;
        {
            constexpr int BNB = 4, BKB = 4;
            const int br = tile / (BNB * p.tiles_k);
            const int r0 = br * BNB, rows_b = min(BNB, p.tiles_n - r0);
            const int t1 = tile - br * BNB * p.tiles_k;
            const int bc = t1 / (rows_b * BKB);
            const int c0 = bc * BKB, cols_b = min(BKB, p.tiles_k - c0);
            const int t2 = t1 - bc * rows_b * BKB;
            tn = r0 + t2 / cols_b;
            tk = c0 + t2 % cols_b;
        }
        const int n0 = tn * TN_BN, k0 = tk * TN_BK;
        const int ms = mt0 * TN_BM;
        const int rows = min(grp.M, mt1 * TN_BM) - ms;
        const long ldp = p.ldp, ldq = p.ldq;

        const u32x4 rP = make_rsrc_words(p.P + (long)ms * ldp, (uint32_t)((long)rows * ldp * 2));
        const u32x4 rQ = make_rsrc_words(p.Q + (long)ms * ldq, (uint32_t)((long)rows * ldq * 2));
        // DMA map: instruction (wave*4 + j) fills rows (wave*4+j)*4 .. +3; lane -> row r = lane>>4, position
        // lane&15, source chunk = position ^ (r<<2)
        uint32_t vP[4], vQ[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 4 + s_r;
            vP[j] = (uint32_t)(row * ldp * 2 + (n0 + s_chunk * 8) * 2);
            vQ[j] = (uint32_t)(row * ldq * 2 + (k0 + s_chunk * 8) * 2);
        }
        const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem + wave * 4096;
        auto stage = [&](int st, int mt) {
            const uint32_t dp = lds0 + st * T2_STAGE_BYTES;
            const uint32_t dq = dp + T2_TILE_BYTES;
            const uint32_t mbP = (uint32_t)((long)mt * TN_BM * ldp * 2);
            const uint32_t mbQ = (uint32_t)((long)mt * TN_BM * ldq * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dma16_bounds(rP, dp + j * 1024, vP[j] + mbP);
                dma16_bounds(rQ, dq + j * 1024, vQ[j] + mbQ);
            }
        };

        f32x16 acc[2][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f;
        }

        const int nmt = mt1 - mt0;
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int mt = 0; mt < nmt; ++mt) {
            const int cur = mt & 1;
            if (mt + 1 < nmt) stage(cur ^ 1, mt + 1);
            const char* st = smem + cur * T2_STAGE_BYTES;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 pf[2], qf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    pf[t] = tr_frag2(st + offP[t] + s * 16 * 256);
                    qf[t] = tr_frag2(st + offQ[t] + s * 16 * 256);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[nt], qf[kt], acc[nt][kt], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next stage landed (this wave's DMA)
            __syncthreads();
        }

        const int ek = k0 + wk * 64 + (lane & 31);
        const int en = n0 + wn * 64 + 4 * (lane >> 5);
        // (a sole writer -- splits == 1 -- could add by plain read-modify-write; measured SLOWER than the no-return
        // float atomics: the 64 dependent cold reads per lane are a latency tail, the atomics are fire-and-forget)
        constexpr bool sole = false;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int k = ek + kt * 32;
                if (k >= p.Kk) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = en + nt * 32 + (r & 3) + 8 * (r >> 2);
                    if (n < p.Nn) {
                        float* o = p.out + (long)n * p.ldo + k;
                        if (sole) *o += acc[nt][kt][r];
                        else atomicAdd(o, acc[nt][kt][r]);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel v3: 256(n) x 256(k) output tile, 16 waves (4 x 4, each 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 as in v2), ONE
// workgroup per CU, LDS ring of 48-row stages x 3 slots (or 32-row stages x 4 slots).  rocprofv3 SQ counters of v2 in the step
// (profiles/r02_pmc_sq_v1.txt): 51 % of the wave-cycles parked on s_waitcnt / barriers, matrix pipe busy 29 % -- the
// 128x128 tiles pull 2.8 GB through L2 per launch with at most 64 KB in flight per CU, i.e. the loop waits for operand
// delivery.  A 256x256 tile halves the operand bytes per FLOP.  One workgroup per CU has no second workgroup to cover
// its stalls, so (first version: two 64-row stages, 823-974 TF/s on cache-warm operands but 440 TF/s in the step, where
// the stashed activations come from HBM) the ring keeps three stages = 96 KB in flight behind the stage being
// multiplied: a DMA has three stages of matrix work to land.  The LDS images, the DMA map, the swizzle and the transposed
// fragment reads are v2's (a stage = four [32][128] images: P0 P1 Q0 Q1).
// Needs Nn, Kk multiples of 256 (every ViT-B/32 / ViT-L/14 block weight is); other shapes stay on v2.
// ------------------------------------------------------------------------------------------
// ROWS contraction rows per stage (32: four-stage ring, 128 KiB; 48: three-stage ring, 144 KiB: fewer barriers per FLOP,
// one stage less in flight), STAGES ring slots.  A stage = four [ROWS][128] images P0 P1 Q0 Q1.
template <int T3_ROWS, int T3_STAGES>
__global__ __launch_bounds__(1024, 4) void gemm_tn3_kernel(TNGroup grp) {
    constexpr int T3_IMG_BYTES = T3_ROWS * 256;
    constexpr int T3_STAGE_BYTES = 4 * T3_IMG_BYTES;
    constexpr int KSTEPS = T3_ROWS / 16;                  // 16-row MFMA k-steps per stage
    constexpr int DPW = T3_ROWS / 16;                     // DMA instructions per wave and stage (ROWS*4 images / 4 rows / 16 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;

    const int m_tiles = (grp.M + TN_BM - 1) / TN_BM;
    const int mps = grp.m_per_split / TN_BM;
    const int s_r = lane >> 4;
    const int s_chunk = (lane & 15) ^ (s_r << 2);
    const int g = lane >> 4, li = lane & 15;
    const int t_row = 8 * (g >> 1) + (li >> 2);
    const int t_sw = (li >> 2) << 2;
    const int t_cb = 2 * (g & 1) + ((li & 3) >> 1);
    int offP[2], offQ[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        offP[t] = (wn >> 1) * T3_IMG_BYTES + t_row * 256 + ((((wn & 1) * 8 + t * 4 + t_cb) ^ t_sw) << 4) + (li & 1) * 8;
        offQ[t] = (2 + (wk >> 1)) * T3_IMG_BYTES + t_row * 256 + ((((wk & 1) * 8 + t * 4 + t_cb) ^ t_sw) << 4) + (li & 1) * 8;
    }

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    int tile = lb / grp.splits;
    const int mt0 = (lb - tile * grp.splits) * mps;
    const int mt1 = min(m_tiles, mt0 + mps);
    if (mt0 >= mt1) return;  // uniform per block
    int pi = 0;
    while (pi + 1 < grp.count && tile >= grp.tile_end[pi]) ++pi;   // block-uniform
    if (pi > 0) tile -= grp.tile_end[pi - 1];
    const TNArgs& p = grp.prob[pi];
    int tn, tk;
    {
        constexpr int BNB = 2, BKB = 2;      // 2x2 blocks of 256^2 tiles = the panel sharing of v2's 4x4 blocks
        const int br = tile / (BNB * p.tiles_k);
        const int r0 = br * BNB, rows_b = min(BNB, p.tiles_n - r0);
        const int t1 = tile - br * BNB * p.tiles_k;
        const int bc = t1 / (rows_b * BKB);
        const int c0 = bc * BKB, cols_b = min(BKB, p.tiles_k - c0);
        const int t2 = t1 - bc * rows_b * BKB;
        tn = r0 + t2 / cols_b;
        tk = c0 + t2 % cols_b;
    }
    const int n0 = tn * 256, k0 = tk * 256;
    const int ms = mt0 * TN_BM;
    const int rows = min(grp.M, mt1 * TN_BM) - ms;

    // DMA: wave w fills image w>>2 (P0 P1 Q0 Q1) of a stage, instructions (w&3)*DPW .. of its 4*DPW (4 rows x 256 B each):
    // every wave issues exactly DPW DMA instructions per stage (the unit of the vmcnt bookkeeping below)
    const int img = wave >> 2;
    const bool isP = img < 2;
    const long ld = isP ? p.ldp : p.ldq;
    const bf16_t* src = isP ? p.P : p.Q;
    const int col0 = (isP ? n0 : k0) + 128 * (img & 1);
    const u32x4 rsrc = make_rsrc_words(src + (long)ms * ld, (uint32_t)((long)rows * ld * 2));
    uint32_t vo[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int row = ((wave & 3) * DPW + j) * 4 + s_r;
        vo[j] = (uint32_t)(row * ld * 2 + (col0 + s_chunk * 8) * 2);
    }
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem + img * T3_IMG_BYTES + (wave & 3) * DPW * 1024;
    auto stage = [&](int slot, int st) {
        const uint32_t d = lds0 + slot * T3_STAGE_BYTES;
        const uint32_t mb = (uint32_t)((long)st * T3_ROWS * ld * 2);
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma16_bounds(rsrc, d + j * 1024, vo[j] + mb);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f;
    }

    // ring of T3_STAGES stages, prefetch distance 3: a DMA has three stages of matrix work (~3 x 1024 cycles per SIMD)
    // to land, and 96 KB are in flight per CU.  Iteration i: wait until this wave's part of stage i has landed (its two
    // later stages may still be in flight: vmcnt(4)), barrier (= every wave's part has landed AND every wave has finished
    // reading stage i-1), refill the slot of stage i-1 with stage i+3, multiply stage i.
    const int nst = (rows + T3_ROWS - 1) / T3_ROWS;
    const int D = min(grp.depth, T3_STAGES - 1);           // stages issued ahead of the one being multiplied
    stage(0, 0);
    if (nst > 1 && D > 1) stage(1, 1);
    if (nst > 2 && D > 2) stage(2, 2);
    int slot = 0;
    for (int i = 0; i < nst; ++i) {
        const int later = min(nst - 1 - i, D - 1);         // stages i+1 .. issued and not yet needed
        if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + D < nst) {
            int sl = slot + D;
            if (sl >= T3_STAGES) sl -= T3_STAGES;
            stage(sl, i + D);
        }
        const char* st = smem + slot * T3_STAGE_BYTES;
        slot = slot + 1 == T3_STAGES ? 0 : slot + 1;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            bf16x8 pf[2], qf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                pf[t] = tr_frag2(st + offP[t] + s * 16 * 256);
                qf[t] = tr_frag2(st + offQ[t] + s * 16 * 256);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[nt], qf[kt], acc[nt][kt], 0, 0, 0);
        }
    }

    const int ek = k0 + wk * 64 + (lane & 31);
    const int en = n0 + wn * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int k = ek + kt * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = en + nt * 32 + (r & 3) + 8 * (r >> 2);
                if (grp.overwrite) __builtin_nontemporal_store(acc[nt][kt][r], p.out + (long)n * p.ldo + k);
                else atomicAdd(p.out + (long)n * p.ldo + k, acc[nt][kt][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------
// TN kernel v3 with LOADER WAVES: the same tile, ring and LDS images; 8 compute waves of 128 (n) x 64 (k) outputs
// (4 x 2 v_mfma_f32_32x32x16_bf16: a quarter fewer fragment reads per FLOP than 64x64) + 4 loader waves, one per image,
// that issue and wait for every LDS-DMA instruction (a workgroup is limited to 16 waves, so 16 compute waves leave
// no room for loaders).  The compute waves only wait at the stage barrier.
// ------------------------------------------------------------------------------------------
template <int T3_ROWS, int T3_STAGES>
__global__ __launch_bounds__(768, 3) void gemm_tn3lw_kernel(TNGroup grp) {
    constexpr int T3_IMG_BYTES = T3_ROWS * 256;
    constexpr int T3_STAGE_BYTES = 4 * T3_IMG_BYTES;
    constexpr int KSTEPS = T3_ROWS / 16;                  // 16-row MFMA k-steps per stage
    constexpr int DPL = T3_ROWS / 4;                      // DMA instructions per LOADER wave and stage: one image, 4 rows x 256 B each
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (wave >> 2) & 1, wk = wave & 3;        // compute waves 0-7: 128 (n) x 64 (k) each; waves 8-11 load

    const int m_tiles = (grp.M + TN_BM - 1) / TN_BM;
    const int mps = grp.m_per_split / TN_BM;
    const int s_r = lane >> 4;
    const int s_chunk = (lane & 15) ^ (s_r << 2);
    const int g = lane >> 4, li = lane & 15;
    const int t_row = 8 * (g >> 1) + (li >> 2);
    const int t_sw = (li >> 2) << 2;
    const int t_cb = 2 * (g & 1) + ((li & 3) >> 1);
    int offP[4], offQ[2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        offP[t] = wn * T3_IMG_BYTES + t_row * 256 + (((t * 4 + t_cb) ^ t_sw) << 4) + (li & 1) * 8;
#pragma unroll
    for (int t = 0; t < 2; ++t)
        offQ[t] = (2 + (wk >> 1)) * T3_IMG_BYTES + t_row * 256 + ((((wk & 1) * 8 + t * 4 + t_cb) ^ t_sw) << 4) + (li & 1) * 8;

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    int tile = lb / grp.splits;
    const int mt0 = (lb - tile * grp.splits) * mps;
    const int mt1 = min(m_tiles, mt0 + mps);
    if (mt0 >= mt1) return;  // uniform per block
    int pi = 0;
    while (pi + 1 < grp.count && tile >= grp.tile_end[pi]) ++pi;   // block-uniform
    if (pi > 0) tile -= grp.tile_end[pi - 1];
    const TNArgs& p = grp.prob[pi];
    int tn, tk;
    {
        constexpr int BNB = 2, BKB = 2;      // 2x2 blocks of 256^2 tiles = the panel sharing of v2's 4x4 blocks
        const int br = tile / (BNB * p.tiles_k);
        const int r0 = br * BNB, rows_b = min(BNB, p.tiles_n - r0);
        const int t1 = tile - br * BNB * p.tiles_k;
        const int bc = t1 / (rows_b * BKB);
        const int c0 = bc * BKB, cols_b = min(BKB, p.tiles_k - c0);
        const int t2 = t1 - bc * rows_b * BKB;
        tn = r0 + t2 / cols_b;
        tk = c0 + t2 % cols_b;
    }
    const int n0 = tn * 256, k0 = tk * 256;
    const int ms = mt0 * TN_BM;
    const int rows = min(grp.M, mt1 * TN_BM) - ms;

    const int nst = (rows + T3_ROWS - 1) / T3_ROWS;
    const int D = min(grp.depth, T3_STAGES - 1);           // stages issued ahead of the one being multiplied
    if (wave >= 8) {
        // ---- loader wave: image wave-8 (P0 P1 Q0 Q1) of every stage, DPL instructions of 4 rows x 256 B
        const int img = wave - 8;
        const bool isP = img < 2;
        const long ld = isP ? p.ldp : p.ldq;
        const bf16_t* src = isP ? p.P : p.Q;
        const int col0 = (isP ? n0 : k0) + 128 * (img & 1);
        const u32x4 rsrc = make_rsrc_words(src + (long)ms * ld, (uint32_t)((long)rows * ld * 2));
        const uint32_t vo0 = (uint32_t)(s_r * ld * 2 + (col0 + s_chunk * 8) * 2);
        const uint32_t vstep = (uint32_t)(4 * ld * 2);
        const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem + img * T3_IMG_BYTES;
        auto stage = [&](int slot, int st) {
            const uint32_t d = lds0 + slot * T3_STAGE_BYTES;
            const uint32_t mb = (uint32_t)((long)st * T3_ROWS * ld * 2);
#pragma unroll
            for (int j = 0; j < DPL; ++j) {
#if CE_DIAG_TN3 == 2      // ablation build (tools/diag/tn3_ablate.sh): no operand traffic
                if (p.ldo < 0)
#endif
                dma16_bounds(rsrc, d + j * 1024, vo0 + j * vstep + mb);
            }
        };
        stage(0, 0);
        if (nst > 1 && D > 1) stage(1, 1);
        if (nst > 2 && D > 2) stage(2, 2);
        int slot = 0;
        for (int i = 0; i < nst; ++i) {
            const int later = min(nst - 1 - i, D - 1);         // stages i+1 .. issued and not yet needed
            if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPL) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // stage i published; stage i-1 has been read by every compute wave
            if (i + D < nst) {
                int sl = slot + D;
                if (sl >= T3_STAGES) sl -= T3_STAGES;
                stage(sl, i + D);
            }
            slot = slot + 1 == T3_STAGES ? 0 : slot + 1;
        }
        return;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[nt][0][r] = 0.f; acc[nt][1][r] = 0.f;
        }
    int slot = 0;
    for (int i = 0; i < nst; ++i) {
        __builtin_amdgcn_s_barrier();
        const char* st = smem + slot * T3_STAGE_BYTES;
        slot = slot + 1 == T3_STAGES ? 0 : slot + 1;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            bf16x8 qf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) qf[t] = tr_frag2(st + offQ[t] + s * 16 * 256);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 pf = tr_frag2(st + offP[nt] + s * 16 * 256);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, qf[kt], acc[nt][kt], 0, 0, 0);
            }
        }
    }

    const int ek = k0 + wk * 64 + (lane & 31);
    const int en = n0 + wn * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int k = ek + kt * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = en + nt * 32 + (r & 3) + 8 * (r >> 2);
#if CE_DIAG_TN3 == 1      // ablation build: no epilogue traffic (the condition keeps the MFMAs alive)
                if (acc[nt][kt][r] == 12345.678f)
#endif
                if (grp.overwrite) __builtin_nontemporal_store(acc[nt][kt][r], p.out + (long)n * p.ldo + k);   // sole writer of a first-touch gradient
                else atomicAdd(p.out + (long)n * p.ldo + k, acc[nt][kt][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------
// probes: raw fragment in / raw accumulator out, so the host can check the lane maps
// ------------------------------------------------------------------------------------------
__global__ void probe_mfma16_kernel(const bf16x8* a, const bf16x8* b, f32x4* out) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    out[threadIdx.x] = c;
}
__global__ void probe_mfma32_kernel(const bf16x8* a, const bf16x8* b, f32x16* out) {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    out[threadIdx.x] = c;
}
__global__ void probe_tr16_kernel(const uint16_t* image, int n_elems, const int* byte_off, s16x4* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* l = reinterpret_cast<uint16_t*>(smem);
    for (int i = threadIdx.x; i < n_elems; i += 64) l[i] = image[i];
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + byte_off[threadIdx.x]));
}

int nt_variant() {   // CE_GEMM_NT=128|256 forces a tile; default: pick per shape
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CE_GEMM_NT");
        v = e ? atoi(e) : 0;
    }
    return v;
}

int g_last_tall = 0, g_last_ts = 0;   // two-height plan of the latest persistent launch (ce_gemm_nt_last_plan)
int g_force_chunk = -2;      // ce_gemm_nt_tune(1000 + ...): walk of the persistent kernel (-2: CE_NT_CHUNK / default)
int g_force_tm = -1;
int force_tm() {   // CE_GEMM_TM / ce_gemm_nt_tune(): 3..8 = tile height (x32 rows) of the 256-column kernel, 32 = the
                   // 160x256x32 two-workgroup kernel, 104 = 160x128 four-wave tile, 160 = three-stage ring; 0 = auto
    if (g_force_tm < 0) {
        const char* e = getenv("CE_GEMM_TM");
        g_force_tm = e ? atoi(e) : 0;
    }
    return g_force_tm;
}

// Tile-variant cost model, fitted to tools/tune_nt.py sweeps (M 8k..20k, both towers' N/K; unit = 0.137 us at
// K = 512, scales with K): one round of 32*TM-row tiles on the 256 CUs costs 28 + 10*TM (the K loop is
// LDS-read bound: a fixed share for the 256-column B fragments plus TM A fragments per k-step); the
// 160x256x32 kernel keeps two workgroups per CU: a co-resident pair costs 146, a lone one 78.
// Tile queues of the persistent kernel's DYNAMIC tile list (NTArgs.tile_queue): one zero-initialised counter per launch, taken from a
// ring per stream; the ring is re-zeroed on its stream when it wraps, behind every launch that used it.  CE_NT_DYNAMIC = 1 (or
// ce_gemm_set_dynamic_tiles) turns the dynamic list on for persistent launches with >= 3 K iterations on the launch-wide walk.
struct TileQueueRing { unsigned int* dev = nullptr; int next = 0; };
constexpr int TQ_RING = 1024;
std::mutex g_tq_mu;
std::map<hipStream_t, TileQueueRing> g_tq;
int g_dynamic = getenv("CE_NT_DYNAMIC") ? atoi(getenv("CE_NT_DYNAMIC")) : 0;
unsigned int* next_tile_queue(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_tq_mu);
    TileQueueRing& r = g_tq[s];
    if (!r.dev) {
        if (hipMalloc(&r.dev, TQ_RING * sizeof(unsigned int)) != hipSuccess) return nullptr;
        r.next = TQ_RING;
    }
    if (r.next >= TQ_RING) {
        if (hipMemsetAsync(r.dev, 0, TQ_RING * sizeof(unsigned int), s) != hipSuccess) return nullptr;
        r.next = 0;
    }
    return r.dev + r.next++;
}

// CU budget of the NT launch policies (ce_gemm_set_cu_budget / CE_GEMM_CUS, default 256 = the whole chip).  Every NT kernel here
// puts ONE 156 KiB workgroup on a CU and sizes its grid to fill the chip exactly once (one-round launches: 226-240 tiles;
// persistent launches: 256 workgroups), so a single CU held by another stream's kernel -- an RCCL channel during a gradient
// all-reduce -- leaves one workgroup without a home until a whole tile list has finished: measured with a 1-CU "hog"
// (ce_cu_hog) every such launch takes 1.6-1.75x as long.  A budget below 256 sizes the one-round and the persistent
// grids for that many CUs, so that the rest may be taken.
int g_cus = getenv("CE_GEMM_CUS") ? atoi(getenv("CE_GEMM_CUS")) : 256;
// epilogues for which the persistent kernel is also built with two tile heights (launch_nt)
constexpr bool nt_two_heights(int epi) {
    return epi == CE_EPI_BIAS_GELU || epi == CE_EPI_GELUGRAD_BF16 || epi == CE_EPI_BIAS_BF16 || epi == CE_EPI_BF16 ||
           epi == CE_EPI_BIAS_RESID_F16 || epi == CE_EPI_BIAS_RESID_F32;
}
inline int cu_budget() { return g_cus >= 32 && g_cus <= 256 ? g_cus : 256; }
inline long nt256_cost(long tiles, int tm) { return ((tiles + cu_budget() - 1) / cu_budget()) * (28 + 10 * tm); }
// Two tile heights for one persistent launch (gemm_nt160p_kernel<EPI, TM, F8, TS>): n_tall row panels of 32 TM rows, the rest in panels
// of 32 TS, chosen so that the longest per-workgroup list (tall tiles first, round-robin over G workgroups) is shortest under the
// launch policy's cost model (32 tm + 48 per tile).  `uniform` = the best single height's cost; true when a split beats it by >=
// min_gain per cent.
inline bool two_height_plan(long M, long tn, long G, int TM, long uniform, int min_gain, long& b_tall, long& b_short, int& b_ts) {
    const long ct = 32 * TM + 48;
    auto span = [&](long n_tall, int ts, long n_short) {      // cost of the longest list
        const long T = n_tall * tn, S = n_short * tn, q = T / G, r = T % G;
        auto shorts = [&](long d) { return d < S ? (S - 1 - d) / G + 1 : 0; };   // short tiles of the workgroup d places behind r
        const long cs = 32 * ts + 48;
        long worst = q * ct + shorts(0) * cs;                  // workgroup r: q tall tiles, the most short ones
        if (r > 0) worst = std::max(worst, (q + 1) * ct + shorts(G - r) * cs);   // workgroup 0: q + 1 tall
        return worst;
    };
    long best = uniform;
    b_tall = -1; b_short = 0; b_ts = 0;
    for (int ts = TM - 1; ts >= 1; --ts)
        for (long n_tall = M / (32 * TM); n_tall >= 1; --n_tall) {
            const long rest = M - n_tall * 32 * TM;
            if (rest <= 0) continue;
            const long n_short = (rest + 32 * ts - 1) / (32 * ts);
            const long c = span(n_tall, ts, n_short);
            if (c < best || (c == best && b_tall < 0)) { best = c; b_tall = n_tall; b_short = n_short; b_ts = ts; }
        }
    return b_tall > 0 && best * 100 <= uniform * (100 - min_gain);
}
inline long nt32_cost(long tiles) {
    const long n = (tiles + 255) / 256;
    return (n / 2) * 146 + (n % 2) * 78;
}

template <int EPI, int TM>
void launch_nt128(NTArgs& a, hipStream_t stream) {      // (32 TM) x 128 tile, 4 waves, two workgroups per CU
    static std::once_flag attr;
    std::call_once(attr, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<EPI, TM, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N2H_LDS_BYTES);
    });
    a.tiles_m = ce_div_up(a.M, 32 * TM);
    hipLaunchKernelGGL((gemm_nt256_kernel<EPI, TM, 2>), dim3(a.tiles_m * a.tiles_n), dim3(256), N2H_LDS_BYTES, stream, a);
}

template <int EPI, int TM>
void launch_nt256(NTArgs& a, hipStream_t stream) {
    static std::once_flag attr;       // forward runs on the caller's thread, backward on autograd's
    std::call_once(attr, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<EPI, TM, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N2_LDS_BYTES);
    });
    a.tiles_m = ce_div_up(a.M, 32 * TM);
    hipLaunchKernelGGL((gemm_nt256_kernel<EPI, TM, 4>), dim3(a.tiles_m * a.tiles_n), dim3(512), N2_LDS_BYTES, stream, a);
}

template <int EPI>
int launch_nt(NTArgs a, hipStream_t stream) {
    static std::once_flag attr_set;
    std::call_once(attr_set, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<EPI, 5, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N2H_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt32_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N3_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160lw_kernel<EPI, 5>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160lw_kernel<EPI, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160lw_kernel<EPI, 3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 5>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
        if constexpr (nt_two_heights(EPI)) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 5, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 5, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 5, 0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 5, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
        }
    });
    const double out_b = (EPI == CE_EPI_F32 || EPI == CE_EPI_BIAS_F32) ? 4.0 : (EPI == CE_EPI_BIAS_RESID_F32 ? 8.0 : (EPI == CE_EPI_BIAS_GELU || EPI == CE_EPI_GELUGRAD_BF16 || EPI == CE_EPI_BIAS_RESID_F16 ? 4.0 : 2.0));
    CeProfScope prof(CE_PROF_GEMM_NT0 + CE_PROF_NT_FAMILIES * EPI, 2.0 * a.M * a.N * a.K, 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + out_b * a.M * a.N, stream);
    const int force = nt_variant();
    const bool can256 = (a.K % N2_BK == 0) && a.N % 8 == 0 && a.ldo % 8 == 0 && a.ldo2 % 8 == 0 && a.ldaux % 8 == 0;
    const bool want256 = force == 256 || (force == 0 && a.M >= 1024 && a.N >= 256);
    if (can256 && want256) {
        a.tiles_n = ce_div_up(a.N, N2_BN);
        // pick the tile height by wave quantisation: cost ~ (rounds over 256 CUs) x (cost of one tile-round)
        const int f = force_tm();
        const long t5 = (long)ce_div_up(a.M, 160) * a.tiles_n;
        int best = 8;
        long best_cost = -1;
        if (f >= 3 && f <= 8) {
            best = f;
        } else {
            for (int tm = 8; tm >= 3; --tm) {
                const long cost = nt256_cost((long)ce_div_up(a.M, 32 * tm) * a.tiles_n, tm);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = tm; }
            }
        }
        const bool use32 = f == 32 || (f == 0 && a.K % N3_BK == 0 && nt32_cost(t5) < best_cost);
        // Where the cost model picks a 256-column tile with one workgroup per CU and the tiles fit one resident round
        // (N = width GEMMs: 240 tiles of 160x256), use the loader-wave kernel (gemm_nt160lw_kernel: -2.7 % on the step
        // against the 160x128 pair below, which was itself 0.5-2 % ahead of the plain 8-wave tile because a workgroup of
        // the OTHER tower's GEMM could share the CU).  CE_NT_POLICY: bit 0 = treat single-round launches specially
        // (default), bit 4 = with the loader-wave kernel (default; without it the 160x128 four-wave pair), bits 5, 6 = the
        // persistent loader-wave kernel for multi-round launches (below; default), bit 1 = also
        // instead of the two-workgroup 160x256x32 kernel (slower), bit 2 = pick the 160x128 family's tile height 96..160 by
        // rounds over the 512 slots (slower in the step), bit 3 = also for multi-round launches (noise).
        static int policy = getenv("CE_NT_POLICY") ? atoi(getenv("CE_NT_POLICY")) : 113;
        const long half_tiles = (long)ce_div_up(a.M, 160) * ce_div_up(a.N, 128);
        const bool half = f == 104 || (f >= 203 && f <= 205) ||
                          (f == 0 && (half_tiles <= 2 * cu_budget() || (policy & 8)) && ((policy & 1) && !use32 || (policy & 2) && use32));
        // the loader-wave kernels address their epilogue operands with 32-bit buffer offsets (EpiBuf): every one must span < 2 GiB
        const auto span = [&](long ld, long esz) { return (long)a.M * ld * esz; };
        const bool fits31 = span(a.ldo, epi_out_bytes(EPI)) < (1l << 31) && span(a.ldo2, 2) < (1l << 31) && span(a.ldaux, 2) < (1l << 31) &&
                            span(a.ldr, 4) < (1l << 31);
        const bool lw = fits31 && (f == 161 || (half && f == 0 && (policy & 16)));       // one 160x256 loader-wave workgroup per CU
        prof.retag(CE_PROF_GEMM_NT0 + CE_PROF_NT_FAMILIES * EPI + (lw ? 5 : (half || f == 104 ? 1 : (use32 ? 3 : 2))));
        // multi-round launches: the persistent loader-wave kernel.  Bit 5 = for the light epilogues (qkv forward: 726 ->
        // 810 TF/s), bit 6 = also for the GELU epilogues (as kernels about equal to the two-workgroup 160x256x32 kernel
        // since their epilogues lost the division and the backward's transcendentals).  Both on by default: B = 256 step
        // 14.15 -> 14.00 (bit 5) -> 13.96 ms (bits 5 + 6), config 4 at B = 64 32.3 -> 31.9 -> 31.5 ms.
        constexpr bool light_epi = EPI == CE_EPI_BF16 || EPI == CE_EPI_F32 || EPI == CE_EPI_BIAS_BF16 || EPI == CE_EPI_BIAS_F32;
        const bool pers = !lw && fits31 && a.K >= 2 * N4_BK && ((f >= 163 && f <= 165) || f == 162 ||
                                                     (f == 0 && !half && ((policy & 32) && light_epi || (policy & 64))));
        // (A two-group "ping-pong" persistent kernel -- 128x256x64 tiles, one group of four waves multiplying while the other
        //  issues the ring's LDS-DMAs and works through the previous tile's epilogue -- was built, parity-tested and measured in
        //  round 3: BIAS_GELU 63.1 vs 55.4 us per launch, qkv 41.0 vs 35.3, step 12.77 vs 12.40 ms, slower in both versions; it
        //  was removed in round 4.  DESIGN 6b keeps the post-mortem.)
        if (pers) {
            // tile height by the longest per-CU tile list: rows of tile work + ~48 rows' worth of epilogue per tile
            int ptm = 5;
            if (f >= 163 && f <= 165) ptm = f - 160;
            else {
                long bc = -1;
                for (int tm = 5; tm >= 3; --tm) {
                    const long tiles = (long)ce_div_up(a.M, 32 * tm) * a.tiles_n;
                    const long cost = ((tiles + cu_budget() - 1) / cu_budget()) * (32 * tm + 48);
                    if (bc < 0 || cost < bc) { bc = cost; ptm = tm; }
                }
            }
            a.tiles_m = ce_div_up(a.M, 32 * ptm);
            static const int strip = getenv("CE_NT_STRIP") ? atoi(getenv("CE_NT_STRIP")) : 0;   // column strips: measured no better (0 / 3 / 6 equal, 4 slower)
            a.tile_strip = strip;
            // XCD-owned walk (persist_walk): CE_NT_CHUNK = 0 (default) the launch-wide walk, -1 = chunks of tiles_m / 8 row
            // panels, n > 0 = chunks of n.  OFF: with sc1 output stores it takes the c_fc GEMM's fetch from 166.8 to 68.3 MB
            // (algorithmic 24.4; the floor of any 8-way partition is 57) and qkv's from 105.5 to 53.5 MB
            // (profiles/r03_pmc_fetch_xcd_walk.txt) and the kernels do not get faster: BIAS_GELU 1.30 -> 1.32 ms/step, qkv
            // 0.945 -> 0.96; with plain stores 1.37 -> 1.46 and 0.92 -> 1.00.  These launches are not bound by operand re-fetch.
            static const int chunk = getenv("CE_NT_CHUNK") ? atoi(getenv("CE_NT_CHUNK")) : 0;
            const int ch = g_force_chunk > -2 ? g_force_chunk : chunk;
            a.tile_chunk = strip > 0 ? 0 : (ch < 0 ? (a.tiles_m >= 8 ? a.tiles_m / 8 : 1) : ch);   // floor: a chunk never spans three XCDs
            const long tiles = (long)a.tiles_m * a.tiles_n;
            // CE_NT_PGRID workgroups walk the tile list (default 256 = one per CU).  More, shorter lists = finer scheduling
            // granularity when some CUs are held by another stream's kernels (or by RCCL): a workgroup that starts late then
            // delays the launch by a shorter list.
            static const int pgrid_env = getenv("CE_NT_PGRID") ? atoi(getenv("CE_NT_PGRID")) : 0;
            const int pgrid = pgrid_env > 0 ? pgrid_env : cu_budget();
            const dim3 grid((unsigned)(tiles < pgrid ? tiles : pgrid)), block(64 * (8 + N4_LOADERS));
            // dynamic tile list (off by default): only where a workgroup walks more than one tile, with >= 3 K iterations
            // (the fetched id is published by the barrier of iteration 1 and needed from iteration nk - 2) on the launch-wide walk
            a.tile_queue = (g_dynamic && tiles > (long)grid.x && a.K >= 3 * N4_BK && a.tile_chunk == 0) ? next_tile_queue(stream) : nullptr;
            prof.retag(CE_PROF_GEMM_NT0 + CE_PROF_NT_FAMILIES * EPI + 6);
            // TWO TILE HEIGHTS (gemm_nt160p_kernel<EPI, 5, 0, TS>): n_tall row panels of 160 rows, the rest in panels of 32 TS, chosen so
            // that the longest per-workgroup list (tall tiles first, round-robin over the grid) is shortest under the same cost
            // model; taken when it beats the best single height by >= 3 %.  12800 x 3072: 960 tiles of 160 rows = 3.75 rounds -> 3
            // rounds of 160 + one of 128.  CE_NT_MIXED=0 switches it off.
            g_last_tall = g_last_ts = 0;
            constexpr bool mixed_epi = nt_two_heights(EPI);
            if constexpr (mixed_epi) {
                static const int mixed = getenv("CE_NT_MIXED") ? atoi(getenv("CE_NT_MIXED")) : 1;
                if (mixed && f == 0 && strip == 0 && a.tile_chunk == 0 && a.M >= 320) {
                    const long G = pgrid, tn = a.tiles_n;
                    long uniform = -1;
                    for (int tm = 5; tm >= 3; --tm) {
                        const long tiles_u = (long)ce_div_up(a.M, 32 * tm) * tn;
                        const long c = ((tiles_u + G - 1) / G) * (32 * tm + 48);
                        if (uniform < 0 || c < uniform) uniform = c;
                    }
                    long b_tall, b_short;
                    int b_ts;
                    static const int min_gain = getenv("CE_NT_MIXED_GAIN") ? atoi(getenv("CE_NT_MIXED_GAIN")) : 3;   // per cent
                    if (two_height_plan(a.M, tn, G, 5, uniform, min_gain, b_tall, b_short, b_ts)) {
                        a.tall_panels = (int)b_tall;
                        g_last_tall = (int)b_tall; g_last_ts = b_ts;
                        a.tiles_m = (int)(b_tall + b_short);
                        const long tiles2 = (long)a.tiles_m * a.tiles_n;
                        const dim3 grid2((unsigned)(tiles2 < pgrid ? tiles2 : pgrid));
                        a.tile_queue = (g_dynamic && tiles2 > (long)grid2.x && a.K >= 3 * N4_BK) ? next_tile_queue(stream) : nullptr;
                        switch (b_ts) {
                            case 1: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 5, 0, 1>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                            case 2: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 5, 0, 2>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                            case 3: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 5, 0, 3>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                            default: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 5, 0, 4>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                        }
                        CE_LAUNCH_CHECK();
                        return 0;
                    }
                }
            }
            switch (ptm) {
                case 3: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 3>), grid, block, N4P_LDS_BYTES, stream, a); break;
                case 4: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 4>), grid, block, N4P_LDS_BYTES, stream, a); break;
                default: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 5>), grid, block, N4P_LDS_BYTES, stream, a); break;
            }
        } else if (lw) {
            // shortest tile whose launch still fits one round of the 256 CUs (the text tower's N = 512 has two tile columns)
            int ltm = 5;
            static const int lw_tm = getenv("CE_NT_LW_TM") ? atoi(getenv("CE_NT_LW_TM")) : 0;
            if (lw_tm >= 3 && lw_tm <= 5) ltm = lw_tm;
            else
                for (int tm = 4; tm >= 3; --tm)
                    if ((long)ce_div_up(a.M, 32 * tm) * a.tiles_n <= cu_budget()) ltm = tm;
            a.tiles_m = ce_div_up(a.M, 32 * ltm);
            const dim3 grid(a.tiles_m * a.tiles_n), block(64 * (8 + N4_LOADERS));
            switch (ltm) {
                case 3: hipLaunchKernelGGL((gemm_nt160lw_kernel<EPI, 3>), grid, block, N4_LDS_BYTES, stream, a); break;
                case 4: hipLaunchKernelGGL((gemm_nt160lw_kernel<EPI, 4>), grid, block, N4_LDS_BYTES, stream, a); break;
                default: hipLaunchKernelGGL((gemm_nt160lw_kernel<EPI, 5>), grid, block, N4_LDS_BYTES, stream, a); break;
            }
        } else if (half) {
            a.tiles_n = ce_div_up(a.N, 128);
            int htm = 5;
            if (f >= 203 && f <= 205) htm = f - 200;
            else if (f == 0 && (policy & 4)) {        // tile height by rounds over the 512 slots
                long bc = -1;
                for (int tm = 5; tm >= 3; --tm) {
                    const long tiles = (long)ce_div_up(a.M, 32 * tm) * a.tiles_n;
                    const long cost = ((tiles + 511) / 512) * (28 + 10 * tm);
                    if (bc < 0 || cost < bc) { bc = cost; htm = tm; }
                }
            }
            switch (htm) {
                case 3: launch_nt128<EPI, 3>(a, stream); break;
                case 4: launch_nt128<EPI, 4>(a, stream); break;
                default: launch_nt128<EPI, 5>(a, stream); break;
            }
        } else if (use32) {
            a.tiles_m = ce_div_up(a.M, N3_BM);
            hipLaunchKernelGGL(gemm_nt32_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(512), N3_LDS_BYTES, stream, a);
        } else if (f == 104) {   // 160x128, two workgroups per CU: measured equal to the 8-wave 160x256 tile, kept as an option
            a.tiles_m = ce_div_up(a.M, 160);
            a.tiles_n = ce_div_up(a.N, 128);
            hipLaunchKernelGGL((gemm_nt256_kernel<EPI, 5, 2>), dim3(a.tiles_m * a.tiles_n), dim3(256), N2H_LDS_BYTES, stream, a);
        } else if (f == 160) {   // three-stage ring: measured 6 % slower than the two-stage loop (the K loop is
                                 // LDS-bandwidth bound, not DMA-latency bound); kept as an option
            a.tiles_m = ce_div_up(a.M, 160);
            hipLaunchKernelGGL(gemm_nt160_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(512), N4_LDS_BYTES, stream, a);
        } else {
            switch (best) {
                case 3: launch_nt256<EPI, 3>(a, stream); break;
                case 4: launch_nt256<EPI, 4>(a, stream); break;
                case 5: launch_nt256<EPI, 5>(a, stream); break;
                case 6: launch_nt256<EPI, 6>(a, stream); break;
                case 7: launch_nt256<EPI, 7>(a, stream); break;
                default: launch_nt256<EPI, 8>(a, stream); break;
            }
        }
    } else if (static const int skinny = getenv("CE_NT_SKINNY") ? atoi(getenv("CE_NT_SKINNY")) : 1;
               skinny && force == 0 && a.M <= 512 && a.K % 256 == 0 && a.N % 8 == 0 && a.ldo % 8 == 0 && a.ldo2 % 8 == 0 &&
               a.ldaux % 8 == 0 && a.ldr % 4 == 0) {
        static std::once_flag sk_attr;
        std::call_once(sk_attr, [] {
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_skinny_kernel<EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES);
        });
        a.tiles_m = ce_div_up(a.M, 64);
        a.tiles_n = ce_div_up(a.N, 64);
        prof.retag(CE_PROF_GEMM_NT0 + CE_PROF_NT_FAMILIES * EPI + 7);
        hipLaunchKernelGGL(gemm_nt_skinny_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(512), SK_LDS_BYTES, stream, a);
    } else {
        hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(256), NT_LDS_BYTES, stream, a);
        if (EPI == CE_EPI_GELUGRAD_BF16 && a.out2) {   // the 128^2 kernel has no fused column sums
            CE_LAUNCH_CHECK();
            return ce_colsum_bf16(a.out, a.ldo, reinterpret_cast<float*>(a.out2), a.M, a.N, stream);
        }
    }
    CE_LAUNCH_CHECK();
    return 0;
}

// e4m3 operands (per-row scales sa / sb applied in the epilogue) on the loader-wave kernels: the tile policy of launch_nt for
// these two families -- one resident round -> gemm_nt160lw_kernel at the shortest tile that still fits the round, more ->
// the persistent gemm_nt160p_kernel.  Returns 1 when the shape is not one these kernels take (the caller falls back).
template <int EPI>
int launch_nt_f8(NTArgs a, hipStream_t stream) {
    static std::once_flag attr_set;
    std::call_once(attr_set, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160lw_kernel<EPI, 5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160lw_kernel<EPI, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160lw_kernel<EPI, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
        if constexpr (nt_two_heights(EPI)) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 4, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 4, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 4, 1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
        }
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt160p_kernel<EPI, 3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, N4P_LDS_BYTES);
    });
    if (!(a.M >= 1024 && a.N >= 256 && a.K % 128 == 0 && a.K >= 256 && a.N % 8 == 0 && a.lda % 16 == 0 && a.ldb % 16 == 0 &&
          a.ldo % 8 == 0 && a.ldo2 % 8 == 0 && a.ldaux % 8 == 0))
        return 1;
    if ((long)a.M * a.ldo * epi_out_bytes(EPI) >= (1l << 31) || (long)a.M * a.ldo2 * 2 >= (1l << 31) || (long)a.M * a.ldaux * 2 >= (1l << 31) ||
        (long)a.M * a.ldr * 4 >= (1l << 31))
        return 1;                                  // 32-bit epilogue offsets (EpiBuf)
    const double out_b = EPI == CE_EPI_BIAS_RESID_F32 ? 8.0 : (EPI == CE_EPI_BIAS_GELU || EPI == CE_EPI_GELUGRAD_BF16 || EPI == CE_EPI_BIAS_RESID_F16 ? 4.0 : 2.0);
    CeProfScope prof(CE_PROF_GEMM_NT0 + CE_PROF_NT_FAMILIES * EPI + 4, 2.0 * a.M * a.N * a.K, 1.0 * ((double)a.M * a.K + (double)a.N * a.K) + out_b * a.M * a.N, stream);
    a.tiles_n = ce_div_up(a.N, N4_BN);
    const long half_tiles = (long)ce_div_up(a.M, 160) * ce_div_up(a.N, 128);
    const dim3 block(64 * (8 + N4_LOADERS));
    if (half_tiles <= 2 * cu_budget()) {           // one resident round of 160 x 256 tiles
        int ltm = 5;
        for (int tm = 4; tm >= 3; --tm)
            if ((long)ce_div_up(a.M, 32 * tm) * a.tiles_n <= cu_budget()) ltm = tm;
        a.tiles_m = ce_div_up(a.M, 32 * ltm);
        const dim3 grid(a.tiles_m * a.tiles_n);
        switch (ltm) {
            case 3: hipLaunchKernelGGL((gemm_nt160lw_kernel<EPI, 3, 1>), grid, block, N4_LDS_BYTES, stream, a); break;
            case 4: hipLaunchKernelGGL((gemm_nt160lw_kernel<EPI, 4, 1>), grid, block, N4_LDS_BYTES, stream, a); break;
            default: hipLaunchKernelGGL((gemm_nt160lw_kernel<EPI, 5, 1>), grid, block, N4_LDS_BYTES, stream, a); break;
        }
    } else {
        int ptm = 4;                               // 160-row tiles spill in the e4m3 form (32-byte fragments)
        long bc = -1;
        for (int tm = 4; tm >= 3; --tm) {
            const long tiles = (long)ce_div_up(a.M, 32 * tm) * a.tiles_n;
            const long cost = ((tiles + cu_budget() - 1) / cu_budget()) * (32 * tm + 48);
            if (bc < 0 || cost < bc) { bc = cost; ptm = tm; }
        }
        a.tiles_m = ce_div_up(a.M, 32 * ptm);
        a.tile_strip = 0;
        a.tile_chunk = 0;
        const long tiles = (long)a.tiles_m * a.tiles_n;
        const dim3 grid((unsigned)(tiles < cu_budget() ? tiles : cu_budget()));
        g_last_tall = g_last_ts = 0;
        if constexpr (nt_two_heights(EPI)) {       // two tile heights (128-row panels + a shorter tail height), as launch_nt
            static const int mixed = getenv("CE_NT_MIXED") ? atoi(getenv("CE_NT_MIXED")) : 1;
            static const int min_gain = getenv("CE_NT_MIXED_GAIN") ? atoi(getenv("CE_NT_MIXED_GAIN")) : 3;
            long b_tall, b_short;
            int b_ts;
            if (mixed && a.M >= 256 && two_height_plan(a.M, a.tiles_n, cu_budget(), 4, bc, min_gain, b_tall, b_short, b_ts)) {
                a.tall_panels = (int)b_tall;
                g_last_tall = (int)b_tall; g_last_ts = b_ts;
                a.tiles_m = (int)(b_tall + b_short);
                const long tiles2 = (long)a.tiles_m * a.tiles_n;
                const dim3 grid2((unsigned)(tiles2 < cu_budget() ? tiles2 : cu_budget()));
                switch (b_ts) {
                    case 1: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 4, 1, 1>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                    case 2: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 4, 1, 2>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                    default: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 4, 1, 3>), grid2, block, N4P_LDS_BYTES, stream, a); break;
                }
                CE_LAUNCH_CHECK();
                return 0;
            }
        }
        switch (ptm) {
            case 3: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 3, 1>), grid, block, N4P_LDS_BYTES, stream, a); break;
            default: hipLaunchKernelGGL((gemm_nt160p_kernel<EPI, 4, 1>), grid, block, N4P_LDS_BYTES, stream, a); break;
        }
    }
    CE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// internal (gemm_fp8.hip): the e4m3 GEMM on the loader-wave kernels; 1 = shape not taken, fall back to gemm_nt8_kernel
extern "C" int ce__gemm_nt_fp8_lw(const void* A8, long lda, const float* sa, const void* B8, long ldb, const float* sb, int M, int N,
                                  int K, int epilogue, const float* bias, const void* resid, long ldr, void* out, long ldo,
                                  void* out2, long ldo2, const void* aux, long ldaux, void* stream) {
    static const int off = getenv("CE_FP8_LW") ? atoi(getenv("CE_FP8_LW")) == 0 : 0;
    if (off) return 1;
    NTArgs a;
    a.A = (const bf16_t*)A8; a.lda = lda; a.B = (const bf16_t*)B8; a.ldb = ldb;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.resid = (const float*)resid; a.ldr = ldr;
    a.out = out; a.ldo = ldo; a.out2 = (bf16_t*)out2; a.ldo2 = ldo2; a.aux = (const bf16_t*)aux; a.ldaux = ldaux;
    a.sa = sa; a.sb = sb;
    a.tiles_m = a.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case CE_EPI_BF16: return launch_nt_f8<CE_EPI_BF16>(a, s);
        case CE_EPI_BIAS_BF16: return launch_nt_f8<CE_EPI_BIAS_BF16>(a, s);
        case CE_EPI_BIAS_RESID_F32: return launch_nt_f8<CE_EPI_BIAS_RESID_F32>(a, s);
        case CE_EPI_BIAS_RESID_F16: return launch_nt_f8<CE_EPI_BIAS_RESID_F16>(a, s);
        case CE_EPI_BIAS_GELU: return launch_nt_f8<CE_EPI_BIAS_GELU>(a, s);
        case CE_EPI_GELUGRAD_BF16: return launch_nt_f8<CE_EPI_GELUGRAD_BF16>(a, s);
        default: return 1;
    }
}

extern "C" int ce_gemm_set_dynamic_tiles(int on) {
    g_dynamic = on < 0 ? (getenv("CE_NT_DYNAMIC") ? atoi(getenv("CE_NT_DYNAMIC")) : 0) : (on != 0);
    return 0;
}

extern "C" int ce_gemm_set_cu_budget(int cus) {
    CE_CHECK_ARG(cus == 0 || (cus >= 32 && cus <= 256), "ce_gemm_set_cu_budget: 32..256 CUs, or 0 for the default (CE_GEMM_CUS / 256)");
    g_cus = cus ? cus : (getenv("CE_GEMM_CUS") ? atoi(getenv("CE_GEMM_CUS")) : 256);
    return 0;
}

// two-height plan of the most recent persistent NT launch (tests): tall panels, short height in 32-row units (0, 0: one height)
extern "C" int ce_gemm_nt_last_plan(int* tall_panels, int* short_tm) {
    if (tall_panels) *tall_panels = g_last_tall;
    if (short_tm) *short_tm = g_last_ts;
    return 0;
}

extern "C" void ce_gemm_nt_tune(int variant) {
    if (variant >= 1000 && variant < 2000) g_force_chunk = variant - 1001;   // 1000: auto chunks, 1001: off, 1001 + n: n panels
    else g_force_tm = variant;
}

extern "C" int ce_gemm_nt(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int epilogue,
                          const float* bias, const void* resid, long ldr, void* out, long ldo, void* out2,
                          long ldo2, const void* aux, long ldaux, void* stream) {
    CE_CHECK_ARG(M > 0 && N > 0 && K > 0, "ce_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    CE_CHECK_ARG(K % 8 == 0 && N % 4 == 0, "ce_gemm_nt: need K%%8==0 and N%%4==0 (K=%d N=%d)", K, N);
    CE_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldo % 4 == 0, "ce_gemm_nt: lda/ldb must be multiples of 8, ldo of 4");
    CE_CHECK_ARG(lda >= K && ldb >= K && ldo >= N, "ce_gemm_nt: leading dimension smaller than the row");
    CE_CHECK_ARG(128L * lda * 2 < (1L << 32) && 128L * ldb * 2 < (1L << 32), "ce_gemm_nt: row panel exceeds 4 GiB");
    NTArgs a;
    a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.resid = (const float*)resid; a.ldr = ldr;
    a.out = out; a.ldo = ldo; a.out2 = (bf16_t*)out2; a.ldo2 = ldo2; a.aux = (const bf16_t*)aux; a.ldaux = ldaux;
    a.tiles_m = ce_div_up(M, NT_BM); a.tiles_n = ce_div_up(N, NT_BN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case CE_EPI_BF16: return launch_nt<CE_EPI_BF16>(a, s);
        case CE_EPI_F32: return launch_nt<CE_EPI_F32>(a, s);
        case CE_EPI_BIAS_BF16:
            CE_CHECK_ARG(bias, "ce_gemm_nt: bias epilogue without bias");
            return launch_nt<CE_EPI_BIAS_BF16>(a, s);
        case CE_EPI_BIAS_F32:
            CE_CHECK_ARG(bias, "ce_gemm_nt: bias epilogue without bias");
            return launch_nt<CE_EPI_BIAS_F32>(a, s);
        case CE_EPI_BIAS_RESID_F32:
            CE_CHECK_ARG(bias && resid && ldr >= N && ldr % 4 == 0, "ce_gemm_nt: residual epilogue needs bias+resid");
            return launch_nt<CE_EPI_BIAS_RESID_F32>(a, s);
        case CE_EPI_BIAS_RESID_F16:
            CE_CHECK_ARG(bias && resid && ldr >= N && ldr % 8 == 0 && ldo % 8 == 0, "ce_gemm_nt: fp16 residual epilogue needs bias+resid, ldr/ldo multiples of 8");
            return launch_nt<CE_EPI_BIAS_RESID_F16>(a, s);
        case CE_EPI_BIAS_GELU:
            CE_CHECK_ARG(bias && out2 && ldo2 >= N && ldo2 % 4 == 0, "ce_gemm_nt: gelu epilogue needs bias+out2");
            return launch_nt<CE_EPI_BIAS_GELU>(a, s);
        case CE_EPI_GELUGRAD_BF16:
            CE_CHECK_ARG(aux && ldaux >= N && ldaux % 4 == 0, "ce_gemm_nt: gelu-grad epilogue needs aux");
            return launch_nt<CE_EPI_GELUGRAD_BF16>(a, s);
        default: CE_CHECK_ARG(false, "ce_gemm_nt: unknown epilogue %d", epilogue);
    }
    return 0;
}

extern "C" int ce_gemm_tn(const void* P, long ldp, const void* Q, long ldq, int M, int Nn, int Kk, float* out,
                          long ldo, int splits, void* stream) {
    return ce_gemm_tn_bias(P, ldp, Q, ldq, M, Nn, Kk, out, ldo, nullptr, splits, stream);
}

extern "C" int ce_gemm_tn_bias(const void* P, long ldp, const void* Q, long ldq, int M, int Nn, int Kk, float* out,
                               long ldo, float* bias_grad, int splits, void* stream) {
    const void* Ps[1] = {P};
    const void* Qs[1] = {Q};
    float* outs[1] = {out};
    int rc = ce_gemm_tn_grouped(1, Ps, &ldp, Qs, &ldq, M, &Nn, &Kk, outs, &ldo, splits, stream);
    if (rc != 0) return rc;
    if (bias_grad) return ce_colsum_bf16(P, ldp, bias_grad, M, Nn, stream);
    return 0;
}

extern "C" int ce_gemm_tn_grouped(int count, const void* const* P, const long* ldp, const void* const* Q,
                                  const long* ldq, int M, const int* Nn, const int* Kk, float* const* out,
                                  const long* ldo, int splits, void* stream) {
    return ce_gemm_tn_grouped_ex(count, P, ldp, Q, ldq, M, Nn, Kk, out, ldo, splits, 0, stream);
}

extern "C" int ce_gemm_tn_grouped_ex(int count, const void* const* P, const long* ldp, const void* const* Q,
                                     const long* ldq, int M, const int* Nn, const int* Kk, float* const* out,
                                     const long* ldo, int splits, int overwrite, void* stream) {
    CE_CHECK_ARG(count >= 1 && count <= CE_TN_MAX_GROUP && M > 0, "ce_gemm_tn_grouped: 1..%d problems, M > 0", CE_TN_MAX_GROUP);
    // overwrite: out = product.  Unsplit 256x256 tiles store their accumulators; every other form (split tiles, the 128x128
    // kernels) zero-fills the outputs first and accumulates as usual.
    auto zero_outputs = [&]() -> int {
        for (int i = 0; i < count; ++i)
            if (hipMemset2DAsync(out[i], (size_t)ldo[i] * 4, 0, (size_t)Kk[i] * 4, (size_t)Nn[i], (hipStream_t)stream) != hipSuccess) {
                ce_set_error("ce_gemm_tn_grouped: zero-fill of output %d failed", i);
                return -5;
            }
        return 0;
    };
    static std::once_flag attr_set;
    static int variant = 3;   // CE_GEMM_TN: 1 = register-staged v1 kernel (one launch per problem), 2 = 128x128 v2,
                              // 3 (default) = 256x256 ring kernel v3 where the shapes allow, v2 elsewhere
    std::call_once(attr_set, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            TN_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            T2_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3_kernel<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            4 * 32 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3_kernel<48, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            3 * 48 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3lw_kernel<48, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            3 * 48 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3lw_kernel<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            4 * 32 * 1024);
        const char* e = getenv("CE_GEMM_TN");
        if (e) variant = atoi(e);
    });
    hipStream_t s = (hipStream_t)stream;
    TNGroup g;
    g.count = count;
    g.M = M;
    int tiles = 0;
    double flops = 0.0, bytes = 0.0;
    long ldmax = 0;
    for (int i = 0; i < count; ++i) {
        CE_CHECK_ARG(Nn[i] > 0 && Kk[i] > 0, "ce_gemm_tn: empty problem");
        CE_CHECK_ARG(Nn[i] % 8 == 0 && Kk[i] % 8 == 0 && ldp[i] % 8 == 0 && ldq[i] % 8 == 0,
                     "ce_gemm_tn: Nn,Kk,ldp,ldq must be multiples of 8");
        CE_CHECK_ARG(ldp[i] >= Nn[i] && ldq[i] >= Kk[i] && ldo[i] >= Kk[i], "ce_gemm_tn: leading dimension smaller than the row");
        TNArgs& a = g.prob[i];
        a.P = (const bf16_t*)P[i]; a.ldp = ldp[i]; a.Q = (const bf16_t*)Q[i]; a.ldq = ldq[i];
        a.out = out[i]; a.ldo = ldo[i]; a.M = M; a.Nn = Nn[i]; a.Kk = Kk[i];
        a.tiles_n = ce_div_up(Nn[i], TN_BN); a.tiles_k = ce_div_up(Kk[i], TN_BK);
        tiles += a.tiles_n * a.tiles_k;
        g.tile_end[i] = tiles;
        flops += 2.0 * M * Nn[i] * Kk[i];
        bytes += 2.0 * ((double)M * Nn[i] + (double)M * Kk[i]) + 8.0 * Nn[i] * Kk[i];
        if (ldp[i] > ldmax) ldmax = ldp[i];
        if (ldq[i] > ldmax) ldmax = ldq[i];
    }
    for (int i = count; i < CE_TN_MAX_GROUP; ++i) g.tile_end[i] = tiles;
    const int m_tiles = ce_div_up(M, TN_BM);
    CE_CHECK_ARG((long)M * ldmax * 2 < (1L << 32), "ce_gemm_tn: operand exceeds 4 GiB");
    if (variant == 1) {
        if (overwrite && zero_outputs() != 0) return -5;
        if (splits <= 0) splits = 512 / tiles;
        if (splits > m_tiles) splits = m_tiles;
        if (splits < 1) splits = 1;
        g.m_per_split = ce_div_up(m_tiles, splits) * TN_BM;
        g.splits = ce_div_up(M, g.m_per_split);
        for (int i = 0; i < count; ++i) {
            TNArgs a = g.prob[i];
            a.splits = g.splits; a.m_per_split = g.m_per_split;
            CeProfScope prof(CE_PROF_GEMM_TN2, 2.0 * M * a.Nn * a.Kk, 0.0, s);
            hipLaunchKernelGGL(gemm_tn_kernel, dim3(a.tiles_n * a.tiles_k * a.splits), dim3(256), TN_LDS_BYTES, s, a);
        }
        CE_LAUNCH_CHECK();
        return 0;
    }
    // v3 (256x256 tiles, one workgroup per CU) when every problem is a multiple of 256 both ways and the contraction is
    // long enough to amortise the 64 KB prologue
    bool can3 = variant == 3 && M >= 2048;
    for (int i = 0; i < count; ++i) can3 = can3 && Nn[i] % 256 == 0 && Kk[i] % 256 == 0;
    if (can3) {
        int tiles3 = 0;
        for (int i = 0; i < count; ++i) {
            TNArgs& a = g.prob[i];
            a.tiles_n = Nn[i] / 256; a.tiles_k = Kk[i] / 256;
            tiles3 += a.tiles_n * a.tiles_k;
            g.tile_end[i] = tiles3;
        }
        for (int i = count; i < CE_TN_MAX_GROUP; ++i) g.tile_end[i] = tiles3;
        static int force_splits = getenv("CE_TN3_SPLITS") ? atoi(getenv("CE_TN3_SPLITS")) : 0;
        // M split by a cost model fitted to tools/bench_tn_group.py (us): a workgroup spends 1.7 per 64-row contraction
        // tile + 3 of prologue; rounds of 256 workgroups; only the LAST round's epilogue is exposed -- 0.20 per tile with
        // float atomics (256 KB at 1.3 TB/s chip-wide), 0.105 as a plain read-modify-write when nothing is split
        auto cost = [&](int sp) {
            const long wgs = (long)tiles3 * sp;
            const long rounds = (wgs + 255) / 256;
            const long tail = wgs - (rounds - 1) * 256;
            return rounds * (ce_div_up(m_tiles, sp) * 1.7 + 3.0) + tail * (sp == 1 ? 0.105 : 0.20);
        };
        int sp = 1;
        if (splits > 0) sp = splits;
        else if (force_splits > 0) sp = force_splits;
        else {
            double best = cost(1);
            for (int c = 2; c <= 16 && c <= m_tiles && (long)tiles3 * c <= 256; ++c)     // split only within one resident round:
                if (cost(c) < best) { best = cost(c); sp = c; }                         // every split tile costs atomic bandwidth
        }
        if (sp > m_tiles) sp = m_tiles;
        if (sp < 1) sp = 1;
        g.m_per_split = ce_div_up(m_tiles, sp) * TN_BM;
        g.splits = ce_div_up(M, g.m_per_split);
        g.overwrite = (overwrite && g.splits == 1) ? 1 : 0;
        if (overwrite && !g.overwrite && zero_outputs() != 0) return -5;
        static int depth = getenv("CE_TN3_DEPTH") ? atoi(getenv("CE_TN3_DEPTH")) : 3;
        g.depth = depth < 1 ? 1 : (depth > 3 ? 3 : depth);
        CeProfScope prof(CE_PROF_GEMM_TN, flops, bytes, s);
        // 48-row stages x 3 slots (default; in the step 993 TF/s) or 32-row stages x 4 slots (CE_TN3_ROWS=32: 935): one
        // stage less in flight costs nothing (prefetch depth 2 = depth 3 above), a third fewer barriers per FLOP pays
        static int rows48 = getenv("CE_TN3_ROWS") ? atoi(getenv("CE_TN3_ROWS")) != 32 : 1;
        static int lw = getenv("CE_TN3_LW") ? atoi(getenv("CE_TN3_LW")) : 1;   // loader-wave form: 1127 -> 1188 TF/s in the step
        if (lw && rows48)
            hipLaunchKernelGGL((gemm_tn3lw_kernel<48, 3>), dim3((unsigned)(tiles3 * g.splits)), dim3(768), 3 * 48 * 1024, s, g);
        else if (lw)
            hipLaunchKernelGGL((gemm_tn3lw_kernel<32, 4>), dim3((unsigned)(tiles3 * g.splits)), dim3(768), 4 * 32 * 1024, s, g);
        else if (rows48)
            hipLaunchKernelGGL((gemm_tn3_kernel<48, 3>), dim3((unsigned)(tiles3 * g.splits)), dim3(1024), 3 * 48 * 1024, s, g);
        else
            hipLaunchKernelGGL((gemm_tn3_kernel<32, 4>), dim3((unsigned)(tiles3 * g.splits)), dim3(1024), 4 * 32 * 1024, s, g);
        CE_LAUNCH_CHECK();
        return 0;
    }
    // v2: one resident round (at most 2 workgroups per CU = 512 slots), never a ragged second round of SPLIT tiles
    if (overwrite && zero_outputs() != 0) return -5;
    if (splits <= 0) splits = 512 / tiles;
    if (splits > m_tiles) splits = m_tiles;
    if (splits < 1) splits = 1;
    g.m_per_split = ce_div_up(m_tiles, splits) * TN_BM;
    g.splits = ce_div_up(M, g.m_per_split);
    CeProfScope prof(CE_PROF_GEMM_TN2, flops, bytes, s);
    hipLaunchKernelGGL(gemm_tn2_kernel, dim3((unsigned)(tiles * g.splits)), dim3(256), T2_LDS_BYTES, s, g);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_probe_mfma(int shape, const void* a_frags, const void* b_frags, float* out, void* stream) {
    if (shape == 16)
        hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16x8*)a_frags,
                           (const bf16x8*)b_frags, (f32x4*)out);
    else if (shape == 32)
        hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16x8*)a_frags,
                           (const bf16x8*)b_frags, (f32x16*)out);
    else
        CE_CHECK_ARG(false, "ce_probe_mfma: shape must be 16 or 32");
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_probe_tr16(const void* image, int n_elems, const int* byte_off, void* out, void* stream) {
    CE_CHECK_ARG(n_elems > 0 && n_elems <= 16384, "ce_probe_tr16: image must hold 1..16384 elements");
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), n_elems * 2, (hipStream_t)stream, (const uint16_t*)image,
                       n_elems, byte_off, (s16x4*)out);
    CE_LAUNCH_CHECK();
    return 0;
}
