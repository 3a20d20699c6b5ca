// Diagnostic build (not part of the library): gemm_nt256_kernel<BF16, TM=5, WN=2> with s_memtime stamps around the
// phases of a K iteration -- where does an iteration of the 160x128 two-workgroup NT kernel spend its cycles?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I clip_event_amd/csrc tools/diag/nt256_stamps.hip -o /tmp/nt256_stamps && /tmp/nt256_stamps
// Stamp values go to a buffer of their own; no output value depends on them (MI355X_MICROARCH.md "DVFS give-back" (6)).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "common.hpp"
#include "../../include/clip_event_hip.h"
void ce_set_error(const char*, ...) {}
int ce_prof_begin(int, double, double, hipStream_t) { return -1; }
void ce_prof_end(int, hipStream_t) {}
void ce_prof_retag(int, int) {}
namespace {
#include "gemm_common.hpp"
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
constexpr int N2_BK = 64;

__device__ __forceinline__ void dma16_asm(const void* gaddr, uint32_t lds_dst) {      // flat global address form
    uint32_t keep;
    asm volatile(
        "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gaddr), "s"(lds_dst)
        : "memory");
}

// MODE 0: builtin LDS-DMA, all issued at the top of the iteration (the library kernel); 1: inline-asm DMA at the top,
// explicit vmcnt(0) before the barrier; 2: inline-asm DMA, ONE instruction after every row of four MFMAs
template <int TM, int WN, int MODE>
__global__ __launch_bounds__(128 * WN, 2) void nt256_stamped(NTArgs p, unsigned long long* stamps) {
    constexpr int NW = 2 * WN;
    constexpr int N2_BN = 64 * WN;
    constexpr int N2_BTILE_BYTES = N2_BN * N2_BK * 2;
    constexpr int N2_BM = 32 * TM;
    constexpr int N2_TILE_BYTES = N2_BM * N2_BK * 2;
    constexpr int N2_STAGE_BYTES = N2_TILE_BYTES + N2_BTILE_BYTES;
    constexpr int A_INSTR = N2_BM / 8;
    constexpr int A_PER_WAVE = (A_INSTR + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * N2_BM, n0 = tn * N2_BN;
    const int s_r = lane >> 3, s_pos = lane & 7, s_chunk = s_pos ^ s_r;
    const bf16_t* gA[A_PER_WAVE];
    const bf16_t* gB[4];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) gA[i] = p.A + (long)min(m0 + (wave + NW * i) * 8 + s_r, p.M - 1) * p.lda + s_chunk * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) gB[i] = p.B + (long)min(n0 + (wave * 4 + i) * 8 + s_r, p.N - 1) * p.ldb + s_chunk * 8;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t*)smem;
    auto issue_one = [&](int d, int st, int kt) {       // DMA d of this wave's 4 + A_PER_WAVE per stage (asm form)
        const int koff = kt * N2_BK;
        const uint32_t base = lds0 + st * N2_STAGE_BYTES;
        if (d < 4) dma16_asm(gB[d] + koff, base + N2_TILE_BYTES + wave * 4096 + d * 1024);
        else if (wave + NW * (d - 4) < A_INSTR) dma16_asm(gA[d - 4] + koff, base + (wave + NW * (d - 4)) * 1024);
    };
    auto stage = [&](int st, int kt) {
        if constexpr (MODE != 0) {
#pragma unroll
            for (int d = 0; d < 4 + A_PER_WAVE; ++d) issue_one(d, st, kt);
            return;
        }
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
    const int nk = p.K / N2_BK;
    unsigned long long rP0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tP0 = __builtin_amdgcn_s_memtime();
    stage(0, 0);
    if constexpr (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long tP1 = __builtin_amdgcn_s_memtime();
    const int f_row = lane & 15, f_kc = lane >> 4, f_sw = lane & 7;
    const int fa_base = (wm * (TM * 16) + f_row) * 128;
    const int fb_base = N2_TILE_BYTES + (wn * 64 + f_row) * 128;
    unsigned long long c_issue = 0, c_mfma = 0, c_wait = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE != 2 && kt + 1 < nk) stage(cur ^ 1, kt + 1);
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
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
                    if (mh * 4 + t < TM) af[t] = *reinterpret_cast<const bf16x8*>(st + fa_base + (mh * 4 + t) * 2048 + coff);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (mh * 4 + t < TM) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mh * 4 + t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[t], acc[mh * 4 + t][nt], 0, 0, 0);
                        if constexpr (MODE == 2) {
                            const int d = ks * TM + mh * 4 + t;
                            if (d < 4 + A_PER_WAVE && kt + 1 < nk) issue_one(d, cur ^ 1, kt + 1);
                        }
                    }
            }
        }
        // keep the MFMA results live up to the stamp
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[TM - 1][3][3]));
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if constexpr (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        c_issue += t1 - t0; c_mfma += t2 - t1; c_wait += t3 - t2;
    }
    unsigned long long tE0 = __builtin_amdgcn_s_memtime();
    // plain epilogue (bf16), enough to keep the result live
    const int em = m0 + wm * (TM * 16) + (lane & 15), en = n0 + wn * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int m = em + mt * 16, n = en + nt * 16;
            if (m < p.M && n < p.N) {
                u32x2 o = {pack_bf2(acc[mt][nt][0], acc[mt][nt][1]), pack_bf2(acc[mt][nt][2], acc[mt][nt][3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + (long)m * p.ldo + n) = o;
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long tE1 = __builtin_amdgcn_s_memtime();
    unsigned long long rE1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned long long* o = stamps + ((long)blockIdx.x * NW + wave) * 8;
        o[0] = tP1 - tP0; o[1] = c_issue; o[2] = c_mfma; o[3] = c_wait; o[4] = tE1 - tE0; o[5] = tE1 - tP0; o[6] = tP0; o[7] = rE1 - rP0;
    }
}
}  // namespace

int main(int argc, char** argv) {
    const int M = 12800, N = 768, K = argc > 1 ? atoi(argv[1]) : 3072;
    const int cold = argc > 2 ? atoi(argv[2]) : 1;
    const float sustain = argc > 3 ? atof(argv[3]) : 0.f;
    constexpr int TM = 5, WN = 2;
    bf16_t *A, *B, *C, *junk;
    unsigned long long* st;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
    const size_t junk_bytes = 1ull << 30;
    hipMalloc(&junk, junk_bytes);
    std::vector<bf16_t> h((size_t)M * K);
    for (auto& v : h) v = (bf16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // random-ish bf16 around +-1
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    NTArgs a{};
    a.A = A; a.lda = K; a.B = B; a.ldb = K; a.M = M; a.N = N; a.K = K; a.out = C; a.ldo = N;
    a.tiles_m = (M + 159) / 160; a.tiles_n = (N + 127) / 128;
    const int grid = a.tiles_m * a.tiles_n, lds = 2 * (160 + 128) * 64 * 2;
    hipMalloc(&st, (size_t)grid * 4 * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        auto kern = mode == 0 ? nt256_stamped<TM, WN, 0> : (mode == 1 ? nt256_stamped<TM, WN, 1> : nt256_stamped<TM, WN, 2>);
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float best = 1e9;
        if (sustain > 0) {                                           // hold the chip under this load first (DVFS settles)
            hipEventRecord(e0, 0);
            for (float el = 0; el < sustain * 1e3f;) {
                for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a, st);
                hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&el, e0, e1);
            }
        }
        for (int it = 0; it < 6; ++it) {
            if (cold) hipMemsetAsync(junk, it, junk_bytes, 0);       // push the operands out of L2 / Infinity Cache
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a, st);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 1 && ms < best) best = ms;
        }
        std::vector<unsigned long long> hs((size_t)grid * 4 * 8);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        double s[6] = {0, 0, 0, 0, 0, 0}, real = 0;
        for (size_t w = 0; w < (size_t)grid * 4; ++w)
        { for (int k = 0; k < 6; ++k) s[k] += (double)hs[w * 8 + k]; real += (double)hs[w * 8 + 7]; }
        const double n = (double)grid * 4, nk = K / 64;
        printf("mode %d M=%d N=%d K=%d %s: %.1f us = %.0f TF/s | prologue %.0f | per K iteration: DMA issue %.1f, reads + 40 MFMA %.1f, "
               "vmcnt+barrier %.1f | epilogue %.0f | wave lifetime %.0f | in-kernel clock %.2f GHz\n", mode, M, N, K, cold ? "cold" : "warm", best * 1e3,
               2.0 * M * N * K / (best * 1e-3) / 1e12, s[0] / n, s[1] / n / nk, s[2] / n / nk, s[3] / n / nk, s[4] / n, s[5] / n, s[5] / real * 0.1);
    }
    return 0;
}
