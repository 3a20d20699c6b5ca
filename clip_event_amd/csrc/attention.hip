// Fused multi-head self-attention forward / backward for short sequences (gfx950, wave64).
//
// Replaces the attention core of nn.MultiheadAttention as called at model_clip.py:175,188
// (need_weights=False; additive causal mask for the text tower, model_clip.py:377-384; no
// padding mask) and its autograd: per (sample, head)  O = softmax(Q K^T / sqrt(64) + mask) V.
// Sequences are 50 (ViT-B/32) or 77 (text) tokens, head_dim 64, so one workgroup owns one
// (sample, head): K and V (and Q, dO in the backward) sit in LDS as [token][64] bf16 images
// with a 160-byte row stride, which is bank-conflict free both for ds_read_b128 row fragments
// and for ds_read_b64_tr_b16 transposed fragments.
//
// Forward, per wave and 16-query strip, all v_mfma_f32_16x16x32_bf16:
//   S^T = K Q^T      (keys on accumulator rows, queries on lanes -> a query's scores live in
//                     4 lanes x 4T registers; softmax row-reduce = in-lane + 2 shuffles)
//   O^T = V^T P^T    (P^T is taken straight from the S^T accumulators as the B operand; the
//                     permuted contraction order is matched by the transposed V reads)
// Backward: phase 1 per query strip recomputes P from the saved log-sum-exp, forms
//   dS = scale * P (dP - delta), writes P^T / dS^T (bf16) to LDS and computes dQ^T = K^T dS^T;
// phase 2 per 16-key tile computes dV^T = dO^T P and dK^T = Q^T dS from the LDS images.
#include <map>
#include <mutex>

#include <type_traits>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

constexpr int HD = 64;          // head dim (both towers, model_clip.py:307 / :600)
constexpr int ROW = 160;        // LDS row stride in bytes: 128 + 32 pad

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    u32x4 v = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, v);
}

// cooperative copy of the K and the V [L][64] head slices (row stride ld elements) into their LDS images, rows >= L zero-filled up to
// Lp.  Bounds-checked buffer loads on one-slice descriptors (a row past L reads as zeros, no branch), BOTH slices' loads in flight
// before the first LDS write: as two `if (r < L) v = load; store` loops, one per slice, every load was waited for inside its
// divergent branch, so the workgroup paid the K and the V memory latency one after the other.
template <int NCH>       // 16-byte chunks per thread and slice: ceil(Lp * 8 / blockDim.x)
__device__ __forceinline__ void stage_kv(char* sK, char* sV, const bf16_t* srcK, const bf16_t* srcV, long ld, int L, int Lp, int tid,
                                         int nthr) {
    const uint32_t span = (uint32_t)(((long)L - 1) * ld * 2 + HD * 2);
    const __amdgpu_buffer_rsrc_t rK = make_rsrc(srcK, span), rV = make_rsrc(srcV, span);
    u32x4 kv[NCH], vv[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int idx = tid + j * nthr, r = idx >> 3, c = idx & 7;
        const uint32_t off = r < L ? (uint32_t)(r * ld * 2 + c * 16) : 0x80000000u;      // (rows L .. Lp - 1 and chunks past the image: zeros)
        kv[j] = __builtin_amdgcn_raw_buffer_load_b128(rK, off, 0, 0);
        vv[j] = __builtin_amdgcn_raw_buffer_load_b128(rV, off, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int idx = tid + j * nthr, r = idx >> 3, c = idx & 7;
        if (idx < Lp * 8) {
            *reinterpret_cast<u32x4*>(sK + r * ROW + c * 16) = kv[j];
            *reinterpret_cast<u32x4*>(sV + r * ROW + c * 16) = vv[j];
        }
    }
}

template <int T>  // T = Lp / 16 key tiles (Lp = L rounded up to 32)
__global__ __launch_bounds__(512) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o,
                                                       long ldo, float* __restrict__ lse,
                                                       const int* __restrict__ cu, int Lmax, int H, int D,
                                                       int causal, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = T * 16;
    char* sK = smem;
    char* sV = smem + Lp * ROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    // packed (variable-length) batches: sample b owns rows cu[b] .. cu[b+1]-1; dense: b*Lmax .. +Lmax-1
    const long row0 = cu ? (long)cu[b] : (long)b * Lmax;
    const int L = cu ? min(cu[b + 1] - cu[b], Lmax) : Lmax;
    if (L <= 0) return;
    const int Tl = ((L + 31) >> 5) << 1;            // live 16-key tiles (even), block-uniform
    const bf16_t* base = qkv + row0 * ld + h * HD;

    // one 16-query strip per wave (the launch uses ceil(Lmax/16) waves); its Q rows are requested before the K / V
    // staging so that the two memory latencies overlap
    const int li = lane & 15, g = lane >> 4;
    const int nstrips = (L + 15) >> 4;
    const int strip = wave;
    const int i = strip * 16 + li;               // this lane's query
    const int iq = min(i, L - 1);
    bf16x8 qf[2];
    if (strip < nstrips) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)iq * ld + ks * 32 + g * 8);
    }
    // (Tl * 16 * 8 chunks per slice over the launch's min(8, ceil(Lmax / 16)) waves: at most four per thread -- L = 1 -- and two or
    //  three at the towers' lengths; the surplus iterations are out-of-range loads that touch no memory)
    stage_kv<4>(sK, sV, base + D, base + 2 * D, ld, L, Tl * 16, tid, blockDim.x);
    __syncthreads();

    if (strip < nstrips) {
        f32x4 s[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < Tl) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
                }
            }
        }
        // lane holds S^T[j = 16t + 4g + r][i]
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 16 + g * 4 + r;
                const bool ok = (j < L) && (!causal || j <= i);
                s[t][r] = ok ? s[t][r] * scale : -INFINITY;
                mx = fmaxf(mx, s[t][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = __expf(s[t][r] - mx);
                sum += s[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        // O^T = V^T P^T : contraction element jj of lane-group g <-> key 32s + 16*(jj>>2) + 4g + (jj&3)
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < T / 2; ++sidx) {
            if (2 * sidx >= Tl) continue;
            bf16x8 pf = pack8(s[2 * sidx], s[2 * sidx + 1]);
            const char* vrow = sV + (32 * sidx + 4 * g + (li >> 2)) * ROW + (4 * (li & 3)) * 2;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 vf = tr_pair(vrow + ct * 32, vrow + ct * 32 + 16 * ROW);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc[ct], 0, 0, 0);
            }
        }
        if (i < L) {
            const float inv = 1.0f / sum;
            bf16_t* orow = o + (row0 + i) * ldo + h * HD + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                u32x2 pk = {pack_bf2(acc[ct][0] * inv, acc[ct][1] * inv), pack_bf2(acc[ct][2] * inv, acc[ct][3] * inv)};
                *reinterpret_cast<u32x2*>(orow + ct * 16) = pk;
            }
            if (g == 0) lse[((long)b * H + h) * Lmax + i] = mx + __logf(sum);
        }
    }
}

template <int T>
__global__ __launch_bounds__(512) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                       const bf16_t* __restrict__ o, long ldo,
                                                       const bf16_t* __restrict__ dout, long lddo,
                                                       const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                       long lddq, float* __restrict__ bias_grad,
                                                       const int* __restrict__ cu, int Lmax, int H, int D,
                                                       int causal, float scale) {
    // LDS: two [Lp][64] operand images (K,V in phase 1; re-filled with Q,dO for phase 2) + the P / dS images
    // [query i][key j]; the 192 column-sum floats live in the 32 pad bytes of the P image's first 24 rows.  72 KiB at L=77 (two
    // workgroups per CU), 40 KiB EXACTLY at L=50: four workgroups per CU (with the sums in 768 bytes of their own it was three).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int Lp = T * 16;
    constexpr int PROW = Lp * 2 + 32;            // row stride of the P / dS images: = 32 (mod 64) bytes
    char* sA = smem;                             // K, then Q
    char* sB = sA + Lp * ROW;                    // V, then dO
    char* sP = sB + Lp * ROW;
    char* sDS = sP + Lp * PROW;
    // [3][64] column sums of dq | dk | dv (in_proj bias gradient): element i in the pad of P-image row i >> 3 (the images use bytes 0 .. 2 Lp - 1 of a row)
    auto csum = [&](int i) __attribute__((always_inline)) -> float* { return reinterpret_cast<float*>(sP + (i >> 3) * PROW + Lp * 2 + (i & 7) * 4); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const long row0 = cu ? (long)cu[b] : (long)b * Lmax;         // packed batches: see attn_fwd_kernel
    const int L = cu ? min(cu[b + 1] - cu[b], Lmax) : Lmax;
    if (L <= 0) return;
    const int Tl = ((L + 31) >> 5) << 1;                         // live 16-row tiles (even), block-uniform
    const int Ll = Tl * 16;
    const bf16_t* base = qkv + row0 * ld + h * HD;
    const bf16_t* dob = dout + row0 * lddo + h * HD;
    const bf16_t* ob = o + row0 * ldo + h * HD;
    bf16_t* dbase = dqkv + row0 * lddq + h * HD;

    for (int i = tid; i < 192; i += blockDim.x) *csum(i) = 0.f;  // the workgroup may have fewer than 192 threads
    const int li = lane & 15, g = lane >> 4;
    // ---- every global read of the workgroup is issued up front (one exposed memory latency instead of three):
    // the K, V, Q, dO head slices (<= 2 16-byte chunks per matrix per thread: blockDim = 64 T >= Lp*8/2) and this
    // wave's own query strip (one strip per wave: the launch uses T waves) ----
    u32x4 rK[2], rV[2], rQ[2], rD[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + c * blockDim.x;
        const int r = idx >> 3, ch = idx & 7;
        const u32x4 z = {0u, 0u, 0u, 0u};
        rK[c] = z; rV[c] = z; rQ[c] = z; rD[c] = z;
        if (idx < Ll * 8 && r < L) {
            rK[c] = *reinterpret_cast<const u32x4*>(base + D + (long)r * ld + ch * 8);
            rV[c] = *reinterpret_cast<const u32x4*>(base + 2 * D + (long)r * ld + ch * 8);
            rQ[c] = *reinterpret_cast<const u32x4*>(base + (long)r * ld + ch * 8);
            rD[c] = *reinterpret_cast<const u32x4*>(dob + (long)r * lddo + ch * 8);
        }
    }
    const int strip = wave;                       // phase 1: one 16-query strip per wave
    const bool has_strip = strip < Tl;
    const int i = strip * 16 + li;
    const bool iok = has_strip && i < L;
    const int iq = min(i, L - 1);
    float dl = 0.f, lsei = 0.f;
    bf16x8 qf[2], df[2];       // B operands: rows of Q and dO straight from global (only this wave needs them)
    if (has_strip) {
        // delta_i = sum_c dO[i][c] O[i][c]; each of the 4 lanes of a query takes 16 columns
        const u32x4* pd = reinterpret_cast<const u32x4*>(dob + (long)iq * lddo + g * 16);
        const u32x4* po = reinterpret_cast<const u32x4*>(ob + (long)iq * ldo + g * 16);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            u32x4 a = pd[v], c = po[v];
#pragma unroll
            for (int e = 0; e < 4; ++e) dl += bf_lo(a[e]) * bf_lo(c[e]) + bf_hi(a[e]) * bf_hi(c[e]);
        }
        lsei = lse[((long)b * H + h) * Lmax + iq];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)iq * ld + ks * 32 + g * 8);
            df[ks] = *reinterpret_cast<const bf16x8*>(dob + (long)iq * lddo + ks * 32 + g * 8);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + c * blockDim.x;
        if (idx < Ll * 8) {
            *reinterpret_cast<u32x4*>(sA + (idx >> 3) * ROW + (idx & 7) * 16) = rK[c];
            *reinterpret_cast<u32x4*>(sB + (idx >> 3) * ROW + (idx & 7) * 16) = rV[c];
        }
    }
    __syncthreads();
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);

    // ---------------- phase 1: this wave's 16-query strip ----------------
    if (has_strip) {
        f32x4 p[T], ds[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            p[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t >= Tl) continue;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(sA + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                bf16x8 vf = *reinterpret_cast<const bf16x8*>(sB + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], p[t], 0, 0, 0);     // S^T
                ds[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df[ks], ds[t], 0, 0, 0);   // dP^T
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 16 + g * 4 + r;
                const bool ok = iok && (j < L) && (!causal || j <= i);
                const float pv = ok ? __expf(p[t][r] * scale - lsei) : 0.f;
                p[t][r] = pv;
                ds[t][r] = pv * (ds[t][r] - dl) * scale;
            }
            // P / dS images [i][j] for phase 2: this lane owns 4 consecutive keys of query i -> one 8-byte store each
            u32x2 pk = {pack_bf2(p[t][0], p[t][1]), pack_bf2(p[t][2], p[t][3])};
            u32x2 dk = {pack_bf2(ds[t][0], ds[t][1]), pack_bf2(ds[t][2], ds[t][3])};
            *reinterpret_cast<u32x2*>(sP + i * PROW + (t * 16 + g * 4) * 2) = pk;
            *reinterpret_cast<u32x2*>(sDS + i * PROW + (t * 16 + g * 4) * 2) = dk;
        }
        // dQ^T = K^T dS^T (dS^T straight from the accumulators, K^T by transposed reads)
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < T / 2; ++sidx) {
            if (2 * sidx >= Tl) continue;
            bf16x8 sf = pack8(ds[2 * sidx], ds[2 * sidx + 1]);
            const char* krow = sA + (32 * sidx + 4 * g + (li >> 2)) * ROW + (4 * (li & 3)) * 2;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 kf = tr_pair(krow + ct * 32, krow + ct * 32 + 16 * ROW);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, sf, acc[ct], 0, 0, 0);
            }
        }
        if (iok) {
            bf16_t* qrow = dbase + (long)i * lddq + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                u32x2 pk = {pack_bf2(acc[ct][0], acc[ct][1]), pack_bf2(acc[ct][2], acc[ct][3])};
                *reinterpret_cast<u32x2*>(qrow + ct * 16) = pk;
            }
        }
        if (bias_grad) {   // sum over the strip's 16 queries (one DPP row); padded queries are exact zeros
            float vals[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) vals[ct * 4 + r] = acc[ct][r];
            const float tot = row16_colsum(vals, li);
            const int k = row16_colsum_index(li);
            atomicAdd(csum((k >> 2) * 16 + 4 * g + (k & 3)), tot);
        }
    }
    __syncthreads();                                             // K, V no longer needed; P / dS images complete
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + c * blockDim.x;
        if (idx < Ll * 8) {
            *reinterpret_cast<u32x4*>(sA + (idx >> 3) * ROW + (idx & 7) * 16) = rQ[c];   // Q
            *reinterpret_cast<u32x4*>(sB + (idx >> 3) * ROW + (idx & 7) * 16) = rD[c];   // dO
        }
    }
    __syncthreads();
    // ---------------- phase 2: per 16-key tile ----------------
    const int nkt = (L + 15) >> 4;
    for (int jt = wave; jt < nkt; jt += nw) {
        const int j = jt * 16 + li;
        f32x4 av[4], ak[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            av[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            ak[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int sidx = 0; sidx < T / 2; ++sidx) {
            if (2 * sidx >= Tl) continue;
            // contraction i = 32s + 8g + jj (natural order) for both operands
            const int irow = 32 * sidx + 8 * g + (li >> 2);
            // B operands: P[i][j], dS[i][j] = 8 consecutive queries of key column j -> transposed reads of the [i][j] images
            const char* pcol = sP + irow * PROW + (jt * 16 + 4 * (li & 3)) * 2;
            const char* scol = sDS + irow * PROW + (jt * 16 + 4 * (li & 3)) * 2;
            bf16x8 pf = tr_pair(pcol, pcol + 4 * PROW);
            bf16x8 sf = tr_pair(scol, scol + 4 * PROW);
            // A operands: dO^T / Q^T rows c
            const int roff = irow * ROW + (4 * (li & 3)) * 2;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 dof = tr_pair(sB + roff + ct * 32, sB + roff + ct * 32 + 4 * ROW);
                bf16x8 qf = tr_pair(sA + roff + ct * 32, sA + roff + ct * 32 + 4 * ROW);
                av[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf, av[ct], 0, 0, 0);
                ak[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, sf, ak[ct], 0, 0, 0);
            }
        }
        if (j < L) {
            bf16_t* krow = dbase + (long)j * lddq + D + 4 * g;
            bf16_t* vrow = dbase + (long)j * lddq + 2 * D + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                u32x2 pk = {pack_bf2(ak[ct][0], ak[ct][1]), pack_bf2(ak[ct][2], ak[ct][3])};
                u32x2 pv = {pack_bf2(av[ct][0], av[ct][1]), pack_bf2(av[ct][2], av[ct][3])};
                *reinterpret_cast<u32x2*>(krow + ct * 16) = pk;
                *reinterpret_cast<u32x2*>(vrow + ct * 16) = pv;
            }
        }
        if (bias_grad) {   // rows j >= L of the tile are exact zeros (their P / dS columns are)
            float vk[16], vv[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vk[ct * 4 + r] = ak[ct][r];
                    vv[ct * 4 + r] = av[ct][r];
                }
            const float tk = row16_colsum(vk, li), tv = row16_colsum(vv, li);
            const int k = row16_colsum_index(li);
            atomicAdd(csum(64 + (k >> 2) * 16 + 4 * g + (k & 3)), tk);
            atomicAdd(csum(128 + (k >> 2) * 16 + 4 * g + (k & 3)), tv);
        }
    }
    if (bias_grad) {
        __syncthreads();
        for (int i = tid; i < 192; i += blockDim.x) atomicAdd(bias_grad + (i >> 6) * D + h * HD + (i & 63), *csum(i));
    }
}

// ------------------------------------------------------------------------------------------
// Sequences longer than 128 tokens (ViT-B/16: 197, ViT-L/14: 257 / 577 at 336 px): the same fragment maps, tiled
// flash-style over 64-key blocks with an online softmax.  One workgroup = 4 waves = 64 queries (forward, dQ) or 64
// keys (dK / dV); K / V (Q / dO) blocks pass through LDS, nothing of size L x L is ever stored.  The backward is two
// kernels so that no gradient needs atomics: A recomputes P per (query block, key block) and accumulates dQ over the
// key blocks; B owns a key block and accumulates dK, dV over the query blocks.
// ------------------------------------------------------------------------------------------
constexpr int LB = 64;                 // rows per block
constexpr int LPROW = LB * 2 + 32;     // P / dS image row stride (bytes)

__device__ __forceinline__ void stage_block(char* dst, const bf16_t* src, long ld, int row0, int L, int tid) {
    // 64 rows x 64 bf16 = 512 16-byte chunks, 256 threads; rows >= L zero-filled
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + c * 256;
        const int r = idx >> 3, ch = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < L) v = *reinterpret_cast<const u32x4*>(src + (long)(row0 + r) * ld + ch * 8);
        *reinterpret_cast<u32x4*>(dst + r * ROW + ch * 16) = v;
    }
}

// the same block in two steps: global -> registers (issued a block ahead, in flight during the MFMAs), registers -> LDS
struct BlockRegs { u32x4 v[2]; };
__device__ __forceinline__ BlockRegs load_block(const bf16_t* src, long ld, int row0, int L, int tid) {
    BlockRegs b;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + c * 256;
        const int r = idx >> 3, ch = idx & 7;
        b.v[c] = u32x4{0u, 0u, 0u, 0u};
        if (row0 + r < L) b.v[c] = *reinterpret_cast<const u32x4*>(src + (long)(row0 + r) * ld + ch * 8);
    }
    return b;
}
__device__ __forceinline__ void store_block(char* dst, const BlockRegs& b, int tid) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = tid + c * 256;
        *reinterpret_cast<u32x4*>(dst + (idx >> 3) * ROW + (idx & 7) * 16) = b.v[c];
    }
}

// (block within the sequence, (sample, head)) of this workgroup.  The grid is (blocks per sequence, B H) and workgroups are handed to the
// eight XCDs round-robin in launch order, x fastest: with the plain decoding the blocks of ONE (sample, head) -- which all stream the same
// K / V (forward, dQ) or Q / dO (dK / dV) slices -- land on eight different L2s.  Here XCD x owns the (sample, head) pairs congruent to x
// modulo 8 and walks their blocks one pair after the other, so a pair's slices are fetched into one L2 once.  Bijective when B H is a
// multiple of 8 (else the plain order).  Speed only, never correctness.  CE_ATTN_XCD=0 (launcher) switches it off.
__device__ __forceinline__ void long_block_coords(int xcd_order, int& blk, int& bh) {
    blk = blockIdx.x;
    bh = blockIdx.y;
    const int nblk = gridDim.x, nbh = gridDim.y;
    if (xcd_order && (nbh & 7) == 0) {
        const int id = blockIdx.y * nblk + blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        bh = (slot / nblk) * 8 + xcd;
        blk = slot - (slot / nblk) * nblk;
    }
}

// NS 16-query strips per wave (a workgroup = 4 waves = 64 NS queries): every K fragment and every transposed V fragment read
// from LDS serves NS strips.  With one strip per wave the 20 resident waves of a CU each re-read the whole 16 KiB K / V block
// for 16 queries -- 320 KiB of LDS reads per CU and key block, 2.5 k cycles at the 128 B/clk the LDS delivers, twice the
// matrix-pipe time of the same step.  MEASURED (ViT-L/14@336, B = 32, same box, attention ms per step): two strips per wave are
// SLOWER in the forward (3.27-3.58 vs 3.00-3.19) and in dK / dV with two key tiles (10.97 vs 9.9), slightly faster in dQ
// (9.34 vs 9.49, 9.86 vs 9.96 for the whole backward) -- the 138-200 registers of the wider forms halve the resident waves and
// these kernels live on occupancy (barriers, dependent exp chains), not on LDS bandwidth.  Defaults: forward 1, dQ 2, dK / dV 1;
// every form is parity-tested (CE_ATTN_NS_FWD / CE_ATTN_NS / CE_ATTN_NK).
template <int NS>
__global__ __launch_bounds__(256) void attn_fwd_long_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o,
                                                            long ldo, float* __restrict__ lse, int L, int H, int D,
                                                            int causal, float scale, int xcd_order) {
    __shared__ __attribute__((aligned(16))) char smem[2 * LB * ROW];
    char* sK = smem;
    char* sV = smem + LB * ROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qb, bh_;
    long_block_coords(xcd_order, qb, bh_);
    const int b = bh_ / H, h = bh_ - b * H;
    const bf16_t* base = qkv + (long)b * L * ld + h * HD;
    const int li = lane & 15, g = lane >> 4;
    constexpr int QB = LB * NS;                                // queries per workgroup
    int i[NS];
    bf16x8 qf[NS][2];
    float m[NS], l[NS];                                        // m: running maximum of the RAW scores q.k
    f32x4 acc[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        i[s] = qb * QB + (wave * NS + s) * 16 + li;
        const int iq = min(i[s], L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[s][ks] = *reinterpret_cast<const bf16x8*>(base + (long)iq * ld + ks * 32 + g * 8);
        m[s] = -INFINITY;
        l[s] = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[s][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = scale * 1.4426950408889634f;             // scale * log2(e)
    const int nkb_all = (L + LB - 1) / LB;
    const int nkb = causal ? min(nkb_all, (qb * QB + QB - 1) / LB + 1) : nkb_all;
    BlockRegs nk = load_block(base + D, ld, 0, L, tid), nv = load_block(base + 2 * D, ld, 0, L, tid);
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb) __syncthreads();
        store_block(sK, nk, tid);
        store_block(sV, nv, tid);
        if (kb + 1 < nkb) {                                   // next block in flight during this block's MFMAs
            nk = load_block(base + D, ld, (kb + 1) * LB, L, tid);
            nv = load_block(base + 2 * D, ld, (kb + 1) * LB, L, tid);
        }
        __syncthreads();
        // Softmax on the RAW scores: m tracks the largest raw score (scale > 0 keeps the order) and every probability is ONE
        // fma + v_exp_f32, exp2(s c - m c) with c = scale log2(e) -- the scaling multiply, the subtraction and (except in the
        // blocks that hold keys past L, or under the causal mask) the per-element mask are gone.  These kernels are bound by
        // their vector ALU work and their LDS reads, not by the matrix pipe (head_dim 64: 256 MFMA FLOPs per score against
        // ~7 vector operations).
        f32x4 sc[NS][4];
        const bool need_mask = causal || (kb + 1) * LB > L;    // block-uniform
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x8 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(sK + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                sc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) sc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[s][ks], sc[s][t], 0, 0, 0);
            }
        }
        bf16x8 pf[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float bm = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (need_mask) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = kb * LB + t * 16 + g * 4 + r;
                        const bool ok = (j < L) && (!causal || j <= i[s]);
                        sc[s][t][r] = ok ? sc[s][t][r] : -INFINITY;
                    }
                }
                bm = fmaxf(fmaxf(bm, fmaxf(sc[s][t][0], sc[s][t][1])), fmaxf(sc[s][t][2], sc[s][t][3]));
            }
            bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float mn = fmaxf(m[s], bm);                 // finite from the first block on: key 0 is never masked
            const float alpha = (m[s] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m[s] - mn) * c2);
            const float mc = mn * c2;
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[s][t][r] = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(sc[s][t][r], c2, -mc));
                    sum += sc[s][t][r];
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            l[s] = l[s] * alpha + sum;
            m[s] = mn;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[s][ct] *= alpha;
            pf[s][0] = pack8(sc[s][0], sc[s][1]);
            pf[s][1] = pack8(sc[s][2], sc[s][3]);
        }
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const char* vrow = sV + (32 * sidx + 4 * g + (li >> 2)) * ROW + (4 * (li & 3)) * 2;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 vf = tr_pair(vrow + ct * 32, vrow + ct * 32 + 16 * ROW);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s][sidx], acc[s][ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (i[s] < L) {
            const float inv = 1.0f / l[s];
            bf16_t* orow = o + ((long)b * L + i[s]) * ldo + h * HD + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                u32x2 pk = {pack_bf2(acc[s][ct][0] * inv, acc[s][ct][1] * inv), pack_bf2(acc[s][ct][2] * inv, acc[s][ct][3] * inv)};
                *reinterpret_cast<u32x2*>(orow + ct * 16) = pk;
            }
            if (g == 0) lse[((long)b * H + h) * L + i[s]] = m[s] * scale + __logf(l[s]);
        }
    }
}

// per-query operands of a 16-query strip: B fragments of Q and dO, log-sum-exp, delta = sum_c dO O
struct StripOps {
    bf16x8 qf[2], df[2];
    float lse, dl;
};
__device__ __forceinline__ StripOps load_strip(const bf16_t* base, long ld, const bf16_t* dob, long lddo, const bf16_t* ob,
                                               long ldo, const float* lse_row, int iq, int g) {
    StripOps s;
    float dl = 0.f;
    const u32x4* pd = reinterpret_cast<const u32x4*>(dob + (long)iq * lddo + g * 16);
    const u32x4* po = reinterpret_cast<const u32x4*>(ob + (long)iq * ldo + g * 16);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        u32x4 a = pd[v], c = po[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) dl += bf_lo(a[e]) * bf_lo(c[e]) + bf_hi(a[e]) * bf_hi(c[e]);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    s.dl = dl;
    s.lse = lse_row[iq];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        s.qf[ks] = *reinterpret_cast<const bf16x8*>(base + (long)iq * ld + ks * 32 + g * 8);
        s.df[ks] = *reinterpret_cast<const bf16x8*>(dob + (long)iq * lddo + ks * 32 + g * 8);
    }
    return s;
}

// backward A: dQ (and its bias column sums), one workgroup per block of 64 NS queries; NS 16-query strips per wave share every
// K / V fragment read (see attn_fwd_long_kernel).  Per probability one fma + v_exp_f32 (p = exp2(s c - lse log2 e), c = scale
// log2 e); the softmax scale of dS is applied once to the finished dQ accumulators (a power of two: the same bits); the
// validity / causal mask only where the key block needs it (block-uniform).
template <int NS>
__global__ __launch_bounds__(256) void attn_bwd_long_dq_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                               const bf16_t* __restrict__ o, long ldo,
                                                               const bf16_t* __restrict__ dout, long lddo,
                                                               const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                               long lddq, float* __restrict__ bias_grad, float* __restrict__ delta_out,
                                                               int L, int H, int D, int causal, float scale, int xcd_order) {
    __shared__ __attribute__((aligned(16))) char smem[2 * LB * ROW + 64 * 4];
    char* sK = smem;
    char* sV = smem + LB * ROW;
    float* csum = reinterpret_cast<float*>(smem + 2 * LB * ROW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qb, bh_;
    long_block_coords(xcd_order, qb, bh_);
    const int b = bh_ / H, h = bh_ - b * H;
    const bf16_t* base = qkv + (long)b * L * ld + h * HD;
    const bf16_t* dob = dout + (long)b * L * lddo + h * HD;
    const bf16_t* ob = o + (long)b * L * ldo + h * HD;
    bf16_t* dbase = dqkv + (long)b * L * lddq + h * HD;
    const int li = lane & 15, g = lane >> 4;
    constexpr int QB = LB * NS;
    if (tid < 64) csum[tid] = 0.f;
    const float c2 = scale * 1.4426950408889634f;
    int i[NS];
    bool iok[NS];
    StripOps q[NS];
    float lse2[NS];
    f32x4 acc[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        i[s] = qb * QB + (wave * NS + s) * 16 + li;
        iok[s] = i[s] < L;
        q[s] = load_strip(base, ld, dob, lddo, ob, ldo, lse + ((long)b * H + h) * L, min(i[s], L - 1), g);
        lse2[s] = q[s].lse * 1.4426950408889634f;
        // delta_i = sum_c dO[i][c] O[i][c], computed here once per query: handed to the dK / dV kernel (launched behind this one),
        // whose every key block would otherwise recompute it for all L queries
        if (delta_out && g == 0 && iok[s]) delta_out[((long)b * H + h) * L + i[s]] = q[s].dl;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[s][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nkb_all = (L + LB - 1) / LB;
    const int nkb = causal ? min(nkb_all, (qb * QB + QB - 1) / LB + 1) : nkb_all;
    BlockRegs nk = load_block(base + D, ld, 0, L, tid), nv = load_block(base + 2 * D, ld, 0, L, tid);
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb) __syncthreads();
        store_block(sK, nk, tid);
        store_block(sV, nv, tid);
        if (kb + 1 < nkb) {
            nk = load_block(base + D, ld, (kb + 1) * LB, L, tid);
            nv = load_block(base + 2 * D, ld, (kb + 1) * LB, L, tid);
        }
        __syncthreads();
        // dS^T / scale of every strip against the 64 keys staged in sK / sV (accumulator layout: key 16t + 4g + r).  A padded
        // query row i >= L repeats row L - 1: finite values, zeroed once at the end, so only keys past L and the causal mask
        // need the per-element mask.
        const bool need_mask = causal || (kb + 1) * LB > L;
        f32x4 ds[NS][4];
        auto p_ds = [&](auto mask_tag) __attribute__((always_inline)) {      // the mask as a compile-time property of the block's code
            constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 kf[2], vf[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    kf[ks] = *reinterpret_cast<const bf16x8*>(sK + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                    vf[ks] = *reinterpret_cast<const bf16x8*>(sV + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    f32x4 p = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], q[s].qf[ks], p, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], q[s].df[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], c2, -lse2[s]));
                        if constexpr (MASK) {
                            const int j = kb * LB + t * 16 + g * 4 + r;
                            const bool ok = (j < L) && (!causal || j <= i[s]);
                            pv = ok ? pv : 0.f;
                        }
                        ds[s][t][r] = pv * (dp[r] - q[s].dl);
                    }
                }
            }
        };
        if (need_mask) p_ds(std::true_type{});
        else p_ds(std::false_type{});
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {                 // dQ^T += K^T dS^T
            bf16x8 sf[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) sf[s] = pack8(ds[s][2 * sidx], ds[s][2 * sidx + 1]);
            const char* krow = sK + (32 * sidx + 4 * g + (li >> 2)) * ROW + (4 * (li & 3)) * 2;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 kf = tr_pair(krow + ct * 32, krow + ct * 32 + 16 * ROW);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, sf[s], acc[s][ct], 0, 0, 0);
            }
        }
    }
    if (bias_grad) __syncthreads();                            // csum zeroed
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[s][ct] = iok[s] ? acc[s][ct] * scale : f32x4{0.f, 0.f, 0.f, 0.f};   // the softmax scale of dS, once; padded queries: exact zeros
        if (iok[s]) {
            bf16_t* qrow = dbase + (long)i[s] * lddq + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                u32x2 pk = {pack_bf2(acc[s][ct][0], acc[s][ct][1]), pack_bf2(acc[s][ct][2], acc[s][ct][3])};
                *reinterpret_cast<u32x2*>(qrow + ct * 16) = pk;
            }
        }
        if (bias_grad) {                                       // block-uniform
            float vals[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) vals[ct * 4 + r] = acc[s][ct][r];
            const float tot = row16_colsum(vals, li);
            const int k = row16_colsum_index(li);
            atomicAdd(&csum[(k >> 2) * 16 + 4 * g + (k & 3)], tot);
        }
    }
    if (bias_grad) {
        __syncthreads();
        if (tid < 64) atomicAdd(bias_grad + h * HD + tid, csum[tid]);
    }
}

// backward B: dK, dV (and their bias column sums), one workgroup per block of 64 NK keys, NK 16-key tiles per wave.
// Here the scores are formed UN-transposed, S[i][j] = Q K^T with the wave's keys on the accumulator columns (K, V rows of
// those keys stay in registers for the whole kernel), so P and dS leave the accumulators already in the layout the second
// products need as B operands (contraction over the queries, in the permuted order the transposed dO^T / Q^T reads match --
// the forward kernel's P V trick with keys and queries swapped): no P / dS images in LDS, only the Q and dO blocks (requested
// one block ahead) and 64 delta / log-sum-exp values.  Every Q / dO fragment read, plain and transposed, serves NK key tiles.
template <int NK>
__global__ __launch_bounds__(256) void attn_bwd_long_dkv_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                                const bf16_t* __restrict__ o, long ldo,
                                                                const bf16_t* __restrict__ dout, long lddo,
                                                                const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                                long lddq, float* __restrict__ bias_grad,
                                                                const float* __restrict__ delta_in, int L, int H, int D,
                                                                int causal, float scale, int xcd_order) {
    __shared__ __attribute__((aligned(16))) char smem[2 * LB * ROW + (2 * LB + 128) * 4];
    char* sQ = smem;
    char* sDO = sQ + LB * ROW;
    float* sLse = reinterpret_cast<float*>(sDO + LB * ROW);
    float* sDelta = sLse + LB;
    float* csum = sDelta + LB;                                  // [2][64]: dk | dv
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kb, bh_;
    long_block_coords(xcd_order, kb, bh_);
    const int b = bh_ / H, h = bh_ - b * H;
    const bf16_t* base = qkv + (long)b * L * ld + h * HD;
    const bf16_t* dob = dout + (long)b * L * lddo + h * HD;
    (void)o; (void)ldo;                                         // (delta arrives from the dQ kernel)
    bf16_t* dbase = dqkv + (long)b * L * lddq + h * HD;
    const float* lse_row = lse + ((long)b * H + h) * L;
    const int li = lane & 15, g = lane >> 4;
    constexpr int KB = LB * NK;                                 // keys per workgroup
    int j[NK];                                                  // this lane's keys (accumulator columns)
    bool jok[NK];
    bf16x8 kf[NK][2], vf[NK][2];                                // B operands: K / V rows of key j
    f32x4 av[NK][4], ak[NK][4];
    if (tid < 128) csum[tid] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        j[kt] = kb * KB + (wave * NK + kt) * 16 + li;
        jok[kt] = j[kt] < L;
        const int jq = min(j[kt], L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kt][ks] = *reinterpret_cast<const bf16x8*>(base + D + (long)jq * ld + ks * 32 + g * 8);
            vf[kt][ks] = *reinterpret_cast<const bf16x8*>(base + 2 * D + (long)jq * ld + ks * 32 + g * 8);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            av[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            ak[kt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int nqb = (L + LB - 1) / LB;
    const int qb0 = causal ? (kb * KB) / LB : 0;
    const float c2 = scale * 1.4426950408889634f;               // scale * log2(e)
    // per-query scalars of the block, one per thread: threads 0 .. 63 the log-sum-exp (stored in log2 units: p = exp2(s c - lse
    // log2 e)), threads 64 .. 127 delta = sum_c dO O as the dQ kernel left it; requested one block ahead like Q and dO
    const float* delta_row = delta_in + ((long)b * H + h) * L;
    auto scalar_of = [&](int qb) __attribute__((always_inline)) -> float {
        if (tid >= 2 * LB) return 0.f;
        const int iq = min(qb * LB + (tid & (LB - 1)), L - 1);
        return tid < LB ? lse_row[iq] * 1.4426950408889634f : delta_row[iq];
    };
    BlockRegs nq = load_block(base, ld, qb0 * LB, L, tid), nd = load_block(dob, lddo, qb0 * LB, L, tid);
    float nsc = scalar_of(qb0);
    const bool keys_past_L = (kb + 1) * KB > L;                 // workgroup-uniform
    for (int qb = qb0; qb < nqb; ++qb) {
        __syncthreads();                                       // previous block's Q / dO / scalars consumed
        store_block(sQ, nq, tid);
        store_block(sDO, nd, tid);
        if (tid < 2 * LB) sLse[tid] = nsc;                      // (sDelta = sLse + LB)
        if (qb + 1 < nqb) {
            nq = load_block(base, ld, (qb + 1) * LB, L, tid);
            nd = load_block(dob, lddo, (qb + 1) * LB, L, tid);
            nsc = scalar_of(qb + 1);
        }
        __syncthreads();                                       // Q / dO / scalars of this block visible
        // one fma + v_exp_f32 per probability; dS without the softmax scale (applied once to the finished dK); the validity /
        // causal mask only in the blocks that need it: the last query block (padded queries repeat row L - 1 and must add
        // nothing to dK / dV), a key block that reaches past L (its column sums must stay zero) and under the causal mask
        const bool need_mask = causal || keys_past_L || (qb + 1) * LB > L;
        f32x4 p[NK][4], ds[NK][4];
        auto p_ds = [&](auto mask_tag) __attribute__((always_inline)) {      // the mask as a compile-time property of the block's code
            constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
            for (int t = 0; t < 4; ++t) {                      // S and dP of queries 16 t .. 16 t + 15 against this wave's keys
                bf16x8 qa[2], da[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    qa[ks] = *reinterpret_cast<const bf16x8*>(sQ + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                    da[ks] = *reinterpret_cast<const bf16x8*>(sDO + (t * 16 + li) * ROW + (ks * 4 + g) * 16);
                }
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + t * 16 + 4 * g);       // lse * log2(e)
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + t * 16 + 4 * g);
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kf[kt][ks], sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[ks], vf[kt][ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#ifdef CE_ABL_NO_EXP
                        float pv = __builtin_fmaf(sc[r], c2, -l4[r]);
#else
                        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c2, -l4[r]));
#endif
                        if constexpr (MASK) {
                            const int i = qb * LB + t * 16 + 4 * g + r;
                            const bool ok = jok[kt] && (i < L) && (!causal || j[kt] <= i);
                            pv = ok ? pv : 0.f;
                        }
                        p[kt][t][r] = pv;
                        ds[kt][t][r] = pv * (dp[r] - d4[r]);
                    }
                }
            }
        };
        if (need_mask) p_ds(std::true_type{});
        else p_ds(std::false_type{});
        // dV^T += dO^T P, dK^T += Q^T dS: contraction element jj of lane group g <-> query 32 s + 16 (jj>>2) + 4 g + (jj&3)
#ifdef CE_ABL_NO_SECOND
        for (int kt = 0; kt < NK; ++kt) for (int t = 0; t < 4; ++t) { av[kt][t] += p[kt][t]; ak[kt][t] += ds[kt][t]; }
#else
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            bf16x8 pf[NK], sf[NK];
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                pf[kt] = pack8(p[kt][2 * sidx], p[kt][2 * sidx + 1]);
                sf[kt] = pack8(ds[kt][2 * sidx], ds[kt][2 * sidx + 1]);
            }
            const int roff = (32 * sidx + 4 * g + (li >> 2)) * ROW + (4 * (li & 3)) * 2;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 dof = tr_pair(sDO + roff + ct * 32, sDO + roff + ct * 32 + 16 * ROW);
                bf16x8 qf = tr_pair(sQ + roff + ct * 32, sQ + roff + ct * 32 + 16 * ROW);
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    av[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf[kt], av[kt][ct], 0, 0, 0);
                    ak[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, sf[kt], ak[kt][ct], 0, 0, 0);
                }
            }
        }
#endif
    }
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) ak[kt][ct] *= scale;    // the softmax scale of dS, once
        if (jok[kt]) {
            bf16_t* krow = dbase + (long)j[kt] * lddq + D + 4 * g;
            bf16_t* vrow = dbase + (long)j[kt] * lddq + 2 * D + 4 * g;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                u32x2 pk = {pack_bf2(ak[kt][ct][0], ak[kt][ct][1]), pack_bf2(ak[kt][ct][2], ak[kt][ct][3])};
                u32x2 pv = {pack_bf2(av[kt][ct][0], av[kt][ct][1]), pack_bf2(av[kt][ct][2], av[kt][ct][3])};
                *reinterpret_cast<u32x2*>(krow + ct * 16) = pk;
                *reinterpret_cast<u32x2*>(vrow + ct * 16) = pv;
            }
        }
        if (bias_grad) {
            float vk[16], vv[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vk[ct * 4 + r] = ak[kt][ct][r];
                    vv[ct * 4 + r] = av[kt][ct][r];
                }
            const float tk = row16_colsum(vk, li), tv = row16_colsum(vv, li);
            const int k = row16_colsum_index(li);
            atomicAdd(&csum[(k >> 2) * 16 + 4 * g + (k & 3)], tk);
            atomicAdd(&csum[64 + (k >> 2) * 16 + 4 * g + (k & 3)], tv);
        }
    }
    if (bias_grad) {
        __syncthreads();
        if (tid < 128) atomicAdd(bias_grad + (1 + (tid >> 6)) * D + h * HD + (tid & 63), csum[tid]);
    }
}

int tiles_for(int L) { return ((L + 31) / 32) * 2; }
int attn_xcd_order() {          // long_block_coords: XCD-owned (sample, head) pairs (default) or the plain launch order (CE_ATTN_XCD=0)
    static const int v = getenv("CE_ATTN_XCD") ? atoi(getenv("CE_ATTN_XCD")) : 1;
    return v;
}

// Per-stream scratch of the long-sequence backward (delta, [B, H, L] floats: 1.2 MB at ViT-L/14@336, B = 32).  Grow-only; kernels
// of one stream run in order, so consecutive calls on a stream may share it.  (hipMallocAsync / hipFreeAsync around the two
// launches measured 25 us slower per call than this.)
struct StreamScratch { float* p = nullptr; size_t cap = 0; };
std::mutex g_scratch_mu;
std::map<hipStream_t, StreamScratch> g_scratch;
float* delta_scratch(hipStream_t s, size_t floats) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    StreamScratch& e = g_scratch[s];
    if (e.cap < floats) {
        if (e.p) hipFree(e.p);                       // (synchronises the device: nothing still reads the old buffer)
        e.p = nullptr; e.cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&e.p), floats * sizeof(float)) != hipSuccess) return nullptr;
        e.cap = floats;
    }
    return e.p;
}

}  // namespace

#define ATTN_DISPATCH(T, CALL)                 \
    switch (T) {                               \
        case 2: CALL(2); break;                \
        case 4: CALL(4); break;                \
        case 6: CALL(6); break;                \
        case 8: CALL(8); break;                \
        default: CE_CHECK_ARG(false, "attention: sequence length %d not supported", L); \
    }

extern "C" int ce_attention_fwd(const void* qkv, long ld, void* o, long ldo, float* lse, const int* cu_seqlens, int B,
                                int L, int H, int causal, void* stream) {
    CE_CHECK_ARG(B > 0 && L > 0 && H > 0, "ce_attention_fwd: empty problem");
    CE_CHECK_ARG(ld % 8 == 0 && ldo % 4 == 0, "ce_attention_fwd: ld must be a multiple of 8, ldo of 4");
    const int D = H * HD;
    if (L > 128) {
        CE_CHECK_ARG(!cu_seqlens, "ce_attention_fwd: packed batches are limited to 128 tokens per sample");
        CE_CHECK_ARG((long)B * H <= 65535, "ce_attention_fwd: B*H = %ld exceeds the grid", (long)B * H);
        hipStream_t sl = (hipStream_t)stream;
        CeProfScope prof(CE_PROF_ATTN_FWD, 4.0 * B * H * (double)L * L * HD, 2.0 * (double)B * L * (4.0 * D), sl);
        // strips per wave (CE_ATTN_NS_FWD = 1 / 2, default 1): a workgroup takes 64 NS queries
        static const int ns = getenv("CE_ATTN_NS_FWD") ? atoi(getenv("CE_ATTN_NS_FWD")) : 1;
        if (ns == 1)
            hipLaunchKernelGGL(attn_fwd_long_kernel<1>, dim3((L + LB - 1) / LB, B * H), dim3(256), 0, sl, (const bf16_t*)qkv, ld,
                               (bf16_t*)o, ldo, lse, L, H, D, causal, 0.125f, attn_xcd_order());
        else
            hipLaunchKernelGGL(attn_fwd_long_kernel<2>, dim3((L + 2 * LB - 1) / (2 * LB), B * H), dim3(256), 0, sl, (const bf16_t*)qkv, ld,
                               (bf16_t*)o, ldo, lse, L, H, D, causal, 0.125f, attn_xcd_order());
        CE_LAUNCH_CHECK();
        return 0;
    }
    const int T = tiles_for(L);
    const int Lp = T * 16;
    int nw = (L + 15) / 16;
    if (nw > 8) nw = 8;
    const size_t lds = 2 * (size_t)Lp * ROW;
    const float scale = 0.125f;  // 1/sqrt(64)
    hipStream_t s = (hipStream_t)stream;
    CeProfScope prof(CE_PROF_ATTN_FWD, 4.0 * B * H * (double)L * L * HD, 2.0 * (double)B * L * (4.0 * D), s);
#define CALL(TT)                                                                                              \
    hipLaunchKernelGGL(attn_fwd_kernel<TT>, dim3(B * H), dim3(64 * nw), lds, s, (const bf16_t*)qkv, ld, (bf16_t*)o, ldo, \
                       lse, cu_seqlens, L, H, D, causal, scale)
    ATTN_DISPATCH(T, CALL);
#undef CALL
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_attention_bwd(const void* qkv, long ld, const void* o, long ldo, const void* dout, long lddo,
                                const float* lse, void* dqkv, long lddq, float* bias_grad, const int* cu_seqlens, int B,
                                int L, int H, int causal, void* stream) {
    CE_CHECK_ARG(B > 0 && L > 0 && H > 0, "ce_attention_bwd: empty problem");
    CE_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0, "ce_attention_bwd: bad leading dimension");
    const int D = H * HD;
    if (L > 128) {
        CE_CHECK_ARG(!cu_seqlens, "ce_attention_bwd: packed batches are limited to 128 tokens per sample");
        CE_CHECK_ARG((long)B * H <= 65535, "ce_attention_bwd: B*H = %ld exceeds the grid", (long)B * H);
        hipStream_t sl = (hipStream_t)stream;
        CeProfScope prof(CE_PROF_ATTN_BWD, 14.0 * B * H * (double)L * L * HD, 2.0 * (double)B * L * (8.0 * D), sl);
        // strips (dQ) / key tiles (dK, dV) per wave: CE_ATTN_NS, CE_ATTN_NK = 1 / 2 (default 2 / 1: see attn_fwd_long_kernel)
        static const int ns = getenv("CE_ATTN_NS") ? atoi(getenv("CE_ATTN_NS")) : 2;
        static const int nk = getenv("CE_ATTN_NK") ? atoi(getenv("CE_ATTN_NK")) : 1;
        const dim3 grid1((L + LB - 1) / LB, B * H), grid2((L + 2 * LB - 1) / (2 * LB), B * H);
        // delta = rowsum(dO O), [B, H, L] floats: written by the dQ kernel, read by the dK / dV kernel behind it on the same stream;
        // one grow-only scratch per stream, so two towers' backward passes on two streams never share it
        float* delta = delta_scratch(sl, (size_t)B * H * L);
        if (!delta) {
            ce_set_error("ce_attention_bwd: no memory for %ld bytes of scratch", (long)B * H * L * 4);
            return -12; /* -ENOMEM */
        }
        if (ns == 1)
            hipLaunchKernelGGL(attn_bwd_long_dq_kernel<1>, grid1, dim3(256), 0, sl, (const bf16_t*)qkv, ld, (const bf16_t*)o, ldo,
                               (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, bias_grad, delta, L, H, D, causal, 0.125f, attn_xcd_order());
        else
            hipLaunchKernelGGL(attn_bwd_long_dq_kernel<2>, grid2, dim3(256), 0, sl, (const bf16_t*)qkv, ld, (const bf16_t*)o, ldo,
                               (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, bias_grad, delta, L, H, D, causal, 0.125f, attn_xcd_order());
        if (nk == 1)
            hipLaunchKernelGGL(attn_bwd_long_dkv_kernel<1>, grid1, dim3(256), 0, sl, (const bf16_t*)qkv, ld, (const bf16_t*)o, ldo,
                               (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, bias_grad, delta, L, H, D, causal, 0.125f, attn_xcd_order());
        else
            hipLaunchKernelGGL(attn_bwd_long_dkv_kernel<2>, grid2, dim3(256), 0, sl, (const bf16_t*)qkv, ld, (const bf16_t*)o, ldo,
                               (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, bias_grad, delta, L, H, D, causal, 0.125f, attn_xcd_order());
        CE_LAUNCH_CHECK();
        return 0;
    }
    const int T = tiles_for(L);
    const int Lp = T * 16;
    int nw = T;
    if (nw > 8) nw = 8;
    const size_t lds = 2 * (size_t)Lp * ROW + 2 * (size_t)Lp * (Lp * 2 + 32);      // (the bias-gradient sums sit in the P image's row pads)
    const float scale = 0.125f;
    hipStream_t s = (hipStream_t)stream;
    CeProfScope prof(CE_PROF_ATTN_BWD, 10.0 * B * H * (double)L * L * HD, 2.0 * (double)B * L * (8.0 * D), s);
#define CALL(TT)                                                                                                   \
    do {                                                                                                           \
        static std::once_flag attr;                                                                                \
        std::call_once(attr, [] {                                                                                  \
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<TT>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                           \
        });                                                                                                        \
        hipLaunchKernelGGL(attn_bwd_kernel<TT>, dim3(B * H), dim3(64 * nw), lds, s, (const bf16_t*)qkv, ld,        \
                           (const bf16_t*)o, ldo, (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, bias_grad, cu_seqlens, L, \
                           H, D, causal, scale);                                                                           \
    } while (0)
    ATTN_DISPATCH(T, CALL);
#undef CALL
    CE_LAUNCH_CHECK();
    return 0;
}
