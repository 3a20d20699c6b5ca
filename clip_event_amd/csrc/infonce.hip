// Fused contrastive head at global-batch scale: scaled similarity tiles, online log-sum-exp, label pick and the
// backward from the saved log-sum-exp, WITHOUT the [B, N*K] logits matrix in HBM (336 MB at N = 4096,
// K = 5).  Replaces, for the 'ce' criterion over the batch, model_clip.py:496-521 (logits) + :633-662 (cross entropy).
//
//   loss += mean_r ( lse_r - s <q_r, k_{y_r}> ),   lse_r = log sum_c exp(s <q_r, k_c>)          q, k already L2-normalised
//   G[r,c] = g/n (exp(s <q_r,k_c> - lse_r) - [c == y_r]);  dq_r = s sum_c G[r,c] k_c;  dk_c = s sum_r G[r,c] q_r;
//   dlogit_scale = sum G[r,c] s <q_r,k_c>
//
// fp32 throughout, products on the fp32 matrix instruction v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 16x below the
// bf16 rate but 3x an LDS-tiled VALU kernel; MI355X_MICROARCH.md "Matrix cores").  One kernel skeleton, three modes:
// a workgroup (4 waves) keeps one 32-row RESIDENT block in LDS and sweeps 32-row STREAMED blocks of the other matrix.
//   * similarity tile: S^T[i][j] = <streamed_i, resident_j>, the streamed rows as the A operand straight from global
//     memory (16 B per lane, L1-resident over the four MFMAs that consume them), the resident rows as the B operand from
//     LDS (row pitch E+4 floats: conflict-free ds_read_b128).  Each wave takes a quarter of E; the four partial tiles are
//     summed through LDS.  Resident index j sits on the lane, streamed index i in the 16 accumulator registers.
//   * mode FWD (resident = queries): online max / sum per lane, partial (max, sum) per split, merged by a second tiny
//     kernel that also picks s <q_r, k_{y_r}> and adds the mean loss.
//   * modes DQ (resident = queries) / DK (resident = keys): G^T in registers is directly the B operand of the gradient
//     product  dres^T[e][j] += sum_i streamed[i][e] G^T[i][j]  (contraction over the accumulator's register index needs no
//     lane movement: cdna_hip_programming.md "An accumulator tile as the next MFMA's operand"); each wave owns a quarter of
//     E; the result leaves through an LDS transpose as whole rows (plain stores when the streamed range is not split,
//     float atomics otherwise).
#include <stdlib.h>

#include <mutex>

#include "common.hpp"
#include "../../include/clip_event_hip.h"

namespace {

constexpr int HB = 32;          // rows per resident / streamed block
enum { MODE_FWD = 0, MODE_DQ = 1, MODE_DK = 2 };

struct HeadArgs {
    const float* res; long ldres; int nres;       // resident matrix (queries for FWD / DQ, keys for DK)
    const float* str; long ldstr; int nstr;       // streamed matrix
    const long* sel;                              // query row gather (nullable): query r = row sel[r] of its matrix
    const long* labels;                           // key index per query SOURCE row (labels[sel[r]])
    const float* lse;                             // [nq]            (DQ / DK)
    const float* logit_scale;
    const float* grad;                            // upstream scalar (DQ / DK)
    float* part;                                  // FWD: [splits][nq][2] partial (max, sum)
    float* dres;                                  // DQ / DK: gradient of the resident matrix [rows, E] (+=)
    float* dls;                                   // DQ: logit_scale gradient (+=)
    int E, nq, splits, blocks_per_split;
};

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // 32x32 C/D row of reg r

// NW waves per workgroup: 4 (E a multiple of 128) or 8 (E a multiple of 256: two waves per SIMD cover the global operand
// loads that feed the MFMAs; each wave then owns an eighth of E)
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void infonce_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = a.E, pitch = E + 4;
    float* res_l = reinterpret_cast<float*>(smem);                       // [32][E+4]
    float* part_l = res_l + HB * pitch;                                  // [NW][32][33]: partial tiles / output transpose
    float* qinfo = part_l + NW * HB * 33;                                 // [2][32]: lse, label (as float bits) of the streamed queries (DK)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * HB;
    const float scale = __expf(*a.logit_scale);

    // ---- resident block -> LDS (gathered through sel when the residents are queries) ----
    for (int idx = tid; idx < HB * (E / 4); idx += 64 * NW) {
        const int r = idx / (E / 4), c = (idx - r * (E / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < a.nres) {
            const long src = (MODE != MODE_DK && a.sel) ? a.sel[r0 + r] : (long)(r0 + r);
            v = *reinterpret_cast<const f32x4*>(a.res + src * a.ldres + c);
        }
        *reinterpret_cast<f32x4*>(res_l + r * pitch + c) = v;
    }
    // per-lane facts about resident row j
    const bool jvalid = r0 + j < a.nres;
    long jsrc = 0, jlabel = -1;
    float jlse = 0.f;
    if (MODE != MODE_DK && jvalid) {
        jsrc = a.sel ? a.sel[r0 + j] : (long)(r0 + j);
        jlabel = a.labels[jsrc];
        if (MODE == MODE_DQ) jlse = a.lse[r0 + j];
    }
    const float coef = (MODE == MODE_FWD) ? 0.f : (*a.grad / (float)a.nq);
    __syncthreads();

    float run_m = -INFINITY, run_l = 0.f;                 // FWD: online statistics of query j over this split
    float dls_acc = 0.f;
    constexpr int GT = 32 / NW;                           // gradient accumulators: up to 1024 / NW / 32 e-tiles of 32 per wave
    f32x16 gacc[MODE == MODE_FWD ? 1 : GT];
    const int etiles = E / (32 * NW);                     // e-tiles of 32 columns per wave (E/NW columns per wave)
    if (MODE != MODE_FWD) {
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[t][r] = 0.f;
    }
    const int e_lo = wave * (E / NW);

    const int sb0 = blockIdx.y * a.blocks_per_split;
    const int sb1 = min(sb0 + a.blocks_per_split, (a.nstr + HB - 1) / HB);
    for (int sb = sb0; sb < sb1; ++sb) {
        const int c0 = sb * HB;
        // streamed row of this lane for the similarity A operand: i = lane & 31
        const bool ivalid = c0 + j < a.nstr;
        long isrc = c0 + j;
        if (MODE == MODE_DK && a.sel && ivalid) isrc = a.sel[c0 + j];
        const float* arow = a.str + isrc * a.ldstr + e_lo + 4 * h;
        // ---- similarity tile, this wave's quarter of E ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* brow = res_l + j * pitch + e_lo + 4 * h;
#pragma unroll 4
        for (int u = 0; u < E / (8 * NW); ++u) {
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (ivalid) av = *reinterpret_cast<const f32x4*>(arow + 8 * u);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 8 * u);
#pragma unroll
            for (int x = 0; x < 4; ++x) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[x], acc, 0, 0, 0);
        }
        // ---- sum the four partial tiles through LDS: part_l[w][row i][col j] ----
        __syncthreads();          // previous iteration's readers of part_l / qinfo writers are done
#pragma unroll
        for (int r = 0; r < 16; ++r) part_l[(wave * HB + row_of(r, h)) * 33 + j] = acc[r];
        if (MODE == MODE_DK && tid < HB) {                // lse / label of the 32 streamed queries
            const bool v = c0 + tid < a.nstr;
            const long src = v ? (a.sel ? a.sel[c0 + tid] : (long)(c0 + tid)) : 0;
            qinfo[tid] = v ? a.lse[c0 + tid] : 0.f;
            qinfo[HB + tid] = v ? (float)a.labels[src] : -1.f;          // key indices < 2^24: exact in fp32
        }
        __syncthreads();
        float S[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = row_of(r, h);
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += part_l[(w * HB + i) * 33 + j];
            S[r] = scale * t;
        }
        if (MODE == MODE_FWD) {
            float mx = run_m;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c0 + row_of(r, h) < a.nstr) mx = fmaxf(mx, S[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c0 + row_of(r, h) < a.nstr) sum += __expf(S[r] - mx);
            sum += __shfl_xor(sum, 32, 64);
            run_l = run_l * __expf(run_m - mx) + sum;      // run_m = -inf on the first block: exp(-inf) = 0
            run_m = mx;
        } else {
            // G^T[i][j] for this lane's resident j and its 16 streamed rows
            float G[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = row_of(r, h);
                float g = 0.f;
                if (MODE == MODE_DQ) {
                    if (jvalid && c0 + i < a.nstr) g = coef * (__expf(S[r] - jlse) - ((long)(c0 + i) == jlabel ? 1.f : 0.f));
                } else {
                    if (jvalid && c0 + i < a.nstr)
                        g = coef * (__expf(S[r] - qinfo[i]) - ((float)(r0 + j) == qinfo[HB + i] ? 1.f : 0.f));
                }
                G[r] = g;
                if (MODE == MODE_DQ) dls_acc += g * S[r];
            }
            // ---- gradient product: dres^T[e][j] += sum_i streamed[i][e] G^T[i][j]; A = streamed[i(r,h)][e0 + (lane&31)] ----
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                if (t < etiles) {
                const int e0 = e_lo + 32 * t + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = c0 + row_of(r, h);
                    float av = 0.f;
                    if (i < a.nstr) {
                        const long src = (MODE == MODE_DK && a.sel) ? a.sel[i] : (long)i;
                        av = a.str[src * a.ldstr + e0];
                    }
                    gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, G[r], gacc[t], 0, 0, 0);
                }
                }
            }
        }
    }

    if (MODE == MODE_FWD) {
        if (wave == 0 && h == 0 && jvalid) {
            float* o = a.part + ((long)blockIdx.y * a.nq + r0 + j) * 2;
            o[0] = run_m;
            o[1] = run_l;
        }
        return;
    }
    // ---- gradient out: transpose each 32(e) x 32(j) tile through LDS, then whole 128-byte row segments ----
    if (MODE == MODE_DQ) {
        dls_acc = wave_sum(dls_acc);
        if (lane == 0 && wave == 0) atomicAdd(a.dls, dls_acc);        // every wave holds the same tile sums: count once
    }
    __syncthreads();
    float* tile = part_l + wave * HB * 33;                            // [j][e'] for this wave
#pragma unroll
    for (int t = 0; t < GT; ++t) {
        if (t >= etiles) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[j * 33 + row_of(r, h)] = gacc[t][r] * scale;
        // lanes: 2 rows per pass (row = pass*2 + h), 32 columns
        for (int pass = 0; pass < 16; ++pass) {
            const int jr = pass * 2 + h;
            if (r0 + jr < a.nres) {
                const long dst = (MODE == MODE_DQ && a.sel) ? a.sel[r0 + jr] : (long)(r0 + jr);
                float* o = a.dres + dst * (long)E + e_lo + 32 * t + j;
                const float v = tile[jr * 33 + j];
                if (a.splits == 1) *o += v;
                else atomicAdd(o, v);
            }
        }
    }
}

// merge the per-split (max, sum), pick the labelled logit, add the mean loss: one wave per query
__global__ __launch_bounds__(256) void infonce_merge_kernel(HeadArgs a, float* __restrict__ lse_out, float* __restrict__ loss) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.nq) return;
    float m = -INFINITY;
    for (int s = 0; s < a.splits; ++s) m = fmaxf(m, a.part[((long)s * a.nq + r) * 2]);
    float l = 0.f;
    for (int s = 0; s < a.splits; ++s) {
        const float* p = a.part + ((long)s * a.nq + r) * 2;
        l += p[1] * __expf(p[0] - m);
    }
    const float lse = m + __logf(l);
    const long src = a.sel ? a.sel[r] : (long)r;
    const long y = a.labels[src];
    const float* q = a.res + src * a.ldres;
    const float* k = a.str + y * a.ldstr;
    float d = 0.f;
    for (int c = lane; c < a.E; c += 64) d += q[c] * k[c];
    d = wave_sum(d);
    if (lane == 0) {
        lse_out[r] = lse;
        atomicAdd(loss, (lse - __expf(*a.logit_scale) * d) / (float)a.nq);
    }
}

size_t head_lds_bytes(int E, int nw) { return (size_t)(HB * (E + 4) + nw * HB * 33 + 2 * HB) * 4; }

template <int MODE>
int launch_head(HeadArgs& a, hipStream_t s) {
    static std::once_flag flag;
    std::call_once(flag, [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(infonce_kernel<MODE, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)head_lds_bytes(1024, 4));
        hipFuncSetAttribute(reinterpret_cast<const void*>(infonce_kernel<MODE, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)head_lds_bytes(768, 8));      // E = 1024 with eight partial tiles exceeds 160 KiB
    });
    const int rb = ce_div_up(a.nres, HB);
    static const int force_nw = getenv("CE_HEAD_WAVES") ? atoi(getenv("CE_HEAD_WAVES")) : 0;
    const bool eight = a.E % 256 == 0 && a.E <= 768 && force_nw != 4;
    if (eight) hipLaunchKernelGGL((infonce_kernel<MODE, 8>), dim3(rb, a.splits), dim3(512), head_lds_bytes(a.E, 8), s, a);
    else hipLaunchKernelGGL((infonce_kernel<MODE, 4>), dim3(rb, a.splits), dim3(256), head_lds_bytes(a.E, 4), s, a);
    return 0;
}

int pick_splits(int nres, int nstr, int cap) {
    const int rb = ce_div_up(nres, HB), sblocks = ce_div_up(nstr, HB);
    int splits = ce_div_up(512, rb);            // about two workgroups per CU
    if (splits > sblocks) splits = sblocks;
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    return splits;
}

int check_common(const char* what, const float* q, long ldq, int nq, const float* k, long ldk, int nk, int E,
                 const float* logit_scale, const int64_t* labels) {
    CE_CHECK_ARG(q && k && logit_scale && labels, "%s: null buffer", what);
    CE_CHECK_ARG(nq > 0 && nk > 0, "%s: empty problem", what);
    CE_CHECK_ARG(E >= 128 && E % 128 == 0 && E <= 1024, "%s: E must be a multiple of 128 in 128..1024 (got %d)", what, E);
    CE_CHECK_ARG(ldq >= E && ldk >= E && ldq % 4 == 0 && ldk % 4 == 0, "%s: leading dimensions must be >= E and multiples of 4", what);
    CE_CHECK_ARG(nk < (1 << 24), "%s: more than 2^24 keys", what);
    return 0;
}

}  // namespace

extern "C" size_t ce_infonce_workspace_bytes(int nq) { return (size_t)CE_INFONCE_MAX_SPLITS * nq * 2 * sizeof(float); }

extern "C" int ce_infonce_fwd(const float* q, long ldq, const int64_t* sel, int nq, const float* k, long ldk, int nk, int E,
                              const float* logit_scale, const int64_t* labels, float* lse, float* loss, void* workspace,
                              void* stream) {
    if (int rc = check_common("ce_infonce_fwd", q, ldq, nq, k, ldk, nk, E, logit_scale, labels)) return rc;
    CE_CHECK_ARG(lse && loss && workspace, "ce_infonce_fwd: null output");
    HeadArgs a{};
    a.res = q; a.ldres = ldq; a.nres = nq; a.str = k; a.ldstr = ldk; a.nstr = nk;
    a.sel = (const long*)sel; a.labels = (const long*)labels; a.logit_scale = logit_scale;
    a.part = (float*)workspace; a.E = E; a.nq = nq;
    a.splits = pick_splits(nq, nk, CE_INFONCE_MAX_SPLITS);
    a.blocks_per_split = ce_div_up(ce_div_up(nk, HB), a.splits);
    a.splits = ce_div_up(ce_div_up(nk, HB), a.blocks_per_split);
    hipStream_t s = (hipStream_t)stream;
    launch_head<MODE_FWD>(a, s);
    CE_LAUNCH_CHECK();
    hipLaunchKernelGGL(infonce_merge_kernel, dim3(ce_div_up(nq, 4)), dim3(256), 0, s, a, lse, loss);
    CE_LAUNCH_CHECK();
    return 0;
}

extern "C" int ce_infonce_bwd(const float* q, long ldq, const int64_t* sel, int nq, const float* k, long ldk, int nk, int E,
                              const float* logit_scale, const int64_t* labels, const float* lse, const float* grad, float* dq,
                              float* dk, float* dlogit_scale, void* stream) {
    if (int rc = check_common("ce_infonce_bwd", q, ldq, nq, k, ldk, nk, E, logit_scale, labels)) return rc;
    CE_CHECK_ARG(lse && grad && dq && dk && dlogit_scale, "ce_infonce_bwd: null buffer");
    hipStream_t s = (hipStream_t)stream;
    HeadArgs a{};
    a.sel = (const long*)sel; a.labels = (const long*)labels; a.logit_scale = logit_scale; a.lse = lse; a.grad = grad;
    a.E = E; a.nq = nq; a.dls = dlogit_scale;
    // dq: resident = queries, streamed = keys
    a.res = q; a.ldres = ldq; a.nres = nq; a.str = k; a.ldstr = ldk; a.nstr = nk; a.dres = dq;
    a.splits = pick_splits(nq, nk, 1 << 20);
    a.blocks_per_split = ce_div_up(ce_div_up(nk, HB), a.splits);
    a.splits = ce_div_up(ce_div_up(nk, HB), a.blocks_per_split);
    launch_head<MODE_DQ>(a, s);
    CE_LAUNCH_CHECK();
    // dk: resident = keys, streamed = queries
    a.res = k; a.ldres = ldk; a.nres = nk; a.str = q; a.ldstr = ldq; a.nstr = nq; a.dres = dk;
    a.splits = pick_splits(nk, nq, 1 << 20);
    a.blocks_per_split = ce_div_up(ce_div_up(nq, HB), a.splits);
    a.splits = ce_div_up(ce_div_up(nq, HB), a.blocks_per_split);
    launch_head<MODE_DK>(a, s);
    CE_LAUNCH_CHECK();
    return 0;
}
