// Row-stationary streaming core of the in-batch sampled-softmax scorer (forward AND flash-style backward).
// Reference: ItemRetrievalScorer.call_outputs (merlin/models/tf/blocks/retrieval/base.py:283-429),
// ContrastiveOutput.outputs (tf/outputs/contrastive.py:276-344), rescore_false_negatives
// (tf/utils/tf_utils.py:126-154), CategoricalCrossEntropy(from_logits=True) (tf/losses/listwise.py:38-52) and
// their gradients under GradientTape (tf/models/base.py:1121-1174).
//
// Structure (gfx950, v_mfma_f32_32x32x2_f32, exact fp32):
//   * a workgroup (8 wavefronts) owns 256 rows of the STATIONARY matrix X; each wavefront keeps its 32 rows as the
//     MFMA B operand in E/2 registers for the whole kernel (E <= 128);
//   * the STREAMED matrix Y arrives in BN-row tiles by direct-to-LDS DMA (global_load_lds_dwordx4: no staging
//     registers, no ds_write pass), double-buffered, ONE barrier per tile; the LDS image is chunk-swizzled by
//     permuting the per-lane SOURCE address, so ds_read_b128 (GEMM 1) and ds_read_b32 (GEMM 2) are conflict-free;
//   * GEMM 1, per 32-row block of the tile: S^T[j, x] = sum_e Y[j, e] X[x, e] as one k-ascending fmaf chain per
//     score.  In the C layout of the MFMA a lane then holds ONE stationary row x (= lane & 31) and 16 streamed
//     rows j = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): the online log-sum-exp state is one (m, s) pair per lane;
//   * GEMM 2 (gradient modes): out[x, :] += sum_j P[x, j] Y[j, :].  The 16 probabilities a lane holds are exactly
//     the A operand of the next MFMA (k-slot = lane >> 5, step r): no LDS round trip, no transpose.
//
// With X = q, Y = items this is the forward (+ dq); with X = items, Y = q it is ditem.  Nothing of size B x Nn is
// ever written: algorithmic bytes = 4 (B E + Nn E) + ids per pass.
#include "mh_common.h"

#include <math.h>

#include <cstring>
#include <type_traits>

namespace {

constexpr int SW = 8;        // wavefronts per workgroup
constexpr int SX = SW * 32;  // stationary rows per workgroup
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;  // finite "minus infinity" in the base-2 domain
constexpr float LAZY_THR = 16.f;     // FWD_GRAD: rescale only when a tile exceeds the reference max by 2^16

enum { SM_FWD = 0, SM_GRAD = 1, SM_FWD_GRAD = 2, SM_FILTER = 3 };

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct StreamArgs {
    const float* X;  // stationary [Nx, E]
    const float* Y;  // streamed   [Ny, E]
    int64_t Nx, Ny;
    const void* x_ids;  // [Nx] or NULL
    const void* y_ids;  // [Ny] or NULL
    const float* lse;   // GRAD: natural-log lse of the softmax rows (stationary side, or streamed side if LSE_STREAM)
    const float* pos;   // FWD_GRAD: positive scores <q, item> [Nx]
    float invT, fns, gscale;
    float* logits;  // FWD: optional [Nx, ld_logits], column 1 + j
    int64_t ld_logits;
    float* part_m;  // FWD / FWD_GRAD: [nsplit, Nx] running max (base 2)
    float* part_s;  //                 [nsplit, Nx] sum of 2^(z2 - m)
    float* opart;   // GRAD / FWD_GRAD: [nsplit, Nx, E] partial outputs
    int bn;               // streamed rows per LDS tile (multiple of 32; 64 or 128)
    int tiles_per_split;  // tiles of bn rows per blockIdx.y
    int prio;             // 1: the second half of the workgroup's wavefronts runs at s_setprio 1
    // logQ sampling correction (outputs/contrastive.py:309-319, transforms/bias.py:238-254): score -= x_corr[x] + y_corr[j]
    // (either may be NULL); corr_after_mask = 1 applies it AFTER the false-negative rescoring (the `post` block form)
    const float* x_corr;
    const float* y_corr;
    int corr_after_mask;
    // SM_FILTER (top-k threshold filter, mh_topk.hip): scores >= tau[x] are appended to the row's compact list
    const float* tau;  // [Nx] current k-th best score of every stationary (query) row
    int* cnt;          // [Nx] entries in the row's list
    float* cs;         // [Nx, cap] scores
    int32_t* ci;       // [Nx, cap] candidate indices (idx0 + streamed row)
    int cap;
    int64_t idx0;
};

// Lane id recomputed on the spot (volatile asm is neither CSE'd nor hoisted): the DMA issue code at the top of a tile
// then keeps no per-lane address state alive across the unit loop.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int E>
__device__ __forceinline__ int swz(int r) {
    return (E >= 64) ? (r & 15) : ((r >> 1) & 7);
}

// One BN x E tile (rows row0 .. row0+BN of Y, clamped to the last valid row) -> LDS, chunk-swizzled.
// LDS chunk position p (16 bytes each) holds chunk (p % CPR) ^ swz(row) of row p / CPR.
template <int E>
__device__ __forceinline__ void issue_tile(const float* __restrict__ Y, int64_t Ny, int64_t row0, int bn,
                                           float* lds_tile, int wave) {
    constexpr int CPR = E / 4;
    const int lane = fresh_lane();
    const int n_instr = bn * CPR / 64;  // wave-instructions of 1 KiB
    for (int i = wave; i < n_instr; i += SW) {
        const int p = i * 64 + lane;
        const int r = p / CPR, slot = p % CPR;
        const int c = slot ^ swz<E>(r);
        int64_t row = row0 + r;
        if (row > Ny - 1) row = Ny - 1;
        const float* g = Y + row * E + c * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds_tile + i * 256), 16, 0, 0);
    }
}

// n 4-byte words src[word0 .. word0+n) (clamped to last_word) -> LDS, 64 words per wave-instruction
__device__ __forceinline__ void issue_words(const void* __restrict__ src, int64_t word0, int64_t last_word, int n,
                                            void* lds_dst, int wave, int wave_shift) {
    const int n_instr = n / 64;
    const int lane = fresh_lane();
    for (int i = (wave + wave_shift) & (SW - 1); i < n_instr; i += SW) {
        int64_t w = word0 + i * 64 + lane;
        if (w > last_word) w = last_word;
        const uint32_t* g = static_cast<const uint32_t*>(src) + w;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(static_cast<uint32_t*>(lds_dst) + i * 64),
                                         4, 0, 0);
    }
}

// LDS accessed by 32-bit byte address in address space 3 (integer XOR on the address must not decay to flat loads)
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ float lds_f32(uint32_t addr) { return *reinterpret_cast<lds_cfloat*>(addr); }

// sel(v, slot): component `slot` of a 16-byte chunk
__device__ __forceinline__ float sel4(const f32x4 v, int slot) {
    const float lo = (slot & 1) ? v.y : v.x;
    const float hi = (slot & 1) ? v.w : v.z;
    return (slot & 2) ? hi : lo;
}

// chunk-swizzle pieces of the GEMM 2 operand address: chunk position = LANE ^ CT (see the kernel)
template <int E>
__device__ __forceinline__ int g2_lane_bits(int slot, int q2) {
    return (E >= 64) ? ((slot << 2) | q2) : (q2 ^ (slot << 1));
}
template <int E>
constexpr int g2_ct_bits(int tn, int reg) {
    return (E >= 64) ? ((tn << 2) | reg) : ((tn << 2) | (reg >> 1));
}

template <int MODE, int E, typename IdT, bool HAS_IDS, bool LSE_STREAM>
__global__ __launch_bounds__(SW * 64, 2) void stream_kernel(const StreamArgs a) {
    constexpr int TN = E / 16;  // 16-column tiles of GEMM 2
    constexpr int NC = E / 4;   // 16-byte chunks per row = MFMA steps of GEMM 1
    constexpr int RB = E * 4;   // bytes per row
    constexpr int IDW = sizeof(IdT) / 4;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int bn = a.bn;
    // byte offsets into smem (tiles first: their 8 KiB-aligned bases let addresses be formed with XOR)
    const int tile_bytes = bn * RB;
    const int ids_off = 2 * tile_bytes;
    const int aux_off = ids_off + 2 * bn * (int)sizeof(IdT);
    const int aux2_off = aux_off + 2 * bn * 4;  // y_corr of the streamed rows
    char* const smem_b = reinterpret_cast<char*>(smem);
    const uint32_t smem_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;  // LDS byte address of smem

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably uniform (LDS-DMA base)
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, slot = lane >> 4;
    const int64_t x0 = (int64_t)blockIdx.x * SX + wave * 32;
    const int split = blockIdx.y;
    const int nt_all = (int)((a.Ny + bn - 1) / bn);
    const int t_beg = split * a.tiles_per_split;
    const int t_end = (t_beg + a.tiles_per_split < nt_all) ? t_beg + a.tiles_per_split : nt_all;

    if (a.prio && wave >= SW / 2) __builtin_amdgcn_s_setprio(1);

    // ---- first tile in flight, then the stationary fragments ------------------------------------------------
    if (t_beg < t_end) {
        issue_tile<E>(a.Y, a.Ny, (int64_t)t_beg * bn, bn, reinterpret_cast<float*>(smem_b), wave);
        if (HAS_IDS)
            issue_words(a.y_ids, (int64_t)t_beg * bn * IDW, a.Ny * IDW - 1, bn * IDW, smem_b + ids_off, wave, 1);
        if (MODE == SM_GRAD && LSE_STREAM) issue_words(a.lse, (int64_t)t_beg * bn, a.Ny - 1, bn, smem_b + aux_off, wave, 3);
        if (MODE != SM_FILTER && a.y_corr) issue_words(a.y_corr, (int64_t)t_beg * bn, a.Ny - 1, bn, smem_b + aux2_off, wave, 5);
    }
    // two sets of 16 stationary rows per wavefront: lane (l15, slot) holds X[x0 + 16 q + l15][4 c + slot], c < NC
    float xf[2][NC];
    bool xvalid[2];
    IdT x_id[2];
    float lse2_x[2], m_run[2], s_run[2], tau_x[2], xc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int64_t xrow = x0 + 16 * q + l15;
        xvalid[q] = xrow < a.Nx;
        if (!xvalid[q]) xrow = a.Nx - 1;
        const float* xp = a.X + xrow * E;
#pragma unroll
        for (int c = 0; c < NC; ++c) xf[q][c] = xp[4 * c + slot];  // dword loads straight into place (16-byte loads + select
                                                                   // would hold 4x the registers in flight and spill)
        x_id[q] = 0;
        if (HAS_IDS) x_id[q] = static_cast<const IdT*>(a.x_ids)[xrow];
        lse2_x[q] = 0.f;
        if (MODE == SM_GRAD && !LSE_STREAM) lse2_x[q] = a.lse[xrow] * LOG2E;
        m_run[q] = NEG_BIG;
        s_run[q] = 0.f;
        tau_x[q] = 0.f;
        xc[q] = (MODE != SM_FILTER && a.x_corr) ? a.x_corr[xrow] : 0.f;
        if (MODE == SM_FILTER) tau_x[q] = xvalid[q] ? a.tau[xrow] : INFINITY;
        if (MODE == SM_FWD_GRAD) m_run[q] = a.pos[xrow] * a.invT * LOG2E;  // the reference max starts at the positive logit
    }
    const float scale2 = a.invT * LOG2E;
    f32x4 o[2][TN];
    if (MODE == SM_GRAD || MODE == SM_FWD_GRAD) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) o[q][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // per-lane LDS byte offsets inside a 16-row unit (unit bases are multiples of 16 RB: bits 4..8 are free for XOR)
    // GEMM 1, A operand Y[l15][4 c + slot]:      a_lane ^ (c << 4)
    const int a_lane = l15 * RB + (swz<E>(l15) << 4) + slot * 4;
    // GEMM 2, B operand Y[4 slot + reg][16 tn + l15]: reg * RB + (b_lane ^ (CT(tn, reg) << 4))
    const int b_lane = 4 * slot * RB + (g2_lane_bits<E>(slot, l15 >> 2) << 4) + (l15 & 3) * 4;

    __syncthreads();  // vmcnt(0) + barrier: tile t_beg has landed for every wavefront

    for (int t = t_beg; t < t_end; ++t) {
        const int odd = (t - t_beg) & 1;
        if (t + 1 < t_end) {
            issue_tile<E>(a.Y, a.Ny, (int64_t)(t + 1) * bn, bn, reinterpret_cast<float*>(smem_b + (odd ^ 1) * tile_bytes), wave);
            if (HAS_IDS)
                issue_words(a.y_ids, (int64_t)(t + 1) * bn * IDW, a.Ny * IDW - 1, bn * IDW,
                            smem_b + ids_off + (odd ^ 1) * bn * (int)sizeof(IdT), wave, 1);
            if (MODE == SM_GRAD && LSE_STREAM)
                issue_words(a.lse, (int64_t)(t + 1) * bn, a.Ny - 1, bn, smem_b + aux_off + (odd ^ 1) * bn * 4, wave, 3);
            if (MODE != SM_FILTER && a.y_corr)
                issue_words(a.y_corr, (int64_t)(t + 1) * bn, a.Ny - 1, bn, smem_b + aux2_off + (odd ^ 1) * bn * 4, wave, 5);
        }
        const int64_t j_tile = (int64_t)t * bn;
        const int nvalid = (j_tile + bn <= a.Ny) ? bn : (int)(a.Ny - j_tile);  // valid streamed rows of this tile
        const int nunits = (nvalid + 15) >> 4;
        const uint32_t tile_a = smem_a + odd * tile_bytes;
        const IdT* ids = reinterpret_cast<const IdT*>(smem_b + ids_off + odd * bn * (int)sizeof(IdT));
        const float* aux = reinterpret_cast<const float*>(smem_b + aux_off + odd * bn * 4);
        const float* aux2 = reinterpret_cast<const float*>(smem_b + aux2_off + odd * bn * 4);
        const bool has_corr = (MODE != SM_FILTER) && (a.x_corr != nullptr || a.y_corr != nullptr);  // wave-uniform

        for (int ju = 0; ju < nunits; ++ju) {
            const uint32_t ub = tile_a + ju * 16 * RB;
            // ---- GEMM 1: S^T unit [16 j, 32 x] = Y[ju*16 .., :] X^T; one k-ascending chain per score, two row sets ------
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const uint32_t ua = ub + a_lane;
            // operand reads run one window of GW steps ahead of the MFMAs; the scheduling barriers bound how far hipcc
            // hoists LDS reads (left alone it fills every free register with them and then spills)
            constexpr int GW = (MODE == SM_FWD_GRAD && sizeof(IdT) == 8) ? 4 : ((NC >= 8) ? 8 : NC), NW = NC / GW;
            float av[2][GW];
#pragma unroll
            for (int i = 0; i < GW; ++i)
                av[0][i] = lds_f32(ua ^ (uint32_t)(i << 4));
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (w + 1 < NW) {
#pragma unroll
                    for (int i = 0; i < GW; ++i)
                        av[(w + 1) & 1][i] = lds_f32(ua ^ (uint32_t)(((w + 1) * GW + i) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GW; ++i) {
                    acc0 = mfma16(av[w & 1][i], xf[0][w * GW + i], acc0);
                    acc1 = mfma16(av[w & 1][i], xf[1][w * GW + i], acc1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: this lane holds x = 16 q + l15 and j = ju*16 + 4 slot + reg ----------------------------------
            const int jl0 = ju * 16 + 4 * slot;
            if (MODE == SM_FILTER) {  // rare survivors (k / n_seen of the scores): one atomic each
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 acc = q ? acc1 : acc0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int jl = jl0 + r;
                        if (acc[r] >= tau_x[q] && jl < nvalid) {
                            const int64_t row = x0 + 16 * q + l15;
                            const int pos = atomicAdd(&a.cnt[row], 1);
                            if (pos < a.cap) {
                                a.cs[row * a.cap + pos] = acc[r];
                                a.ci[row * a.cap + pos] = (int32_t)(a.idx0 + j_tile + jl);
                            }
                        }
                    }
                }
                continue;
            }
            bool msk[4];
            float lsej[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                msk[r] = false;
                lsej[r] = 0.f;
                if (MODE == SM_GRAD && LSE_STREAM) lsej[r] = aux[jl0 + r] * LOG2E;
            }
            IdT yid[4];
            if (HAS_IDS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yid[r] = ids[jl0 + r];
            }
            float yc[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_corr && a.y_corr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yc[r] = aux2[jl0 + r];
            }
            float p[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 acc = q ? acc1 : acc0;
                float post[4] = {0.f, 0.f, 0.f, 0.f};  // correction applied after the mask (`post` block form)
                if (has_corr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float c = xc[q] + yc[r];
                        if (a.corr_after_mask) post[r] = c; else acc[r] -= c;
                    }
                }
                float t2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jl = jl0 + r;
                    bool masked = false;
                    if (HAS_IDS) masked = (yid[r] == x_id[q]);
                    msk[r] = masked;
                    float v2;
                    if (MODE == SM_FWD) {
                        const float z = ((masked ? a.fns : acc[r]) - post[r]) * a.invT;
                        if (a.logits != nullptr) {
                            if (xvalid[q] && jl < nvalid)
                                a.logits[(x0 + 16 * q + l15) * a.ld_logits + 1 + j_tile + jl] = z;
                        }
                        v2 = z * LOG2E;
                    } else {
                        v2 = ((masked ? a.fns : acc[r]) - post[r]) * scale2;
                    }
                    if (nvalid < bn && jl >= nvalid) v2 = -INFINITY;  // only the last tile of Y can be partial (uniform test first)
                    t2[r] = v2;
                }
                if (MODE == SM_FWD) {
                    const float tmax = fmaxf(fmaxf(t2[0], t2[1]), fmaxf(t2[2], t2[3]));
                    // the running maximum settles after the first tiles: the rescale (a quarter-rate exp2 + a multiply per
                    // unit) runs only when some lane's maximum moves; exp2(0) = 1 makes skipping it exact
                    if (__any(tmax > m_run[q])) {
                        const float m_new = fmaxf(m_run[q], tmax);
                        s_run[q] *= fast_exp2(m_run[q] - m_new);
                        m_run[q] = m_new;
                    }
                    float s_new = s_run[q];
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_new += fast_exp2(t2[r] - m_run[q]);
                    s_run[q] = s_new;
                } else if (MODE == SM_GRAD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float l2 = LSE_STREAM ? lsej[r] : lse2_x[q];
                        const float e = fast_exp2(t2[r] - l2) * a.gscale;  // t2 = -inf on invalid rows -> 0
                        p[q][r] = msk[r] ? 0.f : e;
                    }
                } else {  // SM_FWD_GRAD: lazy reference max shared by the four k-slot lanes of a row
                    float tmax = fmaxf(fmaxf(t2[0], t2[1]), fmaxf(t2[2], t2[3]));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    if (__any(tmax > m_run[q] + LAZY_THR)) {
                        const float m_new = (tmax > m_run[q] + LAZY_THR) ? tmax : m_run[q];
                        const float f = fast_exp2(m_run[q] - m_new);
                        s_run[q] *= f;
                        m_run[q] = m_new;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float fr = __shfl(f, 4 * slot + r);  // factor of the row this accumulator element holds
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) o[q][tn][r] *= fr;
                        }
                    }
                    float s_add = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = fast_exp2(t2[r] - m_run[q]);
                        s_add += e;  // rescored false negatives stay in the denominator
                        p[q][r] = msk[r] ? 0.f : e;
                    }
                    s_run[q] += s_add;
                }
            }
            if (MODE == SM_FWD) continue;
            // ---- GEMM 2: o[x, e] += sum_j p[x, j] Y[j, e]; step r contracts j = 4 slot + r (the C layout of GEMM 1) ------
            const uint32_t ubl = ub + b_lane;
            float bv[2][TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[0][tn] = lds_f32(ubl ^ (uint32_t)(g2_ct_bits<E>(tn, 0) << 4));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r + 1 < 4) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        bv[(r + 1) & 1][tn] = lds_f32((ubl ^ (uint32_t)(g2_ct_bits<E>(tn, r + 1) << 4)) + (r + 1) * RB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    o[0][tn] = mfma16(p[0][r], bv[r & 1][tn], o[0][tn]);
                    o[1][tn] = mfma16(p[1][r], bv[r & 1][tn], o[1][tn]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // every wavefront is done with this tile; the next one (vmcnt(0)) has landed
    }

    // ---- results ---------------------------------------------------------------------------------------------
    if (MODE == SM_FWD || MODE == SM_FWD_GRAD) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float mm = m_run[q], ss = s_run[q];
            if (MODE == SM_FWD) {
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float mo = __shfl_xor(mm, off), so = __shfl_xor(ss, off);
                    const float M = fmaxf(mm, mo);
                    ss = ss * fast_exp2(mm - M) + so * fast_exp2(mo - M);
                    mm = M;
                }
            } else {  // the four lanes of a row share m_run
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
            }
            if (slot == 0 && xvalid[q]) {
                a.part_m[(int64_t)split * a.Nx + x0 + 16 * q + l15] = mm;
                a.part_s[(int64_t)split * a.Nx + x0 + 16 * q + l15] = ss;
            }
        }
    }
    if (MODE == SM_GRAD || MODE == SM_FWD_GRAD) {
        // C layout of GEMM 2: column e = 16 tn + l15, row x = 16 q + 4 slot + reg
        float* op = a.opart + (int64_t)split * a.Nx * E;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = x0 + 16 * q + 4 * slot + r;
                if (row < a.Nx) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) op[row * E + tn * 16 + l15] = o[q][tn][r];
                }
            }
    }
}

// ---- combine kernels ------------------------------------------------------------------------------------------
// forward: lse / loss from the per-split partials and the positive logit
__global__ __launch_bounds__(256) void fwd_finalize_kernel(const float* __restrict__ pos, int64_t B, int nsplit,
                                                          const float* __restrict__ part_m,
                                                          const float* __restrict__ part_s, float invT,
                                                          float* __restrict__ logits, int64_t ld_logits,
                                                          float* __restrict__ loss, float* __restrict__ lse) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float z0 = pos[row] * invT;
    const float z0_2 = z0 * LOG2E;
    float M = z0_2;
    for (int k = 0; k < nsplit; ++k) M = fmaxf(M, part_m[(int64_t)k * B + row]);
    float S = fast_exp2(z0_2 - M);
    for (int k = 0; k < nsplit; ++k) S += part_s[(int64_t)k * B + row] * fast_exp2(part_m[(int64_t)k * B + row] - M);
    const float l = (M + log2f(S)) * LN2;
    if (logits) logits[row * ld_logits] = z0;
    if (lse) lse[row] = l;
    if (loss) loss[row] = l - z0;
}

// FWD_GRAD: dq = g * sum_k O_k 2^(m_k - lse2) + ds0 * item ;  ditem = ds0 * q ;  ds0 = (softmax_0 - 1) g
__global__ __launch_bounds__(256) void fwd_grad_combine_kernel(const float* __restrict__ q, const float* __restrict__ item,
                                                              const float* __restrict__ pos, int64_t B, int E, int nsplit,
                                                              const float* __restrict__ part_m,
                                                              const float* __restrict__ part_s,
                                                              const float* __restrict__ opart, float invT, float g,
                                                              float* __restrict__ loss, float* __restrict__ lse,
                                                              float* __restrict__ dq, float* __restrict__ ditem) {
    const int e4 = E / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * e4) return;
    const int64_t row = idx / e4;
    const int c = (int)(idx - row * e4);
    const float z0 = pos[row] * invT;
    const float z0_2 = z0 * LOG2E;
    float M = z0_2;
    for (int k = 0; k < nsplit; ++k) M = fmaxf(M, part_m[(int64_t)k * B + row]);
    float S = fast_exp2(z0_2 - M);
    for (int k = 0; k < nsplit; ++k) S += part_s[(int64_t)k * B + row] * fast_exp2(part_m[(int64_t)k * B + row] - M);
    const float lse2 = M + log2f(S);
    if (c == 0) {
        const float l = lse2 * LN2;
        if (lse) lse[row] = l;
        if (loss) loss[row] = l - z0;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsplit; ++k) {
        const float w = fast_exp2(part_m[(int64_t)k * B + row] - lse2) * g;
        const f32x4 v = *reinterpret_cast<const f32x4*>(opart + ((int64_t)k * B + row) * E + 4 * c);
        acc += v * w;
    }
    const float ds0 = (fast_exp2(z0_2 - lse2) - 1.f) * g;
    const f32x4 iv = *reinterpret_cast<const f32x4*>(item + row * E + 4 * c);
    *reinterpret_cast<f32x4*>(dq + row * E + 4 * c) = acc + iv * ds0;
    if (ditem) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + row * E + 4 * c);
        *reinterpret_cast<f32x4*>(ditem + row * E + 4 * c) = qv * ds0;
    }
}

// GRAD: out = sum_k part_k (+ ds0 * other  when pos != NULL: the positive column of the row side)
__global__ __launch_bounds__(256) void grad_combine_kernel(const float* __restrict__ opart, int64_t N, int E, int nsplit,
                                                          const float* __restrict__ pos, const float* __restrict__ lse,
                                                          const float* __restrict__ other, const float* __restrict__ self,
                                                          float invT, float g, float* __restrict__ out,
                                                          float* __restrict__ out_pos) {
    const int e4 = E / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * e4) return;
    const int64_t row = idx / e4;
    const int c = (int)(idx - row * e4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsplit; ++k) acc += *reinterpret_cast<const f32x4*>(opart + ((int64_t)k * N + row) * E + 4 * c);
    if (pos) {
        const float ds0 = (expf(pos[row] * invT - lse[row]) - 1.f) * g;
        acc += *reinterpret_cast<const f32x4*>(other + row * E + 4 * c) * ds0;
        if (out_pos) *reinterpret_cast<f32x4*>(out_pos + row * E + 4 * c) = *reinterpret_cast<const f32x4*>(self + row * E + 4 * c) * ds0;
    }
    *reinterpret_cast<f32x4*>(out + row * E + 4 * c) = acc;
}

// rows of src [N, E] -> dst [N, Ep] zero-padded (E < Ep): scores are unchanged bit for bit (fma(0, 0, acc) == acc)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int64_t N, int E, int Ep,
                                                      float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * Ep) return;
    const int64_t row = idx / Ep;
    const int e = (int)(idx - row * Ep);
    dst[idx] = e < E ? src[row * E + e] : 0.f;
}
__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* __restrict__ src, int64_t N, int E, int Ep,
                                                        float* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * E) return;
    const int64_t row = idx / E;
    dst[idx] = src[row * Ep + (idx - row * E)];
}


template <int MODE, int E, bool LSE_STREAM>
int32_t launch_mode(const StreamArgs& a, int ids_dtype, dim3 grid, size_t lds, hipStream_t s) {
#define MH_LAUNCH_STREAM(IdT, HAS)                                                                                   \
    do {                                                                                                             \
        auto kern = stream_kernel<MODE, E, IdT, HAS, LSE_STREAM>;                                                    \
        static bool attr_done = false;                                                                               \
        if (!attr_done) {                                                                                            \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    160 * 1024) != hipSuccess) {                                                     \
                mh_set_error("scorer: cannot raise the dynamic LDS limit");                                          \
                return MH_ERR_LAUNCH;                                                                                \
            }                                                                                                        \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        MH_LAUNCH(kern, grid, dim3(SW * 64), lds, s, a);                                                    \
    } while (0)
    if (!a.x_ids) MH_LAUNCH_STREAM(int32_t, false);
    else if (ids_dtype == MH_I32) MH_LAUNCH_STREAM(int32_t, true);
    else MH_LAUNCH_STREAM(int64_t, true);
#undef MH_LAUNCH_STREAM
    return MH_OK;
}

}  // namespace

// ---- internal interface used by mh_scorer.hip ----------------------------------------------------------------------
struct MhStreamPlan {
    int bn, row_tiles, nt, nsplit, tps;
    size_t lds;
};

MhStreamPlan mh_stream_plan(int mode, int64_t Nx, int64_t Ny, int E, int ids_bytes) {
    MhStreamPlan p;
    // forward: 64-row tiles (64 KB of LDS -> two workgroups per CU); gradient modes hold ~200 registers -> one
    // workgroup per CU anyway, so they take 128-row tiles (half as many barriers)
    const bool light = (mode == SM_FWD || mode == SM_FILTER);  // no second GEMM: ~110 registers, two workgroups per CU
    p.bn = light ? 64 : 128;  // streamed rows per tile: measured best of {64, 128} per mode (tools/microbench.py scorer, round 2)
    if (p.bn != 64 && p.bn != 128) p.bn = 128;
    if (Ny <= 64) p.bn = 64;
    p.row_tiles = (int)mh_ceil_div(Nx, SX);
    p.nt = (int)mh_ceil_div(Ny, p.bn);
    if (p.nt < 1) p.nt = 1;
    const int wg_per_cu = (light && p.bn == 64) ? 2 : 1;
    int want = (int)((int64_t)mh_num_cus() * wg_per_cu / p.row_tiles);
    if (want < 1) want = 1;
    if (want > p.nt) want = p.nt;
    p.tps = (int)mh_ceil_div(p.nt, want);
    p.nsplit = (int)mh_ceil_div(p.nt, p.tps);
    p.lds = (size_t)2 * p.bn * E * 4 + (size_t)2 * p.bn * (ids_bytes ? ids_bytes : 4) + (size_t)4 * p.bn * 4;
    return p;
}

// mode: SM_FWD / SM_GRAD / SM_FWD_GRAD; lse_stream only for SM_GRAD
int32_t mh_stream_launch(int mode, int lse_stream, const MhStreamPlan& p, const float* X, int64_t Nx, const float* Y,
                         int64_t Ny, int E, const void* x_ids, const void* y_ids, int ids_dtype, const float* lse,
                         const float* pos, float invT, float fns, float gscale, float* logits, int64_t ld_logits,
                         float* part_m, float* part_s, float* opart, const float* x_corr, const float* y_corr,
                         int corr_after_mask, hipStream_t s) {
    StreamArgs a;
    a.X = X; a.Y = Y; a.Nx = Nx; a.Ny = Ny; a.x_ids = x_ids; a.y_ids = y_ids; a.lse = lse; a.pos = pos;
    a.invT = invT; a.fns = fns; a.gscale = gscale; a.logits = logits; a.ld_logits = ld_logits;
    a.part_m = part_m; a.part_s = part_s; a.opart = opart; a.bn = p.bn; a.tiles_per_split = p.tps;
    a.prio = 1;  // s_setprio around the MFMA section (measured +2 % in round 2)
    a.tau = nullptr; a.cnt = nullptr; a.cs = nullptr; a.ci = nullptr; a.cap = 0; a.idx0 = 0;
    a.x_corr = x_corr; a.y_corr = y_corr; a.corr_after_mask = corr_after_mask;
    dim3 grid((unsigned)p.row_tiles, (unsigned)p.nsplit);
#define MH_MODE_E(EE)                                                                                      \
    do {                                                                                                   \
        if (mode == SM_FWD) return launch_mode<SM_FWD, EE, false>(a, ids_dtype, grid, p.lds, s);           \
        if (mode == SM_FWD_GRAD) return launch_mode<SM_FWD_GRAD, EE, false>(a, ids_dtype, grid, p.lds, s); \
        if (mode == SM_FILTER) return launch_mode<SM_FILTER, EE, false>(a, ids_dtype, grid, p.lds, s);     \
        if (lse_stream) return launch_mode<SM_GRAD, EE, true>(a, ids_dtype, grid, p.lds, s);               \
        return launch_mode<SM_GRAD, EE, false>(a, ids_dtype, grid, p.lds, s);                              \
    } while (0)
    if (E == 128) MH_MODE_E(128);
    if (E == 64) MH_MODE_E(64);
    if (E == 32) MH_MODE_E(32);
#undef MH_MODE_E
    mh_set_error("scorer stream core: E must be 32, 64 or 128 (got %d)", E);
    return MH_ERR_UNSUPPORTED;
}

void mh_stream_fwd_finalize(const float* pos, int64_t B, int nsplit, const float* part_m, const float* part_s, float invT,
                            float* logits, int64_t ld_logits, float* loss, float* lse, hipStream_t s) {
    MH_LAUNCH(fwd_finalize_kernel, dim3((unsigned)mh_ceil_div(B, 256)), dim3(256), 0, s, pos, B, nsplit, part_m,
                       part_s, invT, logits, ld_logits, loss, lse);
}

void mh_stream_fwd_grad_combine(const float* q, const float* item, const float* pos, int64_t B, int E, int nsplit,
                                const float* part_m, const float* part_s, const float* opart, float invT, float g,
                                float* loss, float* lse, float* dq, float* ditem, hipStream_t s) {
    MH_LAUNCH(fwd_grad_combine_kernel, dim3((unsigned)mh_ceil_div(B * (E / 4), 256)), dim3(256), 0, s, q, item, pos,
                       B, E, nsplit, part_m, part_s, opart, invT, g, loss, lse, dq, ditem);
}

void mh_stream_grad_combine(const float* opart, int64_t N, int E, int nsplit, const float* pos, const float* lse,
                            const float* other, const float* self, float invT, float g, float* out, float* out_pos,
                            hipStream_t s) {
    MH_LAUNCH(grad_combine_kernel, dim3((unsigned)mh_ceil_div(N * (E / 4), 256)), dim3(256), 0, s, opart, N, E,
                       nsplit, pos, lse, other, self, invT, g, out, out_pos);
}

void mh_stream_pad_rows(const float* src, int64_t N, int E, int Ep, float* dst, hipStream_t s) {
    MH_LAUNCH(pad_rows_kernel, dim3((unsigned)mh_ceil_div(N * Ep, 256)), dim3(256), 0, s, src, N, E, Ep, dst);
}
void mh_stream_unpad_rows(const float* src, int64_t N, int E, int Ep, float* dst, hipStream_t s) {
    MH_LAUNCH(unpad_rows_kernel, dim3((unsigned)mh_ceil_div(N * E, 256)), dim3(256), 0, s, src, N, E, Ep, dst);
}

// top-k threshold filter on the same core (mh_topk.hip): X = queries (stationary), Y = candidates n_beg .. n_end
int32_t mh_stream_filter(const float* q, int64_t Bq, const float* cand, int64_t n_cand, int E, const float* tau, int* cnt,
                         float* cs, int32_t* ci, int cap, int64_t idx0, hipStream_t s) {
    const MhStreamPlan p = mh_stream_plan(SM_FILTER, Bq, n_cand, E, 0);
    StreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.X = q; a.Y = cand; a.Nx = Bq; a.Ny = n_cand; a.invT = 1.f; a.bn = p.bn; a.tiles_per_split = p.tps;
    a.prio = 1;  // s_setprio around the MFMA section (measured +2 % in round 2)
    a.tau = tau; a.cnt = cnt; a.cs = cs; a.ci = ci; a.cap = cap; a.idx0 = idx0;
    dim3 grid((unsigned)p.row_tiles, (unsigned)p.nsplit);
    if (E == 128) return launch_mode<SM_FILTER, 128, false>(a, MH_I32, grid, p.lds, s);
    if (E == 64) return launch_mode<SM_FILTER, 64, false>(a, MH_I32, grid, p.lds, s);
    if (E == 32) return launch_mode<SM_FILTER, 32, false>(a, MH_I32, grid, p.lds, s);
    mh_set_error("top-k stream filter: E must be 32, 64 or 128");
    return MH_ERR_UNSUPPORTED;
}
