// fp32 MFMA GEMM core, second generation (gfx950, v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
//   C[M, N] = A[M, K] * B        B_NT = false: B = W[K, N] row-major (Dense forward, cross layer)
//                                B_NT = true : B = W[N, K] row-major, C = A W^T (dX of a Dense layer, scorer-style products)
//
// What changed against mh_gemm_core.h (global -> registers -> ds_write staging, 2 LDS stages, 32x64 per wavefront):
//   * tiles travel global -> LDS by DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass, no wait of
//     the issuing wavefront; the LDS image is chunk-swizzled by permuting the per-lane SOURCE address;
//   * a STAGES-deep ring of 16-wide k-tiles with ONE barrier per k-tile: the loads of tile t + STAGES - 1 are issued
//     right after the barrier that publishes tile t, so STAGES - 1 tiles (24 KB each at 256 x 128) are in flight per
//     workgroup -- the HBM-streamed operand of the skinny layers needs that depth;
//   * 64 x 64 outputs per wavefront (2 x 2 accumulators): every A / B fragment feeds two MFMAs, half the LDS reads.
// Numerics are unchanged: every output is ONE k-ascending fp32 fmaf chain from zero (k = 2 s + h inside a step, steps and
// tiles ascending), bit-identical to mh_gemm_core.h and to oracle/oracle_c.c.
//
// Requirements (the launchers in mh_linear*.hip fall back to the first-generation kernels otherwise): K % 4 == 0 (the
// 16-byte chunks at or past K of the last k-tile are fetched from a zero buffer for BOTH operands), 16-byte aligned rows
// (lda % 4 == 0, and ldb % 4 == 0 / N % 4 == 0 for the n-contiguous B), K >= 4, N >= 4.
#pragma once
#include "mh_gemm_core.h"

namespace mhgemm2 {

constexpr int BK = 16;  // default k-tile (the split-M TN kernel); gemm2_kernel takes it as a template parameter

using Epilogue = mhgemm::EpiArgs;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void dma16(const float* g, float* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// source of the 16-byte chunks past K in the last k-tile (K % 16 != 0, K % 4 == 0): both operands read zeros there
static __device__ __attribute__((aligned(16))) float g_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    // "memory": nothing that touches LDS or global memory moves across (the compiler does not know that the DMA
    // instructions issued earlier write the LDS tile read below)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}


// k-major LDS tile [ROWS][BKT k], CH = BKT / 4 chunks of 16 bytes per row: chunk c (k = 4c .. 4c+3) of row r sits at chunk
// position CH r + (c ^ swz(r)) with swz(r) = (r >> 2) & 3 (CH = 4) or (r >> 1) & 7 (CH = 8): the 16 lanes of a quarter
// wavefront (consecutive rows) then hit 16 distinct 4-bank groups on ds_read_b128.
template <int CH>
__device__ __forceinline__ int kmajor_swz(int r) {
    return CH == 4 ? ((r >> 2) & 3) : ((r >> 1) & 7);
}
template <int CH>
__device__ __forceinline__ int kmajor_src_chunk(int p) {  // chunk position -> source chunk c of row p / CH
    return (p % CH) ^ kmajor_swz<CH>(p / CH);
}

// The product of one output tile, accumulators left in registers: acc[tm][tn] is the 32 x 32 block at rows
// row0 + wm * TM * 32 + tm * 32, columns n0 + wn * TN * 32 + tn * 32 of C in the MFMA C layout (lane: column l31, rows
// (r & 3) + 8 (r >> 2) + 4 h).  gemm2_kernel stores them through mhgemm::store_tile; a kernel with its own epilogue (row
// reductions instead of a store: tools/exp/scorer_lab.hip) calls this directly.
template <int BM, int BN, int WM, int WN, bool B_NT, int STAGES, bool PIPE, int BKT, int ABLATE>
__device__ __forceinline__ void gemm2_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                           int64_t M, int N, int K, int64_t row0, int n0, float* smem,
                                           f32x16 (&acc)[BM / WM / 32][BN / WN / 32]) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int BK = BKT, CH = BKT / 4;
    static_assert(BKT == 16 || BKT == 32, "k-tile");
    constexpr int A_FL = BM * BK, B_FL = BN * BK, ST_FL = A_FL + B_FL;
    constexpr int NIA = BM * CH / 64 / NW, NIB = BN * CH / 64 / NW;  // DMA wave-instructions (1 KiB each) per wavefront per tile
    static_assert(NIA >= 1 && NIB >= 1 && (BM * CH / 64) % NW == 0 && (BN * CH / 64) % NW == 0, "tile / wavefront mismatch");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int NI = NIA + NIB;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- per-lane DMA source pointers (tile 0) ------------------------------------------------------------------------
    const float* pa[NIA];
    const float* pb[NIB];
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        const int p = (wave + j * NW) * 64 + lane;
        int64_t row = row0 + p / CH;
        if (row > M - 1) row = M - 1;
        pa[j] = A + row * lda + 4 * kmajor_src_chunk<CH>(p);
    }
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        const int p = (wave + j * NW) * 64 + lane;
        if (B_NT) {
            int n = n0 + p / CH;
            if (n > N - 1) n = N - 1;
            pb[j] = B + (int64_t)n * ldb + 4 * kmajor_src_chunk<CH>(p);
        } else {
            // n-major tile [16 k][BN]: chunk position p = k * (BN/4) + pc holds source chunk pc ^ (8 (k & 1)): the two
            // k-slots of an MFMA step (rows k, k+1) read 32 consecutive floats each, 32 banks apart
            constexpr int CPR = BN / 4;
            const int k = p / CPR, pc = p % CPR;
            int n = n0 + 4 * (pc ^ (8 * (k & 1)));
            const int nlast = ((N - 4) / 4) * 4;
            if (n > nlast) n = nlast;
            pb[j] = B + (int64_t)k * ldb + n;
        }
    }
    const int nk = (K + BK - 1) / BK;
    // k offset (inside a tile) of the chunk each DMA lane fetches: chunks at or past K read g_zero_chunk (last tile only)
    int ka[NIA], kb[NIB];
#pragma unroll
    for (int j = 0; j < NIA; ++j) ka[j] = 4 * kmajor_src_chunk<CH>((wave + j * NW) * 64 + lane);
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        const int p = (wave + j * NW) * 64 + lane;
        kb[j] = B_NT ? 4 * kmajor_src_chunk<CH>(p) : p / (BN / 4);
    }
    auto issue = [&](int kt) {
        float* st = smem + (kt % STAGES) * ST_FL;
        const int k0 = kt * BK;
        const bool tail = k0 + BK > K;  // uniform
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const float* g = pa[j] + k0;
            if (tail && k0 + ka[j] >= K) g = g_zero_chunk;
            dma16(g, st + (wave + j * NW) * 256);
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const float* g = B_NT ? pb[j] + k0 : pb[j] + (int64_t)k0 * ldb;
            if (tail && k0 + kb[j] >= K) g = g_zero_chunk;
            dma16(g, st + A_FL + (wave + j * NW) * 256);
        }
    };

    // ---- fragment addresses (floats, relative to the stage base) -------------------------------------------------------
    int fa[TM], fb[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = wm * TM * 32 + tm * 32 + l31;
        fa[tm] = r * BK + (kmajor_swz<CH>(r) << 2);  // chunk c of this row: fa ^ (c << 2)
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = wn * TN * 32 + tn * 32 + l31;
        if (B_NT)
            fb[tn] = A_FL + n * BK + (kmajor_swz<CH>(n) << 2);
        else
            fb[tn] = A_FL + h * BN + ((((n >> 2) ^ (8 * h)) << 2) | (n & 3));  // step s: + 2 s BN
    }

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- prologue: STAGES - 1 tiles in flight -------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);

    // fragments of chunk c (MFMA steps 2c, 2c + 1): [0] = step 2c, [1] = step 2c + 1
    auto load_frags = [&](const float* st, int c, float (&a)[2][TM], float (&b)[2][TN]) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(st + (fa[tm] ^ (c << 2)));
            a[0][tm] = h ? v.y : v.x;  // k = 4c + h
            a[1][tm] = h ? v.w : v.z;  // k = 4c + 2 + h
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            if (B_NT) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(st + (fb[tn] ^ (c << 2)));
                b[0][tn] = h ? v.y : v.x;
                b[1][tn] = h ? v.w : v.z;
            } else {
                b[0][tn] = st[fb[tn] + (4 * c) * BN];
                b[1][tn] = st[fb[tn] + (4 * c + 2) * BN];
            }
        }
    };
    auto compute = [&](const float* st) {
        float a[2][2][TM], b[2][2][TN];  // [buffer][step][tile]
        load_frags(st, 0, a[0], b[0]);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            // the reads of chunk c + 1 are issued BEFORE the MFMAs of chunk c (a wavefront issues in order: reads placed
            // after the MFMAs only leave once the matrix pipe has accepted all of them, and their latency is then exposed)
            if (c + 1 < CH) load_frags(st, c + 1, a[(c + 1) & 1], b[(c + 1) & 1]);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(a[c & 1][j][tm], b[c & 1][j][tn], acc[tm][tn]);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- main loop, unrolled by the ring depth so that stage bases are immediates ---------------------------------------
    for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            const int kt = kt0 + s;
            if (kt < nk) {
                // tile kt has landed when at most the loads of the STAGES - 2 tiles issued after it are outstanding
                if (kt + STAGES - 2 <= nk - 1)
                    wait_vm_and_barrier<(STAGES - 2) * NI>();
                else
                    wait_vm_and_barrier<0>();
                // every wavefront is past tile kt - 1: its stage is free for tile kt + STAGES - 1
                // ABLATE (tools/exp/gemm_lab only): 1 = no tile loads after the prologue, 2 = no MFMAs -- which side bounds the loop
                if (ABLATE != 1 && kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
                if (ABLATE != 2) compute(smem + s * ST_FL);
            }
        }
    }

}

template <int BM, int BN, int WM, int WN, bool B_NT, int STAGES, bool PIPE = true, int BKT = 16, int ABLATE = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb, int64_t M, int N, int K,
                                                          float* __restrict__ C, int64_t ldc, const Epilogue ep,
                                                          int ncol_tiles) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    // column tiles fastest: the workgroups resident at one time share few A row panels (read once from HBM) and sweep
    // all of B (L2 / Infinity Cache resident)
    const int64_t row0 = (int64_t)(blockIdx.x / ncol_tiles) * BM;
    const int n0 = (int)(blockIdx.x % ncol_tiles) * BN;
    f32x16 acc[TM][TN];
    gemm2_tile<BM, BN, WM, WN, B_NT, STAGES, PIPE, BKT, ABLATE>(A, lda, B, ldb, M, N, K, row0, n0, smem, acc);
    // ---- epilogue: bias / activation / cross / folded activation derivative (mh_gemm_core.h) -------------------------------
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    mhgemm::store_tile<TM, TN>(acc, C, ldc, row0 + (wave / WN) * TM * 32, n0 + (wave % WN) * TN * 32, M, N, (int)(threadIdx.x & 63), ep);
}

template <int BM, int BN, int WM, int WN, bool B_NT, int STAGES, bool PIPE = true, int BKT = 16, int ABLATE = 0>
inline hipError_t launch(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int N, int K, float* C,
                         int64_t ldc, const Epilogue& ep, hipStream_t s) {
    auto kern = gemm2_kernel<BM, BN, WM, WN, B_NT, STAGES, PIPE, BKT, ABLATE>;
    const size_t lds = (size_t)STAGES * (BM + BN) * BKT * sizeof(float);
    static bool attr_done = false;  // per instantiation
    if (!attr_done && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int ncol = (int)((N + BN - 1) / BN);
    const int64_t nrow = (M + BM - 1) / BM;
    MH_LAUNCH(kern, dim3((unsigned)(nrow * ncol)), dim3(WM * WN * 64), lds, s, A, lda, B, ldb, M, N, K, C, ldc, ep, ncol);
    return hipGetLastError();
}

// ---- split-M "TN" product: part[s][K, N] = X[m in slice s][K]^T Z[m in slice s][N]  (dW of a Dense layer) ---------------
// Both operands are n-major tiles [16 contraction rows][cols] (the B-operand layout of the NN product: odd rows rotated by
// 8 chunks, fragments are conflict-free ds_read_b32); rows at or past the end of the slice read the zero chunk.  The
// contraction runs over the batch in ascending row order inside a slice (deterministic); the slices are summed by
// reduce_partials_kernel.  db_part (optional): column sums of Z per slice, accumulated by the blockIdx.x == 0 workgroups.
template <int BMO, int BNO, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_tn_kernel(const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ Z, int64_t ldz, int64_t M, int K,
                                                             int N, int64_t rows_per_split, float* __restrict__ part,
                                                             float* __restrict__ db_part) {
    constexpr int NW = WM * WN;
    constexpr int TM = BMO / WM / 32, TN = BNO / WN / 32;
    constexpr int A_FL = BMO * BK, B_FL = BNO * BK, ST_FL = A_FL + B_FL;
    constexpr int NIA = BMO / 16 / NW, NIB = BNO / 16 / NW;
    static_assert(NIA >= 1 && NIB >= 1 && (BMO / 16) % NW == 0 && (BNO / 16) % NW == 0, "tile / wavefront mismatch");
    constexpr int NI = NIA + NIB;
    extern __shared__ __attribute__((aligned(1024))) float smem[];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int k0 = blockIdx.x * BMO;  // output row tile (over K)
    const int n0 = blockIdx.y * BNO;  // output column tile (over N)
    const int sl = blockIdx.z;
    const int64_t m_beg = (int64_t)sl * rows_per_split;
    const int64_t m_end = (m_beg + rows_per_split < M) ? m_beg + rows_per_split : M;
    const int nk = (int)((m_end - m_beg + BK - 1) / BK);

    // per-lane DMA sources: row (inside a tile) and clamped, rotated column chunk
    const float* pa[NIA];
    const float* pb[NIB];
    int ra[NIA], rb[NIB];
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
        constexpr int CPR = BMO / 4;
        const int p = (wave + j * NW) * 64 + lane;
        const int r = p / CPR, pc = p % CPR;
        int c = k0 + 4 * (pc ^ (8 * (r & 1)));
        // chunks past K only feed outputs that are never stored: clamp them to the chunk that holds column K - 1 (an
        // ld-padded operand owns it: ldx % 4 == 0), never beyond -- a column-offset view must not be read past its rows
        int clast = (K + 3) / 4 * 4 - 4;
        if (clast > (int)((ldx - 4) / 4) * 4) clast = (int)((ldx - 4) / 4) * 4;
        if (c > clast) c = clast;
        ra[j] = r;
        pa[j] = X + (m_beg + r) * ldx + c;
    }
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
        constexpr int CPR = BNO / 4;
        const int p = (wave + j * NW) * 64 + lane;
        const int r = p / CPR, pc = p % CPR;
        int c = n0 + 4 * (pc ^ (8 * (r & 1)));
        int clast = (N + 3) / 4 * 4 - 4;
        if (clast > (int)((ldz - 4) / 4) * 4) clast = (int)((ldz - 4) / 4) * 4;
        if (c > clast) c = clast;
        rb[j] = r;
        pb[j] = Z + (m_beg + r) * ldz + c;
    }
    const int64_t nrows = m_end - m_beg;
    auto issue = [&](int kt) {
        float* st = smem + (kt % STAGES) * ST_FL;
        const int64_t r0 = (int64_t)kt * BK;
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            const float* g = pa[j] + r0 * ldx;
            if (r0 + ra[j] >= nrows) g = g_zero_chunk;
            dma16(g, st + (wave + j * NW) * 256);
        }
#pragma unroll
        for (int j = 0; j < NIB; ++j) {
            const float* g = pb[j] + r0 * ldz;
            if (r0 + rb[j] >= nrows) g = g_zero_chunk;
            dma16(g, st + A_FL + (wave + j * NW) * 256);
        }
    };

    int fa[TM], fb[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int c = wm * TM * 32 + tm * 32 + l31;
        fa[tm] = h * BMO + ((((c >> 2) ^ (8 * h)) << 2) | (c & 3));  // step s: + 2 s BMO
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int c = wn * TN * 32 + tn * 32 + l31;
        fb[tn] = A_FL + h * BNO + ((((c >> 2) ^ (8 * h)) << 2) | (c & 3));
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    float dbacc = 0.f;
    const bool do_db = db_part != nullptr && blockIdx.x == 0 && (int)threadIdx.x < BNO;
    const int dbc = threadIdx.x;  // column of the Z tile this thread sums

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s);

    auto compute = [&](const float* st) {
        if (do_db) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < BK; ++r) cs += st[A_FL + r * BNO + ((((dbc >> 2) ^ (8 * (r & 1))) << 2) | (dbc & 3))];
            dbacc += cs;
        }
#pragma unroll
        for (int sp = 0; sp < BK / 2; sp += 2) {  // two MFMA steps per round: reads of the round are issued together
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) a[u][tm] = st[fa[tm] + 2 * (sp + u) * BMO];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) b[u][tn] = st[fb[tn] + 2 * (sp + u) * BNO];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(a[u][tm], b[u][tn], acc[tm][tn]);
        }
    };

    for (int kt0 = 0; kt0 < nk; kt0 += STAGES) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            const int kt = kt0 + s;
            if (kt < nk) {
                if (kt + STAGES - 2 <= nk - 1)
                    wait_vm_and_barrier<(STAGES - 2) * NI>();
                else
                    wait_vm_and_barrier<0>();
                if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1);
                compute(smem + s * ST_FL);
            }
        }
    }
    if (do_db && n0 + dbc < N) db_part[(int64_t)sl * N + n0 + dbc] = dbacc;
    float* P = part + (int64_t)sl * K * N;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + l31;
        if (col >= N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < K) P[(int64_t)row * N + col] = acc[tm][tn][r];
            }
    }
}

template <int BMO, int BNO, int WM, int WN, int STAGES>
inline hipError_t launch_tn(const float* X, int64_t ldx, const float* Z, int64_t ldz, int64_t M, int K, int N,
                            int64_t rows_per_split, int splits, float* part, float* db_part, hipStream_t s) {
    auto kern = gemm2_tn_kernel<BMO, BNO, WM, WN, STAGES>;
    const size_t lds = (size_t)STAGES * (BMO + BNO) * BK * sizeof(float);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)((K + BMO - 1) / BMO), (unsigned)((N + BNO - 1) / BNO), (unsigned)splits);
    MH_LAUNCH(kern, grid, dim3(WM * WN * 64), lds, s, X, ldx, Z, ldz, M, K, N, rows_per_split, part, db_part);
    return hipGetLastError();
}

}  // namespace mhgemm2
