// Backward of the one-hot embedding lookup fused with the sparse optimizer step (gfx950).
// Reference: the IndexedSlices branch of BaseModel.train_step (merlin/models/tf/models/base.py:
// 1121-1174): Keras sums duplicate indices first (_deduplicate_indexed_slices) and then applies
// the optimizer row-wise, so the update must see the SUM of a row's gradients exactly once.
//
// Pipeline (all on one stream, no host sync):
//   1. build COMPACT keys: key = first_row[table] + id, i.e. the row number in the concatenation of the
//      distinct tables (features sharing one table share its key range, so shared rows are updated once),
//      + the packed (feature, sample) of every entry; out-of-range ids get the all-ones sentinel and sort
//      to the end.  Keys are 32-bit whenever the tables hold < 2^32 - 1 rows together (64-bit otherwise);
//   2. ONE stable rocPRIM radix sort of the (key, sample) pairs over ceil(log2(total rows + 1)) bits --
//      23 bits = 3 Onesweep passes for the 26 Criteo tables of the headline config (library plumbing);
//   3. segmented reduce over PIECES (a run cut at every 16th sorted index): one D/4-lane group per piece
//      sums its gradient rows in registers; a piece that is a whole run is applied to the table row
//      directly (exclusive owner, no atomics); the pieces of a run crossing a chunk boundary add to
//      carry[home chunk] (home = chunk holding the run's first entry, from a max-scan of chunk flags)
//      -- long runs of hot ids are pre-summed 16:1 in registers and 16:1 again through LDS;
//   4. each chunk that is home to a crossing run applies the carried sum.
// HBM traffic: grad rows read once (random), table (+state) rows read+written once per UNIQUE id.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mh_common.h"

namespace {

constexpr int CHUNK = 16;

template <typename KeyT>
struct KeyTraits {
    static constexpr KeyT sentinel = (KeyT)~(KeyT)0;
};

struct BwdArgs {
    float* table[MH_MAX_FEATURES];
    float* state[MH_MAX_FEATURES];
    float* state2[MH_MAX_FEATURES];  // Adam second moment
    const void* ids[MH_MAX_FEATURES];
    int64_t rows[MH_MAX_FEATURES];
    int64_t offset[MH_MAX_FEATURES];  // float offset of the feature inside a grad row
    int64_t first[MH_MAX_FEATURES];   // first compact key of the feature's table (shared tables share it)
};

template <typename IdT, typename KeyT>
__global__ __launch_bounds__(256) void build_keys_kernel(const BwdArgs a, int64_t B, int F, KeyT* __restrict__ keys,
                                                        uint32_t* __restrict__ vals,
                                                        unsigned int* __restrict__ counter) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx == 0) *counter = 0u;  // piece counter of step 3 (saves a memset launch)
    if (idx >= B * F) return;
    const int f = (int)(idx / B);
    const int64_t b = idx - (int64_t)f * B;
    const int64_t id = (int64_t) static_cast<const IdT*>(a.ids[f])[b];
    const bool ok = id >= 0 && id < a.rows[f];
    keys[idx] = ok ? (KeyT)(a.first[f] + id) : KeyTraits<KeyT>::sentinel;
    vals[idx] = ((uint32_t)f << 26) | (uint32_t)b;
}

struct OptHyper {
    float lr, eps, beta1, beta2;
    const float* lr_dev;  // optional device scalar overriding lr (graph-replayable bias correction)
};

// A table row update in two halves, so that the row (+ optimizer state) loads can be issued BEFORE the gradient
// rows they do not depend on: load_row() fetches, finish_row() applies the optimizer and stores.
struct RowRmw {
    float* w;
    float* s1;
    float* s2;
    f32x4 wv, m, v;
};

// f: any feature of the run (features sharing a table share pointers and key range); key: compact key.
__device__ __forceinline__ void load_row(const BwdArgs& a, int f, int64_t key, int D, int c4, int opt, RowRmw& r) {
    const int64_t off = (key - a.first[f]) * D + c4 * 4;
    r.w = a.table[f] + off;
    r.wv = *reinterpret_cast<const f32x4*>(r.w);
    if (opt != MH_OPT_SGD) {
        r.s1 = a.state[f] + off;
        r.m = *reinterpret_cast<const f32x4*>(r.s1);
    }
    if (opt == MH_OPT_ADAM) {
        r.s2 = a.state2[f] + off;
        r.v = *reinterpret_cast<const f32x4*>(r.s2);
    }
}

__device__ __forceinline__ void finish_row(RowRmw& r, f32x4 g, int opt, const OptHyper& hp) {
    const float lr = hp.lr_dev ? *hp.lr_dev : hp.lr;
    const float eps = hp.eps;
    f32x4 wv = r.wv;
    if (opt == MH_OPT_ADAGRAD) {
        const f32x4 sv = r.m + g * g;
        *reinterpret_cast<f32x4*>(r.s1) = sv;
        wv.x -= lr * g.x / (sqrtf(sv.x) + eps);
        wv.y -= lr * g.y / (sqrtf(sv.y) + eps);
        wv.z -= lr * g.z / (sqrtf(sv.z) + eps);
        wv.w -= lr * g.w / (sqrtf(sv.w) + eps);
    } else if (opt == MH_OPT_ADAM) {
        // LazyAdam._resource_apply_sparse (blocks/optimizer.py:412-437): only the touched rows' moments move
        const f32x4 m = r.m * hp.beta1 + g * (1.f - hp.beta1);
        const f32x4 v = r.v * hp.beta2 + (g * g) * (1.f - hp.beta2);
        *reinterpret_cast<f32x4*>(r.s1) = m;
        *reinterpret_cast<f32x4*>(r.s2) = v;
        wv.x -= lr * m.x / (sqrtf(v.x) + eps);
        wv.y -= lr * m.y / (sqrtf(v.y) + eps);
        wv.z -= lr * m.z / (sqrtf(v.z) + eps);
        wv.w -= lr * m.w / (sqrtf(v.w) + eps);
    } else {
        wv -= g * lr;
    }
    *reinterpret_cast<f32x4*>(r.w) = wv;
}


// chunk c: v[c] = -1 if its first run continues from the previous chunk AND the whole chunk is that one
// run (the run's home lies further back), else c.  An inclusive max-scan of v gives lasthome[c] = home
// chunk of the LAST run of chunk c; the head piece of a continuing chunk c then belongs to lasthome[c-1].
template <typename KeyT>
__global__ __launch_bounds__(256) void chunk_flags_kernel(const KeyT* __restrict__ keys, int64_t n, int64_t nchunks,
                                                         int* __restrict__ v) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    const int64_t c0 = c * CHUNK;
    const int64_t c1 = (c0 + CHUNK < n) ? c0 + CHUNK : n;
    const bool cont = (c0 > 0) && (keys[c0 - 1] == keys[c0]);
    const bool whole = keys[c1 - 1] == keys[c0];  // sorted: first == last  <=>  one run
    v[c] = (cont && whole) ? -1 : (int)c;
}

// A PIECE is a maximal range of sorted entries with one key inside one chunk of 16 (cuts at run starts and at
// every 16th index).  piece_list_kernel enumerates the pieces of the valid (non-sentinel) prefix into a
// compact list -- one 16-byte record {start:32 | len:5 | starts_run:1 | ends_run:1 | feature:6, key} each (one
// load gives the consumer everything the table-row address needs).  A workgroup walks LIST_TILES tiles of 256
// entries and bumps the global counter ONCE (same-address atomics retire at ~12 ns each: one bump per 256
// entries cost 79 us for 1.7M entries); the list is ordered inside a workgroup, unordered across workgroups.
constexpr int LIST_TILES = 8;

template <typename KeyT>
__global__ __launch_bounds__(256) void piece_list_kernel(const KeyT* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, int64_t n,
                                                         ulonglong2* __restrict__ pieces,
                                                         unsigned int* __restrict__ counter) {
    constexpr KeyT SENT = KeyTraits<KeyT>::sentinel;
    __shared__ unsigned int wave_cnt[LIST_TILES][4];
    __shared__ unsigned int block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint64_t rec[LIST_TILES];
    KeyT kk[LIST_TILES];
    unsigned int rank[LIST_TILES];
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * LIST_TILES + t) * 256 + threadIdx.x;
        const KeyT k = (i < n) ? keys[i] : SENT;
        const bool valid = k != SENT;
        const bool run_start = valid && (i == 0 || keys[i - 1] != k);
        const bool cut = valid && (run_start || (i & (CHUNK - 1)) == 0);
        const uint64_t cuts = __ballot(cut);
        const uint64_t valids = __ballot(valid);
        rec[t] = 0;
        kk[t] = k;
        if (cut) {
            // the piece ends before the next cut of this wave, or with the wave's valid entries (64 | chunk
            // size, so a wave boundary is always a cut or the end of the valid prefix)
            const uint64_t later = (lane == 63) ? 0ull : (cuts >> (lane + 1));
            const int nvalid = __popcll(valids);  // valid entries are a prefix of the wave (sentinels sort last)
            const int len = later ? (__ffsll((unsigned long long)later)) : (nvalid - lane);
            const int64_t e = i + len;
            const bool ends = (e >= n) || (keys[e] != k);
            rec[t] = (uint64_t)i | ((uint64_t)len << 32) | ((uint64_t)(run_start ? 1 : 0) << 37) |
                     ((uint64_t)(ends ? 1 : 0) << 38) | ((uint64_t)(vals[i] >> 26) << 39) | (1ull << 63);
        }
        rank[t] = __popcll(cuts & lt_mask);
        if (lane == 0) wave_cnt[t][wave] = __popcll(cuts);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int t = 0; t < LIST_TILES; ++t) tot += wave_cnt[t][0] + wave_cnt[t][1] + wave_cnt[t][2] + wave_cnt[t][3];
        block_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    unsigned int o = block_base;
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        unsigned int before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[t][w];
        if (rec[t]) pieces[o + before + rank[t]] = make_ulonglong2(rec[t], (uint64_t)kk[t]);
        o += wave_cnt[t][0] + wave_cnt[t][1] + wave_cnt[t][2] + wave_cnt[t][3];
    }
}

// One D/4-lane group per piece (grid-stride over the list): the piece's gradient rows are summed in sorted
// order, four loads in flight; a piece that is a whole run is applied to its table row directly (exclusive
// owner, no atomics); a piece of a run that crosses a chunk boundary adds its sum to carry[home chunk]
// (home from the scanned chunk flags).  Every piece is independent, so the random 256-byte read-modify-writes
// of different rows overlap freely -- tools/exp/rmw_bench.hip measures the same traffic pattern at ~5 TB/s,
// which the earlier one-group-per-chunk serial walk (2.3 TB/s) could not reach.
__global__ __launch_bounds__(256) void piece_reduce_apply_kernel(const BwdArgs a, const uint32_t* __restrict__ vals,
                                                                int D, int LPR, const float* __restrict__ grad,
                                                                int64_t grad_row_stride, float* __restrict__ carry,
                                                                const int* __restrict__ lasthome,
                                                                const ulonglong2* __restrict__ pieces,
                                                                const unsigned int* __restrict__ counter, int opt,
                                                                const OptHyper hp) {
    __shared__ f32x4 part_s[256];      // partial sum of every group (one f32x4 per thread)
    __shared__ uint64_t part_key[64];  // key of a group's partial-run piece, ~0 if it has none
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    const int64_t np = (int64_t)*counter;
    const int64_t stride = (int64_t)gridDim.x * groups;
    auto row = [&](uint32_t v) -> f32x4 {
        const int f = (int)(v >> 26);
        return *reinterpret_cast<const f32x4*>(grad + (int64_t)(v & ((1u << 26) - 1)) * grad_row_stride + a.offset[f] +
                                               c4 * 4);
    };
    int64_t base = (int64_t)blockIdx.x * groups;  // block-uniform: the loop carries barriers
    ulonglong2 next = make_ulonglong2(0ull, ~0ull);
    if (gi < groups && base + gi < np) next = pieces[base + gi];
    for (; base < np; base += stride) {
        const int64_t p = base + gi;
        const bool active = gi < groups && p < np;
        const ulonglong2 cur = next;
        if (gi < groups && p + stride < np) next = pieces[p + stride];  // next record in flight during this piece
        const uint64_t rec = cur.x, key = cur.y;
        const int64_t s0 = (int64_t)(rec & 0xffffffffull);
        const int len = active ? (int)((rec >> 32) & 31) : 0;
        const bool starts = (rec >> 37) & 1, ends = (rec >> 38) & 1;
        const int fk = (int)((rec >> 39) & 63);
        const bool whole = active && starts && ends;
        const bool partial = active && !whole;
        RowRmw rr;
        if (whole) load_row(a, fk, (int64_t)key, D, c4, opt, rr);  // independent of the gradient rows: overlaps them
        const uint32_t* vv = vals + s0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int i = 0;
        for (; i + 4 <= len; i += 4) {
            const uint32_t v0 = vv[i], v1 = vv[i + 1], v2 = vv[i + 2], v3 = vv[i + 3];
            const f32x4 r0 = row(v0), r1 = row(v1), r2 = row(v2), r3 = row(v3);
            acc += r0;
            acc += r1;
            acc += r2;
            acc += r3;
        }
        if (i + 2 <= len) {
            const uint32_t v0 = vv[i], v1 = vv[i + 1];
            const f32x4 r0 = row(v0), r1 = row(v1);
            acc += r0;
            acc += r1;
            i += 2;
        }
        if (i < len) acc += row(vv[i]);
        if (whole) finish_row(rr, acc, opt, hp);
        // Pieces of ONE long run sit next to each other in the list, i.e. in neighbouring groups of this block:
        // they are summed through LDS first and the leader issues a single set of atomics.  Without this a hot
        // row (a 3-row table takes 21K gradients of a 64K batch) serialises ~1400 same-line atomics in L2 and
        // the whole kernel waits for it (measured: ~330 us floor independent of everything else).
        part_s[threadIdx.x] = acc;
        if (c4 == 0 && gi < 64) part_key[gi] = partial ? key : ~0ull;
        __syncthreads();
        if (partial && (gi == 0 || part_key[gi - 1] != key)) {
            f32x4 sum = acc;
            for (int g2 = gi + 1; g2 < groups && part_key[g2] == key; ++g2) sum += part_s[g2 * LPR + c4];
            const int64_t chunk = s0 / CHUNK;
            const int64_t home = starts ? chunk : (int64_t)lasthome[chunk - 1];
            float* cr = carry + home * D + c4 * 4;
            atomicAdd(cr + 0, sum.x);
            atomicAdd(cr + 1, sum.y);
            atomicAdd(cr + 2, sum.z);
            atomicAdd(cr + 3, sum.w);
        }
        __syncthreads();
    }
}

// Chunk c is home to a carried sum iff a run starts inside it (flags[c] == c, see chunk_flags_kernel) and its last
// run continues into chunk c + 1.
template <typename KeyT>
__global__ __launch_bounds__(256) void carry_apply_kernel(const BwdArgs a, const KeyT* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals,
                                                         const int* __restrict__ flags, int64_t n, int D, int LPR,
                                                         const float* __restrict__ carry, int opt, const OptHyper hp) {
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    if (gi >= groups) return;
    const int64_t chunk = (int64_t)blockIdx.x * groups + gi;
    const int64_t c1 = (chunk + 1) * CHUNK;
    if (c1 >= n) return;  // the last chunk cannot be crossed
    if (flags[chunk] < 0) return;  // one run that began earlier: an earlier chunk is its home
    const KeyT key = keys[c1 - 1];
    if (key == KeyTraits<KeyT>::sentinel || keys[c1] != key) return;  // last run ends here
    const f32x4 g = *reinterpret_cast<const f32x4*>(carry + chunk * D + c4 * 4);
    RowRmw rr;
    load_row(a, (int)(vals[c1 - 1] >> 26), (int64_t)key, D, c4, opt, rr);
    finish_row(rr, g, opt, hp);
}

struct WsLayout {
    int64_t n, nchunks;
    size_t off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_carry, off_flags, off_home, off_pieces, off_counter,
        off_tmp, tmp_bytes, total;
};

// rocPRIM switches to a merge sort (dozens of 5 us launches) below 1M items; Onesweep already wins from ~64K.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 65536>;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Sized for the 64-bit key variant (the 32-bit one uses the front of the same buffers).
bool ws_layout(int64_t B, int F, int D, WsLayout* L) {
    L->n = B * F;
    L->nchunks = mh_ceil_div(L->n, CHUNK);
    size_t tmp = 0, tmp32 = 0;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                                         (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)L->n, 0, 64);
    if (e != hipSuccess) return false;
    e = rocprim::radix_sort_pairs<SortConfig>(nullptr, tmp32, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                              (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)L->n, 0, 32);
    if (e != hipSuccess) return false;
    if (tmp32 > tmp) tmp = tmp32;
    size_t scan = 0;
    e = rocprim::inclusive_scan(nullptr, scan, (const int*)nullptr, (int*)nullptr, (size_t)L->nchunks,
                                rocprim::maximum<int>());
    if (e != hipSuccess) return false;
    if (scan > tmp) tmp = scan;
    size_t o = 0;
    L->off_keys_a = o; o = align_up(o + (size_t)L->n * 8, 256);
    L->off_keys_b = o; o = align_up(o + (size_t)L->n * 8, 256);
    L->off_vals_a = o; o = align_up(o + (size_t)L->n * 4, 256);
    L->off_vals_b = o; o = align_up(o + (size_t)L->n * 4, 256);
    L->off_carry = o; o = align_up(o + (size_t)L->nchunks * D * 4, 256);
    L->off_flags = o; o = align_up(o + (size_t)L->nchunks * 4, 256);
    L->off_home = o; o = align_up(o + (size_t)L->nchunks * 4, 256);
    L->off_pieces = o; o = align_up(o + ((size_t)L->n + (size_t)L->nchunks) * 16, 256);  // <= one cut per run + per chunk
    L->off_counter = o; o = align_up(o + 4, 256);
    L->off_tmp = o; L->tmp_bytes = tmp; o = align_up(o + tmp, 256);
    L->total = o;
    return true;
}

template <typename KeyT>
int32_t run_pipeline(const BwdArgs& a, const WsLayout& L, char* ws, int ids_dtype, int64_t B, int F, int D, int bits,
                     const float* grad, int64_t grad_row_stride, int optimizer, const OptHyper& hp, hipStream_t s) {
    KeyT* keys_a = reinterpret_cast<KeyT*>(ws + L.off_keys_a);
    KeyT* keys_b = reinterpret_cast<KeyT*>(ws + L.off_keys_b);
    uint32_t* vals_a = reinterpret_cast<uint32_t*>(ws + L.off_vals_a);
    uint32_t* vals_b = reinterpret_cast<uint32_t*>(ws + L.off_vals_b);
    float* carry = reinterpret_cast<float*>(ws + L.off_carry);
    int* flags = reinterpret_cast<int*>(ws + L.off_flags);
    int* lasthome = reinterpret_cast<int*>(ws + L.off_home);
    ulonglong2* pieces = reinterpret_cast<ulonglong2*>(ws + L.off_pieces);
    unsigned int* counter = reinterpret_cast<unsigned int*>(ws + L.off_counter);

    dim3 gk((unsigned)mh_ceil_div(L.n, 256));
    if (ids_dtype == MH_I32)
        hipLaunchKernelGGL((build_keys_kernel<int32_t, KeyT>), gk, dim3(256), 0, s, a, B, F, keys_a, vals_a, counter);
    else
        hipLaunchKernelGGL((build_keys_kernel<int64_t, KeyT>), gk, dim3(256), 0, s, a, B, F, keys_a, vals_a, counter);
    // stable sort over the key bits in use only; the all-ones sentinel stays last because no valid key has all
    // of those bits set (bits = ceil(log2(total rows + 1)))
    size_t tmp_bytes = L.tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(ws + L.off_tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b,
                                                         (size_t)L.n, 0, bits, s);
    if (e != hipSuccess) {
        mh_set_error("mh_embedding_gather_bwd: radix sort failed: %s", hipGetErrorString(e));
        return MH_ERR_LAUNCH;
    }
    (void)hipMemsetAsync(carry, 0, (size_t)L.nchunks * D * sizeof(float), s);
    const int LPR = D / 4;
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    dim3 gs((unsigned)mh_ceil_div(L.nchunks, groups));
    hipLaunchKernelGGL((chunk_flags_kernel<KeyT>), dim3((unsigned)mh_ceil_div(L.nchunks, 256)), dim3(256), 0, s, keys_b,
                       L.n, L.nchunks, flags);
    size_t scan_bytes = L.tmp_bytes;
    e = rocprim::inclusive_scan(ws + L.off_tmp, scan_bytes, flags, lasthome, (size_t)L.nchunks, rocprim::maximum<int>(), s);
    if (e != hipSuccess) {
        mh_set_error("mh_embedding_gather_bwd: scan failed: %s", hipGetErrorString(e));
        return MH_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((piece_list_kernel<KeyT>), dim3((unsigned)mh_ceil_div(L.n, 256 * LIST_TILES)), dim3(256), 0, s,
                       keys_b, vals_b, L.n, pieces, counter);
    {
        int64_t nb = mh_ceil_div(L.n, groups);  // never more groups than entries
        static int resident = 0;  // workgroups per CU the kernel's register budget allows: exactly one resident wave
        if (resident == 0) {
            int r = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, piece_reduce_apply_kernel, 256, 0) != hipSuccess || r < 1) r = 4;
            resident = r;
        }
        const int64_t cap = (int64_t)mh_num_cus() * resident;
        if (nb > cap) nb = cap;
        hipLaunchKernelGGL(piece_reduce_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, a, vals_b, D, LPR, grad,
                           grad_row_stride, carry, lasthome, pieces, counter, optimizer, hp);
    }
    hipLaunchKernelGGL((carry_apply_kernel<KeyT>), gs, dim3(256), 0, s, a, keys_b, vals_b, flags, L.n, D, LPR, carry, optimizer, hp);
    MH_CHECK_LAUNCH("mh_embedding_gather_bwd");
    return MH_OK;
}

// ---- ragged / list lookups: backward = one-hot backward over the nnz values ------------------------------
// d table[values[j]] += grad[bag(j)] / div(bag(j))   with div = 1 | kept | sqrt(kept) (sum | mean | sqrtn).
// bag_scale_kernel: one wave per bag counts the kept (non-negative) ids; bag_expand_kernel writes the scaled
// gradient row of every value, which then goes through the sort / segment-reduce / fused-optimizer pipeline
// above with B := nnz, F := 1.
template <typename IdT>
__global__ __launch_bounds__(256) void bag_scale_kernel(const IdT* __restrict__ values,
                                                        const IdT* __restrict__ offsets, int64_t L, int64_t B,
                                                        int combiner, float* __restrict__ scale) {
    const int64_t bag = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bag >= B) return;
    const int lane = threadIdx.x & 63;
    int64_t kept;
    if (offsets) {  // safe_embedding_lookup_sparse: negative ids are pruned and not counted
        const int64_t beg = (int64_t)offsets[bag], end = (int64_t)offsets[bag + 1];
        int cnt = 0;
        for (int64_t p = beg + lane; p < end; p += 64) cnt += (values[p] >= 0) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        kept = cnt;
    } else {
        kept = L;  // dense list: every position counts
    }
    if (lane == 0) {
        // the DIVISOR of the forward (sum / n, sum / sqrt(n)): the backward divides too, like the gradient of
        // div_no_nan in the reference graph (a bag with nothing kept has no value that receives an update)
        float dv = 1.f;
        if (combiner == MH_COMBINER_MEAN && kept > 0) dv = (float)kept;
        if (combiner == MH_COMBINER_SQRTN && kept > 0) dv = sqrtf((float)kept);
        scale[bag] = dv;
    }
}

template <typename IdT>
__global__ __launch_bounds__(256) void bag_expand_kernel(const IdT* __restrict__ offsets, int64_t L, int64_t B,
                                                         int64_t nnz, int LPR, const float* __restrict__ scale,
                                                         const float* __restrict__ grad, int64_t ldg,
                                                         float* __restrict__ gexp) {
    const int groups = 256 / LPR;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    if (gi >= groups) return;
    for (int64_t j = (int64_t)blockIdx.x * groups + gi; j < nnz; j += (int64_t)gridDim.x * groups) {
        int64_t bag;
        if (offsets) {  // last bag whose offset is <= j (empty bags share an offset with their successor)
            int64_t lo = 0, hi = B;  // invariant: offsets[lo] <= j < offsets[hi]
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)offsets[mid] <= j) lo = mid; else hi = mid;
            }
            bag = lo;
        } else {
            bag = j / L;
        }
        const float dv = scale[bag];
        const f32x4 g = *reinterpret_cast<const f32x4*>(grad + bag * ldg + c4 * 4);
        *reinterpret_cast<f32x4*>(gexp + j * (int64_t)(LPR * 4) + c4 * 4) = g / dv;
    }
}

}  // namespace

extern "C" {

int64_t mh_embedding_bwd_workspace_bytes(int64_t B, int32_t F, int32_t D) {
    if (B <= 0 || F <= 0 || D <= 0) return 0;
    WsLayout L;
    if (!ws_layout(B, F, D, &L)) return -1;
    return (int64_t)L.total;
}

int32_t mh_embedding_gather_bwd(float* const* tables, float* const* state, const int64_t* table_rows,
                                const void* const* ids, int32_t ids_dtype, int64_t B, int32_t F, int32_t D,
                                const float* grad, int64_t grad_row_stride, const int64_t* grad_offset,
                                int32_t optimizer, float lr, float eps, float* const* state2, float beta1, float beta2,
                                const float* lr_device, void* workspace, int64_t workspace_bytes,
                                mh_stream_t stream) {
    MH_REQUIRE(tables && table_rows && ids && grad && grad_offset, "mh_embedding_gather_bwd: null argument");
    MH_REQUIRE(F >= 1 && F < MH_MAX_FEATURES, "mh_embedding_gather_bwd: F=%d outside [1,%d]", F, MH_MAX_FEATURES - 1);
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "mh_embedding_gather_bwd: D=%d must be a multiple of 4 in [4,1024]", D);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_gather_bwd: bad ids_dtype");
    MH_REQUIRE(optimizer >= MH_OPT_SGD && optimizer <= MH_OPT_ADAM, "mh_embedding_gather_bwd: bad optimizer %d", optimizer);
    MH_REQUIRE(optimizer == MH_OPT_SGD || state, "mh_embedding_gather_bwd: Adagrad / Adam need state tables");
    MH_REQUIRE(optimizer != MH_OPT_ADAM || state2, "mh_embedding_gather_bwd: Adam needs the second-moment tables");
    MH_REQUIRE(grad_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0,
               "mh_embedding_gather_bwd: grad must be 16-byte aligned with grad_row_stride %% 4 == 0");
    if (B <= 0) return MH_OK;
    MH_REQUIRE(B < (1ll << 26), "mh_embedding_gather_bwd: B must be < 2^26");
    MH_REQUIRE(B * F < (1ll << 31), "mh_embedding_gather_bwd: B*F must be < 2^31");
    WsLayout L;
    MH_REQUIRE(ws_layout(B, F, D, &L), "mh_embedding_gather_bwd: rocprim size query failed");
    if (!workspace || workspace_bytes < (int64_t)L.total) {
        mh_set_error("mh_embedding_gather_bwd: workspace too small (%lld < %zu)", (long long)workspace_bytes, L.total);
        return MH_ERR_WORKSPACE;
    }
    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    int64_t total_rows = 0;  // rows of the distinct tables back to back = the compact key space
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(table_rows[f] >= 1, "mh_embedding_gather_bwd: table %d has no rows", f);
        MH_REQUIRE(tables[f] && ids[f], "mh_embedding_gather_bwd: null table/ids for feature %d", f);
        MH_REQUIRE(optimizer == MH_OPT_SGD || state[f], "mh_embedding_gather_bwd: null optimizer state for feature %d", f);
        MH_REQUIRE(optimizer != MH_OPT_ADAM || state2[f], "mh_embedding_gather_bwd: null second moment for feature %d", f);
        a.table[f] = tables[f];
        a.state[f] = state ? state[f] : nullptr;
        a.state2[f] = state2 ? state2[f] : nullptr;
        a.ids[f] = ids[f];
        a.rows[f] = table_rows[f];
        MH_REQUIRE(grad_offset[f] >= 0 && grad_offset[f] % 4 == 0 && grad_offset[f] + D <= grad_row_stride,
                   "mh_embedding_gather_bwd: grad offset of feature %d misaligned or out of row", f);
        a.offset[f] = grad_offset[f];
        a.first[f] = -1;
        for (int g = 0; g < f; ++g)
            if (tables[g] == tables[f]) {
                MH_REQUIRE(table_rows[g] == table_rows[f], "mh_embedding_gather_bwd: features %d and %d share a table but not its row count", g, f);
                a.first[f] = a.first[g];
                break;
            }
        if (a.first[f] < 0) {
            a.first[f] = total_rows;
            total_rows += table_rows[f];
        }
    }
    char* ws = static_cast<char*>(workspace);
    hipStream_t s = mh_stream(stream);
    const OptHyper hp = {lr, eps, beta1, beta2, lr_device};
    int bits = 1;
    while (bits < 64 && (1ull << bits) < (uint64_t)total_rows + 1) ++bits;
    if ((uint64_t)total_rows < 0xffffffffull)
        return run_pipeline<uint32_t>(a, L, ws, ids_dtype, B, F, D, bits, grad, grad_row_stride, optimizer, hp, s);
    return run_pipeline<uint64_t>(a, L, ws, ids_dtype, B, F, D, bits, grad, grad_row_stride, optimizer, hp, s);
}

int64_t mh_embedding_bag_bwd_workspace_bytes(int64_t B, int64_t nnz, int32_t D) {
    if (B <= 0 || nnz <= 0 || D <= 0) return 0;
    const int64_t inner = mh_embedding_bwd_workspace_bytes(nnz, 1, D);
    if (inner < 0) return -1;
    return (int64_t)align_up((size_t)B * sizeof(float), 256) + (int64_t)align_up((size_t)nnz * D * sizeof(float), 256) + inner;
}

int32_t mh_embedding_bag_bwd(float* table, float* state, float* state2, int64_t rows, const void* values,
                             int64_t nnz, const void* offsets, int64_t L, int32_t ids_dtype, int64_t B, int32_t D,
                             int32_t combiner, const float* grad, int64_t grad_row_stride, int32_t optimizer, float lr,
                             float eps, float beta1, float beta2, const float* lr_device, void* workspace,
                             int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(table && grad, "mh_embedding_bag_bwd: null argument");
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_bag_bwd: bad ids_dtype");
    MH_REQUIRE(combiner >= MH_COMBINER_SUM && combiner <= MH_COMBINER_SQRTN, "mh_embedding_bag_bwd: bad combiner %d", combiner);
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "mh_embedding_bag_bwd: D=%d must be a multiple of 4 in [4,1024]", D);
    MH_REQUIRE(grad_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0,
               "mh_embedding_bag_bwd: grad must be 16-byte aligned with grad_row_stride %% 4 == 0");
    MH_REQUIRE(offsets || L >= 1, "mh_embedding_bag_bwd: need CSR offsets or a list length L >= 1");
    if (B <= 0 || nnz <= 0) return MH_OK;
    MH_REQUIRE(values && workspace, "mh_embedding_bag_bwd: null values / workspace");
    MH_REQUIRE(offsets || nnz == B * L, "mh_embedding_bag_bwd: dense list needs nnz == B*L");
    MH_REQUIRE(nnz < (1ll << 26), "mh_embedding_bag_bwd: nnz must be < 2^26");
    const int64_t need = mh_embedding_bag_bwd_workspace_bytes(B, nnz, D);
    MH_REQUIRE(need >= 0 && workspace_bytes >= need, "mh_embedding_bag_bwd: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)need);
    hipStream_t s = mh_stream(stream);
    char* ws = static_cast<char*>(workspace);
    float* scale = reinterpret_cast<float*>(ws);
    ws += align_up((size_t)B * sizeof(float), 256);
    float* gexp = reinterpret_cast<float*>(ws);
    ws += align_up((size_t)nnz * D * sizeof(float), 256);
    const int LPR = D / 4;
    const int groups = 256 / LPR;
    const dim3 gs((unsigned)mh_ceil_div(B, 4));
    int64_t nb = mh_ceil_div(nnz, groups);
    const int64_t cap = (int64_t)mh_num_cus() * 16;
    if (nb > cap) nb = cap;
    if (ids_dtype == MH_I32) {
        hipLaunchKernelGGL((bag_scale_kernel<int32_t>), gs, dim3(256), 0, s, (const int32_t*)values,
                           (const int32_t*)offsets, L, B, combiner, scale);
        hipLaunchKernelGGL((bag_expand_kernel<int32_t>), dim3((unsigned)nb), dim3(256), 0, s, (const int32_t*)offsets, L,
                           B, nnz, LPR, scale, grad, grad_row_stride, gexp);
    } else {
        hipLaunchKernelGGL((bag_scale_kernel<int64_t>), gs, dim3(256), 0, s, (const int64_t*)values,
                           (const int64_t*)offsets, L, B, combiner, scale);
        hipLaunchKernelGGL((bag_expand_kernel<int64_t>), dim3((unsigned)nb), dim3(256), 0, s, (const int64_t*)offsets, L,
                           B, nnz, LPR, scale, grad, grad_row_stride, gexp);
    }
    MH_CHECK_LAUNCH("mh_embedding_bag_bwd");
    float* tabs[1] = {table};
    float* st1[1] = {state};
    float* st2[1] = {state2};
    const int64_t trows[1] = {rows};
    const void* idp[1] = {values};
    const int64_t off[1] = {0};
    return mh_embedding_gather_bwd(tabs, state ? st1 : nullptr, trows, idp, ids_dtype, nnz, 1, D, gexp, D, off, optimizer,
                                   lr, eps, state2 ? st2 : nullptr, beta1, beta2, lr_device, ws,
                                   workspace_bytes - (ws - static_cast<char*>(workspace)), stream);
}


}  // extern "C"
