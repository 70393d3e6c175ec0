// Backward of the one-hot embedding lookup fused with the sparse optimizer step (gfx950).
// Reference: the IndexedSlices branch of BaseModel.train_step (merlin/models/tf/models/base.py:
// 1121-1174): Keras sums duplicate indices first (_deduplicate_indexed_slices) and then applies
// the optimizer row-wise, so the update must see the SUM of a row's gradients exactly once.
//
// Pipeline (all on one stream, no host sync):
//   1. build 64-bit keys (table << 40 | id) + packed (feature, sample) for every (feature, sample)
//      -- features sharing one table share its key space, so shared rows are updated once;
//      out-of-range ids get the all-ones sentinel and sort to the end;
//   2. rocPRIM radix sort of the (key, sample) pairs (46 key bits) -- library plumbing;
//   3. segmented reduce over fixed chunks of 16 sorted entries: one D/4-lane group per chunk sums
//      the gradient rows of each run in registers; a run wholly inside the chunk is applied to
//      the table row directly (exclusive owner, no atomics); a run crossing a chunk boundary adds
//      its piece to carry[home chunk] (home = chunk holding the run's first entry, found by a
//      binary search for continuing runs) -- long runs of hot ids are pre-summed 16:1;
//   4. each chunk that is home to a crossing run applies the carried sum.
// HBM traffic: grad rows read once (random), table (+state) rows read+written once per UNIQUE id.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mh_common.h"

namespace {

constexpr int CHUNK = 16;
constexpr uint64_t SENTINEL = ~0ull;
constexpr int KEY_BITS = 46;
constexpr uint64_t ID_MASK = (1ull << 40) - 1;

struct BwdArgs {
    float* table[MH_MAX_FEATURES];
    float* state[MH_MAX_FEATURES];
    float* state2[MH_MAX_FEATURES];  // Adam second moment
    const void* ids[MH_MAX_FEATURES];
    int64_t rows[MH_MAX_FEATURES];
    int64_t offset[MH_MAX_FEATURES];  // float offset of the feature inside a grad row
    int32_t tid[MH_MAX_FEATURES];  // first feature sharing the same table (shared embeddings)
};

template <typename IdT>
__global__ __launch_bounds__(256) void build_keys_kernel(const BwdArgs a, int64_t B, int F,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * F) return;
    const int f = (int)(idx / B);
    const int64_t b = idx - (int64_t)f * B;
    const int64_t id = (int64_t) static_cast<const IdT*>(a.ids[f])[b];
    const bool ok = id >= 0 && id < a.rows[f];
    keys[idx] = ok ? (((uint64_t)a.tid[f] << 40) | (uint64_t)id) : SENTINEL;
    vals[idx] = ((uint32_t)f << 26) | (uint32_t)b;
}

struct OptHyper {
    float lr, eps, beta1, beta2;
    const float* lr_dev;  // optional device scalar overriding lr (graph-replayable bias correction)
};

__device__ __forceinline__ void apply_update(const BwdArgs& a, uint64_t key, f32x4 g, int D, int c4, int opt,
                                             const OptHyper& hp) {
    const float lr = hp.lr_dev ? *hp.lr_dev : hp.lr;
    const float eps = hp.eps;
    const int f = (int)(key >> 40);
    const int64_t id = (int64_t)(key & ID_MASK);
    float* w = a.table[f] + id * D + c4 * 4;
    f32x4 wv = *reinterpret_cast<f32x4*>(w);
    if (opt == MH_OPT_ADAGRAD) {
        float* st = a.state[f] + id * D + c4 * 4;
        f32x4 sv = *reinterpret_cast<f32x4*>(st);
        sv += g * g;
        *reinterpret_cast<f32x4*>(st) = sv;
        wv.x -= lr * g.x / (sqrtf(sv.x) + eps);
        wv.y -= lr * g.y / (sqrtf(sv.y) + eps);
        wv.z -= lr * g.z / (sqrtf(sv.z) + eps);
        wv.w -= lr * g.w / (sqrtf(sv.w) + eps);
    } else if (opt == MH_OPT_ADAM) {
        // LazyAdam._resource_apply_sparse (blocks/optimizer.py:412-437): only the touched rows' moments move
        float* mp = a.state[f] + id * D + c4 * 4;
        float* vp = a.state2[f] + id * D + c4 * 4;
        f32x4 m = *reinterpret_cast<f32x4*>(mp);
        f32x4 v = *reinterpret_cast<f32x4*>(vp);
        m = m * hp.beta1 + g * (1.f - hp.beta1);
        v = v * hp.beta2 + (g * g) * (1.f - hp.beta2);
        *reinterpret_cast<f32x4*>(mp) = m;
        *reinterpret_cast<f32x4*>(vp) = v;
        wv.x -= lr * m.x / (sqrtf(v.x) + eps);
        wv.y -= lr * m.y / (sqrtf(v.y) + eps);
        wv.z -= lr * m.z / (sqrtf(v.z) + eps);
        wv.w -= lr * m.w / (sqrtf(v.w) + eps);
    } else {
        wv -= g * lr;
    }
    *reinterpret_cast<f32x4*>(w) = wv;
}

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ keys, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// chunk c: v[c] = -1 if its first run continues from the previous chunk AND the whole chunk is that one
// run (the run's home lies further back), else c.  An inclusive max-scan of v gives lasthome[c] = home
// chunk of the LAST run of chunk c; the head piece of a continuing chunk c then belongs to lasthome[c-1].
__global__ __launch_bounds__(256) void chunk_flags_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                         int64_t nchunks, int* __restrict__ v) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    const int64_t c0 = c * CHUNK;
    const int64_t c1 = (c0 + CHUNK < n) ? c0 + CHUNK : n;
    const bool cont = (c0 > 0) && (keys[c0 - 1] == keys[c0]);
    const bool whole = keys[c1 - 1] == keys[c0];  // sorted: first == last  <=>  one run
    v[c] = (cont && whole) ? -1 : (int)c;
}

// One D/4-lane group per chunk of 16 sorted entries: run sums are formed in registers; a run wholly
// inside the chunk is applied to its table row directly (exclusive owner, no atomics); a run crossing
// a chunk boundary adds its piece to carry[home chunk] (home from the scanned chunk flags).
// The block's keys / packed values are staged in LDS with coalesced loads (one dependent HBM round trip
// per block instead of two per entry) and gradient rows are fetched four at a time ahead of the sequential
// run logic: enough memory-level parallelism at ~70 VGPRs (fully batching all 16 rows + the table
// read-modify-writes needed 250 VGPRs and ran slower -- profiles/r1_notes.md).
__global__ __launch_bounds__(256) void segment_reduce_apply_kernel(const BwdArgs a, const uint64_t* __restrict__ keys,
                                                                  const uint32_t* __restrict__ vals, int64_t n,
                                                                  int D, int LPR, const float* __restrict__ grad,
                                                                  int64_t grad_row_stride, float* __restrict__ carry,
                                                                  const int* __restrict__ lasthome,
                                                                  int opt, const OptHyper hp) {
    __shared__ uint64_t key_s[64 * CHUNK + 2];
    __shared__ uint32_t val_s[64 * CHUNK];
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    const int64_t blk0 = (int64_t)blockIdx.x * groups * CHUNK;
    const int span = groups * CHUNK;
    for (int t = threadIdx.x; t < span + 2; t += 256) {
        const int64_t i = blk0 - 1 + t;
        key_s[t] = (i >= 0 && i < n) ? keys[i] : SENTINEL;
    }
    for (int t = threadIdx.x; t < span; t += 256) {
        const int64_t i = blk0 + t;
        val_s[t] = (i < n) ? vals[i] : 0u;
    }
    __syncthreads();
    if (gi >= groups) return;
    const int64_t chunk = (int64_t)blockIdx.x * groups + gi;
    const int64_t c0 = chunk * CHUNK;
    if (c0 >= n) return;
    const int cnt_all = (int)(((c0 + CHUNK < n) ? c0 + CHUNK : n) - c0);
    const uint64_t* kk = key_s + 1 + gi * CHUNK;  // kk[-1]: key before the chunk, kk[cnt_all]: key after
    const uint32_t* vv = val_s + gi * CHUNK;
    uint64_t cur = kk[0];
    if (cur == SENTINEL) return;
    const bool head_cont = (c0 > 0) && (kk[-1] == cur);
    int run_start = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};

    auto flush = [&](uint64_t key, int s_, int e_) {
        const bool starts_here = (s_ > 0) || !head_cont;
        const bool ends_here = (e_ < cnt_all) || (c0 + e_ == n) || (kk[e_] != key);
        if (starts_here && ends_here) {
            apply_update(a, key, acc, D, c4, opt, hp);
        } else {
            const int64_t home = starts_here ? chunk : (int64_t)lasthome[chunk - 1];
            float* cr = carry + home * D + c4 * 4;
            atomicAdd(cr + 0, acc.x);
            atomicAdd(cr + 1, acc.y);
            atomicAdd(cr + 2, acc.z);
            atomicAdd(cr + 3, acc.w);
        }
    };

    int i = 0;
    bool done = false;
    for (int base = 0; base < CHUNK && !done; base += 4) {
        f32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = base + j;
            r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < cnt_all && kk[u] != SENTINEL) {
                const uint32_t v = vv[u];
                const int f = (int)(v >> 26);
                r[j] = *reinterpret_cast<const f32x4*>(grad + (int64_t)(v & ((1u << 26) - 1)) * grad_row_stride +
                                                       a.offset[f] + c4 * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = base + j;
            if (done || u >= cnt_all) continue;
            const uint64_t k = kk[u];
            if (k == SENTINEL) {
                done = true;
                continue;
            }
            if (k != cur) {
                flush(cur, run_start, u);
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                cur = k;
                run_start = u;
            }
            acc += r[j];
            i = u + 1;
        }
    }
    flush(cur, run_start, i);
}

__global__ __launch_bounds__(256) void carry_apply_kernel(const BwdArgs a, const uint64_t* __restrict__ keys,
                                                         int64_t n, int D, int LPR, const float* __restrict__ carry,
                                                         int opt, const OptHyper hp) {
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    if (gi >= groups) return;
    const int64_t chunk = (int64_t)blockIdx.x * groups + gi;
    const int64_t c0 = chunk * CHUNK;
    if (c0 >= n) return;
    const int64_t c1 = (c0 + CHUNK < n) ? c0 + CHUNK : n;
    if (c1 == n) return;  // the last chunk cannot be crossed
    const uint64_t key = keys[c1 - 1];
    if (key == SENTINEL || keys[c1] != key) return;  // last run ends here
    int64_t s = c1 - 1;
    while (s > c0 && keys[s - 1] == key) --s;
    const bool starts_here = (s > c0) || (c0 == 0) || (keys[c0 - 1] != key);
    if (!starts_here) return;  // an earlier chunk is this run's home
    const f32x4 g = *reinterpret_cast<const f32x4*>(carry + chunk * D + c4 * 4);
    apply_update(a, key, g, D, c4, opt, hp);
}

struct WsLayout {
    int64_t n, nchunks;
    size_t off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_carry, off_flags, off_home, off_tmp, tmp_bytes,
        scan_bytes, total;
    int key_bits;
};

// rocPRIM switches to a merge sort (dozens of 5 us launches) below 1M items; Onesweep already wins from ~64K.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 65536>;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool ws_layout(int64_t B, int F, int D, WsLayout* L) {
    L->n = B * F;
    L->nchunks = mh_ceil_div(L->n, CHUNK);
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)L->n, 0, KEY_BITS);
    if (e != hipSuccess) return false;
    size_t scan = 0;
    e = rocprim::inclusive_scan(nullptr, scan, (const int*)nullptr, (int*)nullptr, (size_t)L->nchunks,
                                rocprim::maximum<int>());
    if (e != hipSuccess) return false;
    L->scan_bytes = scan;
    if (scan > tmp) tmp = scan;
    size_t o = 0;
    L->off_keys_a = o; o = align_up(o + (size_t)L->n * 8, 256);
    L->off_keys_b = o; o = align_up(o + (size_t)L->n * 8, 256);
    L->off_vals_a = o; o = align_up(o + (size_t)L->n * 4, 256);
    L->off_vals_b = o; o = align_up(o + (size_t)L->n * 4, 256);
    L->off_carry = o; o = align_up(o + (size_t)L->nchunks * D * 4, 256);
    L->off_flags = o; o = align_up(o + (size_t)L->nchunks * 4, 256);
    L->off_home = o; o = align_up(o + (size_t)L->nchunks * 4, 256);
    L->off_tmp = o; L->tmp_bytes = tmp; o = align_up(o + tmp, 256);
    L->total = o;
    return true;
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
    WsLayout L;
    MH_REQUIRE(ws_layout(B, F, D, &L), "mh_embedding_gather_bwd: rocprim size query failed");
    if (!workspace || workspace_bytes < (int64_t)L.total) {
        mh_set_error("mh_embedding_gather_bwd: workspace too small (%lld < %zu)", (long long)workspace_bytes, L.total);
        return MH_ERR_WORKSPACE;
    }
    BwdArgs a;
    for (int f = 0; f < F; ++f) {
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
        a.tid[f] = f;
        for (int g = 0; g < f; ++g)
            if (tables[g] == tables[f]) {
                a.tid[f] = g;
                break;
            }
    }
    char* ws = static_cast<char*>(workspace);
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(ws + L.off_keys_a);
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(ws + L.off_keys_b);
    uint32_t* vals_a = reinterpret_cast<uint32_t*>(ws + L.off_vals_a);
    uint32_t* vals_b = reinterpret_cast<uint32_t*>(ws + L.off_vals_b);
    float* carry = reinterpret_cast<float*>(ws + L.off_carry);
    hipStream_t s = mh_stream(stream);
    const OptHyper hp = {lr, eps, beta1, beta2, lr_device};

    dim3 gk((unsigned)mh_ceil_div(L.n, 256));
    if (ids_dtype == MH_I32)
        hipLaunchKernelGGL((build_keys_kernel<int32_t>), gk, dim3(256), 0, s, a, B, F, keys_a, vals_a);
    else
        hipLaunchKernelGGL((build_keys_kernel<int64_t>), gk, dim3(256), 0, s, a, B, F, keys_a, vals_a);
    size_t tmp_bytes = L.tmp_bytes;
    // sort only over the key bits in use: id bits of the largest table + the table index at bit 40..
    // (keys are (table << 40 | id); the sentinel is all ones, so it stays last under any bit window
    //  that includes the table field) -- two radix passes over [0, idbits) and [40, 40 + tbits) would
    // need a stable two-stage sort; rocPRIM's single call over [0, 46) costs 6 passes, so instead the
    // id field is sorted first and the (narrow) table field second, both stable.
    int idbits = 1;
    {
        int64_t maxrows = 1;
        for (int f = 0; f < F; ++f) maxrows = table_rows[f] > maxrows ? table_rows[f] : maxrows;
        while ((1ll << idbits) < maxrows) ++idbits;
        if (idbits > 40) idbits = 40;
    }
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(ws + L.off_tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)L.n, 0,
                                             idbits, s);
    if (e == hipSuccess) {
        tmp_bytes = L.tmp_bytes;
        e = rocprim::radix_sort_pairs<SortConfig>(ws + L.off_tmp, tmp_bytes, keys_b, keys_a, vals_b, vals_a, (size_t)L.n, 40,
                                      KEY_BITS, s);
    }
    {   // results are back in the *_a buffers
        uint64_t* tk = keys_a; keys_a = keys_b; keys_b = tk;
        uint32_t* tv = vals_a; vals_a = vals_b; vals_b = tv;
    }
    if (e != hipSuccess) {
        mh_set_error("mh_embedding_gather_bwd: radix sort failed: %s", hipGetErrorString(e));
        return MH_ERR_LAUNCH;
    }
    (void)hipMemsetAsync(carry, 0, (size_t)L.nchunks * D * sizeof(float), s);
    const int LPR = D / 4;
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    dim3 gs((unsigned)mh_ceil_div(L.nchunks, groups));
    int* flags = reinterpret_cast<int*>(ws + L.off_flags);
    int* lasthome = reinterpret_cast<int*>(ws + L.off_home);
    hipLaunchKernelGGL(chunk_flags_kernel, dim3((unsigned)mh_ceil_div(L.nchunks, 256)), dim3(256), 0, s, keys_b, L.n,
                       L.nchunks, flags);
    size_t scan_bytes = L.tmp_bytes;
    e = rocprim::inclusive_scan(ws + L.off_tmp, scan_bytes, flags, lasthome, (size_t)L.nchunks, rocprim::maximum<int>(), s);
    if (e != hipSuccess) {
        mh_set_error("mh_embedding_gather_bwd: scan failed: %s", hipGetErrorString(e));
        return MH_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(segment_reduce_apply_kernel, gs, dim3(256), 0, s, a, keys_b, vals_b, L.n, D, LPR, grad,
                       grad_row_stride, carry, lasthome, optimizer, hp);
    hipLaunchKernelGGL(carry_apply_kernel, gs, dim3(256), 0, s, a, keys_b, L.n, D, LPR, carry, optimizer, hp);
    MH_CHECK_LAUNCH("mh_embedding_gather_bwd");
    return MH_OK;
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
