// Row-sharded embedding exchange: build the all-to-all send order on the device.
//
// A request is entry e = f*B + b of F id columns; it goes to rank owner = id % W and asks for local row
// id / W of feature f (key = f << 40 | local row).  The send buffer must hold the requests grouped by owner;
// this is a STABLE counting sort by owner (W <= 64 buckets), so the order -- and therefore the order in which
// the owner's fused backward sums duplicate rows -- is a pure function of the ids:
//   pass 1  per-tile owner histogram                      -> hist[w][tile]
//   scan    exclusive prefix over the (w, tile) sequence  -> first send slot of every (owner, tile) pair (one workgroup)
//   pass 2  per-tile stable ranks (wave ballots), scatter keys / inverse permutation / gradient source rows
// Replaces ~45 framework launches (stack, shifts, remainder, merge sort, bincount, index ...) per step.
//
// De-duplicating route (mh_route_build_dedup): a row that many samples of the batch ask for travels ONCE per (sender, owner)
// pair -- SparseOperationKit does the same behind merlin/models/tf/distributed/embedding.py:144-148.  On Criteo-like ids
// (594 K distinct keys among 1.70 M requests at B = 65 536) that is ~35 % of the row and row-gradient bytes on xGMI.
//   pass 0  every request inserts its key (feature << 40 | id) into an open-addressing table (CAS on the key word) and
//           atomicMin's its entry number into the slot: the request with the SMALLEST entry number is the key's leader
//   pass 1-3 the counting sort above over the LEADERS only: a key's send slot is the rank of its first occurrence among its
//           owner's distinct keys -- a pure function of the ids, whatever order the atomics ran in; the leader notes the slot
//   pass 4  every request reads its key's send slot -> pos_of (the forward gather and the backward segment sum use it)
#include <cstring>

#include "mh_common.h"

namespace {

constexpr int TILE_IT = 8;
constexpr int TILE = 256 * TILE_IT;  // entries per workgroup
constexpr int MAX_W = 64;

struct RouteArgs {
    const void* ids[MH_MAX_FEATURES];
    int32_t slot[MH_MAX_FEATURES];
};

// ids are row numbers (>= 0); a stray negative id must not index outside the histogram
__device__ __forceinline__ int owner_of(int64_t id, int W) {
    const int o = (int)(id % W);
    return o < 0 ? o + W : o;
}

template <typename IdT>
__device__ __forceinline__ int64_t load_id(const RouteArgs& a, int64_t e, int64_t B, int& f, int64_t& b) {
    f = (int)(e / B);
    b = e - (int64_t)f * B;
    return (int64_t) reinterpret_cast<const IdT*>(a.ids[f])[b];
}

// The TILE entries of a workgroup lie inside ONE feature whenever B is a multiple of TILE (and almost always otherwise): the
// id column and the first sample are then workgroup-uniform, and a thread's TILE_IT ids are TILE_IT independent loads from a
// scalar base, issued together (index clamped, validity applied by the caller).  Entry by entry -- a 64-bit division, a
// pointer fetched from the kernel-argument table with a per-lane index, then the id, each behind a wait -- the loop was
// 2 x TILE_IT sequential memory round trips per thread (seen in the ISA).
template <typename IdT>
struct TileIds {
    bool one;   // workgroup-uniform
    int f0;
    int64_t b0;
    IdT raw[TILE_IT];
    __device__ __forceinline__ void load(const RouteArgs& a, int64_t e0, int64_t n, int64_t B) {
        const int64_t last = (e0 + TILE < n ? e0 + TILE : n) - 1;  // >= e0: a launched tile has entries
        f0 = (int)(e0 / B);
        b0 = e0 - (int64_t)f0 * B;
        one = last / B == f0;
        if (one) {
            const IdT* col = reinterpret_cast<const IdT*>(a.ids[f0]) + b0;
            const int64_t nt = last - e0 + 1;
#pragma unroll
            for (int it = 0; it < TILE_IT; ++it) {
                int64_t i = it * 256 + threadIdx.x;
                if (i > nt - 1) i = nt - 1;
                raw[it] = col[i];
            }
        }
    }
    // entry e = e0 + it * 256 + threadIdx.x < n
    __device__ __forceinline__ int64_t get(const RouteArgs& a, int it, int64_t e, int64_t B, int& f, int64_t& b) const {
        if (one) {
            f = f0;
            b = b0 + it * 256 + threadIdx.x;
            return (int64_t)raw[it];
        }
        return load_id<IdT>(a, e, B, f, b);
    }
};

constexpr unsigned long long EMPTY_KEY = ~0ull;

__device__ __forceinline__ uint32_t hash_key(unsigned long long x) {  // murmur3 finaliser
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return (uint32_t)x;
}

// pass 0 of the de-duplicating route.  slot_of[e] = the table slot of entry e's key (-1: a negative id, which no owner has)
template <typename IdT>
__global__ __launch_bounds__(256) void route_dedup_insert_kernel(const RouteArgs a, int64_t n, int64_t B,
                                                                 unsigned long long* __restrict__ tab_key,
                                                                 int* __restrict__ tab_first, uint32_t hmask,
                                                                 int* __restrict__ slot_of) {
    const int64_t e0 = (int64_t)blockIdx.x * TILE;
    TileIds<IdT> t;
    t.load(a, e0, n, B);
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        const int64_t e = e0 + it * 256 + threadIdx.x;
        if (e >= n) continue;
        int f;
        int64_t b;
        const int64_t id = t.get(a, it, e, B, f, b);
        if (id < 0) {
            slot_of[e] = -1;
            continue;
        }
        const unsigned long long key = ((unsigned long long)f << 40) | (unsigned long long)id;
        uint32_t h = hash_key(key) & hmask;
        for (;;) {  // the table has >= 2 n slots: a probe sequence ends
            const unsigned long long seen = tab_key[h];
            if (seen == key) break;
            if (seen == EMPTY_KEY) {
                const unsigned long long old = atomicCAS(tab_key + h, EMPTY_KEY, key);
                if (old == EMPTY_KEY || old == key) break;
            }
            h = (h + 1) & hmask;
        }
        atomicMin(tab_first + h, (int)e);
        slot_of[e] = (int)h;
    }
}

// pass 4: every request takes the send slot its key's leader was given (-1: dropped by a full window / negative id)
__global__ __launch_bounds__(256) void route_dedup_follow_kernel(const int* __restrict__ slot_of, const int* __restrict__ tab_pos,
                                                                 int64_t n, int64_t* __restrict__ pos_of) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int sl = slot_of[e];
    pos_of[e] = sl < 0 ? -1 : (int64_t)tab_pos[sl];
}

template <typename IdT, bool DEDUP>
__global__ __launch_bounds__(256) void route_count_kernel(const RouteArgs a, int64_t n, int64_t B, int W,
                                                          int64_t ntiles, int* __restrict__ hist,
                                                          const int* __restrict__ slot_of,
                                                          const int* __restrict__ tab_first) {
    __shared__ int h[MAX_W];
    if (threadIdx.x < MAX_W) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * TILE;
    TileIds<IdT> t;
    t.load(a, e0, n, B);
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        const int64_t e = e0 + it * 256 + threadIdx.x;
        if (e < n) {
            int f;
            int64_t b;
            const int64_t id = t.get(a, it, e, B, f, b);
            if (DEDUP) {  // only the leader of a key counts
                const int sl = slot_of[e];
                if (sl < 0 || tab_first[sl] != (int)e) continue;
            }
            atomicAdd(&h[owner_of(id, W)], 1);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < W) hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// counts[w] = requests for owner w = scan[(w+1)*ntiles] - scan[w*ntiles] (exclusive scan; last from n)
// hist != NULL (de-duplicating route): the total is the number of leaders, not n
__global__ void route_counts_kernel(const int* __restrict__ scan, int64_t ntiles, int W, int64_t n,
                                    int64_t* __restrict__ counts, const int* __restrict__ hist) {
    const int w = threadIdx.x;
    if (w >= W) return;
    const int64_t lo = scan[(int64_t)w * ntiles];
    const int64_t total = hist ? (int64_t)scan[(int64_t)W * ntiles - 1] + hist[(int64_t)W * ntiles - 1] : n;
    const int64_t hi = (w + 1 < W) ? (int64_t)scan[(int64_t)(w + 1) * ntiles] : total;
    counts[w] = hi - lo;
}

template <typename IdT, bool DEDUP>
__global__ __launch_bounds__(256) void route_scatter_kernel(const RouteArgs a, int64_t n, int64_t B, int W,
                                                            int64_t ntiles, int F_total,
                                                            const int* __restrict__ scan,
                                                            int64_t* __restrict__ send_keys,
                                                            int64_t* __restrict__ pos_of,
                                                            int64_t* __restrict__ src_row, int64_t cap,
                                                            int* __restrict__ overflow,
                                                            const int* __restrict__ slot_of,
                                                            const int* __restrict__ tab_first,
                                                            int* __restrict__ tab_pos) {
    __shared__ int run[MAX_W];         // slots of this tile already handed out, per owner
    __shared__ int wave_cnt[2][4][MAX_W];  // per-wave counts, double-buffered by iteration parity
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // dense mode (cap == 0): owner w's requests start at the exclusive prefix scan[w][0]; fixed-capacity mode: at w * cap
    if ((int)threadIdx.x < W) {
        const int64_t first = scan[(int64_t)threadIdx.x * ntiles];
        run[threadIdx.x] = scan[(int64_t)threadIdx.x * ntiles + blockIdx.x] - (cap > 0 ? (int)first : 0);
    }
    const int64_t e0 = (int64_t)blockIdx.x * TILE;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    TileIds<IdT> t;
    t.load(a, e0, n, B);
#pragma unroll
    for (int it = 0; it < TILE_IT; ++it) {
        const int64_t e = e0 + it * 256 + threadIdx.x;
        int owner = -1, f = 0;
        int64_t b = 0, id = 0;
        int sl = -1;
        if (e < n) {
            id = t.get(a, it, e, B, f, b);
            owner = owner_of(id, W);
            if (DEDUP) {  // followers take no slot: pass 4 gives them their leader's
                sl = slot_of[e];
                if (sl < 0 || tab_first[sl] != (int)e) owner = -1;
            }
        }
        int rank = 0;
        for (int w = 0; w < W; ++w) {
            const uint64_t m = __ballot(owner == w);
            if (owner == w) rank = __popcll(m & lt_mask);
            if (lane == 0) wave_cnt[it & 1][wave][w] = __popcll(m);
        }
        __syncthreads();  // wave_cnt complete; run[] of the previous iteration visible
        if (owner >= 0) {
            int64_t p = run[owner] + rank;
            for (int wv = 0; wv < wave; ++wv) p += wave_cnt[it & 1][wv][owner];
            if (cap > 0) {
                if (p >= cap) {  // this owner's fixed window is full: the request is dropped (zero row, no update)
                    if (!DEDUP) pos_of[e] = -1;
                    if (overflow) atomicOr(overflow, 1);
                    p = -1;
                } else {
                    p += (int64_t)owner * cap;
                }
            }
            if (DEDUP) tab_pos[sl] = (int)p;
            if (p >= 0) {
                send_keys[p] = ((int64_t)f << 40) | (id / W);
                if (!DEDUP) {
                    pos_of[e] = p;
                    src_row[p] = b * F_total + a.slot[f];
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < W)
            run[threadIdx.x] += wave_cnt[it & 1][0][threadIdx.x] + wave_cnt[it & 1][1][threadIdx.x] +
                                wave_cnt[it & 1][2][threadIdx.x] + wave_cnt[it & 1][3][threadIdx.x];
    }
}

// rows[i] = base[f] + local row, or -1 (a row the gather reads as zeros and the fused update skips) for the padding
// keys of the fixed-capacity exchange (key < 0) and for local rows outside feature f's shard (an id >= the table's
// cardinality must not reach the NEXT feature's shard of the concatenated buffer)
__global__ void route_local_rows_kernel(const int64_t* __restrict__ keys, int64_t n,
                                        const int64_t* __restrict__ base, const int64_t* __restrict__ shard_rows,
                                        int F, int64_t* __restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t k = keys[i];
    const int f = (int)(k >> 40);
    const int64_t r = k & ((1ll << 40) - 1);
    const bool ok = k >= 0 && f < F && (!shard_rows || r < shard_rows[f]);
    rows[i] = ok ? base[f] + r : -1;
}

// exclusive prefix of the (owner, tile) histogram, one workgroup (the sequence is W * ntiles ints: a few thousand)
__global__ __launch_bounds__(1024) void route_scan_kernel(const int* __restrict__ in, int* __restrict__ out, int64_t len) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < len; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = (i < len) ? in[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < len) out[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + x;
        __syncthreads();
    }
}

struct RouteWs {
    int64_t ntiles, hslots;
    size_t off_hist, off_scan, off_key, off_first, off_pos, off_slot, total;
};

size_t up(size_t v) { return (v + 255) / 256 * 256; }

bool route_ws(int64_t n, int W, RouteWs* L, bool dedup = false) {
    L->ntiles = mh_ceil_div(n, TILE);
    const size_t cells = (size_t)L->ntiles * W;
    L->off_hist = 0;
    L->off_scan = up(cells * sizeof(int));
    L->total = L->off_scan + up(cells * sizeof(int));
    L->hslots = 0;
    if (dedup) {  // open addressing at a load factor <= 1/2
        int64_t h = 1024;
        while (h < 2 * n) h <<= 1;
        L->hslots = h;
        L->off_key = L->total;
        L->off_first = L->off_key + up((size_t)h * 8);
        L->off_pos = L->off_first + up((size_t)h * 4);
        L->off_slot = L->off_pos + up((size_t)h * 4);
        L->total = L->off_slot + up((size_t)n * 4);
    }
    return true;
}

}  // namespace

extern "C" {

int64_t mh_route_workspace_bytes(int64_t n, int32_t W) {
    if (n <= 0 || W <= 0 || W > MAX_W) return 0;
    RouteWs L;
    if (!route_ws(n, W, &L)) return -1;
    return (int64_t)L.total;
}

int32_t mh_route_build(const void* const* ids, int32_t ids_dtype, int32_t F, int64_t B, int32_t W,
                       const int32_t* slots, int32_t F_total, int64_t capacity, int64_t* send_keys, int64_t* pos_of,
                       int64_t* src_row, int64_t* counts, int32_t* overflow, void* workspace, int64_t workspace_bytes,
                       mh_stream_t stream) {
    MH_REQUIRE(ids && slots && counts, "mh_route_build: null argument");
    MH_REQUIRE(F >= 1 && F <= MH_MAX_FEATURES, "mh_route_build: F=%d outside [1,%d]", F, MH_MAX_FEATURES);
    MH_REQUIRE(W >= 1 && W <= MAX_W, "mh_route_build: world size %d outside [1,%d]", W, MAX_W);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_route_build: bad ids_dtype");
    MH_REQUIRE(F_total >= 1, "mh_route_build: F_total must be >= 1");
    hipStream_t s = mh_stream(stream);
    if (B <= 0) {
        return mh_fill_words(counts, 0u, 2 * (int64_t)W, s);
    }
    MH_REQUIRE(send_keys && pos_of && src_row && workspace, "mh_route_build: null output");
    MH_REQUIRE(capacity >= 0, "mh_route_build: negative capacity");
    const int64_t n = B * F;
    if (capacity > 0) {  // padding slots: key -1 (no row), source row -1 (zero gradient)
        int32_t st = mh_fill_words(send_keys, 0xffffffffu, 2 * (int64_t)W * capacity, s);
        if (st == MH_OK) st = mh_fill_words(src_row, 0xffffffffu, 2 * (int64_t)W * capacity, s);
        if (st != MH_OK) return st;
    }
    MH_REQUIRE(n < (1ll << 31), "mh_route_build: F*B must be < 2^31");
    RouteWs L;
    route_ws(n, W, &L);
    MH_REQUIRE(workspace_bytes >= (int64_t)L.total, "mh_route_build: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)L.total);
    RouteArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(ids[f], "mh_route_build: ids[%d] is null", f);
        MH_REQUIRE(slots[f] >= 0 && slots[f] < F_total, "mh_route_build: slot %d outside [0,%d)", slots[f], F_total);
        a.ids[f] = ids[f];
        a.slot[f] = slots[f];
    }
    char* ws = static_cast<char*>(workspace);
    int* hist = reinterpret_cast<int*>(ws + L.off_hist);
    int* scan = reinterpret_cast<int*>(ws + L.off_scan);
    const dim3 grid((unsigned)L.ntiles);
    const int* none = nullptr;
    int* none_w = nullptr;
    if (ids_dtype == MH_I32)
        MH_LAUNCH((route_count_kernel<int32_t, false>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, hist, none, none);
    else
        MH_LAUNCH((route_count_kernel<int64_t, false>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, hist, none, none);
    MH_LAUNCH(route_scan_kernel, dim3(1), dim3(1024), 0, s, hist, scan, L.ntiles * W);
    MH_LAUNCH(route_counts_kernel, dim3(1), dim3(64), 0, s, scan, L.ntiles, W, n, counts, none);
    if (ids_dtype == MH_I32)
        MH_LAUNCH((route_scatter_kernel<int32_t, false>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, F_total, scan,
                           send_keys, pos_of, src_row, capacity, overflow, none, none, none_w);
    else
        MH_LAUNCH((route_scatter_kernel<int64_t, false>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, F_total, scan,
                           send_keys, pos_of, src_row, capacity, overflow, none, none, none_w);
    MH_CHECK_LAUNCH("mh_route_build");
    return MH_OK;
}

int64_t mh_route_dedup_workspace_bytes(int64_t n, int32_t W) {
    if (n <= 0 || W <= 0 || W > MAX_W) return 0;
    RouteWs L;
    if (!route_ws(n, W, &L, true)) return -1;
    return (int64_t)L.total;
}

int32_t mh_route_build_dedup(const void* const* ids, int32_t ids_dtype, int32_t F, int64_t B, int32_t W, int64_t capacity,
                             int64_t* send_keys, int64_t* pos_of, int64_t* counts, int32_t* overflow, void* workspace,
                             int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(ids && counts, "mh_route_build_dedup: null argument");
    MH_REQUIRE(F >= 1 && F <= MH_MAX_FEATURES, "mh_route_build_dedup: F=%d outside [1,%d]", F, MH_MAX_FEATURES);
    MH_REQUIRE(W >= 1 && W <= MAX_W, "mh_route_build_dedup: world size %d outside [1,%d]", W, MAX_W);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_route_build_dedup: bad ids_dtype");
    hipStream_t s = mh_stream(stream);
    if (B <= 0) return mh_fill_words(counts, 0u, 2 * (int64_t)W, s);
    MH_REQUIRE(send_keys && pos_of && workspace, "mh_route_build_dedup: null output");
    MH_REQUIRE(capacity >= 0, "mh_route_build_dedup: negative capacity");
    const int64_t n = B * F;
    MH_REQUIRE(n < (1ll << 30), "mh_route_build_dedup: F*B must be < 2^30");
    RouteWs L;
    route_ws(n, W, &L, true);
    MH_REQUIRE(workspace_bytes >= (int64_t)L.total, "mh_route_build_dedup: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)L.total);
    RouteArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(ids[f], "mh_route_build_dedup: ids[%d] is null", f);
        a.ids[f] = ids[f];
    }
    char* ws = static_cast<char*>(workspace);
    int* hist = reinterpret_cast<int*>(ws + L.off_hist);
    int* scan = reinterpret_cast<int*>(ws + L.off_scan);
    unsigned long long* tab_key = reinterpret_cast<unsigned long long*>(ws + L.off_key);
    int* tab_first = reinterpret_cast<int*>(ws + L.off_first);
    int* tab_pos = reinterpret_cast<int*>(ws + L.off_pos);
    int* slot_of = reinterpret_cast<int*>(ws + L.off_slot);
    int32_t st = MH_OK;
    if (capacity > 0) st = mh_fill_words(send_keys, 0xffffffffu, 2 * (int64_t)W * capacity, s);  // padding slots: key -1
    if (st == MH_OK) st = mh_fill_words(tab_key, 0xffffffffu, 2 * L.hslots, s);
    if (st == MH_OK) st = mh_fill_words(tab_first, 0x7fffffffu, L.hslots, s);
    if (st != MH_OK) return st;
    const dim3 grid((unsigned)L.ntiles);
    const uint32_t hmask = (uint32_t)(L.hslots - 1);
    const int* c_slot = slot_of;
    const int* c_first = tab_first;
    const int* c_hist = hist;
    int64_t* no_src = nullptr;
    if (ids_dtype == MH_I32) {
        MH_LAUNCH(route_dedup_insert_kernel<int32_t>, grid, dim3(256), 0, s, a, n, B, tab_key, tab_first, hmask, slot_of);
        MH_LAUNCH((route_count_kernel<int32_t, true>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, hist, c_slot, c_first);
    } else {
        MH_LAUNCH(route_dedup_insert_kernel<int64_t>, grid, dim3(256), 0, s, a, n, B, tab_key, tab_first, hmask, slot_of);
        MH_LAUNCH((route_count_kernel<int64_t, true>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, hist, c_slot, c_first);
    }
    MH_LAUNCH(route_scan_kernel, dim3(1), dim3(1024), 0, s, hist, scan, L.ntiles * W);
    MH_LAUNCH(route_counts_kernel, dim3(1), dim3(64), 0, s, scan, L.ntiles, W, n, counts, c_hist);
    if (ids_dtype == MH_I32)
        MH_LAUNCH((route_scatter_kernel<int32_t, true>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, 1, scan, send_keys,
                  pos_of, no_src, capacity, overflow, c_slot, c_first, tab_pos);
    else
        MH_LAUNCH((route_scatter_kernel<int64_t, true>), grid, dim3(256), 0, s, a, n, B, W, L.ntiles, 1, scan, send_keys,
                  pos_of, no_src, capacity, overflow, c_slot, c_first, tab_pos);
    const int* c_pos = tab_pos;
    MH_LAUNCH(route_dedup_follow_kernel, dim3((unsigned)mh_ceil_div(n, 256)), dim3(256), 0, s, c_slot, c_pos, n, pos_of);
    MH_CHECK_LAUNCH("mh_route_build_dedup");
    return MH_OK;
}

int32_t mh_route_local_rows(const int64_t* recv_keys, int64_t n, const int64_t* base, const int64_t* shard_rows,
                            int32_t F, int64_t* rows, mh_stream_t stream) {
    if (n <= 0) return MH_OK;
    MH_REQUIRE(recv_keys && base && rows, "mh_route_local_rows: null argument");
    MH_REQUIRE(F >= 1 && F <= MH_MAX_FEATURES, "mh_route_local_rows: F=%d outside [1,%d]", F, MH_MAX_FEATURES);
    MH_LAUNCH(route_local_rows_kernel, dim3((unsigned)mh_ceil_div(n, 256)), dim3(256), 0, mh_stream(stream),
                       recv_keys, n, base, shard_rows, F, rows);
    MH_CHECK_LAUNCH("mh_route_local_rows");
    return MH_OK;
}

}  // extern "C"
