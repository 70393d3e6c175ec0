// Backward of the one-hot embedding lookup fused with the sparse optimizer step (gfx950).
// Reference: the IndexedSlices branch of BaseModel.train_step (merlin/models/tf/models/base.py:
// 1121-1174): Keras sums duplicate indices first (_deduplicate_indexed_slices) and then applies
// the optimizer row-wise, so the update must see the SUM of a row's gradients exactly once.
//
// Pipeline (all on one stream, no host sync, no scratch memory: the whole launch replays from a hipGraph):
//   1. SEGMENTED stable LSD radix sort of the (id, packed (feature, sample)) pairs, hand-written (radix_*_kernel
//      below).  A segment = the entries of the features that share one table (the host orders features so that a
//      segment is contiguous); its sort key is the id itself (invalid ids -> rows, which sorts last), so the
//      number of passes follows the LARGEST table (20 bits = 2 passes of 10 bits for 1M-row tables), not the
//      total row count.  Pass 1 reads the ids directly (no key-build pass); the last pass emits COMPACT keys
//      first_row[table] + id (the row number in the concatenation of the distinct tables; sentinel = all ones).
//      Per pass: per-tile digit histograms -> per-segment exclusive scan -> stable scatter (wave-private digit
//      counters in LDS, ranks from a ballot match).  Pass 0 of the histogram kernel also clears what the later
//      stages accumulate into (carried rows, piece counter): no separate fill launch;
//   2. PIECES (a run cut at every 16th sorted index) enumerated into a compact list, each with the chunk its run
//      starts in (found inside the workgroup's window, or by one binary search per workgroup for a run that began
//      before it);
//   3. segmented reduce over the pieces: one D/4-lane group per piece sums its gradient rows in registers (record
//      and sample indices prefetched two / one iteration ahead: only the gradient-row loads are a dependent chain); a
//      piece that is a whole run is applied to the table row directly (exclusive owner, no atomics); the pieces of
//      a run crossing a chunk boundary add to carry[home chunk] -- long runs of hot ids are pre-summed 16:1 in
//      registers and 16:1 again through LDS;
//   4. each chunk that is home to a crossing run applies the carried sum.
//   MERLIN_HIP_DETERMINISTIC=1: step 3 skips the crossing pieces and step 4 walks every crossing run in sorted
//   (sample) order instead -- no float atomics, bit-reproducible, slow for very hot rows (parity runs).
// HBM traffic: grad rows read once (random), table (+state) rows read+written once per UNIQUE id.
#include <cstdlib>
#include <cstring>

#include "mh_common.h"
#include <stdlib.h>
#include <type_traits>

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

// Optional indirection between a lookup position and its gradient row (multi-hot lookups, mh_embedding_bag_bwd_multi): position
// p of feature f reads gradient row map[f * map_stride + p] (its BAG) and divides it by scale[f * scale_stride + row] (the bag's
// combiner divisor) -- exactly the value bag_expand_kernel would have materialised, without the nnz x D round trip through HBM.
struct GradMap {
    const int32_t* map;  // handed to the SORT as its payload (SortArgs::pmap: the sorted payload of an entry is its bag); nullptr: one-hot lookups
    int64_t map_stride;
    // multi != 0: the gradient handed to the pipeline is the PRE-SCALED, FEATURE-MAJOR copy gs[f][bag][D] = grad[bag][f] / divisor(f, bag)
    // (bag_prescale_kernel: one division per gradient element, where round 5 divided per VALUE -- 20 x more divisions, ~0.5 ms of the
    // reduce kernel's 3.4 -- and read the divisor with a scattered load per value): row stride D, a.offset[f] = f * B * D.
    int multi;
    int64_t bags;  // multi: rows of one feature's block of the gradient copy (the walk kernel forms bag * D in 32 bits where that fits)
};

// ---- segmented LSD radix sort ---------------------------------------------------------------------------------------
constexpr int RTILE = 4096;     // entries per workgroup tile (256 threads x 16)
constexpr int RITEMS = 16;
constexpr int RBITS_MAX = 11;   // digit width: 4 wave-private counter sets of 2^11 ints = 32 KB of LDS
constexpr int MAX_SEG = MH_MAX_FEATURES;

struct SortArgs {
    const void* ids[MH_MAX_FEATURES];  // id column of feature f (features ordered so that a segment is contiguous)
    int64_t rows[MAX_SEG];             // rows of the segment's table = its sentinel key
    int64_t first[MAX_SEG];            // compact key of the table's row 0
    int seg_f0[MAX_SEG + 1];           // first feature of segment s
    int tile0[MAX_SEG + 1];            // first tile of segment s
    int npass[MAX_SEG];                // passes this segment needs (its key bits / the digit widths); later passes skip it
    int nseg;
    int64_t B;
    // Optional PAYLOAD of the sort (multi-hot lookups): entry b of feature f carries (f << 26 | pmap[f * pmap_stride + b]) -- its BAG -- instead
    // of (f << 26 | b).  Read coalesced, once, where pass 0 forms its pairs; the kernels behind the sort then find the gradient row of
    // an entry in the sorted payload itself (round 5 carried the position and looked the bag up per entry in the reduce kernel: 34 M
    // scattered 4-byte loads, one more link in its dependent chain).  The sort is stable: the order inside a run is the same either way.
    const int32_t* pmap;
    int64_t pmap_stride;
};

__device__ __forceinline__ int seg_of_tile(const SortArgs& a, int tile) {
    int s = 0;
    while (s + 1 < a.nseg && a.tile0[s + 1] <= tile) ++s;
    return s;
}

// (feature offset inside the segment, sample) of entry e: one 64-bit division per call -- callers hoist it out of
// their per-entry loops with EntryPos
struct EntryPos {
    int64_t fo, b;  // feature offset, sample
    __device__ __forceinline__ void init(int64_t e, int64_t B) {
        fo = e / B;
        b = e - fo * B;
    }
    __device__ __forceinline__ void advance(int64_t step, int64_t B) {  // e += step
        b += step;
        while (b >= B) {
            b -= B;
            ++fo;
        }
    }
};

template <typename IdT>
__device__ __forceinline__ uint32_t id_to_local_key(const SortArgs& a, int s, int64_t fo, int64_t b) {
    const int64_t id = (int64_t) static_cast<const IdT*>(a.ids[a.seg_f0[s] + fo])[b];
    return (id >= 0 && id < a.rows[s]) ? (uint32_t)id : (uint32_t)a.rows[s];
}

// Pass 0 reads the id columns themselves.  The RTILE / 4 consecutive entries of a wavefront almost always lie inside ONE
// feature (always when B is a multiple of RTILE / 4): then the column pointer and the first sample are wave-uniform and an
// entry is one coalesced load from a scalar base.  Otherwise (a wavefront that straddles two features of a shared table, or
// the switch MERLIN_HIP_SORT_FASTLOAD=0) every lane walks its own (feature, sample) position.
template <typename IdT>
struct Pass0 {
    bool one;          // wave-uniform: fast path
    const IdT* col;    // fast path: column of the wavefront's feature, at its first sample
    int64_t b0;        // fast path: first sample
    int feat;          // fast path: feature index (position in SortArgs::ids)
    EntryPos pos;      // slow path
    __device__ __forceinline__ void init(const SortArgs& a, int s, int64_t e0, int64_t n_s, int lane, int fast) {
        int64_t last = e0 + RTILE / 4;
        if (last > n_s) last = n_s;
        const int64_t fo0 = e0 / a.B;
        b0 = e0 - fo0 * a.B;
        one = fast && e0 < n_s && b0 + (last - e0) <= a.B;
        feat = a.seg_f0[s] + (int)fo0;
        if (one)
            col = static_cast<const IdT*>(a.ids[feat]) + b0;
        else
            pos.init(e0 + lane, a.B);
    }
    // entry e0 + it * 64 + lane (the caller checked e < n_s): local key and, for the scatter, the packed (feature, sample)
    __device__ __forceinline__ uint32_t key(const SortArgs& a, int s, int it, int lane) const {
        if (one) {
            const int64_t id = (int64_t)col[it * 64 + lane];
            return (id >= 0 && id < a.rows[s]) ? (uint32_t)id : (uint32_t)a.rows[s];
        }
        return id_to_local_key<IdT>(a, s, pos.fo, pos.b);
    }
    // The two halves of key() for the fast path, so that a tile's loads can ALL be issued before the first of them is used:
    // a load whose value is compared right behind it, inside a per-entry `if (e < n_s)`, is followed by s_waitcnt vmcnt(0) --
    // the 16 loads of a thread were 16 sequential round trips (seen in the ISA).  raw(): branch-free, the entry index is
    // clamped into the wavefront's range (n >= 1 entries); finish(): the bounds test.
    __device__ __forceinline__ IdT raw(int it, int lane, int n) const {
        int i = it * 64 + lane;
        if (i > n - 1) i = n - 1;
        return col[i];
    }
    __device__ __forceinline__ uint32_t finish(const SortArgs& a, int s, IdT r) const {
        const int64_t id = (int64_t)r;
        return (id >= 0 && id < a.rows[s]) ? (uint32_t)id : (uint32_t)a.rows[s];
    }
    __device__ __forceinline__ uint32_t val(const SortArgs& a, int s, int it, int lane) const {
        if (one) return ((uint32_t)feat << 26) | (uint32_t)(b0 + it * 64 + lane);
        const int f = a.seg_f0[s] + (int)pos.fo;
        if (a.pmap) return ((uint32_t)f << 26) | (uint32_t)a.pmap[(int64_t)f * a.pmap_stride + pos.b];
        return ((uint32_t)f << 26) | (uint32_t)pos.b;
    }
    __device__ __forceinline__ void next(const SortArgs& a) {
        if (!one) pos.advance(64, a.B);
    }
};

// cnt[(tile0[s] + t) * 2^RBITS_MAX + d] = number of entries of tile t of segment s whose digit is d (tile-major: a
// tile's counters are one contiguous, coalesced block for the histogram, the scan and the scatter alike)
// cnt1 != NULL (pass 0 of a LOOK-BACK sort, see radix_scatter_kernel): the tile's counts of the SECOND digit (bits rbits ..
// 2 rbits - 1 of the same keys) go to cnt1 as well -- the digit totals of a segment do not depend on the order of its entries,
// so the scan that follows pass 0's histogram can already produce pass 1's digit bases, and pass 1 needs neither a histogram
// nor a scan launch of its own.
template <typename IdT>
__global__ __launch_bounds__(256) void radix_hist_kernel(const SortArgs a, const void* keys0, const void* keys1, int pass,
                                                        int shift, int rbits, int* __restrict__ cnt, int fast,
                                                        uint4* __restrict__ clear, int64_t clear_vec,
                                                        unsigned int* __restrict__ clear_word, int* __restrict__ cnt1) {
    __shared__ int hist[1 << RBITS_MAX];
    __shared__ int hist1[1 << RBITS_MAX];
    const int s = seg_of_tile(a, blockIdx.x);
    // Pass 0 (every workgroup runs it) also zeroes what the later stages accumulate into -- the carried rows and the piece
    // counter: each workgroup clears its slice with fire-and-forget 16-byte stores before its own loads (a separate fill
    // kernel was 7 us of the launch; a kernel, not a memset node -- see mh_fill_words)
    if (clear_vec > 0) {
        const int64_t per = (clear_vec + gridDim.x - 1) / gridDim.x;
        const int64_t beg = (int64_t)blockIdx.x * per;
        const int64_t end = beg + per < clear_vec ? beg + per : clear_vec;
        for (int64_t i = beg + threadIdx.x; i < end; i += 256) clear[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) clear_word[0] = clear_word[1] = 0u;  // piece counters (whole / partial, see piece_list_kernel)
    if (pass >= a.npass[s]) return;  // this segment is already sorted
    // input of pass p = output of pass p - 1 (see radix_scatter_kernel for the buffer parity)
    const uint32_t* keys_in = static_cast<const uint32_t*>(((a.npass[s] - pass) & 1) ? keys0 : keys1);
    const int R = 1 << rbits;
    const bool dual = cnt1 != nullptr && pass == 0 && a.npass[s] >= 2;  // block-uniform
    for (int d = threadIdx.x; d < R; d += 256) {
        hist[d] = 0;
        if (dual) hist1[d] = 0;
    }
    const int t = blockIdx.x - a.tile0[s];
    const int64_t seg_base = (int64_t)a.seg_f0[s] * a.B;
    const int64_t n_s = (int64_t)(a.seg_f0[s + 1] - a.seg_f0[s]) * a.B;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e0 = (int64_t)t * RTILE + wave * (RTILE / 4);
    uint32_t k[RITEMS];
    Pass0<IdT> p0;
    if (pass == 0) p0.init(a, s, e0, n_s, lane, fast);
    const int64_t nw = n_s - e0;  // entries of this wavefront (<= 0: none)
    if (pass == 0 && p0.one) {  // wave-uniform
        IdT r[RITEMS];
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) r[it] = p0.raw(it, lane, (int)(nw < RTILE / 4 ? nw : RTILE / 4));
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) k[it] = (it * 64 + lane < nw) ? p0.finish(a, s, r[it]) : 0xffffffffu;
    } else if (pass > 0 && nw > 0) {  // wave-uniform
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) {
            int64_t i = it * 64 + lane;
            if (i > nw - 1) i = nw - 1;
            k[it] = keys_in[2 * (seg_base + e0 + i)];  // (key, val) pairs
        }
#pragma unroll
        for (int it = 0; it < RITEMS; ++it)
            if (it * 64 + lane >= nw) k[it] = 0xffffffffu;
    } else {
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) {  // a wavefront that straddles two features of a shared table (or has no entries)
            const int64_t e = e0 + it * 64 + lane;
            k[it] = 0xffffffffu;
            if (e < n_s && pass == 0) k[it] = p0.key(a, s, it, lane);
            if (pass == 0) p0.next(a);
        }
    }
    __syncthreads();
    // plain LDS atomics: peeling hot digits with ballots (one aggregated atomic per distinct digit) was measured SLOWER
    // (27 vs 20 us on the Criteo tables) -- the ballot loop costs more than the serialised same-address atomics save
#pragma unroll
    for (int it = 0; it < RITEMS; ++it)
        if (e0 + it * 64 + lane < n_s) {
            atomicAdd(&hist[(k[it] >> shift) & (uint32_t)(R - 1)], 1);
            if (dual) atomicAdd(&hist1[(k[it] >> (shift + rbits)) & (uint32_t)(R - 1)], 1);
        }
    __syncthreads();
    int* out = cnt + (int64_t)blockIdx.x * (1 << RBITS_MAX);
    for (int d = threadIdx.x; d < R; d += 256) out[d] = hist[d];
    if (dual) {
        int* out1 = cnt1 + (int64_t)blockIdx.x * (1 << RBITS_MAX);
        for (int d = threadIdx.x; d < R; d += 256) out1[d] = hist1[d];
    }
}

// one workgroup per segment: cnt[t][d] := first output slot of (digit d, tile t) inside the segment, i.e. the exclusive
// prefix in digit-major order.  A thread owns 2 digits: tile totals (coalesced loop over the tiles), block scan over the
// digits, then the running prefix over the tiles.
// cnt1 != NULL (pass 0 of a look-back sort): the segment's totals of the SECOND digit, exclusive-scanned over the digits, go to
// base1[s][d] (first output slot of digit d in pass 1), and the tile flags of the segment are cleared for pass 1's look-back.
__global__ __launch_bounds__(1024) void radix_scan_kernel(const SortArgs a, int pass, int rbits, int* __restrict__ cnt,
                                                         int keep_regs, const int* __restrict__ cnt1, int* __restrict__ base1,
                                                         int* __restrict__ flags) {
    constexpr int RS = 1 << RBITS_MAX;
    __shared__ int wsum[16];
    const int s = blockIdx.x;
    if (pass >= a.npass[s]) return;
    if (cnt1 != nullptr && pass == 0 && a.npass[s] >= 2) {  // block-uniform
        __shared__ int wsum1[16];
        const int R = 1 << rbits;
        const int nt = a.tile0[s + 1] - a.tile0[s];
        const int* c1 = cnt1 + (int64_t)a.tile0[s] * RS;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int d0 = 2 * threadIdx.x;
        const bool on = d0 < R;
        int t0 = 0, t1 = 0;
        if (on) {
            int t = 0;
            for (; t + 8 <= nt; t += 8) {
                int2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const int2*>(c1 + (int64_t)(t + u) * RS + d0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    t0 += v[u].x;
                    t1 += v[u].y;
                }
            }
            for (; t < nt; ++t) {
                const int2 v = *reinterpret_cast<const int2*>(c1 + (int64_t)t * RS + d0);
                t0 += v.x;
                t1 += v.y;
            }
        }
        const int tot = t0 + t1;
        int x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum1[wave] = x;
        __syncthreads();
        int run0 = x - tot;
        for (int w = 0; w < wave; ++w) run0 += wsum1[w];
        if (on) *reinterpret_cast<int2*>(base1 + (int64_t)s * RS + d0) = make_int2(run0, run0 + t0);
        for (int t = threadIdx.x; t < nt; t += 1024) flags[a.tile0[s] + t] = 0;
        __syncthreads();  // wsum (below) is a different array, but keep the two phases apart for the sake of `on` reuse
    }
    const int R = 1 << rbits;
    const int nt = a.tile0[s + 1] - a.tile0[s];
    int* c = cnt + (int64_t)a.tile0[s] * RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = 2 * threadIdx.x;  // digits d0, d0 + 1: the pair's totals scan as one element
    const bool on = d0 < R;  // R is a power of two >= 2: d1 < R as well
    // both tile loops run 8 loads ahead: one dependent load per iteration made a 128-tile segment (the single segment of
    // the row-sharded update: 524 K requests) cost 35 us per pass; the per-table segments of the one-GPU path have 16 tiles
    int t0 = 0, t1 = 0;
    constexpr int NREG = 16;  // a one-hot feature of a 64 K batch: 16 tiles -- ONE round of loads, the counts stay in registers
    int2 keep[NREG];
    const bool small = keep_regs && nt <= NREG;  // block-uniform
    if (on && small) {
#pragma unroll
        for (int u = 0; u < NREG; ++u) keep[u] = (u < nt) ? *reinterpret_cast<const int2*>(c + (int64_t)u * RS + d0) : make_int2(0, 0);
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            t0 += keep[u].x;
            t1 += keep[u].y;
        }
    } else if (on) {
        int t = 0;
        for (; t + 8 <= nt; t += 8) {
            int2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const int2*>(c + (int64_t)(t + u) * RS + d0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t0 += v[u].x;
                t1 += v[u].y;
            }
        }
        for (; t < nt; ++t) {
            const int2 v = *reinterpret_cast<const int2*>(c + (int64_t)t * RS + d0);
            t0 += v.x;
            t1 += v.y;
        }
    }
    const int tot = t0 + t1;
    int x = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int run0 = x - tot;
    for (int w = 0; w < wave; ++w) run0 += wsum[w];
    int run1 = run0 + t0;
    if (on && small) {
#pragma unroll
        for (int u = 0; u < NREG; ++u)
            if (u < nt) {
                *reinterpret_cast<int2*>(c + (int64_t)u * RS + d0) = make_int2(run0, run1);
                run0 += keep[u].x;
                run1 += keep[u].y;
            }
    } else if (on) {
        int t = 0;
        for (; t + 8 <= nt; t += 8) {
            int2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const int2*>(c + (int64_t)(t + u) * RS + d0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                *reinterpret_cast<int2*>(c + (int64_t)(t + u) * RS + d0) = make_int2(run0, run1);
                run0 += v[u].x;
                run1 += v[u].y;
            }
        }
        for (; t < nt; ++t) {
            int2* q = reinterpret_cast<int2*>(c + (int64_t)t * RS + d0);
            const int2 v = *q;
            *q = make_int2(run0, run1);
            run0 += v.x;
            run1 += v.y;
        }
    }
}

// stable scatter of one tile: rank of an entry = entries with the same digit earlier in the tile (wave-private
// counters, 16 rounds of 64 consecutive entries per wavefront; inside a round a ballot match gives the lanes that
// share the digit) + the scanned global offset of (digit, tile).  A segment's LAST pass emits compact keys.  The
// passes of a segment alternate between the two buffers such that its last pass lands in buffer 1.
// LOOK-BACK mode (lb_hist != NULL, used for pass 1): there was no histogram / scan launch for this pass.  The tile forms its
// digit counts itself (it needs them for the ranks anyway: wcnt), publishes them to lb_hist[tile] and raises lb_flags[tile]; the
// first output slot of (digit, tile) is lb_base[segment][digit] (digit totals of the segment, scanned beside pass 0: they do not
// depend on the order) + the counts of the EARLIER tiles of the segment, which the tile reads once their flags are up.  A tile
// only ever waits for tiles with a LOWER workgroup index: the dispatcher hands workgroups out in index order, so whatever a
// waiting tile needs is already running (or done) -- no deadlock, whatever else shares the chip.
template <typename IdT, typename KeyT>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const SortArgs a, int pass, int shift, int rbits,
                                                           const int* __restrict__ cnt, void* keys0, uint32_t* vals0,
                                                           void* keys1, uint32_t* vals1, int fast, int* __restrict__ lb_hist,
                                                           const int* __restrict__ lb_base, int* __restrict__ lb_flags) {
    __shared__ int wcnt[4][1 << RBITS_MAX];
    const int s = seg_of_tile(a, blockIdx.x);
    const int np = a.npass[s];
    if (pass >= np) return;
    const int R = 1 << rbits;
    for (int d = threadIdx.x; d < 4 * R; d += 256) wcnt[d / R][d % R] = 0;
    const bool last = (pass == np - 1);
    const int dst = ((np - 1 - pass) & 1) ? 0 : 1;  // last pass -> buffer 1, the one before -> 0, ...
    const uint32_t* keys_in = static_cast<const uint32_t*>(dst ? keys0 : keys1);
    void* keys_out = dst ? keys1 : keys0;
    uint32_t* vals_out = dst ? vals1 : vals0;
    const int t = blockIdx.x - a.tile0[s];
    const int64_t seg_base = (int64_t)a.seg_f0[s] * a.B;
    const int64_t n_s = (int64_t)(a.seg_f0[s + 1] - a.seg_f0[s]) * a.B;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t e0 = (int64_t)t * RTILE + wave * (RTILE / 4);
    uint32_t key[RITEMS], val[RITEMS];
    int off[RITEMS];  // digit | rank-inside-(wave, digit) << 11
    Pass0<IdT> p0;
    if (pass == 0) p0.init(a, s, e0, n_s, lane, fast);
    const int64_t nw = n_s - e0;  // entries of this wavefront (<= 0: none)
    if (pass == 0 && p0.one) {  // wave-uniform; loads first, branch-free (see Pass0::raw)
        IdT r[RITEMS];
        const int nwi = (int)(nw < RTILE / 4 ? nw : RTILE / 4);
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) r[it] = p0.raw(it, lane, nwi);
        if (a.pmap) {  // kernel-uniform: the payload is the entry's bag (SortArgs::pmap), loaded like the ids -- all up front, branch-free
            const int32_t* pm = a.pmap + (int64_t)p0.feat * a.pmap_stride + p0.b0;
            int32_t m[RITEMS];
#pragma unroll
            for (int it = 0; it < RITEMS; ++it) {
                int i = it * 64 + lane;
                if (i > nwi - 1) i = nwi - 1;
                m[it] = pm[i];
            }
#pragma unroll
            for (int it = 0; it < RITEMS; ++it) val[it] = ((uint32_t)p0.feat << 26) | (uint32_t)m[it];
        } else {
#pragma unroll
            for (int it = 0; it < RITEMS; ++it) val[it] = p0.val(a, s, it, lane);
        }
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) {
            const bool live = it * 64 + lane < nw;
            key[it] = live ? p0.finish(a, s, r[it]) : 0u;
            if (!live) val[it] = 0u;
        }
    } else if (pass > 0 && nw > 0) {  // wave-uniform
        uint2 kv[RITEMS];
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) {
            int64_t i = it * 64 + lane;
            if (i > nw - 1) i = nw - 1;
            kv[it] = reinterpret_cast<const uint2*>(keys_in)[seg_base + e0 + i];  // pairs from the previous pass
        }
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) {
            const bool live = it * 64 + lane < nw;
            key[it] = live ? kv[it].x : 0u;
            val[it] = live ? kv[it].y : 0u;
        }
    } else {
#pragma unroll
        for (int it = 0; it < RITEMS; ++it) {  // a wavefront that straddles two features of a shared table (or has no entries)
            const int64_t e = e0 + it * 64 + lane;
            key[it] = 0;
            val[it] = 0;
            if (e < n_s && pass == 0) {
                key[it] = p0.key(a, s, it, lane);
                val[it] = p0.val(a, s, it, lane);
            }
            if (pass == 0) p0.next(a);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RITEMS; ++it) {
        const int64_t e = e0 + it * 64 + lane;
        const bool live = e < n_s;
        const uint32_t k = key[it];
        const int d = (int)((k >> shift) & (uint32_t)(R - 1));
        // lanes of this round that hold the same digit (dead lanes match nobody)
        uint64_t peers = __ballot(live);
        for (int bit = 0; bit < rbits; ++bit) {
            const uint64_t m = __ballot((d >> bit) & 1);
            peers &= ((d >> bit) & 1) ? m : ~m;
        }
        int rank = 0;
        if (live) {
            const int leader = __ffsll((unsigned long long)peers) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&wcnt[wave][d], __popcll(peers));  // wave-private: only this wave adds
            base = __shfl(base, leader);
            rank = base + __popcll(peers & lt_mask);
        }
        off[it] = live ? (d | (rank << RBITS_MAX)) : -1;
    }
    __syncthreads();
    // wcnt[w][d] := global offset of (d, tile) + entries of waves < w with digit d
    const int* c = cnt + (int64_t)blockIdx.x * (1 << RBITS_MAX);
    if (lb_hist != nullptr) {  // kernel-uniform: look-back instead of a scanned count
        constexpr int RS = 1 << RBITS_MAX;
        int* mine = lb_hist + (int64_t)blockIdx.x * RS;
        for (int d = threadIdx.x; d < R; d += 256) mine[d] = (wcnt[0][d] + wcnt[1][d]) + (wcnt[2][d] + wcnt[3][d]);
        __threadfence();   // the counts are visible device-wide before the flag
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&lb_flags[blockIdx.x], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int tfirst = a.tile0[s];
        // every thread waits for the earlier tiles itself (one flag per tile, read past the L1).  The spin is bounded; when the
        // bound is hit (workgroups NOT dispatched in index order: the assumption of this opt-in pass is broken) the wavefront
        // TRAPS: the launch fails with a HIP error at the next synchronisation instead of handing a corrupted sort to the fused
        // optimizer (round-4 advisor finding: silently wrong rows updated)
        for (int tp = tfirst + (int)(threadIdx.x & 63); tp < (int)blockIdx.x; tp += 64) {
            int spins = 0;
            while (__hip_atomic_load(&lb_flags[tp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                if (++spins >= (1 << 24)) __builtin_trap();
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        const int* bs = lb_base + (int64_t)s * RS;
        // plain loads (the acquire above invalidated the L1; a line of lb_hist is only ever read after its tile's flag), eight
        // in flight per digit: as relaxed ATOMIC loads they were issued one at a time -- 120 dependent L2 round trips per thread,
        // the whole pass 48 us slower than the three classic launches it replaces
        const int* hp = lb_hist + (int64_t)tfirst * RS;
        const int nprev = (int)blockIdx.x - tfirst;
        for (int d = threadIdx.x; d < R; d += 256) {
            int run = bs[d];
            int tp = 0;
            for (; tp + 8 <= nprev; tp += 8) {
                int v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = hp[(int64_t)(tp + u) * RS + d];
                run += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; tp < nprev; ++tp) run += hp[(int64_t)tp * RS + d];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int n = wcnt[w][d];
                wcnt[w][d] = run;
                run += n;
            }
        }
    } else {
        for (int d = threadIdx.x; d < R; d += 256) {
            int run = c[d];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int n = wcnt[w][d];
                wcnt[w][d] = run;
                run += n;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RITEMS; ++it) {
        if (off[it] < 0) continue;
        const int d = off[it] & ((1 << RBITS_MAX) - 1);
        const int64_t p = seg_base + wcnt[wave][d] + (off[it] >> RBITS_MAX);
        if (last) {
            const KeyT out = (key[it] < (uint32_t)a.rows[s]) ? (KeyT)(a.first[s] + (int64_t)key[it]) : KeyTraits<KeyT>::sentinel;
            static_cast<KeyT*>(keys_out)[p] = out;
            vals_out[p] = val[it];
        } else {
            static_cast<uint2*>(keys_out)[p] = make_uint2(key[it], val[it]);  // the key buffers hold 8 bytes per entry
        }
    }
}

struct OptHyper {
    float lr, eps, beta1, beta2;
    const float* lr_dev;  // optional device scalar overriding lr (graph-replayable bias correction)
};

// A table row update in two halves, so that the row (+ optimizer state) loads can be issued BEFORE the gradient
// rows they do not depend on: load_row() fetches, finish_row() applies the optimizer and stores.
// pointers into the tables carry the GLOBAL address space explicitly: a pointer that went through LDS (FeatRow below) is
// generic to the compiler, and its accesses would be flat_load / flat_store
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f32x4 gf32x4;
struct RowRmw {
    gfloat* w;
    gfloat* s1;
    gfloat* s2;
    f32x4 wv, m, v;
};

// f: any feature of the run (features sharing a table share pointers and key range); key: compact key.
__device__ __forceinline__ void load_row(const BwdArgs& a, int f, int64_t key, int D, int c4, int opt, RowRmw& r) {
    const int64_t off = (key - a.first[f]) * D + c4 * 4;
    r.w = (gfloat*)(a.table[f] + off);
    r.wv = *reinterpret_cast<const gf32x4*>(r.w);
    if (opt != MH_OPT_SGD) {
        r.s1 = (gfloat*)(a.state[f] + off);
        r.m = *reinterpret_cast<const gf32x4*>(r.s1);
    }
    if (opt == MH_OPT_ADAM) {
        r.s2 = (gfloat*)(a.state2[f] + off);
        r.v = *reinterpret_cast<const gf32x4*>(r.s2);
    }
}

// What a row update needs of one feature.  The piece kernel copies the BwdArgs columns into LDS once per workgroup: indexing
// the kernel-argument arrays with a per-lane feature number compiles to GLOBAL loads from the argument segment, and every
// row address then hangs behind such a load -- measured in the ISA of the first version: pointer fetch -> wait -> weight row
// -> pointer fetch -> wait for everything -> state row -> offset fetch -> wait -> gradient rows, three to four memory round
// trips per piece.  From LDS the pointers arrive in ~100 cycles and all rows of a piece are fetched in ONE round trip.
struct FeatRow {
    float* table;
    float* state;
    float* state2;
    int64_t first, offset;
};

__device__ __forceinline__ void load_row(const FeatRow& ft, int64_t key, int D, int c4, int opt, RowRmw& r) {
    const int64_t off = (key - ft.first) * D + c4 * 4;
    r.w = (gfloat*)(ft.table + off);
    r.wv = *reinterpret_cast<const gf32x4*>(r.w);
    if (opt != MH_OPT_SGD) {
        r.s1 = (gfloat*)(ft.state + off);
        r.m = __builtin_nontemporal_load(reinterpret_cast<const gf32x4*>(r.s1));  // optimizer state: touched by this kernel only, once per step
    }
    if (opt == MH_OPT_ADAM) {
        r.s2 = (gfloat*)(ft.state2 + off);
        r.v = *reinterpret_cast<const gf32x4*>(r.s2);
    }
}

__device__ __forceinline__ void finish_row(RowRmw& r, f32x4 g, int opt, const OptHyper& hp) {
    const float lr = hp.lr_dev ? *hp.lr_dev : hp.lr;
    const float eps = hp.eps;
    f32x4 wv = r.wv;
    if (opt == MH_OPT_ADAGRAD) {
        const f32x4 sv = r.m + g * g;
        __builtin_nontemporal_store(sv, reinterpret_cast<gf32x4*>(r.s1));
        wv.x -= lr * g.x / (sqrtf(sv.x) + eps);
        wv.y -= lr * g.y / (sqrtf(sv.y) + eps);
        wv.z -= lr * g.z / (sqrtf(sv.z) + eps);
        wv.w -= lr * g.w / (sqrtf(sv.w) + eps);
    } else if (opt == MH_OPT_ADAM) {
        // LazyAdam._resource_apply_sparse (blocks/optimizer.py:412-437): only the touched rows' moments move
        const f32x4 m = r.m * hp.beta1 + g * (1.f - hp.beta1);
        const f32x4 v = r.v * hp.beta2 + (g * g) * (1.f - hp.beta2);
        *reinterpret_cast<gf32x4*>(r.s1) = m;
        *reinterpret_cast<gf32x4*>(r.s2) = v;
        wv.x -= lr * m.x / (sqrtf(v.x) + eps);
        wv.y -= lr * m.y / (sqrtf(v.y) + eps);
        wv.z -= lr * m.z / (sqrtf(v.z) + eps);
        wv.w -= lr * m.w / (sqrtf(v.w) + eps);
    } else {
        wv -= g * lr;
    }
    *reinterpret_cast<gf32x4*>(r.w) = wv;
}


// A PIECE is a maximal range of sorted entries with one key inside one chunk of 16 (cuts at run starts and at
// every 16th index; invalid entries -- the sentinels at the end of each segment -- belong to no piece).
// piece_list_kernel enumerates the pieces into a compact list -- one 16-byte record {start:32 | len:5 |
// starts_run:1 | ends_run:1 | feature:6, key} each (one load gives the consumer everything the table-row address
// needs) -- plus, for a piece that continues a run begun earlier, home[p] = the chunk holding the run's first entry
// (where the pieces of a crossing run meet: carry[home]).  The run start is the latest run start at or before the
// piece inside the workgroup's window of LIST_TILES x 256 entries; a run that began before the window is located by
// ONE binary search per workgroup (valid keys are sorted inside a segment, and everything before a valid entry of
// a segment is valid).  A workgroup bumps the global counter ONCE (same-address atomics retire at ~12 ns each: one
// bump per 256 entries cost 79 us for 1.7M entries); the list is ordered inside a workgroup, unordered across.
constexpr int LIST_TILES = 8;

// SPLIT (the multi-hot update): the list comes in TWO regions of the same array -- pieces that are a whole run (applied to their table row
// directly) from the front, counter[0] of them; pieces of runs that cross a chunk boundary from the BACK (logical index q at cap - 1 - q),
// counter[1] of them -- so that each kind is walked by a kernel that holds nothing of the other's state in registers.
template <typename KeyT, bool SPLIT = false>
__global__ __launch_bounds__(256) void piece_list_kernel(const SortArgs sa, const KeyT* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, int64_t n,
                                                         ulonglong2* __restrict__ pieces, int* __restrict__ home,
                                                         unsigned int* __restrict__ counter, int64_t cap = 0) {
    __shared__ unsigned int wave_cnt_w[LIST_TILES][4];  // SPLIT: whole pieces of (tile, wave)
    __shared__ unsigned int block_base_p;
    unsigned int rank_w[LIST_TILES];
    constexpr KeyT SENT = KeyTraits<KeyT>::sentinel;
    __shared__ unsigned int wave_cnt[LIST_TILES][4];
    __shared__ int64_t wave_last_start[LIST_TILES][4];  // last run start inside (tile, wave), -1 if none
    __shared__ int64_t start_before[LIST_TILES][4];     // latest run start before (tile, wave): in-window or searched
    __shared__ unsigned int block_base;
    __shared__ int64_t pre_start;  // start of the run that reaches into this window from before it, -1 if none
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t le_mask = lt_mask | (1ull << lane);
    uint64_t rec[LIST_TILES];
    KeyT kk[LIST_TILES];
    unsigned int rank[LIST_TILES];
    int64_t my_start[LIST_TILES];  // run start of this entry if it lies in the same wave, else -1
    const int64_t w0 = (int64_t)blockIdx.x * LIST_TILES * 256;
    if (wave == 3) {
        // The run that reaches into this window from before it (if any): lower bound of its key inside the segment, found
        // by a 64-ary search of this wavefront (a 64 K-entry segment takes 3 dependent probes; the binary search of one
        // thread took 16, and the windows of a tiny table -- runs of thousands -- all need it).
        int64_t pre = -1;
        if (w0 > 0 && w0 < n) {
            const KeyT k0 = keys[w0];
            if (k0 != SENT && keys[w0 - 1] == k0) {
                int sg = 0;
                while (sg + 1 < sa.nseg && (int64_t)sa.seg_f0[sg + 1] * sa.B <= w0) ++sg;
                // invariant: the answer lies in [lo, hi] and keys[hi] == k0 (valid keys are sorted inside a segment, and
                // everything before a valid entry of a segment is valid)
                int64_t lo = (int64_t)sa.seg_f0[sg] * sa.B, hi = w0 - 1;
                while (lo < hi) {
                    const int64_t step = (hi - lo) / 64 + 1;
                    int64_t q = lo + lane * step;
                    if (q > hi) q = hi;
                    const int c = __popcll(__ballot(keys[q] < k0));  // sorted: the probes below k0 are the first c
                    if (c == 0) {
                        hi = lo;
                    } else {
                        const int64_t nhi = lo + c * step;  // c == 64: past hi (64 step > hi - lo)
                        lo = lo + (c - 1) * step + 1;
                        if (nhi < hi) hi = nhi;
                    }
                }
                pre = lo;
            }
        }
        if (lane == 0) pre_start = pre;
    }
    // Loads in two rounds, each branch-free and issued for all LIST_TILES tiles before any of them is used (one load per
    // `if`, used right behind it, compiled to a load + s_waitcnt vmcnt(0) each: 4 x LIST_TILES sequential round trips per
    // thread in the first version).  Round 1: the entry's key and its predecessor's; round 2: the key behind the piece and
    // the entry's value.  Indices are clamped into [0, n - 1]; what a clamped load returns is never used.
    KeyT k1[LIST_TILES], kp[LIST_TILES];
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        const int64_t i = w0 + t * 256 + threadIdx.x;
        const int64_t ic = i < n ? i : n - 1;
        k1[t] = keys[ic];
        kp[t] = keys[ic > 0 ? ic - 1 : 0];
    }
    int64_t pe[LIST_TILES];  // first index behind the piece that starts at this entry (if one does)
    int plen[LIST_TILES];
    bool pcut[LIST_TILES], pstart[LIST_TILES];
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        const int64_t i = w0 + t * 256 + threadIdx.x;
        const KeyT k = (i < n) ? k1[t] : SENT;
        const bool valid = k != SENT;
        const bool run_start = valid && (i == 0 || kp[t] != k);
        const bool cut = valid && (run_start || (i & (CHUNK - 1)) == 0);
        const uint64_t cuts = __ballot(cut);
        const uint64_t stops = cuts | ~__ballot(valid);  // a piece ends before the next cut or the next invalid entry
        const uint64_t starts = __ballot(run_start);
        rec[t] = 0;
        kk[t] = k;
        my_start[t] = -1;
        const uint64_t later = (lane == 63) ? 0ull : (stops >> (lane + 1));
        const int len = later ? (__ffsll((unsigned long long)later)) : (64 - lane);  // 64 | chunk: a wave end is a cut
        plen[t] = len;
        pcut[t] = cut;
        pstart[t] = run_start;
        pe[t] = cut ? i + len : i;
        if (cut) {
            const uint64_t sb = starts & le_mask;
            if (sb) my_start[t] = i - lane + (63 - __clzll((unsigned long long)sb));
        }
        rank[t] = __popcll(cuts & lt_mask);
        if (lane == 0) {
            wave_cnt[t][wave] = __popcll(cuts);
            wave_last_start[t][wave] = starts ? (i + (63 - __clzll((unsigned long long)starts))) : -1;
        }
    }
    KeyT ke[LIST_TILES];
    uint32_t pv[LIST_TILES];
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        const int64_t i = w0 + t * 256 + threadIdx.x;
        ke[t] = keys[pe[t] < n ? pe[t] : n - 1];
        pv[t] = vals[i < n ? i : n - 1];
    }
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        const bool ends = (pe[t] >= n) || (ke[t] != kk[t]);
        if (pcut[t]) {
            const int64_t i = w0 + t * 256 + threadIdx.x;
            rec[t] = (uint64_t)i | ((uint64_t)plen[t] << 32) | ((uint64_t)(pstart[t] ? 1 : 0) << 37) |
                     ((uint64_t)(ends ? 1 : 0) << 38) | ((uint64_t)(pv[t] >> 26) << 39) | (1ull << 63);
        }
        if (SPLIT) {
            const uint64_t wh = __ballot(pcut[t] && pstart[t] && ends);
            rank_w[t] = __popcll(wh & lt_mask);
            if (lane == 0) wave_cnt_w[t][wave] = __popcll(wh);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0, tot_w = 0;
        for (int t = 0; t < LIST_TILES; ++t) tot += wave_cnt[t][0] + wave_cnt[t][1] + wave_cnt[t][2] + wave_cnt[t][3];
        if (SPLIT) {
            for (int t = 0; t < LIST_TILES; ++t) tot_w += wave_cnt_w[t][0] + wave_cnt_w[t][1] + wave_cnt_w[t][2] + wave_cnt_w[t][3];
            block_base = tot_w ? atomicAdd(counter, tot_w) : 0u;
            block_base_p = (tot - tot_w) ? atomicAdd(counter + 1, tot - tot_w) : 0u;
        } else
        block_base = tot ? atomicAdd(counter, tot) : 0u;
        int64_t run = pre_start;
        for (int t = 0; t < LIST_TILES; ++t)
            for (int w = 0; w < 4; ++w) {
                start_before[t][w] = run;
                if (wave_last_start[t][w] >= 0) run = wave_last_start[t][w];
            }
    }
    __syncthreads();
    unsigned int o = block_base, op = SPLIT ? block_base_p : 0u;
#pragma unroll
    for (int t = 0; t < LIST_TILES; ++t) {
        unsigned int before = 0, before_w = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[t][w];
        if (SPLIT)
            for (int w = 0; w < wave; ++w) before_w += wave_cnt_w[t][w];
        if (rec[t]) {
            int64_t p = (int64_t)o + before + rank[t];
            if (SPLIT) {
                const bool whole = ((rec[t] >> 37) & 3) == 3;
                // a partial piece's rank among the partial pieces = its rank among all pieces - its rank among the whole ones
                p = whole ? (int64_t)o + before_w + rank_w[t] : cap - 1 - ((int64_t)op + (before - before_w) + (rank[t] - rank_w[t]));
            }
            pieces[p] = make_ulonglong2(rec[t], (uint64_t)kk[t]);
            const int64_t st = (my_start[t] >= 0) ? my_start[t] : start_before[t][wave];
            home[p] = (int)(st / CHUNK);
        }
        const unsigned int all = wave_cnt[t][0] + wave_cnt[t][1] + wave_cnt[t][2] + wave_cnt[t][3];
        if (SPLIT) {
            const unsigned int allw = wave_cnt_w[t][0] + wave_cnt_w[t][1] + wave_cnt_w[t][2] + wave_cnt_w[t][3];
            o += allw;
            op += all - allw;
        } else {
            o += all;
        }
    }
}

// One D/4-lane group per piece (grid-stride over the list): the piece's gradient rows are summed in sorted
// order, four loads in flight; a piece that is a whole run is applied to its table row directly (exclusive
// owner, no atomics); a piece of a run that crosses a chunk boundary adds its sum to carry[home chunk]
// (home from the scanned chunk flags).  Every piece is independent, so the random 256-byte read-modify-writes
// of different rows overlap freely -- tools/exp/rmw_bench.hip measures the same traffic pattern at ~5 TB/s,
// which the earlier one-group-per-chunk serial walk (2.3 TB/s) could not reach.
// VMODE 1 (D / 4 in {16, 32, 64}: a group is an aligned part of ONE wavefront at least as wide as a chunk): the dependent
// chain record -> sample indices -> gradient rows is cut to its last link.  The record is fetched TWO iterations ahead
// and the piece's <= 16 sample indices ONE iteration ahead, by the group's first 16 lanes in one coalesced load; the
// gradient-row loop takes them from those lanes with wavefront shuffles.  The LDS partials are double-buffered, so an
// iteration has one barrier instead of two.  Sums are formed in the same order: results are bit-identical to VMODE 0.
// RUNS 1 (the multi-hot update, `gm` set: entries are values of bags, a 4-row table of a 1.3 M-value feature has runs of 20 000
// pieces; RUNS 0 compiles the bag indirection out -- the one-hot kernel keeps its 96 registers): a workgroup takes the list in
// tiles of 2^tile_log2 consecutive iterations and carries the partial sum of the run that leaves an iteration into the next one
// (`pend`, double-buffered in LDS) instead of adding it to carry[] at once: one set of same-line atomics per run and TILE instead of
// one per iteration, while all workgroups still walk one window of the list (the gradient block of one or two features stays cached).
template <int VMODE, int RUNS>
__global__ __launch_bounds__(256) void piece_reduce_apply_kernel(const BwdArgs a, const uint32_t* __restrict__ vals,
                                                                int D, int LPR, const float* __restrict__ grad,
                                                                int64_t grad_row_stride, float* __restrict__ carry,
                                                                const int* __restrict__ home,
                                                                const ulonglong2* __restrict__ pieces,
                                                                const unsigned int* __restrict__ counter, int opt,
                                                                const OptHyper hp, int deterministic, const GradMap gm,
                                                                int tile_log2) {
    constexpr int NBUF = VMODE ? 2 : 1;
    __shared__ f32x4 pend_s[2][64];  // RUNS: sum of the run that left the previous iteration through its last group
    __shared__ uint64_t pend_key[2];
    __shared__ int pend_home[2];
    __shared__ f32x4 part_s[NBUF][256];      // partial sum of every group (one f32x4 per thread)
    __shared__ uint64_t part_key[NBUF][64];  // key of a group's partial-run piece, ~0 if it has none
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    const int glane0 = (int)(threadIdx.x & 63) - c4;  // VMODE 1: first lane of the group inside its wavefront
    __shared__ FeatRow feat[MH_MAX_FEATURES];
    if (threadIdx.x < MH_MAX_FEATURES) {
        const int f = threadIdx.x;
        feat[f].table = a.table[f];
        feat[f].state = a.state[f];
        feat[f].state2 = a.state2[f];
        feat[f].first = a.first[f];
        feat[f].offset = a.offset[f];
    }
    if (RUNS && threadIdx.x < 2) pend_key[threadIdx.x] = ~0ull;
    __syncthreads();
    const int64_t np = (int64_t)*counter;
    // first piece of iteration `it` of this workgroup: grid-stride over iterations (RUNS 0) or over tiles of iterations (RUNS 1)
    auto piece0 = [&](int64_t it) -> int64_t {
        if (!RUNS) return ((int64_t)blockIdx.x + it * gridDim.x) * groups;
        const int64_t tile = it >> tile_log2, r = it - (tile << tile_log2);
        return (((tile * gridDim.x + blockIdx.x) << tile_log2) + r) * groups;
    };
    constexpr uint32_t PMASK = (1u << 26) - 1;
    // multi-hot lookups: the position of an entry is replaced by its gradient row (bag) where the entry is fetched from the sorted
    // list -- one iteration ahead of its use in VMODE 1 -- so that the row loads below keep their chain length
    // (multi-hot lookups: the sorted payload already IS the bag -- SortArgs::pmap)
    auto remap = [&](uint32_t v) -> uint32_t { return v; };
    auto row = [&](uint32_t v) -> f32x4 {
        const int f = (int)(v >> 26);
        if (RUNS)  // bag gradients (pre-scaled, GradMap::multi) are re-read by every value of the bag: cached loads
            return *reinterpret_cast<const f32x4*>(grad + (int64_t)(v & PMASK) * grad_row_stride + feat[f].offset + c4 * 4);
        // gradient rows are read exactly once: streaming loads (the table / state rows of hot ids stay cached)
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grad + (int64_t)(v & PMASK) * grad_row_stride +
                                                                         feat[f].offset + c4 * 4));
    };
    // records as scalar pairs (rec, key); a record of length 0 stands for "no piece"
    int64_t it = 0;
    int64_t base = piece0(0);  // block-uniform: the loop carries barriers
    uint64_t next_rec = 0, next_key = ~0ull, next2_rec = 0, next2_key = ~0ull;
    uint32_t myv = 0;  // VMODE 1: lane c4 < len holds the sample index of entry c4 of the CURRENT piece
    if (gi < groups && base + gi < np) {
        const ulonglong2 r = pieces[base + gi];
        next_rec = r.x;
        next_key = r.y;
    }
    if (VMODE) {
        if (gi < groups && piece0(1) + gi < np) {
            const ulonglong2 r = pieces[piece0(1) + gi];
            next2_rec = r.x;
            next2_key = r.y;
        }
        if (c4 < (int)((next_rec >> 32) & 31)) myv = remap(vals[(int64_t)(next_rec & 0xffffffffull) + c4]);
    }
    int buf = 0, pb = 0;
    for (; base < np; base = piece0(++it)) {
        const int64_t p = base + gi;
        const bool active = gi < groups && p < np;
        const uint64_t rec = next_rec, key = next_key;
        uint32_t nv = 0;
        if (VMODE) {
            next_rec = next2_rec;
            next_key = next2_key;
            if (c4 < (int)((next_rec >> 32) & 31)) nv = remap(vals[(int64_t)(next_rec & 0xffffffffull) + c4]);  // indices of the next piece
            next2_rec = 0;
            next2_key = ~0ull;
            const int64_t p2 = piece0(it + 2) + gi;
            if (gi < groups && p2 < np) {
                const ulonglong2 r = pieces[p2];
                next2_rec = r.x;
                next2_key = r.y;
            }
        } else if (gi < groups && piece0(it + 1) + gi < np) {  // next record in flight during this piece
            const ulonglong2 r = pieces[piece0(it + 1) + gi];
            next_rec = r.x;
            next_key = r.y;
        } else {
            next_rec = 0;
            next_key = ~0ull;
        }
        const int64_t s0 = (int64_t)(rec & 0xffffffffull);
        const int len = active ? (int)((rec >> 32) & 31) : 0;
        const bool starts = (rec >> 37) & 1, ends = (rec >> 38) & 1;
        const int fk = (int)((rec >> 39) & 63);
        const bool whole = active && starts && ends;
        const bool partial = active && !whole && !deterministic;  // deterministic: carry_apply walks crossing runs itself
        RowRmw rr;
        if (whole) load_row(feat[fk], (int64_t)key, D, c4, opt, rr);  // independent of the gradient rows: overlaps them
        const uint32_t* vv = vals + s0;
        // entry i of the piece: VMODE 1 reads it from lane i of the group (every lane of a group runs the same trip count, so
        // the source lanes are active), VMODE 0 from memory
        auto idx = [&](int i) -> uint32_t { return VMODE ? (uint32_t)__shfl((int)myv, glane0 + i) : remap(vv[i]); };
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int i = 0;
        for (; i + 4 <= len; i += 4) {
            const uint32_t v0 = idx(i), v1 = idx(i + 1), v2 = idx(i + 2), v3 = idx(i + 3);
            const f32x4 r0 = row(v0), r1 = row(v1), r2 = row(v2), r3 = row(v3);
            acc += r0;
            acc += r1;
            acc += r2;
            acc += r3;
        }
        if (i + 2 <= len) {
            const uint32_t v0 = idx(i), v1 = idx(i + 1);
            const f32x4 r0 = row(v0), r1 = row(v1);
            acc += r0;
            acc += r1;
            i += 2;
        }
        if (i < len) acc += row(idx(i));
        if (whole) finish_row(rr, acc, opt, hp);
        if (VMODE) myv = nv;
        // Pieces of ONE long run sit next to each other in the list, i.e. in neighbouring groups of this block:
        // they are summed through LDS first and the leader issues a single set of atomics.  Without this a hot
        // row (a 3-row table takes 21K gradients of a 64K batch) serialises ~1400 same-line atomics in L2 and
        // the whole kernel waits for it (measured: ~330 us floor independent of everything else).
        part_s[buf][threadIdx.x] = acc;
        if (c4 == 0 && gi < 64) part_key[buf][gi] = partial ? key : ~0ull;
        __syncthreads();
        auto add_carry = [&](int hm, const f32x4 sum) {
            float* cr = carry + (int64_t)hm * D + c4 * 4;
            atomicAdd(cr + 0, sum.x);
            atomicAdd(cr + 1, sum.y);
            atomicAdd(cr + 2, sum.z);
            atomicAdd(cr + 3, sum.w);
        };
        if (partial && (gi == 0 || part_key[buf][gi - 1] != key)) {
            f32x4 sum = acc;
            int g2 = gi + 1;
            for (; g2 < groups && part_key[buf][g2] == key; ++g2) sum += part_s[buf][g2 * LPR + c4];
            if (RUNS) {
                if (gi == 0 && pend_key[pb ^ 1] == key) sum = pend_s[pb ^ 1][c4] + sum;  // the run entered through group 0
                if (g2 == groups) {  // ... and leaves through the last group: its sum waits for the next iteration
                    pend_s[pb][c4] = sum;
                    if (c4 == 0) {
                        pend_key[pb] = key;
                        pend_home[pb] = home[p];
                    }
                } else {
                    add_carry(home[p], sum);
                }
            } else {
                add_carry(home[p], sum);
            }
        }
        if (RUNS) {
            // a waiting sum whose run did not continue through group 0 is added now; no run leaves this iteration: nothing waits
            if (gi == 0 && pend_key[pb ^ 1] != ~0ull && !(partial && pend_key[pb ^ 1] == key)) add_carry(pend_home[pb ^ 1], pend_s[pb ^ 1][c4]);
            if (gi == groups - 1 && c4 == 0 && !partial) pend_key[pb] = ~0ull;
            pb ^= 1;  // written after this iteration's barrier, read after the next one's, rewritten after the one after that
        }
        // two buffers: the next iteration writes the other one, and the one after that is separated from the reads above by
        // the next iteration's barrier
        if (VMODE) buf ^= 1; else __syncthreads();
    }
    if (RUNS) {
        __syncthreads();
        if (gi == 0 && pend_key[pb ^ 1] != ~0ull) {
            float* cr = carry + (int64_t)pend_home[pb ^ 1] * D + c4 * 4;
            const f32x4 sum = pend_s[pb ^ 1][c4];
            atomicAdd(cr + 0, sum.x);
            atomicAdd(cr + 1, sum.y);
            atomicAdd(cr + 2, sum.z);
            atomicAdd(cr + 3, sum.w);
        }
    }
}

// The MULTI-HOT update (GradMap::multi; D / 4 in {16, 32, 64}).  Round 5 ran piece_reduce_apply_kernel<1, 1> here: 3.45 ms for 34 M values
// in 6.5 M pieces, 0.15 of the HBM peak for the whole update.  What the round-6 ablations showed (tools/gpu_bag_ablate.sh, profiles/r6_notes.md):
// removing the workgroup barrier of that kernel and putting all 16 rows of a piece in flight changed nothing -- with EVERY memory access and
// every division switched off the loop still took half its time: it was bound by the instructions issued per value (a shuffle for the payload
// and one for the divisor, 64-bit address arithmetic, four IEEE divisions, 32 divergent branch regions per piece).  Hence:
//   * the gradient arrives PRE-SCALED and feature-major (bag_prescale_kernel): no divisor, no division, a 32-bit row offset;
//   * the payload IS the bag (SortArgs::pmap): no per-entry map lookup;
//   * the trip count of the row loop is the WAVE-UNIFORM maximum of the piece lengths of the wavefront's lane groups (one scalar branch per
//     NR rows; a group whose piece is shorter predicates its loads): every shuffle source is active, no divergent branch regions;
//   * the list comes SPLIT (piece_list_kernel<SPLIT>) and each half has its own kernel: the whole-run pieces (the 1-2-value runs of the large
//     tables: 4.7 of the 6.5 M pieces, each a read-modify-write of a table and a state row) need many wavefronts in flight and little else;
//     the pieces of crossing runs (the long runs of the small tables) need 16 gradient rows in flight and the run's partial sum in registers.
//     One kernel for both held the registers of both (166: three wavefronts per SIMD) everywhere.
// Whole pieces are summed in sorted order, one chain, over the same quotients as before: bit-identical to the round-5 kernel in
// deterministic mode (where crossing runs are left to carry_apply_kernel's ordered walk).
template <int NR>
__global__ __launch_bounds__(256, 5) void piece_whole_apply_kernel(const BwdArgs a, const uint32_t* __restrict__ vals, int D, int LPR,
                                                               const float* __restrict__ grad, const ulonglong2* __restrict__ pieces,
                                                               const unsigned int* __restrict__ counter, int opt, const OptHyper hp) {
    __shared__ FeatRow feat[MH_MAX_FEATURES];
    if (threadIdx.x < MH_MAX_FEATURES) {
        const int f = threadIdx.x;
        feat[f].table = a.table[f];
        feat[f].state = a.state[f];
        feat[f].state2 = a.state2[f];
        feat[f].first = a.first[f];
        feat[f].offset = a.offset[f];
    }
    __syncthreads();
    const int groups = 256 / LPR;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    const int glane0 = (int)(threadIdx.x & 63) - c4;
    const int64_t np = (int64_t)counter[0];
    constexpr uint32_t PMASK = (1u << 26) - 1;
    const ulonglong2 none = make_ulonglong2(0ull, ~0ull);
    // iteration `it` of this workgroup takes `groups` neighbouring pieces (coalesced records); grid-stride over iterations
    auto piece_of = [&](int64_t it) -> int64_t { return (it * gridDim.x + blockIdx.x) * groups + gi; };
    auto rec_at = [&](int64_t q) -> ulonglong2 { return q < np ? pieces[q] : none; };
    auto vals_of = [&](const ulonglong2& r) -> uint32_t {
        return (c4 < (int)((r.x >> 32) & 31)) ? vals[(int64_t)(r.x & 0xffffffffull) + c4] : 0u;
    };
    ulonglong2 r0 = rec_at(piece_of(0)), r1 = rec_at(piece_of(1)), r2 = rec_at(piece_of(2));
    uint32_t v0 = vals_of(r0), v1 = vals_of(r1);
    const uint32_t Du = (uint32_t)D;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int64_t it = 0; (it * gridDim.x + blockIdx.x) * groups < np; ++it) {  // workgroup-uniform
        const uint64_t rec = r0.x, key = r0.y;
        const uint32_t myv = v0;
        r0 = r1; r1 = r2; r2 = rec_at(piece_of(it + 3));
        v0 = v1; v1 = vals_of(r1);
        const int len = (int)((rec >> 32) & 31);
        const int l0 = __builtin_amdgcn_readlane(len, 0), l1 = __builtin_amdgcn_readlane(len, 16), l2 = __builtin_amdgcn_readlane(len, 32),
                  l3 = __builtin_amdgcn_readlane(len, 48);
        const int m01 = l0 > l1 ? l0 : l1, m23 = l2 > l3 ? l2 : l3;
        const int maxlen = m01 > m23 ? m01 : m23;  // wave-uniform (scalar)
        if (maxlen == 0) continue;
        const int fk = (int)((rec >> 39) & 63);
        RowRmw rr;
        if (len > 0) load_row(feat[fk], (int64_t)key, D, c4, opt, rr);  // independent of the gradient rows: overlaps them
        const float* gbase = grad + feat[fk].offset + c4 * 4;
        f32x4 acc = zero;
        for (int i = 0; i < maxlen; i += NR) {
            f32x4 q[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const uint32_t v = (uint32_t)__shfl((int)myv, glane0 + ((i + u) & 15));
                q[u] = zero;
                if (i + u < len) q[u] = *reinterpret_cast<const f32x4*>(gbase + (size_t)((v & PMASK) * Du));
            }
#pragma unroll
            for (int u = 0; u < NR; ++u)
                if (i + u < len) acc += q[u];  // sorted order, one chain; rows past the piece's end are skipped, not added (x + 0 is not x for -0)
        }
        if (len > 0) finish_row(rr, acc, opt, hp);
    }
}

// the pieces of runs that cross a chunk boundary: a lane group WALKS ITS OWN contiguous stretch of T pieces (no LDS, no barrier after the feature
// table is staged); the partial sums of a run are carried IN REGISTERS along the stretch and leave as ONE set of float atomics when the run ends,
// another run begins, or the stretch ends (the list is ordered inside a piece_list window only: continuation is decided by the KEY); record
// (+ home chunk) three pieces ahead, the piece's <= 16 payloads two ahead.  Logical piece q lies at pieces[cap - 1 - q] (piece_list_kernel<SPLIT>).
template <int NR>
__global__ __launch_bounds__(256) void piece_partial_walk_kernel(const BwdArgs a, const uint32_t* __restrict__ vals, int D, int LPR,
                                                                const float* __restrict__ grad, float* __restrict__ carry,
                                                                const int* __restrict__ home, const ulonglong2* __restrict__ pieces,
                                                                const unsigned int* __restrict__ counter, int64_t cap, int T) {
    __shared__ int64_t foff[MH_MAX_FEATURES];
    if (threadIdx.x < MH_MAX_FEATURES) foff[threadIdx.x] = a.offset[threadIdx.x];
    __syncthreads();
    const int groups = 256 / LPR;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    const int glane0 = (int)(threadIdx.x & 63) - c4;
    const int64_t np = (int64_t)counter[1];
    // stretches of the wavefront's groups: a group past the list's end keeps running with empty pieces (wave-uniform control flow below)
    const int64_t p_begin = ((int64_t)blockIdx.x * groups + gi) * T;
    const int64_t wave_begin = ((int64_t)blockIdx.x * groups + (gi - (glane0 / LPR))) * T;  // first group of this wavefront
    if (wave_begin >= np) return;  // the whole wavefront has nothing
    const int64_t p_end = (p_begin + T < np) ? p_begin + T : (p_begin < np ? np : p_begin);
    constexpr uint32_t PMASK = (1u << 26) - 1;
    auto add_carry = [&](int hm, const f32x4 sum) {
        float* cr = carry + (int64_t)hm * D + c4 * 4;
        atomicAdd(cr + 0, sum.x);
        atomicAdd(cr + 1, sum.y);
        atomicAdd(cr + 2, sum.z);
        atomicAdd(cr + 3, sum.w);
    };
    const ulonglong2 none = make_ulonglong2(0ull, ~0ull);
    auto rec_at = [&](int64_t q) -> ulonglong2 { return q < p_end ? pieces[cap - 1 - q] : none; };
    auto home_at = [&](int64_t q) -> int { return q < p_end ? home[cap - 1 - q] : 0; };
    auto vals_of = [&](const ulonglong2& r) -> uint32_t {
        return (c4 < (int)((r.x >> 32) & 31)) ? vals[(int64_t)(r.x & 0xffffffffull) + c4] : 0u;
    };
    ulonglong2 r0 = rec_at(p_begin), r1 = rec_at(p_begin + 1), r2 = rec_at(p_begin + 2);
    int h0 = home_at(p_begin), h1 = home_at(p_begin + 1), h2 = home_at(p_begin + 2);
    uint32_t v0 = vals_of(r0), v1 = vals_of(r1);
    f32x4 run_sum = {0.f, 0.f, 0.f, 0.f};
    uint64_t run_key = ~0ull;  // ~0: no run is open
    int run_home = 0;
    const uint32_t Du = (uint32_t)D;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < T; ++j) {  // wave-uniform trip count; a group past its stretch sees records of length 0
        const int64_t p = p_begin + j;
        const uint64_t rec = r0.x, key = r0.y;
        const int hm = h0;
        const uint32_t myv = v0;
        r0 = r1; r1 = r2; r2 = rec_at(p + 3);
        h0 = h1; h1 = h2; h2 = home_at(p + 3);
        v0 = v1; v1 = vals_of(r1);
        const int len = (int)((rec >> 32) & 31);
        const int l0 = __builtin_amdgcn_readlane(len, 0), l1 = __builtin_amdgcn_readlane(len, 16), l2 = __builtin_amdgcn_readlane(len, 32),
                  l3 = __builtin_amdgcn_readlane(len, 48);
        const int m01 = l0 > l1 ? l0 : l1, m23 = l2 > l3 ? l2 : l3;
        const int maxlen = m01 > m23 ? m01 : m23;
        if (maxlen == 0) break;  // a real piece has a length: every group of the wavefront is past the end of its stretch
        const bool ends = (rec >> 38) & 1;
        const int fk = (int)((rec >> 39) & 63);
        const float* gbase = grad + foff[fk] + c4 * 4;  // every entry of a piece belongs to the piece's feature
        f32x4 acc = zero;
        for (int i = 0; i < maxlen; i += NR) {  // scalar loop: all lanes of the wavefront take every trip, so every shuffle source is active
            f32x4 q[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const uint32_t v = (uint32_t)__shfl((int)myv, glane0 + ((i + u) & 15));
                q[u] = zero;
                if (i + u < len) q[u] = *reinterpret_cast<const f32x4*>(gbase + (size_t)((v & PMASK) * Du));
            }
#pragma unroll
            for (int u = 0; u < NR; ++u)
                if (i + u < len) acc += q[u];
        }
        if (len > 0) {
            if (run_key != ~0ull && run_key != key) {  // the open run continues somewhere else in the list (or has ended there)
                add_carry(run_home, run_sum);
                run_key = ~0ull;
            }
            if (run_key == ~0ull) {
                run_sum = acc;
                run_key = key;
                run_home = hm;
            } else {
                run_sum += acc;
            }
            if (ends) {
                add_carry(run_home, run_sum);
                run_key = ~0ull;
            }
        }
    }
    if (run_key != ~0ull) add_carry(run_home, run_sum);
}

// Chunk c is home to a carried sum iff its last run starts inside it and continues into chunk c + 1.
// deterministic != 0: nothing was carried; the group walks the whole crossing run in sorted (sample) order.
template <typename KeyT>
__global__ __launch_bounds__(256) void carry_apply_kernel(const BwdArgs a, const KeyT* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, int64_t n, int D, int LPR,
                                                         const float* __restrict__ carry, int opt, const OptHyper hp,
                                                         const float* __restrict__ grad, int64_t grad_row_stride,
                                                         int deterministic, const GradMap gm) {
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    if (gi >= groups) return;
    const int64_t chunk = (int64_t)blockIdx.x * groups + gi;
    const int64_t c0 = chunk * CHUNK;
    const int64_t c1 = c0 + CHUNK;
    if (c1 >= n) return;  // the last chunk cannot be crossed
    const KeyT key = keys[c1 - 1], after = keys[c1];  // both before the test: `a || keys[c1] != key` would load the second behind a wait for the first
    if ((key == KeyTraits<KeyT>::sentinel) | (after != key)) return;  // last run ends here (`|`: no short circuit, see above)
    // everything else that depends on the chunk index alone is fetched together (one round trip, not a chain of four)
    const KeyT kfirst = keys[c0];
    const KeyT kbefore = (c0 > 0) ? keys[c0 - 1] : KeyTraits<KeyT>::sentinel;  // the sentinel differs from `key`
    const uint32_t vlast = vals[c1 - 1];
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (!deterministic) g = *reinterpret_cast<const f32x4*>(carry + chunk * D + c4 * 4);
    if (kfirst == key && kbefore == key) return;  // sorted: the whole chunk is this run, which began in an earlier chunk (its home)
    if (deterministic) {
        int64_t st = c0;  // first entry of that run inside the chunk
        if (kfirst != key) {
            st = c1 - 1;
            while (keys[st - 1] == key) --st;  // stops inside the chunk: keys[c0] differs
        }
        for (int64_t i = st; i < n && keys[i] == key; ++i) {
            const uint32_t v = vals[i];
            const int64_t r = (int64_t)(v & ((1u << 26) - 1));  // multi-hot: the bag (SortArgs::pmap)
            f32x4 gv = *reinterpret_cast<const f32x4*>(grad + r * grad_row_stride + a.offset[v >> 26] + c4 * 4);

            g += gv;
        }
    }
    RowRmw rr;
    load_row(a, (int)(vlast >> 26), (int64_t)key, D, c4, opt, rr);
    finish_row(rr, g, opt, hp);
}

struct WsLayout {
    int64_t n, nchunks, max_tiles;
    size_t off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_carry, off_home, off_pieces, off_counter, off_cnt, off_cnt1, off_base1,
        off_flags, total;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// digit width of the sort for n entries: 11 bits, narrower for very long inputs (bounds the counter array)
// (measured, round 6, the 34 M values of the multi-hot bench: two 11-bit passes over the 20-bit keys take 1.0 ms -- scatter 276 + scan 140 + histogram
// 87 us each -- against 0.69 ms for three 8-bit passes: 2048 bins per 4096-entry tile scatter two entries per bin)
int radix_bits_for(int64_t n) {
    static int lim = -1;
    if (lim < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_SORT_WIDE_LOG2");
        lim = e ? atoi(e) : 24;
    }
    return n > (1ll << lim) ? 8 : RBITS_MAX;
}

// Sized for 64-bit final keys (the 32-bit variant uses the front of the same buffers).
bool ws_layout(int64_t B, int F, int D, WsLayout* L) {
    L->n = B * F;
    L->nchunks = mh_ceil_div(L->n, CHUNK);
    L->max_tiles = mh_ceil_div(L->n, RTILE) + F;  // every segment rounds its last tile up
    size_t o = 0;
    L->off_keys_a = o; o = align_up(o + (size_t)L->n * 8, 256);
    L->off_keys_b = o; o = align_up(o + (size_t)L->n * 8, 256);
    L->off_vals_a = o; o = align_up(o + (size_t)L->n * 4, 256);
    L->off_vals_b = o; o = align_up(o + (size_t)L->n * 4, 256);
    L->off_carry = o; o = align_up(o + (size_t)L->nchunks * D * 4, 256);
    L->off_counter = o; o = align_up(o + 8, 256);  // right behind the carry rows: ONE fill kernel clears both (two counters: whole / partial pieces)
    L->off_pieces = o; o = align_up(o + ((size_t)L->n + (size_t)L->nchunks) * 16, 256);  // <= one cut per run + per chunk
    L->off_home = o; o = align_up(o + ((size_t)L->n + (size_t)L->nchunks) * 4, 256);
    L->off_cnt = o; o = align_up(o + (size_t)L->max_tiles * ((size_t)1 << RBITS_MAX) * 4, 256);
    // look-back sort: second-digit tile counts (then pass 1's published tile counts), per-segment digit bases, tile flags
    L->off_cnt1 = o; o = align_up(o + (size_t)L->max_tiles * ((size_t)1 << RBITS_MAX) * 4, 256);
    L->off_base1 = o; o = align_up(o + (size_t)MAX_SEG * ((size_t)1 << RBITS_MAX) * 4, 256);
    L->off_flags = o; o = align_up(o + (size_t)L->max_tiles * 4, 256);
    L->total = o;
    return true;
}

// phases of one launch: PREPARE = everything that depends on the ids only (sort, cleared carry rows, piece list -- left in the
// workspace); APPLY = the gradient-dependent part (segmented reduce + fused optimizer, carried runs)
enum { PH_PREPARE = 1, PH_APPLY = 2, PH_ALL = 3 };

// MERLIN_HIP_DETERMINISTIC=1 at load time, or mh_set_deterministic() at run time (no getenv on the launch path)
int g_deterministic = [] {
    const char* v = getenv("MERLIN_HIP_DETERMINISTIC");
    return (v && v[0] == '1') ? 1 : 0;
}();
bool deterministic_mode() { return g_deterministic != 0; }

template <typename IdT, typename KeyT>
int32_t run_pipeline_t(const BwdArgs& a, const SortArgs& sa, int npass, int rbits, const WsLayout& L, char* ws, int64_t B, int F,
                       int D, const float* grad, int64_t grad_row_stride, int optimizer, const OptHyper& hp,
                       hipStream_t s, int phases, const GradMap gm = GradMap{nullptr, 0, 0, 0}) {
    void* kbuf[2] = {ws + L.off_keys_a, ws + L.off_keys_b};
    uint32_t* vbuf[2] = {reinterpret_cast<uint32_t*>(ws + L.off_vals_a), reinterpret_cast<uint32_t*>(ws + L.off_vals_b)};
    float* carry = reinterpret_cast<float*>(ws + L.off_carry);
    int* home = reinterpret_cast<int*>(ws + L.off_home);
    ulonglong2* pieces = reinterpret_cast<ulonglong2*>(ws + L.off_pieces);
    unsigned int* counter = reinterpret_cast<unsigned int*>(ws + L.off_counter);
    int* cnt = reinterpret_cast<int*>(ws + L.off_cnt);
    const int det = deterministic_mode() ? 1 : 0;

    // ---- 1. segmented stable LSD radix sort: digit width rbits per pass; a segment runs only the passes its own
    //         key bits need (sa.npass) and alternates buffers so that its last pass lands in buffer 1 ---------------------
    const int ntiles = sa.tile0[sa.nseg];
    constexpr int fast = 1;  // pass 0 reads the id columns from a wave-uniform base (decided in round 2: 396 -> 385 us)
    constexpr int lean = 1;  // pass 0 of the histogram kernel clears the carried rows; the scan keeps tile counts in registers
    // LOOK-BACK second pass (OPT-IN: MERLIN_HIP_SORT=lookback; asked for by the round-3 review as "3 launches instead of 6"): pass
    // 0's histogram also counts the second digit, its scan also produces the second pass's digit bases, and the second scatter
    // finds its tile offsets by looking back at the earlier tiles of its segment -- 4 sort launches instead of 6.  Correct (the
    // whole sparse-update suite passes with it) and SLOWER, same box, alternating: C2 Adagrad launch 308 -> 342 us, 26 x 1 M rows
    // 391 -> 445 us; the look-back scatter takes ~67 us against 41 us for the histogram + scan + scatter it replaces.  On this
    // chip every XCD has its own L2: publishing a tile's counts to the other XCDs needs an agent-scope release (write back the
    // XCD's dirty L2 lines -- pass 0 has just written 13.6 MB of keys) and reading them an agent-scope acquire (invalidate), and
    // those cost more than two kernel boundaries, which do the same thing once for everybody.  (Relaxed atomic loads for the
    // counts were worse still: issued one at a time, 120 dependent round trips per thread.)
    const char* sort_env = MH_LAB_ENV("MERLIN_HIP_SORT");  // read per launch (host-side string test): tests switch it in-process
    const bool classic = !(sort_env && !strcmp(sort_env, "lookback"));
    const bool lookback = !classic && npass >= 2;
    int* cnt1 = reinterpret_cast<int*>(ws + L.off_cnt1);
    int* base1 = reinterpret_cast<int*>(ws + L.off_base1);
    int* flags = reinterpret_cast<int*>(ws + L.off_flags);
    for (int p = 0; (phases & PH_PREPARE) && p < npass; ++p) {
        const int shift = p * rbits;
        // pass 0 clears the carried rows (not accumulated into in deterministic mode) and the piece counter on the way
        const bool clr = (p == 0) && lean;
        const bool lb = lookback && p == 1;  // this pass runs as ONE launch
        if (!lb) {
            MH_LAUNCH((radix_hist_kernel<IdT>), dim3((unsigned)ntiles), dim3(256), 0, s, sa, (const void*)kbuf[0],
                               (const void*)kbuf[1], p, shift, rbits, cnt, fast, reinterpret_cast<uint4*>(carry),
                               (clr && !det) ? (int64_t)((L.off_counter - L.off_carry) / 16) : (int64_t)0,
                               clr ? counter : (unsigned int*)nullptr, (lookback && p == 0) ? cnt1 : (int*)nullptr);
            MH_LAUNCH(radix_scan_kernel, dim3((unsigned)sa.nseg), dim3(1024), 0, s, sa, p, rbits, cnt, lean,
                               (lookback && p == 0) ? (const int*)cnt1 : (const int*)nullptr, base1, flags);
        }
        MH_LAUNCH((radix_scatter_kernel<IdT, KeyT>), dim3((unsigned)ntiles), dim3(256), 0, s, sa, p, shift, rbits, cnt,
                           kbuf[0], vbuf[0], kbuf[1], vbuf[1], fast, lb ? cnt1 : (int*)nullptr, (const int*)base1, flags);
    }
    const KeyT* keys = static_cast<const KeyT*>(kbuf[1]);
    const uint32_t* vals = vbuf[1];

    // ---- 2. piece list, 3. segmented reduce + fused optimizer, 4. carried runs -------------------------------------------
    if ((phases & PH_PREPARE) && !lean) {  // a kernel, not a memset node (see mh_fill_words)
        const int32_t st = det ? mh_fill_words(counter, 0u, 2, s)
                               : mh_fill_words(carry, 0u, (int64_t)(L.off_counter + 8 - L.off_carry) / 4, s);
        if (st != MH_OK) return st;
    }
    const int LPR = D / 4;
    const int groups = (256 / LPR) < 64 ? (256 / LPR) : 64;
    // multi-hot update with a wave-friendly row width: the list is split into whole-run pieces and pieces of crossing runs
    const bool split_list = gm.multi && (LPR == 16 || LPR == 32 || LPR == 64) && gm.bags * D < (1ll << 32);
    const int64_t pieces_cap = L.n + L.nchunks;  // a cut per run start + per chunk boundary at most (the list's allocation)
    if (phases & PH_PREPARE) {
        if (split_list)
            MH_LAUNCH((piece_list_kernel<KeyT, true>), dim3((unsigned)mh_ceil_div(L.n, 256 * LIST_TILES)), dim3(256), 0, s, sa,
                               keys, vals, L.n, pieces, home, counter, pieces_cap);
        else
            MH_LAUNCH((piece_list_kernel<KeyT, false>), dim3((unsigned)mh_ceil_div(L.n, 256 * LIST_TILES)), dim3(256), 0, s, sa,
                               keys, vals, L.n, pieces, home, counter, (int64_t)0);
    }
    if (!(phases & PH_APPLY)) {
        MH_CHECK_LAUNCH("mh_embedding_gather_bwd_prepare");
        return MH_OK;
    }
    if (split_list) {
        // multi-hot update: the two halves of the split list, each with its own kernel (kernel comments)
        static int walk_t = -1;
        if (walk_t < 0) {
            const char* e = MH_LAB_ENV("MERLIN_HIP_APPLY_WALK_T");
            walk_t = e ? atoi(e) : 64;
            if (walk_t < 1 || walk_t > 65536) walk_t = 64;
        }
        static int res_whole = 0;
        if (res_whole == 0) {
            int r = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, piece_whole_apply_kernel<4>, 256, 0) != hipSuccess || r < 1) r = 4;
            res_whole = r;
        }
        int64_t nbw = mh_ceil_div(pieces_cap, (int64_t)(256 / LPR));
        const int64_t capw = (int64_t)mh_num_cus() * res_whole;  // persistent: exactly one resident wave of workgroups
        if (nbw > capw) nbw = capw;
        MH_LAUNCH(piece_whole_apply_kernel<4>, dim3((unsigned)nbw), dim3(256), 0, s, a, vals, D, LPR, grad, pieces, counter, optimizer, hp);
        if (!det) {  // deterministic mode: crossing runs are walked in sorted order by carry_apply_kernel
            const int64_t nbp = mh_ceil_div(pieces_cap, (int64_t)(256 / LPR) * walk_t);
            MH_REQUIRE(nbp < (1ll << 31), "mh_embedding_bag_bwd_multi: grid too large");
            MH_LAUNCH(piece_partial_walk_kernel<16>, dim3((unsigned)nbp), dim3(256), 0, s, a, vals, D, LPR, grad, carry, home, pieces, counter,
                      pieces_cap, walk_t);
        }
    } else {
        // the group's lanes fetch and hand out a piece's sample indices (kernel comment) where a group is an aligned 16- / 32-lane
        // part of a wavefront
        const bool vmode = (LPR == 16 || LPR == 32);  // D = 64 / 128 (GPU-tested); 64-lane groups (D = 256) keep mode 0 until a test covers them
        // multi-hot updates (a value list per sample: long runs in every small table) carry run sums across iterations
        const int runs = gm.multi ? 1 : 0;
        static int tile_log2 = -1;
        if (tile_log2 < 0) {
            const char* e = MH_LAB_ENV("MERLIN_HIP_APPLY_TILE_LOG2");
            tile_log2 = e ? atoi(e) : 3;
            if (tile_log2 < 0 || tile_log2 > 16) tile_log2 = 3;
        }
        auto kern = runs ? (vmode ? piece_reduce_apply_kernel<1, 1> : piece_reduce_apply_kernel<0, 1>)
                         : (vmode ? piece_reduce_apply_kernel<1, 0> : piece_reduce_apply_kernel<0, 0>);
        int64_t nb = mh_ceil_div(L.n, groups);  // never more groups than entries
        static int resident_of[4] = {0, 0, 0, 0};  // workgroups per CU the kernel's register budget allows: exactly one resident wave
        int* resident = resident_of + 2 * runs;
        if (resident[vmode] == 0) {
            int r = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, kern, 256, 0) != hipSuccess || r < 1) r = 4;
            resident[vmode] = r;
        }
        // MERLIN_HIP_APPLY_RESIDENT = n caps the persistent grid at n workgroups per compute unit (default: all the register budget
        // allows, 5): a full grid leaves no registers on any SIMD for kernels of OTHER streams until it ends (the step's dW GEMM
        // needs 120 per lane, this kernel 96 x 5 of 512)
        static int cap_env = -1;
        if (cap_env < 0) {
            const char* e = MH_LAB_ENV("MERLIN_HIP_APPLY_RESIDENT");
            cap_env = e ? atoi(e) : 0;
        }
        const int res = (cap_env > 0 && cap_env < resident[vmode]) ? cap_env : resident[vmode];
        const int64_t cap = (int64_t)mh_num_cus() * res;
        if (nb > cap) nb = cap;
        MH_LAUNCH(kern, dim3((unsigned)nb), dim3(256), 0, s, a, vals, D, LPR, grad, grad_row_stride, carry, home,
                           pieces, counter, optimizer, hp, det, gm, tile_log2);
    }
    MH_LAUNCH((carry_apply_kernel<KeyT>), dim3((unsigned)mh_ceil_div(L.nchunks, groups)), dim3(256), 0, s, a, keys,
                       vals, L.n, D, LPR, carry, optimizer, hp, grad, grad_row_stride, det, gm);
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

// Dense list with the "max" combiner: gexp[b*L + l][d] = grad[b][d] / ties(b, d) if row(ids[b, l])[d] is the maximum of
// component d over the list, else 0 (tf.reduce_max gradient: indicators / number selected).  One LPR-lane group per bag;
// the rows are read three times (maximum, tie count, expansion) -- they stay in L1 / L2.
template <typename IdT>
__global__ __launch_bounds__(256) void bag_expand_max_kernel(const float* __restrict__ table, int64_t rows,
                                                            const IdT* __restrict__ ids, int64_t L, int64_t B, int LPR,
                                                            const float* __restrict__ grad, int64_t ldg,
                                                            float* __restrict__ gexp) {
    const int groups = 256 / LPR;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    if (gi >= groups) return;
    const int64_t D = (int64_t)LPR * 4;
    auto row = [&](int64_t j) -> f32x4 {
        const int64_t id = (int64_t)ids[j];
        if (id < 0 || id >= rows) return f32x4{0.f, 0.f, 0.f, 0.f};
        return *reinterpret_cast<const f32x4*>(table + id * D + c4 * 4);
    };
    for (int64_t b = (int64_t)blockIdx.x * groups + gi; b < B; b += (int64_t)gridDim.x * groups) {
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int64_t l = 0; l < L; ++l) {
            const f32x4 x = row(b * L + l);
            m.x = fmaxf(m.x, x.x); m.y = fmaxf(m.y, x.y); m.z = fmaxf(m.z, x.z); m.w = fmaxf(m.w, x.w);
        }
        f32x4 n = {0.f, 0.f, 0.f, 0.f};
        for (int64_t l = 0; l < L; ++l) {
            const f32x4 x = row(b * L + l);
            n.x += (x.x == m.x); n.y += (x.y == m.y); n.z += (x.z == m.z); n.w += (x.w == m.w);
        }
        f32x4 g = *reinterpret_cast<const f32x4*>(grad + b * ldg + c4 * 4);
        g.x /= n.x; g.y /= n.y; g.z /= n.z; g.w /= n.w;
        for (int64_t l = 0; l < L; ++l) {
            const f32x4 x = row(b * L + l);
            const f32x4 o = {x.x == m.x ? g.x : 0.f, x.y == m.y ? g.y : 0.f, x.z == m.z ? g.z : 0.f, x.w == m.w ? g.w : 0.f};
            *reinterpret_cast<f32x4*>(gexp + (b * L + l) * D + c4 * 4) = o;
        }
    }
}


// ---- several multi-hot features in ONE sparse update (mh_embedding_bag_bwd_multi) -----------------------------------------------
struct BwdArgsOffsets {
    int64_t v[MH_MAX_FEATURES];
    int sidx[MH_MAX_FEATURES];  // row of `scale` (original feature index) of the f-th feature of the launch
};
struct BagMultiArgs {
    const void* values[MH_MAX_FEATURES];
    const void* offsets[MH_MAX_FEATURES];  // CSR offsets [B + 1] of the feature, or nullptr: dense list of length L
    int64_t nnz[MH_MAX_FEATURES];
};

// scale[f][bag] = the combiner's divisor of the bag (1 | kept | sqrt(kept)) for every feature in one launch.  A LANE per bag (a
// wavefront per bag, as bag_scale_kernel does it, spends 64 lanes on ~20 ids: 0.38 ms for 26 x 65 536 bags); the lanes of a wavefront
// walk 64 neighbouring bags side by side, bags longer than 64 ids are then counted by the whole wavefront one after the other.
template <typename IdT>
__global__ __launch_bounds__(256) void bag_scale_multi_kernel(const BagMultiArgs a, int64_t L, int64_t B, int combiner,
                                                              float* __restrict__ scale) {
    const int f = blockIdx.y;
    const int64_t bag = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const IdT* values = static_cast<const IdT*>(a.values[f]);
    const IdT* offsets = static_cast<const IdT*>(a.offsets[f]);
    int64_t kept = L;  // dense list: every position counts
    if (offsets && combiner != MH_COMBINER_SUM) {  // safe_embedding_lookup_sparse: negative ids are pruned and not counted
        int64_t beg = 0, end = 0;
        if (bag < B) {
            beg = (int64_t)offsets[bag];
            end = (int64_t)offsets[bag + 1];
        }
        const bool long_bag = end - beg > 64;
        int cnt = 0;
        if (!long_bag)
            for (int64_t p = beg; p < end; ++p) cnt += (values[p] >= 0) ? 1 : 0;
        unsigned long long todo = __ballot(long_bag);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t b2 = __shfl(beg, src), e2 = __shfl(end, src);
            int c2 = 0;
            for (int64_t p = b2 + lane; p < e2; p += 64) c2 += (values[p] >= 0) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c2 += __shfl_xor(c2, o);
            if (lane == src) cnt = c2;
        }
        kept = cnt;
    }
    if (bag >= B) return;
    float dv = 1.f;
    if (combiner == MH_COMBINER_MEAN && kept > 0) dv = (float)kept;
    if (combiner == MH_COMBINER_SQRTN && kept > 0) dv = sqrtf((float)kept);
    scale[(int64_t)f * B + bag] = dv;
}

// map[f][j] = bag of value j, idpad[f][j] = its id for j < nnz[f]; the padding up to Bp entries carries id -1 (no row: the sort
// sends it to the sentinel key, nothing is read for it).  A workgroup owns 256 neighbouring BAGS: their offsets go to LDS, the
// values between the first and the last of them are walked with coalesced accesses and every value finds its bag in the LDS window
// (8 steps).  (A workgroup per 256 VALUES needs two 17-step searches of the global offsets before it can start: 0.40 ms for 34 M
// values, all of it latency.)  Dense lists: bag = j / L.  The workgroups past the last bag write the padding.
template <typename IdT>
__global__ __launch_bounds__(256) void bag_index_pad_kernel(const BagMultiArgs a, int64_t L, int64_t B, int64_t Bp,
                                                            int32_t* __restrict__ map, IdT* __restrict__ idpad) {
    __shared__ int64_t win[257];
    const int f = blockIdx.y;
    const int64_t n = a.nnz[f];
    const IdT* offsets = static_cast<const IdT*>(a.offsets[f]);
    const IdT* values = static_cast<const IdT*>(a.values[f]);
    int32_t* mp = map + (int64_t)f * Bp;
    IdT* ip = idpad + (int64_t)f * Bp;
    const int64_t nbag_blocks = (B + 255) / 256;
    if ((int64_t)blockIdx.x >= nbag_blocks) {  // padding: entries n .. Bp - 1, split over the remaining workgroups
        const int64_t nb = (int64_t)gridDim.x - nbag_blocks;
        int64_t covered = offsets ? (int64_t)offsets[B] : B * L;  // values past the last bag (a caller's nnz > offsets[B]) are padding too
        if (covered > n) covered = n;
        for (int64_t j = covered + ((int64_t)blockIdx.x - nbag_blocks) * 256 + threadIdx.x; j < Bp; j += nb * 256) {
            mp[j] = 0;
            ip[j] = (IdT)-1;
        }
        return;
    }
    const int64_t b0 = (int64_t)blockIdx.x * 256;
    const int nb = (int)((B - b0 < 256) ? B - b0 : 256);
    if (!offsets) {
        for (int64_t j = b0 * L + threadIdx.x; j < (b0 + nb) * L; j += 256) {
            mp[j] = (int32_t)(j / L);
            ip[j] = values[j];
        }
        return;
    }
    for (int t = threadIdx.x; t <= nb; t += 256) win[t] = (int64_t)offsets[b0 + t];
    __syncthreads();
    const int64_t j0 = win[0], j1 = win[nb] < n ? win[nb] : n;
    if (blockIdx.x == 0) {  // offsets of a sliced CSR view need not start at 0: the values in front of the first bag belong to no bag
        const int64_t lead = j0 < n ? j0 : n;
        for (int64_t j = threadIdx.x; j < lead; j += 256) {
            mp[j] = 0;
            ip[j] = (IdT)-1;
        }
    }
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
        int lo = 0, hi = nb;  // win[lo] <= j < win[hi]; empty bags share an offset with their successor: the LAST bag with offset <= j
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (win[mid] <= j) lo = mid; else hi = mid;
        }
        mp[j] = (int32_t)(b0 + lo);
        ip[j] = values[j];
    }
}

// ---- multi-hot update of SMALL tables (D = 64, rows <= 128) as a count-matrix GEMM on the fp32 matrix pipe ------------------------------
// A table of a few rows under 1.3 M values (Criteo: eight tables of 4 .. 96 rows) is one run of tens of thousands of values per row.  Through
// the sort pipeline every one of those values is sorted, listed, and re-reads its bag's 256-byte gradient row from the caches: 31 % of the
// values of the multi-hot bench, none of which needs a sort to find its duplicates.  Bag-major instead:
//     acc[rows, D] += C^T[rows, 64 bags] . G'[64 bags, D],   C[b][r] = how often id r occurs in bag b,   G'[b] = grad[b] / divisor(b)
// per block of 64 neighbouring bags -- a GEMM whose contraction runs over the BAGS.  A wavefront builds C for its block in LDS (lane = bag:
// a lane walks ITS bag's ids and bumps its own column of 16-bit counts: plain read-modify-write, nobody else touches the column; row stride
// rows + 2 shorts = an odd number of words, so the lanes' writes of one id fall into 64 different banks), streams the block's gradient rows
// ONCE straight from global memory as the B operand of v_mfma_f32_32x32x2f32 (lane (n, k): one float of bag k -- 128 contiguous bytes per
// bag), reads the counts back as the A operand, and keeps acc (<= 4 x 2 tiles of 32 x 32: 128 registers) for all its blocks.  Counts are
// small integers: count x g is one exact-to-rounding fp32 product, the same sum in another association as adding g count times.
// What was tried before (all measured on the eight small tables, 10.9 M values; profiles/r6_notes.md): LDS accumulators with ds_add_f32,
// shared or in 16 bank-skewed replicas 3.4 ms (the LDS float atomic itself); wavefront-private LDS accumulators, plain read / add / write with
// register forwarding 1.2 ms; register accumulators fed by ballot counts 0.72 ms.  Not used in deterministic mode (the partial blocks of the
// wavefronts are added in a fixed order, but the sort pipeline's ordered walk is the reproducible statement of this update).
constexpr int SMALL_TABLE_BYTES = 32 * 1024;  // rows * D * 4
constexpr int SMALL_BLOCKS = 256;             // wavefronts (= partial blocks) per small feature
constexpr int SMALL_MAX_ROWS = 128;
constexpr int SMALL_CSTRIDE = SMALL_MAX_ROWS + 2;  // shorts per bag column of the count matrix (odd number of 32-bit words)

struct BagSmallArgs {
    const void* values[MH_MAX_FEATURES];
    const void* offsets[MH_MAX_FEATURES];
    float* table[MH_MAX_FEATURES];
    float* state[MH_MAX_FEATURES];
    float* state2[MH_MAX_FEATURES];
    int64_t goff[MH_MAX_FEATURES];   // float offset of the feature inside a gradient row
    int64_t soff[MH_MAX_FEATURES];   // float offset of the feature's partial blocks in gsum: [nblk][rows * D sums | rows flags]
    int rows[MH_MAX_FEATURES];
    int feat[MH_MAX_FEATURES];       // original feature index (row of `scale`)
};

template <typename IdT>
__global__ __launch_bounds__(256, 2) void bag_small_gemm_kernel(const BagSmallArgs a, int64_t L, int64_t B, const float* __restrict__ grad,
                                                               int64_t grad_row_stride, const float* __restrict__ scale,
                                                               float* __restrict__ gsum) {
    constexpr int D = 64;
    __shared__ uint16_t cnt_all[4][64 * SMALL_CSTRIDE];  // per wavefront: [bag][row] counts
    const int fs = blockIdx.y;
    const int rows = a.rows[fs];
    const int RT = (rows + 31) / 32;  // row tiles (wave-uniform)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int nwaves = (int)gridDim.x * 4, wid = (int)blockIdx.x * 4 + wave;
    uint16_t* cnt = cnt_all[wave];
    const IdT* values = static_cast<const IdT*>(a.values[fs]);
    const IdT* offsets = static_cast<const IdT*>(a.offsets[fs]);
    const float* sc = scale + (int64_t)a.feat[fs] * B;
    const float* gcol = grad + a.goff[fs] + l31;
    f32x16 acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;
    float seen[4] = {0.f, 0.f, 0.f, 0.f};  // sum of the counts this lane fed for row rt * 32 + l31: > 0 <=> the row was looked up
    const int64_t nblocks = (B + 63) / 64;
    for (int64_t blk = wid; blk < nblocks; blk += nwaves) {
        const int64_t bb = blk * 64;
        const int nbag = (int)((B - bb < 64) ? B - bb : 64);
        const int64_t bl = bb + (lane < nbag ? lane : nbag - 1);
        const int64_t beg = offsets ? (int64_t)offsets[bl] : bl * L;
        const int64_t end = offsets ? (int64_t)offsets[bl + 1] : beg + L;
        const int len = lane < nbag ? (int)(end - beg) : 0;
        const float dvl = sc[bl];
        // zero this wavefront's count matrix (64 x SMALL_CSTRIDE shorts = 4160 words)
        for (int i = lane; i < 64 * SMALL_CSTRIDE / 2; i += 64) reinterpret_cast<uint32_t*>(cnt)[i] = 0u;
        int maxlen = len;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // C: lane = bag walks its own ids, four loads in flight (branch-free: a position past the bag re-reads its first id, masked)
        uint16_t* mine = cnt + lane * SMALL_CSTRIDE;
        for (int t0 = 0; t0 < maxlen; t0 += 4) {
            int64_t idv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = t0 + u < len;
                const int64_t v = (int64_t)values[live ? beg + t0 + u : (len > 0 ? beg : 0)];
                idv[u] = (live && v >= 0 && v < rows) ? v : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (idv[u] >= 0) mine[idv[u]] = (uint16_t)(mine[idv[u]] + 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // acc += C^T G': 32 k-steps of two bags; B operand: lane (n = l31, k = h) = G'[bag 2 s + h][ct * 32 + n]
#pragma unroll 1
        for (int s0 = 0; s0 < 32; s0 += 8) {
            float bv[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kb = 2 * (s0 + u) + h;
                const int kc = kb < nbag ? kb : nbag - 1;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) bv[u][ct] = gcol[(bb + kc) * grad_row_stride + ct * 32];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kb = 2 * (s0 + u) + h;
                // (the builtin is an INTEGER lane read: a float handed to it is converted, i.e. truncated -- sqrt(30) became 5)
                const float d0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dvl), 2 * (s0 + u)));
                const float d1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dvl), 2 * (s0 + u) + 1));
                const float dv = h ? d1 : d0;
                const bool live = kb < nbag;
                const float b0 = live ? bv[u][0] / dv : 0.f, b1 = live ? bv[u][1] / dv : 0.f;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    if (rt < RT) {  // wave-uniform
                        const float av = live ? (float)cnt[kb * SMALL_CSTRIDE + rt * 32 + l31] : 0.f;
                        seen[rt] += av;
                        acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[rt][0], 0, 0, 0);
                        acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[rt][1], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the counts are read: the next block may zero them
    }
    // partial block of this wavefront: [rows][D] sums + [rows] flags; C layout: entry i of lane (l31, h) = row (i >> 2) * 8 + h * 4 + (i & 3), column l31
    float* out = gsum + a.soff[fs] + (int64_t)wid * ((int64_t)rows * D + ((rows + 3) & ~3));
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        if (rt < RT) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = rt * 32 + (i >> 2) * 8 + h * 4 + (i & 3);
                    if (r < rows) out[(int64_t)r * D + ct * 32 + l31] = acc[rt][ct][i];
                }
            const float tot = seen[rt] + __shfl_xor(seen[rt], 32);
            const int r = rt * 32 + l31;
            if (h == 0 && r < rows) reinterpret_cast<uint32_t*>(out + (int64_t)rows * D)[r] = tot > 0.f ? 1u : 0u;
        }
    }
}

// optimizer step of the touched rows of the small tables (lazy: an untouched row keeps weights AND state): the row's gradient is the sum of
// the workgroups' partial blocks in index order
__global__ __launch_bounds__(256) void bag_small_apply_kernel(const BagSmallArgs a, int nblk, int D, int LPR, const float* __restrict__ gsum,
                                                             int opt, const OptHyper hp) {
    const int fs = blockIdx.y;
    const int groups = 256 / LPR;
    const int gi = threadIdx.x / LPR;
    const int c4 = threadIdx.x - gi * LPR;
    if (gi >= groups) return;
    const int r = blockIdx.x * groups + gi;
    const int rows = a.rows[fs];
    if (r >= rows) return;
    const float* part = gsum + a.soff[fs];
    const int64_t blk = (int64_t)rows * D + ((rows + 3) & ~3);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    uint32_t any = 0;
    for (int q = 0; q < nblk; q += 4) {  // four blocks in flight
        f32x4 v[4];
        uint32_t t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int qq = (q + u < nblk) ? q + u : nblk - 1;
            v[u] = *reinterpret_cast<const f32x4*>(part + qq * blk + (int64_t)r * D + c4 * 4);
            t[u] = reinterpret_cast<const uint32_t*>(part + qq * blk + (int64_t)rows * D)[r];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q + u < nblk) {
                g += v[u];
                any |= t[u];
            }
    }
    if (!any) return;
    FeatRow ft;
    ft.table = a.table[fs];
    ft.state = a.state[fs];
    ft.state2 = a.state2[fs];
    ft.first = 0;
    ft.offset = 0;
    RowRmw rr;
    load_row(ft, (int64_t)r, D, c4, opt, rr);
    finish_row(rr, g, opt, hp);
}

// gs[f][bag][:] = grad[bag][offset f ..] / scale[f][bag]: the gradient row every value of the bag adds to its table row, formed ONCE per bag
// (the reduce kernel of round 5 divided once per value).  Streaming: 16 bytes per thread in, 16 out.
__global__ __launch_bounds__(256) void bag_prescale_kernel(const float* __restrict__ grad, int64_t grad_row_stride, const BwdArgsOffsets off,
                                                          const float* __restrict__ scale, int64_t B, int LPR, float* __restrict__ gs) {
    const int f = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t bag = t / LPR;
    const int c4 = (int)(t - bag * LPR);
    if (bag >= B) return;
    const f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grad + bag * grad_row_stride + off.v[f] + c4 * 4));
    const float dv = scale[(int64_t)off.sidx[f] * B + bag];
    *reinterpret_cast<f32x4*>(gs + ((int64_t)f * B + bag) * (LPR * 4) + c4 * 4) = g / dv;
}

struct BagMultiWs {
    int64_t Bp, off_scale, off_map, off_ids, off_gs, off_small, small_words, off_inner, total;
};

bool bag_multi_ws(int64_t B, int64_t max_nnz, int F, int D, BagMultiWs* w) {
    w->Bp = (max_nnz + 63) / 64 * 64;
    const int64_t inner = mh_embedding_bwd_workspace_bytes(w->Bp, F, D);
    if (inner < 0) return false;
    int64_t o = 0;
    w->off_scale = o; o = (int64_t)align_up((size_t)(o + (int64_t)F * B * 4), 256);
    w->off_map = o;   o = (int64_t)align_up((size_t)(o + (int64_t)F * w->Bp * 4), 256);
    w->off_ids = o;   o = (int64_t)align_up((size_t)(o + (int64_t)F * w->Bp * 8), 256);
    w->off_gs = o;    o = (int64_t)align_up((size_t)(o + (int64_t)F * B * D * 4), 256);  // pre-scaled feature-major gradient
    // small tables: per feature SMALL_BLOCKS partial blocks of [rows, D] sums + [rows] flags, rows * D * 4 <= SMALL_TABLE_BYTES
    w->small_words = (int64_t)F * SMALL_BLOCKS * (SMALL_TABLE_BYTES / 4 + SMALL_TABLE_BYTES / 16 + 4);
    w->off_small = o; o = (int64_t)align_up((size_t)(o + w->small_words * 4), 256);
    w->off_inner = o; o += inner;
    w->total = o;
    return true;
}

}  // namespace

extern "C" {

int32_t mh_set_deterministic(int32_t on) {
    g_deterministic = on ? 1 : 0;
    return MH_OK;
}

int64_t mh_embedding_bwd_workspace_bytes(int64_t B, int32_t F, int32_t D) {
    if (B <= 0 || F <= 0 || D <= 0) return 0;
    WsLayout L;
    if (!ws_layout(B, F, D, &L)) return -1;
    return (int64_t)L.total;
}

static int32_t gather_bwd_impl(float* const* tables, float* const* state, const int64_t* table_rows,
                               const void* const* ids, int32_t ids_dtype, int64_t B, int32_t F, int32_t D,
                               const float* grad, int64_t grad_row_stride, const int64_t* grad_offset,
                               int32_t optimizer, float lr, float eps, float* const* state2, float beta1, float beta2,
                               const float* lr_device, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream, int phases, const GradMap gm = GradMap{nullptr, 0, 0, 0}) {
    if (phases == PH_PREPARE) {  // ids only: no gradient, no optimizer state yet
        static const int64_t zero_off[MH_MAX_FEATURES] = {0};
        MH_REQUIRE(tables && table_rows && ids, "mh_embedding_gather_bwd_prepare: null argument");
        grad = reinterpret_cast<const float*>(tables[0]);  // any 16-byte aligned address: never dereferenced by this phase
        grad_row_stride = D;
        grad_offset = zero_off;
        optimizer = MH_OPT_SGD;
        state = state2 = nullptr;
    }
    if (B <= 0) return MH_OK;  // before the pointer checks: an empty gradient buffer has no address
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
    ws_layout(B, F, D, &L);
    if (!workspace || workspace_bytes < (int64_t)L.total) {
        mh_set_error("mh_embedding_gather_bwd: workspace too small (%lld < %zu)", (long long)workspace_bytes, L.total);
        return MH_ERR_WORKSPACE;
    }
    // features ordered so that those sharing a table are adjacent: a SEGMENT of the sort (first-appearance order)
    int order[MH_MAX_FEATURES], seg_of[MH_MAX_FEATURES], nseg = 0, pos = 0;
    bool placed[MH_MAX_FEATURES] = {false};
    BwdArgs a;
    SortArgs sa;
    std::memset(&a, 0, sizeof(a));
    std::memset(&sa, 0, sizeof(sa));
    int64_t total_rows = 0;  // rows of the distinct tables back to back = the compact key space
    int64_t max_rows = 1;
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(table_rows[f] >= 1, "mh_embedding_gather_bwd: table %d has no rows", f);
        MH_REQUIRE(table_rows[f] < 0xffffffffll, "mh_embedding_gather_bwd: table %d has 2^32 - 1 rows or more", f);
        MH_REQUIRE(tables[f] && ids[f], "mh_embedding_gather_bwd: null table/ids for feature %d", f);
        MH_REQUIRE(optimizer == MH_OPT_SGD || state[f], "mh_embedding_gather_bwd: null optimizer state for feature %d", f);
        MH_REQUIRE(optimizer != MH_OPT_ADAM || state2[f], "mh_embedding_gather_bwd: null second moment for feature %d", f);
        MH_REQUIRE(grad_offset[f] >= 0 && grad_offset[f] % 4 == 0 && (gm.multi || grad_offset[f] + D <= grad_row_stride),
                   "mh_embedding_gather_bwd: grad offset of feature %d misaligned or out of row", f);
        if (placed[f]) continue;
        sa.seg_f0[nseg] = pos;
        sa.rows[nseg] = table_rows[f];
        sa.first[nseg] = total_rows;
        for (int g = f; g < F; ++g)
            if (!placed[g] && tables[g] == tables[f]) {
                MH_REQUIRE(table_rows[g] == table_rows[f], "mh_embedding_gather_bwd: features %d and %d share a table but not its row count", f, g);
                placed[g] = true;
                order[pos] = g;
                seg_of[pos] = nseg;
                ++pos;
            }
        total_rows += table_rows[f];
        if (table_rows[f] > max_rows) max_rows = table_rows[f];
        ++nseg;
    }
    sa.seg_f0[nseg] = F;
    sa.nseg = nseg;
    sa.B = B;
    sa.pmap = gm.map;  // multi-hot: the sort's payload is the entry's bag (features of such a call use distinct tables: order[] is the identity)
    sa.pmap_stride = gm.map_stride;
    int tiles = 0;
    for (int sg = 0; sg < nseg; ++sg) {
        sa.tile0[sg] = tiles;
        tiles += (int)mh_ceil_div((int64_t)(sa.seg_f0[sg + 1] - sa.seg_f0[sg]) * B, RTILE);
    }
    sa.tile0[nseg] = tiles;
    for (int i = 0; i < F; ++i) {  // position i of the kernel-side arrays = original feature order[i]
        const int f = order[i];
        a.table[i] = tables[f];
        a.state[i] = state ? state[f] : nullptr;
        a.state2[i] = state2 ? state2[f] : nullptr;
        a.ids[i] = ids[f];
        a.rows[i] = table_rows[f];
        a.offset[i] = grad_offset[f];
        a.first[i] = sa.first[seg_of[i]];
        sa.ids[i] = ids[f];
    }
    char* ws = static_cast<char*>(workspace);
    hipStream_t s = mh_stream(stream);
    const OptHyper hp = {lr, eps, beta1, beta2, lr_device};
    int max_bits = 1;  // bits of the largest local key: ids 0 .. rows-1 and the sentinel `rows`
    while (max_bits < 32 && (1ull << max_bits) < (uint64_t)max_rows + 1) ++max_bits;
    const int rmax = radix_bits_for(B * F);
    const int npass = (max_bits + rmax - 1) / rmax;
    const int rbits = (npass == 1) ? max_bits : rmax;  // full-width digits first: a <= 2^rmax-row table is done in ONE pass
    for (int sg = 0; sg < nseg; ++sg) {
        int bits = 1;
        while (bits < 32 && (1ull << bits) < (uint64_t)sa.rows[sg] + 1) ++bits;
        sa.npass[sg] = (bits + rbits - 1) / rbits;
    }
    const bool wide = (uint64_t)total_rows >= 0xffffffffull;
    if (ids_dtype == MH_I32) {
        if (!wide) return run_pipeline_t<int32_t, uint32_t>(a, sa, npass, rbits, L, ws, B, F, D, grad, grad_row_stride, optimizer, hp, s, phases, gm);
        return run_pipeline_t<int32_t, uint64_t>(a, sa, npass, rbits, L, ws, B, F, D, grad, grad_row_stride, optimizer, hp, s, phases, gm);
    }
    if (!wide) return run_pipeline_t<int64_t, uint32_t>(a, sa, npass, rbits, L, ws, B, F, D, grad, grad_row_stride, optimizer, hp, s, phases, gm);
    return run_pipeline_t<int64_t, uint64_t>(a, sa, npass, rbits, L, ws, B, F, D, grad, grad_row_stride, optimizer, hp, s, phases, gm);
}

int32_t mh_embedding_gather_bwd(float* const* tables, float* const* state, const int64_t* table_rows,
                                const void* const* ids, int32_t ids_dtype, int64_t B, int32_t F, int32_t D,
                                const float* grad, int64_t grad_row_stride, const int64_t* grad_offset,
                                int32_t optimizer, float lr, float eps, float* const* state2, float beta1, float beta2,
                                const float* lr_device, void* workspace, int64_t workspace_bytes,
                                mh_stream_t stream) {
    return gather_bwd_impl(tables, state, table_rows, ids, ids_dtype, B, F, D, grad, grad_row_stride, grad_offset, optimizer, lr,
                           eps, state2, beta1, beta2, lr_device, workspace, workspace_bytes, stream, PH_ALL);
}

// The two halves of mh_embedding_gather_bwd as separate calls: _prepare needs the ids only (sort + piece list, left in the
// workspace), so a caller can issue it at the START of the step on a second stream, beside the forward pass; _apply (same
// tables / ids / shapes / workspace, after the gradient exists) does the segmented reduce + fused optimizer.
int32_t mh_embedding_gather_bwd_prepare(float* const* tables, const int64_t* table_rows, const void* const* ids,
                                        int32_t ids_dtype, int64_t B, int32_t F, int32_t D, void* workspace,
                                        int64_t workspace_bytes, mh_stream_t stream) {
    return gather_bwd_impl(tables, nullptr, table_rows, ids, ids_dtype, B, F, D, nullptr, 0, nullptr, MH_OPT_SGD, 0.f, 0.f,
                           nullptr, 0.f, 0.f, nullptr, workspace, workspace_bytes, stream, PH_PREPARE);
}

int32_t mh_embedding_gather_bwd_apply(float* const* tables, float* const* state, const int64_t* table_rows,
                                      const void* const* ids, int32_t ids_dtype, int64_t B, int32_t F, int32_t D,
                                      const float* grad, int64_t grad_row_stride, const int64_t* grad_offset,
                                      int32_t optimizer, float lr, float eps, float* const* state2, float beta1,
                                      float beta2, const float* lr_device, void* workspace, int64_t workspace_bytes,
                                      mh_stream_t stream) {
    return gather_bwd_impl(tables, state, table_rows, ids, ids_dtype, B, F, D, grad, grad_row_stride, grad_offset, optimizer, lr,
                           eps, state2, beta1, beta2, lr_device, workspace, workspace_bytes, stream, PH_APPLY);
}

int64_t mh_embedding_bag_bwd_workspace_bytes(int64_t B, int64_t nnz, int32_t D) {
    if (B <= 0 || nnz <= 0 || D <= 0) return 0;
    const int64_t inner = mh_embedding_bwd_workspace_bytes(nnz, 1, D);
    if (inner < 0) return -1;
    return (int64_t)align_up((size_t)B * sizeof(float), 256) + (int64_t)align_up((size_t)nnz * D * sizeof(float), 256) + inner;
}

// Expansion half of the list backward on its own: gexp[j] = the gradient row of value j (grad[bag(j)] / div(bag(j)), or
// the max-combiner split).  mh_embedding_bag_bwd = this + mh_embedding_gather_bwd over the nnz values; callers that must
// merge several lookups of ONE table into a single dedup + optimizer step (a one-hot feature and a list feature sharing a
// table) expand first, concatenate, and call mh_embedding_gather_bwd once.  scale_ws: B floats.
int32_t mh_embedding_bag_expand(const float* table, int64_t rows, const void* values, int64_t nnz, const void* offsets,
                                int64_t L, int32_t ids_dtype, int64_t B, int32_t D, int32_t combiner, const float* grad,
                                int64_t grad_row_stride, float* gexp, float* scale_ws, mh_stream_t stream) {
    MH_REQUIRE(grad && gexp, "mh_embedding_bag_expand: null argument");
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_bag_expand: bad ids_dtype");
    MH_REQUIRE(combiner >= MH_COMBINER_SUM && combiner <= MH_COMBINER_MAX, "mh_embedding_bag_expand: bad combiner %d", combiner);
    MH_REQUIRE(combiner != MH_COMBINER_MAX || (!offsets && table),
               "mh_embedding_bag_expand: the max combiner is defined for dense lists only (and needs the table)");
    MH_REQUIRE(combiner == MH_COMBINER_MAX || scale_ws, "mh_embedding_bag_expand: scale workspace (B floats) required");
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "mh_embedding_bag_expand: D=%d must be a multiple of 4 in [4,1024]", D);
    MH_REQUIRE(grad_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0,
               "mh_embedding_bag_expand: grad must be 16-byte aligned with grad_row_stride %% 4 == 0");
    MH_REQUIRE(offsets || L >= 1, "mh_embedding_bag_expand: need CSR offsets or a list length L >= 1");
    if (B <= 0 || nnz <= 0) return MH_OK;
    MH_REQUIRE(values, "mh_embedding_bag_expand: null values");
    MH_REQUIRE(offsets || nnz == B * L, "mh_embedding_bag_expand: dense list needs nnz == B*L");
    hipStream_t s = mh_stream(stream);
    const int LPR = D / 4;
    const int groups = 256 / LPR;
    const dim3 gs((unsigned)mh_ceil_div(B, 4));
    int64_t nb = mh_ceil_div(nnz, groups);
    const int64_t cap = (int64_t)mh_num_cus() * 16;
    if (nb > cap) nb = cap;
    float* scale = scale_ws;
    if (combiner == MH_COMBINER_MAX) {
        if (ids_dtype == MH_I32)
            MH_LAUNCH((bag_expand_max_kernel<int32_t>), dim3((unsigned)nb), dim3(256), 0, s, table, rows,
                               (const int32_t*)values, L, B, LPR, grad, grad_row_stride, gexp);
        else
            MH_LAUNCH((bag_expand_max_kernel<int64_t>), dim3((unsigned)nb), dim3(256), 0, s, table, rows,
                               (const int64_t*)values, L, B, LPR, grad, grad_row_stride, gexp);
    } else if (ids_dtype == MH_I32) {
        MH_LAUNCH((bag_scale_kernel<int32_t>), gs, dim3(256), 0, s, (const int32_t*)values,
                           (const int32_t*)offsets, L, B, combiner, scale);
        MH_LAUNCH((bag_expand_kernel<int32_t>), dim3((unsigned)nb), dim3(256), 0, s, (const int32_t*)offsets, L,
                           B, nnz, LPR, scale, grad, grad_row_stride, gexp);
    } else {
        MH_LAUNCH((bag_scale_kernel<int64_t>), gs, dim3(256), 0, s, (const int64_t*)values,
                           (const int64_t*)offsets, L, B, combiner, scale);
        MH_LAUNCH((bag_expand_kernel<int64_t>), dim3((unsigned)nb), dim3(256), 0, s, (const int64_t*)offsets, L,
                           B, nnz, LPR, scale, grad, grad_row_stride, gexp);
    }
    MH_CHECK_LAUNCH("mh_embedding_bag_expand");
    return MH_OK;
}

int32_t mh_embedding_bag_bwd(float* table, float* state, float* state2, int64_t rows, const void* values,
                             int64_t nnz, const void* offsets, int64_t L, int32_t ids_dtype, int64_t B, int32_t D,
                             int32_t combiner, const float* grad, int64_t grad_row_stride, int32_t optimizer, float lr,
                             float eps, float beta1, float beta2, const float* lr_device, void* workspace,
                             int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(table && grad, "mh_embedding_bag_bwd: null argument");
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_bag_bwd: bad ids_dtype");
    MH_REQUIRE(combiner >= MH_COMBINER_SUM && combiner <= MH_COMBINER_MAX, "mh_embedding_bag_bwd: bad combiner %d", combiner);
    MH_REQUIRE(combiner != MH_COMBINER_MAX || !offsets, "mh_embedding_bag_bwd: the max combiner is defined for dense lists only");
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
    char* ws = static_cast<char*>(workspace);
    float* scale = reinterpret_cast<float*>(ws);
    ws += align_up((size_t)B * sizeof(float), 256);
    float* gexp = reinterpret_cast<float*>(ws);
    ws += align_up((size_t)nnz * D * sizeof(float), 256);
    {
        const int32_t st = mh_embedding_bag_expand(table, rows, values, nnz, offsets, L, ids_dtype, B, D, combiner, grad,
                                                   grad_row_stride, gexp, scale, stream);
        if (st != MH_OK) return st;
    }
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



int64_t mh_embedding_bag_bwd_multi_workspace_bytes(int64_t B, int64_t max_nnz, int32_t F, int32_t D) {
    if (B <= 0 || max_nnz <= 0 || F <= 0 || D <= 0) return 0;
    BagMultiWs w;
    if (!bag_multi_ws(B, max_nnz, F, D, &w)) return -1;
    return w.total;
}

// F multi-hot features over DISTINCT tables in one fused update: the combiner divisors of all bags (one launch), bag index + padded
// ids of all values (one launch), then ONE segmented sort / piece list / segmented reduce + optimizer over the F x max(nnz) values
// (mh_embedding_gather_bwd's pipeline) that reads the gradient row of a value through its bag index -- no expanded [nnz, D] gradient.
int32_t mh_embedding_bag_bwd_multi(float* const* tables, float* const* state, float* const* state2, const int64_t* table_rows,
                                   const void* const* values, const int64_t* nnz, const void* const* offsets, int64_t L,
                                   int32_t ids_dtype, int64_t B, int32_t F, int32_t D, int32_t combiner, const float* grad,
                                   int64_t grad_row_stride, const int64_t* grad_offset, int32_t optimizer, float lr, float eps,
                                   float beta1, float beta2, const float* lr_device, void* workspace, int64_t workspace_bytes,
                                   mh_stream_t stream) {
    MH_REQUIRE(tables && table_rows && values && nnz && grad && grad_offset, "mh_embedding_bag_bwd_multi: null argument");
    MH_REQUIRE(F >= 1 && F < MH_MAX_FEATURES, "mh_embedding_bag_bwd_multi: F=%d outside [1,%d]", F, MH_MAX_FEATURES - 1);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_bag_bwd_multi: bad ids_dtype");
    MH_REQUIRE(combiner >= MH_COMBINER_SUM && combiner <= MH_COMBINER_SQRTN, "mh_embedding_bag_bwd_multi: combiner must be sum, mean or sqrtn");
    MH_REQUIRE(offsets || L >= 1, "mh_embedding_bag_bwd_multi: need CSR offsets or a list length L >= 1");
    if (B <= 0) return MH_OK;
    int64_t max_nnz = 0;
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(nnz[f] >= 0 && (nnz[f] == 0 || values[f]), "mh_embedding_bag_bwd_multi: null values of feature %d", f);
        MH_REQUIRE(offsets == nullptr || offsets[f], "mh_embedding_bag_bwd_multi: null offsets of feature %d", f);
        MH_REQUIRE(offsets != nullptr || nnz[f] == B * L, "mh_embedding_bag_bwd_multi: dense list needs nnz == B * L");
        for (int g = 0; g < f; ++g)
            MH_REQUIRE(tables[g] != tables[f], "mh_embedding_bag_bwd_multi: features %d and %d share a table (one call per table set)", g, f);
        if (nnz[f] > max_nnz) max_nnz = nnz[f];
    }
    if (max_nnz == 0) return MH_OK;
    BagMultiWs w;
    MH_REQUIRE(bag_multi_ws(B, max_nnz, F, D, &w), "mh_embedding_bag_bwd_multi: sizes out of range");
    MH_REQUIRE(workspace && workspace_bytes >= w.total, "mh_embedding_bag_bwd_multi: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)w.total);
    MH_REQUIRE(w.Bp < (1ll << 26) && B < (1ll << 26), "mh_embedding_bag_bwd_multi: B and the longest value list must be < 2^26");
    hipStream_t s = mh_stream(stream);
    char* ws = static_cast<char*>(workspace);
    float* scale = reinterpret_cast<float*>(ws + w.off_scale);
    int32_t* map = reinterpret_cast<int32_t*>(ws + w.off_map);
    void* idpad = ws + w.off_ids;
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024 && grad_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0,
               "mh_embedding_bag_bwd_multi: grad must be 16-byte aligned with D and grad_row_stride multiples of 4");
    MH_REQUIRE(optimizer >= MH_OPT_SGD && optimizer <= MH_OPT_ADAM, "mh_embedding_bag_bwd_multi: bad optimizer %d", optimizer);
    MH_REQUIRE(optimizer == MH_OPT_SGD || state, "mh_embedding_bag_bwd_multi: Adagrad / Adam need state tables");
    MH_REQUIRE(optimizer != MH_OPT_ADAM || state2, "mh_embedding_bag_bwd_multi: Adam needs the second-moment tables");
    // combiner divisors of every bag of every feature
    BagMultiArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(table_rows[f] >= 1 && tables[f], "mh_embedding_bag_bwd_multi: table %d is null or has no rows", f);
        MH_REQUIRE(optimizer == MH_OPT_SGD || state[f], "mh_embedding_bag_bwd_multi: null optimizer state for feature %d", f);
        MH_REQUIRE(optimizer != MH_OPT_ADAM || state2[f], "mh_embedding_bag_bwd_multi: null second moment for feature %d", f);
        MH_REQUIRE(grad_offset[f] >= 0 && grad_offset[f] % 4 == 0 && grad_offset[f] + D <= grad_row_stride,
                   "mh_embedding_bag_bwd_multi: grad offset of feature %d misaligned or out of row", f);
        ba.values[f] = values[f];
        ba.offsets[f] = offsets ? offsets[f] : nullptr;
        ba.nnz[f] = nnz[f];
    }
    const dim3 gs((unsigned)mh_ceil_div(B, 256), (unsigned)F);
    if (ids_dtype == MH_I32) MH_LAUNCH((bag_scale_multi_kernel<int32_t>), gs, dim3(256), 0, s, ba, L, B, combiner, scale);
    else MH_LAUNCH((bag_scale_multi_kernel<int64_t>), gs, dim3(256), 0, s, ba, L, B, combiner, scale);

    // ---- small tables (D = 64, <= 128 rows): the count-matrix GEMM (kernel comment); everything else: the sort pipeline ------------------
    const int LPR = D / 4;
    bool small_ok = !deterministic_mode() && D == 64;
    if (const char* e = MH_LAB_ENV("MERLIN_HIP_BAG_SMALL")) small_ok = small_ok && atoi(e) != 0;
    BagSmallArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    int nsmall = 0, nbig = 0, big[MH_MAX_FEATURES];
    int64_t sw = 0;  // words used of the small-table buffers
    int max_small_rows = 0;
    // one partial block per wavefront: SMALL_BLOCKS wavefronts per feature at most (64 bags per block of work)
    int64_t nwg = SMALL_BLOCKS / 4;
    const int64_t max_wg = mh_ceil_div(mh_ceil_div(B, 64), 4);
    if (nwg > max_wg) nwg = max_wg;
    const int64_t nblk = nwg * 4;
    for (int f = 0; f < F; ++f) {
        const bool is_small = small_ok && nnz[f] > 0 && table_rows[f] <= SMALL_MAX_ROWS && table_rows[f] * D * 4 <= SMALL_TABLE_BYTES;
        if (!is_small) {
            big[nbig++] = f;
            continue;
        }
        const int k = nsmall++;
        sa.values[k] = values[f];
        sa.offsets[k] = offsets ? offsets[f] : nullptr;
        sa.table[k] = tables[f];
        sa.state[k] = state ? state[f] : nullptr;
        sa.state2[k] = state2 ? state2[f] : nullptr;
        sa.goff[k] = grad_offset[f];
        sa.rows[k] = (int)table_rows[f];
        sa.feat[k] = f;
        sa.soff[k] = sw;
        sw += nblk * (table_rows[f] * D + ((table_rows[f] + 3) & ~3ll));  // nblk partial blocks of [rows, D] sums + [rows] flags
        if ((int)table_rows[f] > max_small_rows) max_small_rows = (int)table_rows[f];
    }
    if (nsmall > 0) {
        float* gsum = reinterpret_cast<float*>(ws + w.off_small);
        MH_REQUIRE(sw <= w.small_words, "mh_embedding_bag_bwd_multi: small-table buffers out of range");
        const OptHyper hp = {lr, eps, beta1, beta2, lr_device};
        const dim3 ga((unsigned)nwg, (unsigned)nsmall);
        if (ids_dtype == MH_I32) MH_LAUNCH((bag_small_gemm_kernel<int32_t>), ga, dim3(256), 0, s, sa, L, B, grad, grad_row_stride, scale, gsum);
        else MH_LAUNCH((bag_small_gemm_kernel<int64_t>), ga, dim3(256), 0, s, sa, L, B, grad, grad_row_stride, scale, gsum);
        MH_LAUNCH(bag_small_apply_kernel, dim3((unsigned)mh_ceil_div(max_small_rows, 256 / LPR), (unsigned)nsmall), dim3(256), 0, s, sa, (int)nblk, D, LPR,
                  gsum, optimizer, hp);
        MH_CHECK_LAUNCH("mh_embedding_bag_bwd_multi (small tables)");
    }
    if (nbig == 0) return MH_OK;

    // ---- the other features: bag of every value + padded ids, pre-scaled feature-major gradient, ONE sort / piece list / walk ---------
    int64_t big_nnz = 0;
    BagMultiArgs bb;
    std::memset(&bb, 0, sizeof(bb));
    BwdArgsOffsets go;
    std::memset(&go, 0, sizeof(go));
    float* btab[MH_MAX_FEATURES];
    float* bst[MH_MAX_FEATURES];
    float* bst2[MH_MAX_FEATURES];
    int64_t brows[MH_MAX_FEATURES], fm_off[MH_MAX_FEATURES];
    for (int g = 0; g < nbig; ++g) {
        const int f = big[g];
        bb.values[g] = values[f];
        bb.offsets[g] = offsets ? offsets[f] : nullptr;
        bb.nnz[g] = nnz[f];
        if (nnz[f] > big_nnz) big_nnz = nnz[f];
        go.v[g] = grad_offset[f];
        go.sidx[g] = f;
        btab[g] = tables[f];
        bst[g] = state ? state[f] : nullptr;
        bst2[g] = state2 ? state2[f] : nullptr;
        brows[g] = table_rows[f];
        fm_off[g] = (int64_t)g * B * D;
    }
    if (big_nnz == 0) return MH_OK;
    const int64_t Bp = (big_nnz + 63) / 64 * 64;  // <= w.Bp: the buffers were sized for the longest list of ALL features
    // one workgroup per 256 bags + the workgroups that write the padding (at most Bp - min nnz entries)
    const dim3 gi((unsigned)(mh_ceil_div(B, 256) + 64), (unsigned)nbig);
    const void* idp[MH_MAX_FEATURES];
    if (ids_dtype == MH_I32) {
        MH_LAUNCH((bag_index_pad_kernel<int32_t>), gi, dim3(256), 0, s, bb, L, B, Bp, map, static_cast<int32_t*>(idpad));
        for (int g = 0; g < nbig; ++g) idp[g] = static_cast<int32_t*>(idpad) + (int64_t)g * Bp;
    } else {
        MH_LAUNCH((bag_index_pad_kernel<int64_t>), gi, dim3(256), 0, s, bb, L, B, Bp, map, static_cast<int64_t*>(idpad));
        for (int g = 0; g < nbig; ++g) idp[g] = static_cast<int64_t*>(idpad) + (int64_t)g * Bp;
    }
    float* gsc = reinterpret_cast<float*>(ws + w.off_gs);
    MH_LAUNCH(bag_prescale_kernel, dim3((unsigned)mh_ceil_div(B * LPR, 256), (unsigned)nbig), dim3(256), 0, s, grad, grad_row_stride, go, scale, B,
              LPR, gsc);
    const GradMap gm{map, Bp, 1, B};
    return gather_bwd_impl(btab, state ? bst : nullptr, brows, idp, ids_dtype, Bp, nbig, D, gsc, D, fm_off, optimizer, lr, eps,
                           state2 ? bst2 : nullptr, beta1, beta2, lr_device, ws + w.off_inner, workspace_bytes - w.off_inner, stream, PH_ALL, gm);
}

}  // extern "C"
