// Categorical embedding lookup kernels for gfx950 (HBM-bound).
//
// Reference semantics: EmbeddingTable._call_table (merlin/models/tf/inputs/embedding.py:424-471),
// EmbeddingFeatures.lookup_feature (:1126-1156), process_str_sequence_combiner (:1556-1587).
//
// Data layout: a table is [rows, D] fp32 row-major; one embedding row is D*4 bytes, moved as
// D/4 16-byte lanes ("LPR" = lanes per row).  A 64-wide wavefront therefore moves 64/LPR whole
// rows per vector-memory instruction (4 rows of 256 B at D=64), each row a contiguous,
// 16-B-aligned segment -- the widest coalescing a random-row gather admits.  Every thread keeps
// R independent row loads in flight before the first store so that a CU has
// 256 threads x R x 16 B = 32 KiB outstanding per workgroup, enough to cover HBM latency.
//
// Algorithmic bytes: per looked-up row  D*4 read + D*4 write + id bytes.
#include "mh_common.h"

namespace {

struct GatherArgs {
    const float* table[MH_MAX_FEATURES];
    const void* ids[MH_MAX_FEATURES];
    int64_t rows[MH_MAX_FEATURES];
    int64_t offset[MH_MAX_FEATURES];  // float offset of the feature inside an output row
};

// grid.x = sample tiles, grid.y = feature.  block = 256 threads = (256/LPR) rows x LPR lanes.
template <typename IdT, int R>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const GatherArgs args, int64_t B, int LPR,
                                                        float* __restrict__ out,
                                                        int64_t out_row_stride) {
    const int f = blockIdx.y;
    const float* __restrict__ table = args.table[f];
    const IdT* __restrict__ ids = static_cast<const IdT*>(args.ids[f]);
    const int64_t rows = args.rows[f];
    const int rows_per_pass = 256 / LPR;
    const int t = threadIdx.x;
    const int r_in = t / LPR;
    const int c = t - r_in * LPR;
    if (r_in >= rows_per_pass) return;
    const int64_t b0 = (int64_t)blockIdx.x * (rows_per_pass * R) + r_in;
    float* __restrict__ obase = out + args.offset[f] + c * 4;

    // ids first, RAW and branch-free (sample index clamped): a predicated load with its sign extension right behind it was
    // followed by s_waitcnt vmcnt(0) -- the R id loads were R sequential round trips (seen in the ISA)
    IdT raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t b = b0 + (int64_t)r * rows_per_pass;
        raw[r] = ids[b < B ? b : B - 1];
    }
    int64_t id[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t b = b0 + (int64_t)r * rows_per_pass;
        id[r] = (b < B) ? (int64_t)raw[r] : -1;
    }
    f32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = (id[r] >= 0) && (id[r] < rows);
        v[r] = ok ? *reinterpret_cast<const f32x4*>(table + id[r] * (int64_t)(LPR * 4) + c * 4)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t b = b0 + (int64_t)r * rows_per_pass;
        // written once, read by a later kernel after hundreds of MB more: streaming store (keeps the hot table rows in L2 / MALL)
        if (b < B) __builtin_nontemporal_store(v[r], reinterpret_cast<f32x4*>(obase + b * out_row_stride));
    }
}

// Bag (multi-hot) lookup, short bags: one LPR-lane group per bag, sequential accumulate (4 independent row loads in flight).  Long bags
// (>= 8 ids on average) take bag_fwd_coop_kernel below.
//   offsets == nullptr  -> dense list of fixed length L (bag b = [b*L, (b+1)*L)), negatives
//                          are not pruned (tf.gather semantics, zero row, still counted).
template <typename IdT>
__global__ __launch_bounds__(256) void bag_fwd_kernel(const float* __restrict__ table, int64_t rows,
                                                     const IdT* __restrict__ values,
                                                     const IdT* __restrict__ offsets, int64_t L,
                                                     int64_t B, int LPR, int combiner,
                                                     float* __restrict__ out,
                                                     int64_t out_row_stride) {
    const int t = threadIdx.x;
    const bool prune_neg = (offsets != nullptr);
    const int groups = 256 / LPR;
    const int r_in = t / LPR;
    const int c = t - r_in * LPR;
    const int64_t bag = (r_in < groups) ? (int64_t)blockIdx.x * groups + r_in : B;
    if (bag >= B) return;
    const int64_t beg = offsets ? (int64_t)offsets[bag] : bag * L;
    const int64_t end = offsets ? (int64_t)offsets[bag + 1] : beg + L;
    // MAX (dense lists only, process_str_sequence_combiner "max", inputs/embedding.py:1579-1580): elementwise maximum
    // over the L rows; a position whose id is out of range contributes the zero row, like every other combiner here
    const bool is_max = combiner == MH_COMBINER_MAX;
    const float a0 = is_max ? -INFINITY : 0.f;
    f32x4 acc = {a0, a0, a0, a0};
    auto comb = [is_max](f32x4& a, const f32x4 v) {
        if (is_max) {
            a.x = fmaxf(a.x, v.x); a.y = fmaxf(a.y, v.y); a.z = fmaxf(a.z, v.z); a.w = fmaxf(a.w, v.w);
        } else {
            a += v;
        }
    };
    int cnt = 0;
    const int64_t stride = (int64_t)LPR * 4;
    int64_t p = beg;
    // 4 independent loads per trip
    for (; p + 3 < end; p += 4) {
        int64_t i0 = values[p], i1 = values[p + 1], i2 = values[p + 2], i3 = values[p + 3];
        const bool k0 = !(prune_neg && i0 < 0), k1 = !(prune_neg && i1 < 0),
                   k2 = !(prune_neg && i2 < 0), k3 = !(prune_neg && i3 < 0);
        const bool o0 = i0 >= 0 && i0 < rows, o1 = i1 >= 0 && i1 < rows, o2 = i2 >= 0 && i2 < rows,
                   o3 = i3 >= 0 && i3 < rows;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 v0 = o0 ? *reinterpret_cast<const f32x4*>(table + i0 * stride + c * 4) : z;
        f32x4 v1 = o1 ? *reinterpret_cast<const f32x4*>(table + i1 * stride + c * 4) : z;
        f32x4 v2 = o2 ? *reinterpret_cast<const f32x4*>(table + i2 * stride + c * 4) : z;
        f32x4 v3 = o3 ? *reinterpret_cast<const f32x4*>(table + i3 * stride + c * 4) : z;
        comb(acc, v0);
        comb(acc, v1);
        comb(acc, v2);
        comb(acc, v3);
        cnt += (int)k0 + (int)k1 + (int)k2 + (int)k3;
    }
    for (; p < end; ++p) {
        int64_t i0 = values[p];
        const bool k0 = !(prune_neg && i0 < 0);
        const bool o0 = i0 >= 0 && i0 < rows;
        const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
        comb(acc, o0 ? *reinterpret_cast<const f32x4*>(table + i0 * stride + c * 4) : zz);
        cnt += (int)k0;
    }
    if (is_max && acc.x == -INFINITY) acc = f32x4{0.f, 0.f, 0.f, 0.f};  // a group that saw no position (cannot happen for L >= 1)
    if (cnt > 0) {
        if (combiner == MH_COMBINER_MEAN) {
            const float n = (float)cnt;
            acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
        } else if (combiner == MH_COMBINER_SQRTN) {
            const float n = sqrtf((float)cnt);
            acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
        }
    }
    *reinterpret_cast<f32x4*>(out + bag * out_row_stride + c * 4) = acc;
}

// The wave-cooperative lookup as a PERSISTENT kernel (round 6): a wavefront walks bags w, w + W, w + 2 W, ... and keeps three loads of three
// different bags in flight -- the offsets of bag i + 2, the ids of bag i + 1 (its range arrived an iteration ago) and the rows of bag i -- so a
// bag costs ONE memory round trip instead of the three dependent ones (offsets -> ids -> rows) of a wavefront that handles a single bag
// (the round-5 form: 0.52-0.57 of the HBM peak on a 12.8 GB table; this one measures the same 0.57 -- the random 256-byte row reads
// themselves, 4.3 TB/s of them, are what a cold table allows; tools/gpu_bagfwd.py).  Same additions in the same order: group g takes entries g, g + G, ...
// of every 64-id chunk, the groups combine by wavefront shuffles -- bit-identical results.
template <typename IdT>
__global__ __launch_bounds__(256) void bag_fwd_coop_kernel(const float* __restrict__ table, int64_t rows, const IdT* __restrict__ values,
                                                          const IdT* __restrict__ offsets, int64_t L, int64_t B, int LPR, int combiner,
                                                          float* __restrict__ out, int64_t out_row_stride) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int G = 64 / LPR, g = lane / LPR, c = lane - g * LPR;
    const int64_t W = (int64_t)gridDim.x * 4;
    int64_t bag = (int64_t)blockIdx.x * 4 + wave;
    if (bag >= B) return;  // the whole wavefront
    const bool prune_neg = (offsets != nullptr);
    const bool is_max = combiner == MH_COMBINER_MAX;
    const float a0 = is_max ? -INFINITY : 0.f;
    const int64_t stride = (int64_t)LPR * 4;
    auto comb = [is_max](f32x4& a, const f32x4 v) {
        if (is_max) {
            a.x = fmaxf(a.x, v.x); a.y = fmaxf(a.y, v.y); a.z = fmaxf(a.z, v.z); a.w = fmaxf(a.w, v.w);
        } else {
            a += v;
        }
    };
    auto range = [&](int64_t b, int64_t& beg, int64_t& end) {  // b clamped: a bag past the end re-reads the last one (never used)
        const int64_t bb = b < B ? b : B - 1;
        beg = offsets ? (int64_t)offsets[bb] : bb * L;
        end = offsets ? (int64_t)offsets[bb + 1] : beg + L;
    };
    auto ids_of = [&](int64_t beg, int64_t end) -> IdT { return (beg + lane < end) ? values[beg + lane] : (IdT)0; };
    int64_t beg, end, nbeg, nend, n2beg, n2end;
    range(bag, beg, end);
    range(bag + W, nbeg, nend);
    IdT myid = ids_of(beg, end);
    for (; bag < B; bag += W) {
        range(bag + 2 * W, n2beg, n2end);                                        // offsets two bags ahead
        const IdT nid = (bag + W < B) ? ids_of(nbeg, nend) : (IdT)0;             // ids one bag ahead
        f32x4 acc = {a0, a0, a0, a0};
        int cnt = 0;
        IdT cid = myid;
        for (int64_t c0 = beg; c0 < end; c0 += 64) {
            const int nchunk = (int)((end - c0) < 64 ? (end - c0) : 64);
            if (c0 != beg) cid = (lane < nchunk) ? values[c0 + lane] : (IdT)0;   // bags longer than a wavefront: the later chunks
            for (int kb = 0; kb < nchunk; kb += 8 * G) {  // wave-uniform trip count: every shuffle source lane is active
                const int k0 = kb + g;
                f32x4 v[8];
                int kept[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * G;
                    const bool live = k < nchunk;
                    const int64_t id = (int64_t)__shfl(cid, live ? k : 0);
                    const bool ok = live && id >= 0 && id < rows;
                    kept[u] = (live && !(prune_neg && id < 0)) ? 1 : 0;
                    v[u] = ok ? *reinterpret_cast<const f32x4*>(table + id * stride + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if (is_max && !live) v[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + u * G < nchunk) comb(acc, v[u]);
                    cnt += kept[u];
                }
            }
        }
        for (int off = LPR; off < 64; off <<= 1) {
            const f32x4 o = {__shfl_xor(acc.x, off), __shfl_xor(acc.y, off), __shfl_xor(acc.z, off), __shfl_xor(acc.w, off)};
            comb(acc, o);
            cnt += __shfl_xor(cnt, off);
        }
        if (g == 0) {
            if (is_max && acc.x == -INFINITY) acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cnt > 0) {
                if (combiner == MH_COMBINER_MEAN) {
                    const float n = (float)cnt;
                    acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
                } else if (combiner == MH_COMBINER_SQRTN) {
                    const float n = sqrtf((float)cnt);
                    acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
                }
            }
            *reinterpret_cast<f32x4*>(out + bag * out_row_stride + c * 4) = acc;
        }
        beg = nbeg; end = nend;
        nbeg = n2beg; nend = n2end;
        myid = nid;
    }
}

template <typename IdT>
int launch_bag(const float* table, int64_t rows, const void* values, const void* offsets,
               int64_t L, int64_t nnz_hint, int64_t B, int D, int combiner, float* out,
               int64_t out_row_stride, hipStream_t s) {
    const int LPR = D / 4;
    const bool pow2 = (LPR & (LPR - 1)) == 0;
    const bool coop = pow2 && LPR <= 32 && nnz_hint >= 8 * B;
    if (coop) {
        // persistent: every CU holds its share of wavefronts for the whole launch, each walking bags W apart (kernel comment)
        int64_t nb = mh_ceil_div(B, 4);
        const int64_t cap = (int64_t)mh_num_cus() * 5;  // what the kernel's registers let a CU hold: one resident wave of workgroups
        if (nb > cap) nb = cap;
        MH_LAUNCH((bag_fwd_coop_kernel<IdT>), dim3((unsigned)nb), dim3(256), 0, s, table, rows,
                           (const IdT*)values, (const IdT*)offsets, L, B, LPR, combiner, out,
                           out_row_stride);
    } else {
        const int groups = 256 / LPR;
        dim3 grid((unsigned)mh_ceil_div(B, groups));
        MH_LAUNCH((bag_fwd_kernel<IdT>), grid, dim3(256), 0, s, table, rows,
                           (const IdT*)values, (const IdT*)offsets, L, B, LPR, combiner, out,
                           out_row_stride);
    }
    return MH_OK;
}

}  // namespace

extern "C" {

int32_t mh_embedding_gather_fwd(const float* const* tables, const int64_t* table_rows,
                                const void* const* ids, int32_t ids_dtype, int64_t B, int32_t F,
                                int32_t D, float* out, int64_t out_row_stride,
                                const int64_t* out_offset, mh_stream_t stream) {
    MH_REQUIRE(tables && table_rows && ids && out && out_offset, "mh_embedding_gather_fwd: null argument");
    MH_REQUIRE(F >= 1 && F <= MH_MAX_FEATURES, "mh_embedding_gather_fwd: F=%d outside [1,%d]", F,
               MH_MAX_FEATURES);
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "mh_embedding_gather_fwd: D=%d must be a multiple of 4 in [4,1024]", D);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_gather_fwd: bad ids_dtype %d", ids_dtype);
    MH_REQUIRE(B >= 0, "mh_embedding_gather_fwd: negative batch");
    MH_REQUIRE(out_row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "mh_embedding_gather_fwd: out must be 16-byte aligned with out_row_stride %% 4 == 0");
    if (B == 0) return MH_OK;
    GatherArgs a;
    for (int f = 0; f < F; ++f) {
        MH_REQUIRE(tables[f] && ids[f], "mh_embedding_gather_fwd: null table/ids for feature %d", f);
        MH_REQUIRE((reinterpret_cast<uintptr_t>(tables[f]) & 15) == 0, "mh_embedding_gather_fwd: table %d not 16-byte aligned", f);
        MH_REQUIRE(out_offset[f] >= 0 && out_offset[f] % 4 == 0 && out_offset[f] + D <= out_row_stride,
                   "mh_embedding_gather_fwd: offset %lld of feature %d is misaligned or exceeds out_row_stride", (long long)out_offset[f], f);
        a.table[f] = tables[f];
        a.ids[f] = ids[f];
        a.rows[f] = table_rows[f];
        a.offset[f] = out_offset[f];
    }
    const int LPR = D / 4;
    constexpr int R = 8;
    const int rows_per_block = (256 / LPR) * R;
    dim3 grid((unsigned)mh_ceil_div(B, rows_per_block), (unsigned)F);
    hipStream_t s = mh_stream(stream);
    if (ids_dtype == MH_I32)
        MH_LAUNCH((gather_fwd_kernel<int32_t, R>), grid, dim3(256), 0, s, a, B, LPR, out, out_row_stride);
    else
        MH_LAUNCH((gather_fwd_kernel<int64_t, R>), grid, dim3(256), 0, s, a, B, LPR, out, out_row_stride);
    MH_CHECK_LAUNCH("mh_embedding_gather_fwd");
    return MH_OK;
}

int32_t mh_embedding_bag_fwd(const float* table, int64_t rows, const void* values, int64_t nnz,
                             const void* offsets, int32_t ids_dtype, int64_t B, int32_t D,
                             int32_t combiner, float* out, int64_t out_row_stride,
                             mh_stream_t stream) {
    MH_REQUIRE(table && offsets && out, "mh_embedding_bag_fwd: null argument");
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "mh_embedding_bag_fwd: D=%d must be a multiple of 4 in [4,1024]", D);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_bag_fwd: bad ids_dtype %d", ids_dtype);
    MH_REQUIRE(combiner >= MH_COMBINER_SUM && combiner <= MH_COMBINER_SQRTN, "mh_embedding_bag_fwd: bad combiner %d", combiner);
    MH_REQUIRE(out_row_stride % 4 == 0 && out_row_stride >= D, "mh_embedding_bag_fwd: bad out_row_stride");
    if (B <= 0) return MH_OK;
    hipStream_t s = mh_stream(stream);
    MH_REQUIRE(nnz == 0 || values, "mh_embedding_bag_fwd: null values");
    const int64_t nnz_hint = nnz;  // long bags (>= 8 ids on average) take the wave-cooperative kernel
    if (ids_dtype == MH_I32)
        launch_bag<int32_t>(table, rows, values, offsets, 0, nnz_hint, B, D, combiner, out, out_row_stride, s);
    else
        launch_bag<int64_t>(table, rows, values, offsets, 0, nnz_hint, B, D, combiner, out, out_row_stride, s);
    MH_CHECK_LAUNCH("mh_embedding_bag_fwd");
    return MH_OK;
}

int32_t mh_embedding_dense_list_fwd(const float* table, int64_t rows, const void* ids,
                                    int32_t ids_dtype, int64_t B, int32_t L, int32_t D,
                                    int32_t combiner, float* out, int64_t out_row_stride,
                                    mh_stream_t stream) {
    MH_REQUIRE(table && ids && out, "mh_embedding_dense_list_fwd: null argument");
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "mh_embedding_dense_list_fwd: D=%d must be a multiple of 4 in [4,1024]", D);
    MH_REQUIRE(L >= 1, "mh_embedding_dense_list_fwd: L must be >= 1");
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_embedding_dense_list_fwd: bad ids_dtype %d", ids_dtype);
    MH_REQUIRE(combiner == MH_COMBINER_SUM || combiner == MH_COMBINER_MEAN || combiner == MH_COMBINER_MAX,
               "mh_embedding_dense_list_fwd: combiner must be sum, mean or max");
    MH_REQUIRE(out_row_stride % 4 == 0 && out_row_stride >= D, "mh_embedding_dense_list_fwd: bad out_row_stride");
    if (B <= 0) return MH_OK;
    hipStream_t s = mh_stream(stream);
    if (ids_dtype == MH_I32)
        launch_bag<int32_t>(table, rows, ids, nullptr, L, (int64_t)L * B, B, D, combiner, out, out_row_stride, s);
    else
        launch_bag<int64_t>(table, rows, ids, nullptr, L, (int64_t)L * B, B, D, combiner, out, out_row_stride, s);
    MH_CHECK_LAUNCH("mh_embedding_dense_list_fwd");
    return MH_OK;
}

}  // extern "C"
