// Device-side negative samplers of the retrieval path (gfx950).
// Reference: PopularityBasedSamplerV2.sample (merlin/models/tf/outputs/sampling/popularity.py:118-137) calls
// tf.random.log_uniform_candidate_sampler(range_max = max_id - min_id, num_sampled, unique): class k is drawn with
// P(k) = (log(k + 2) - log(k + 1)) / log(range_max + 1) as floor(exp(u log(range_max + 1))) - 1; with unique = True draws are
// rejected until num_sampled DISTINCT classes were seen, returned in first-appearance order.
//
// Here: draw i of call c is a pure function of (seed, c, i) -- Philox4x32-10 with key = seed, counter = (i, c) -- so the
// sequence d_0, d_1, ... of a call is defined without any sequential state, and "the first n distinct values of that
// sequence, in order" can be produced by parallel rounds: one workgroup draws 1024 values per round, inserts them into an
// open-addressing hash set (value -> smallest draw index that produced it, atomicCAS + atomicMin), keeps the draws that own
// their slot (first occurrences, this round or any earlier one excluded) and appends them in draw order by a block scan.
// The call counter lives in device memory and is bumped by the kernel: a captured hipGraph replays a NEW sample every time.
// No host synchronisation, no host loop (the first version drew on the CPU and converted with .tolist() every step).
#include "mh_common.h"

namespace {

constexpr int ST = 1024;  // threads of the (single) workgroup = draws per round

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0;
    o1 = c1;
}

// class of draw i: floor(exp(u log(range_max + 1))) - 1 clamped to [0, range_max - 1], u in [0, 1) with 53 random bits
__device__ __forceinline__ int64_t log_uniform_draw(uint64_t i, uint64_t call, uint64_t seed, double log_range, int64_t range_max) {
    uint32_t a, b;
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), a, b);
    const double u = (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
    int64_t k = (int64_t)floor(exp(u * log_range)) - 1;
    if (k < 0) k = 0;
    if (k > range_max - 1) k = range_max - 1;
    return k;
}

__device__ __forceinline__ uint32_t hash64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    return (uint32_t)x;
}

// state[0] = seed, state[1] = calls so far.  table: cap (power of two) slots of {key + 1 (0 = empty), owner draw index}
__global__ __launch_bounds__(ST) void log_uniform_sample_kernel(int64_t range_max, int64_t min_id, int64_t n, int unique,
                                                               uint64_t* __restrict__ state, int64_t* __restrict__ out,
                                                               unsigned long long* __restrict__ tkey,
                                                               unsigned long long* __restrict__ town, uint32_t cap_mask,
                                                               int* __restrict__ status) {
    __shared__ int wave_tot[ST / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t seed = state[0], call = state[1];
    const double log_range = log((double)range_max + 1.0);
    if (!unique) {
        for (int64_t i = tid; i < n; i += ST) out[i] = min_id + log_uniform_draw((uint64_t)i, call, seed, log_range, range_max);
        __syncthreads();  // every thread has read state[1]
        if (tid == 0) state[1] = call + 1;
        return;
    }
    for (uint32_t s = tid; s <= cap_mask; s += ST) {
        tkey[s] = 0ull;
        town[s] = ~0ull;
    }
    __syncthreads();
    int64_t done = 0;
    // each round adds at least one new class while classes remain (n <= range_max is checked by the caller); the bound
    // only guards against a broken argument combination: status = 1 then
    for (uint64_t round = 0; done < n && round < (1ull << 22); ++round) {
        const uint64_t d = round * ST + tid;
        const int64_t v = log_uniform_draw(d, call, seed, log_range, range_max);
        uint32_t h = hash64((uint64_t)v) & cap_mask;
        for (;;) {
            const unsigned long long old = atomicCAS(&tkey[h], 0ull, (unsigned long long)v + 1ull);
            if (old == 0ull || old == (unsigned long long)v + 1ull) break;
            h = (h + 1) & cap_mask;
        }
        atomicMin(&town[h], (unsigned long long)d);
        __syncthreads();
        // agent-scope load: the atomics above were performed in L2, a plain load could be served by a stale L1 line
        const bool fresh = __hip_atomic_load(&town[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)d;  // first draw ever of this class
        const uint64_t m = __ballot(fresh);
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < ST / 64; ++w) {
            const int c = wave_tot[w];
            if (w < wave) before += c;
            tot += c;
        }
        const int64_t pos = done + before + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        if (fresh && pos < n) out[pos] = min_id + v;
        done += tot;
        __syncthreads();  // wave_tot is rewritten by the next round
    }
    if (tid == 0) {
        state[1] = call + 1;
        if (status) *status = done < n ? 1 : 0;
    }
}

uint32_t table_capacity(int64_t n) {  // distinct classes ever inserted < n + ST; load factor <= 1/2
    uint64_t need = 2ull * (uint64_t)(n + ST), cap = 1024;
    while (cap < need) cap <<= 1;
    return (uint32_t)cap;
}

}  // namespace

extern "C" {

int64_t mh_log_uniform_sample_workspace_bytes(int64_t n, int32_t unique) {
    if (n <= 0) return 0;
    return 256 + (unique ? (int64_t)table_capacity(n) * 16 : 0);
}

int32_t mh_log_uniform_sample(int64_t range_max, int64_t min_id, int64_t n, int32_t unique, uint64_t* rng_state,
                              int64_t* out_ids, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(range_max >= 1, "mh_log_uniform_sample: range_max must be >= 1");
    MH_REQUIRE(n >= 0 && n < (1ll << 31), "mh_log_uniform_sample: bad sample count");
    MH_REQUIRE(!unique || n <= range_max, "mh_log_uniform_sample: cannot draw %lld distinct classes out of %lld",
               (long long)n, (long long)range_max);
    if (n == 0) return MH_OK;
    MH_REQUIRE(rng_state && out_ids, "mh_log_uniform_sample: null argument");
    const int64_t need = mh_log_uniform_sample_workspace_bytes(n, unique);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_log_uniform_sample: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    char* ws = static_cast<char*>(workspace);
    int* status = reinterpret_cast<int*>(ws);
    const uint32_t cap = unique ? table_capacity(n) : 1;
    unsigned long long* tkey = reinterpret_cast<unsigned long long*>(ws + 256);
    unsigned long long* town = tkey + cap;
    MH_LAUNCH(log_uniform_sample_kernel, dim3(1), dim3(ST), 0, mh_stream(stream), range_max, min_id, n, (int)unique,
                       rng_state, out_ids, tkey, town, cap - 1, status);
    MH_CHECK_LAUNCH("mh_log_uniform_sample");
    return MH_OK;
}

}  // extern "C"
