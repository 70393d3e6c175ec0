// Internal helpers shared by the HIP translation units of libmerlin_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/merlin_hip.h"

#define MH_WAVE 64

void mh_set_error(const char* fmt, ...);

#define MH_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            mh_set_error(__VA_ARGS__);        \
            return MH_ERR_INVALID_ARGUMENT;   \
        }                                     \
    } while (0)

#define MH_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            mh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return MH_ERR_LAUNCH;                                                   \
        }                                                                           \
    } while (0)

static inline hipStream_t mh_stream(mh_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- launch recorder (mh_record.hip) ------------------------------------------------------------------------------------
// Every kernel launch of the library goes through MH_LAUNCH.  While a recording is open (mh_record_begin .. mh_record_end) the
// launch is ALSO kept -- kernel, geometry, stream and a by-value copy of its arguments -- and mh_record_replay re-issues the whole
// sequence, in order, on the streams it was issued on, with the recorded event hand-offs in between: the launch sequence of a train
// step replayed from C at a few microseconds of host time per launch, with the multi-stream overlap of the eager step (a hipGraph
// of the same step replays on ONE hardware queue under ROCm 7).  Same contract as graph capture: fixed shapes, persistent buffers.
#include <functional>
#ifdef MH_STANDALONE  // tools/exp labs that include the GEMM headers without linking the library
static inline bool mh_recording() { return false; }
static inline void mh_record_op(std::function<void()>&&) {}
#else
bool mh_recording();
void mh_record_op(std::function<void()>&& op);
#endif

#define MH_LAUNCH(kern, grid, block, lds, stream, ...)                                                     \
    do {                                                                                                   \
        if (mh_recording()) {                                                                              \
            auto mh_k_ = kern;                                                                             \
            const dim3 mh_g_ = (grid), mh_b_ = (block);                                                    \
            const size_t mh_l_ = (size_t)(lds);                                                            \
            hipStream_t mh_s_ = (stream);                                                                  \
            mh_record_op([=]() { hipLaunchKernelGGL(mh_k_, mh_g_, mh_b_, mh_l_, mh_s_, __VA_ARGS__); });   \
        }                                                                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                   \
    } while (0)

// Tuning / ablation knobs of the labs (tools/exp, profiles/*_notes.md) exist only in a library built with -DMH_LAB
// (MH_LAB=1 python -c "from models_amd import build; build.build(force=True)").  The shipped library reads the switches documented in
// README.md ("Environment switches", enforced by tests/test_abi.py) and nothing else: MH_LAB_ENV is a constant nullptr there.
#ifdef MH_LAB
#define MH_LAB_ENV(name) getenv(name)
#else
#define MH_LAB_ENV(name) (static_cast<const char*>(nullptr))
#endif

static inline int64_t mh_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Fill of `words` 32-bit words as a KERNEL on the stream (mh_misc.hip).  hipMemsetAsync is not used anywhere in the library:
// captured into a hipGraph it becomes a memset node, and ROCm 7 does not keep such a node ordered with the kernel
// nodes around it on every replay (measured: a replayed train step whose sort counters are cleared by a memset node
// sporadically sees them cleared late -- garbage piece counts, inf accumulators; a kernel node in the chain is ordered).
int32_t mh_fill_words(void* dst, uint32_t value, int64_t words, hipStream_t s);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// hi = bf16(v), lo = bf16(v - hi) for a PAIR of values, packed (first value in the low half): the 3-term split of the opt-in
// bf16x3 arithmetic.  gfx950 converts two floats to two round-to-nearest-even bf16 in one instruction (v_cvt_pk_bf16_f32): five
// instructions per pair where the bit-twiddling form (add 0x7fff + lsb, shift, class test) needs ~30 -- the epilogue of the split
// scorer was vector-ALU bound on it.  Same results for finite values.
#if defined(__HIPCC__)
typedef __bf16 mh_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float mh_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mh_split_pair(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    const mh_f32x2_t v = {v0, v1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mh_bf16x2_t));
    const mh_f32x2_t hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - hf, mh_bf16x2_t));
}
// h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) for a PAIR of values: the three bf16 pieces of the six-term "bf16x6" arithmetic -- together they
// hold the 24-bit significand (|v - h - m - l| <= 2^-27 |v|); both subtractions are exact in fp32
__device__ __forceinline__ void mh_split3_pair(float v0, float v1, uint32_t& h, uint32_t& m, uint32_t& l) {
    const mh_f32x2_t v = {v0, v1};
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mh_bf16x2_t));
    const mh_f32x2_t hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const mh_f32x2_t r1 = v - hf;
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, mh_bf16x2_t));
    const mh_f32x2_t mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    const mh_f32x2_t r2 = r1 - mf;
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, mh_bf16x2_t));
}
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Number of workgroups that fill the chip a few times over for grid-stride kernels.
int mh_num_cus();

// ---- internal GEMM launchers shared across translation units (fp32 MFMA, k-ascending chains) ----
// y[M,N] = act(x[M,K] W[K,N] + b)  (optionally the cross epilogue x0 * (.) + xres)
int32_t mh_internal_linear(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int K, int N,
                           int act, float* y, int64_t ldy, const float* x0, const float* xres, hipStream_t s);
// C[M,Nout] = A[M,Kc] B[Nout,Kc]^T
int32_t mh_internal_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int Nout, int Kc,
                            float* C, int64_t ldc, hipStream_t s);
// out[K,N] = X[M,K]^T Z[M,N], single slice (no split): deterministic k-ascending over M
int32_t mh_internal_gemm_tn(const float* X, int64_t ldx, const float* Z, int64_t ldz, int64_t M, int K, int N,
                            float* out, hipStream_t s);
