// Entry points declared in include/merlin_hip.h whose kernels have not landed yet.
// They fail loudly (MH_ERR_UNSUPPORTED + message); nothing falls back to a CPU path.
#include "mh_common.h"

#define MH_STUB(name)                                  \
    mh_set_error(name ": not implemented in this build"); \
    return MH_ERR_UNSUPPORTED

extern "C" {

int32_t mh_cross_layer_fwd(const float*, const float*, const float*, const float*, int64_t, int32_t,
                           float*, mh_stream_t) {
    MH_STUB("mh_cross_layer_fwd");
}
int32_t mh_inbatch_softmax_fwd(const float*, const float*, const float*, const void*, const void*,
                               int32_t, int64_t, int64_t, int32_t, float, float, float*, int64_t,
                               float*, float*, mh_stream_t) {
    MH_STUB("mh_inbatch_softmax_fwd");
}
int32_t mh_inbatch_softmax_bwd(const float*, const float*, const float*, const void*, const void*,
                               int32_t, int64_t, int64_t, int32_t, float, float, const float*, float,
                               float*, float*, float*, mh_stream_t) {
    MH_STUB("mh_inbatch_softmax_bwd");
}
int64_t mh_topk_workspace_bytes(int64_t, int64_t, int32_t) { return 0; }
int32_t mh_topk_dot(const float*, const float*, const int32_t*, int64_t, int64_t, int32_t, int32_t,
                    float*, int32_t*, int32_t*, void*, int64_t, mh_stream_t) {
    MH_STUB("mh_topk_dot");
}

}  // extern "C"
