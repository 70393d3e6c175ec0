// Collectives behind the C ABI (SURVEY.md section 8b minimum set): RCCL over xGMI, one communicator per process
// (one process per GPU), every call asynchronous on the caller's stream and with HOST-KNOWN sizes only -- so a
// multi-GPU step that uses them is a fixed launch sequence and can be captured into a hipGraph.
//
// Reference role: SparseOperationKit behind `merlin/models/tf/distributed/embedding.py:117-149` (sok.lookup_sparse
// exchanges ids and vectors between the GPUs holding the rows) and Horovod's gradient all-reduce around the optimizer
// (`tf/models/base.py:476-508`).  RCCL itself is the vendor collective library (plumbing, like the HIP runtime); it is
// resolved at run time with dlopen so that the process keeps ONE copy (the one PyTorch already loaded, if any).
//
//   mh_sharded_lookup_fwd = mh_route_build (fixed windows)  ->  all-to-all(keys)  ->  mh_route_local_rows
//                           ->  gather of the local shards  ->  all-to-all(rows)  ->  scatter into the caller's layout
//   mh_sharded_lookup_bwd = gather of the gradient rows in owner order -> all-to-all -> fused dedup + optimizer update
//   mh_allreduce_dense    = in-place SUM: reduce-scatter + all-gather (every xGMI link carries 1/W per phase)
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "mh_common.h"

namespace {

// the slice of the NCCL / RCCL ABI used here (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE)
struct NcclUniqueId {
    char internal[128];
};
typedef void* ncclComm_t;
enum { NCCL_INT8 = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {  // a copy that is already resident (PyTorch's) wins: one RCCL per process
        api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (api.handle) break;
    }
    for (int i = 0; !api.handle && i < 3; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) return nullptr;
#define MH_SYM(field, name) *reinterpret_cast<void**>(&api.field) = dlsym(api.handle, name)
    MH_SYM(GetUniqueId, "ncclGetUniqueId");
    MH_SYM(CommInitRank, "ncclCommInitRank");
    MH_SYM(CommDestroy, "ncclCommDestroy");
    MH_SYM(GroupStart, "ncclGroupStart");
    MH_SYM(GroupEnd, "ncclGroupEnd");
    MH_SYM(Send, "ncclSend");
    MH_SYM(Recv, "ncclRecv");
    MH_SYM(AllReduce, "ncclAllReduce");
    MH_SYM(ReduceScatter, "ncclReduceScatter");
    MH_SYM(AllGather, "ncclAllGather");
    MH_SYM(GetErrorString, "ncclGetErrorString");
#undef MH_SYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd ||
        !api.AllReduce || !api.ReduceScatter || !api.AllGather) {
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}

struct MhComm {
    ncclComm_t comm;
    int rank, world;
};

#define MH_RCCL(call, what)                                                                              \
    do {                                                                                                 \
        const int rc_ = (call);                                                                          \
        if (rc_ != 0) {                                                                                  \
            mh_set_error("%s: RCCL error %d (%s)", what, rc_, R->GetErrorString ? R->GetErrorString(rc_) : "?"); \
            return MH_ERR_LAUNCH;                                                                        \
        }                                                                                                \
    } while (0)

// equal windows of `bytes` per peer; W == 1 without an RCCL communicator is a device copy (never part of a captured
// multi-GPU step: the product path at W == 1 aliases instead, see models_amd/distributed.py).  A one-rank communicator that
// was created WITH a unique id owns a real RCCL communicator and takes the RCCL path below (send / recv to itself): the
// single-GPU execution of exactly the code an N-rank job runs.
int32_t alltoall_issue(RcclApi* R, MhComm* c, const void* send, void* recv, int64_t bytes, hipStream_t s, const char* what);

// While the library records a step's launch sequence (mh_record_begin), a collective is part of that sequence: it is kept as a
// closure over its by-value arguments like a kernel launch and re-issued by mh_record_replay in the same place.
int32_t alltoall_bytes(RcclApi* R, MhComm* c, const void* send, void* recv, int64_t bytes, hipStream_t s, const char* what) {
    if (bytes <= 0) return MH_OK;
    if (mh_recording()) mh_record_op([=]() { (void)alltoall_issue(R, c, send, recv, bytes, s, what); });
    return alltoall_issue(R, c, send, recv, bytes, s, what);
}

int32_t alltoall_issue(RcclApi* R, MhComm* c, const void* send, void* recv, int64_t bytes, hipStream_t s, const char* what) {
    if (c->world == 1 && !c->comm) {
        if (hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            mh_set_error("%s: device copy failed", what);
            return MH_ERR_LAUNCH;
        }
        return MH_OK;
    }
    if (!R) {
        mh_set_error("%s: librccl.so not found", what);
        return MH_ERR_UNSUPPORTED;
    }
    MH_RCCL(R->GroupStart(), what);
    for (int p = 0; p < c->world; ++p) {
        MH_RCCL(R->Send(static_cast<const char*>(send) + (int64_t)p * bytes, (size_t)bytes, NCCL_INT8, p, c->comm, s), what);
        MH_RCCL(R->Recv(static_cast<char*>(recv) + (int64_t)p * bytes, (size_t)bytes, NCCL_INT8, p, c->comm, s), what);
    }
    MH_RCCL(R->GroupEnd(), what);
    return MH_OK;
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct LookupWs {  // persists from _fwd to _bwd of one step
    size_t off_send_keys, off_recv_keys, off_pos_of, off_src_row, off_counts, off_rows, off_gath, off_back, off_route,
        off_bwd, route_bytes, bwd_bytes, total;
};

LookupWs lookup_ws(int64_t B, int F, int W, int64_t cap, int D) {
    LookupWs L;
    const size_t nw = (size_t)W * cap;
    size_t o = 0;
    L.off_send_keys = o; o = up256(o + nw * 8);
    L.off_recv_keys = o; o = up256(o + nw * 8);
    L.off_pos_of = o; o = up256(o + (size_t)F * B * 8);
    L.off_src_row = o; o = up256(o + nw * 8);
    L.off_counts = o; o = up256(o + (size_t)W * 8);
    L.off_rows = o; o = up256(o + nw * 8);
    L.off_gath = o; o = up256(o + nw * D * 4);   // rows gathered for the peers / gradient rows received from them
    L.off_back = o; o = up256(o + nw * D * 4);   // rows returned by the owners / gradient rows sent to them
    L.route_bytes = (size_t)mh_route_workspace_bytes((int64_t)F * B, W);
    L.off_route = o; o = up256(o + L.route_bytes);
    L.bwd_bytes = (size_t)mh_embedding_bwd_workspace_bytes((int64_t)nw, 1, D);
    L.off_bwd = o; o = up256(o + L.bwd_bytes);
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

int32_t mh_comm_unique_id(void* id128) {
    MH_REQUIRE(id128, "mh_comm_unique_id: null argument");
    RcclApi* R = rccl();
    MH_REQUIRE(R, "mh_comm_unique_id: librccl.so not found");
    NcclUniqueId id;
    MH_RCCL(R->GetUniqueId(&id), "mh_comm_unique_id");
    std::memcpy(id128, &id, sizeof(id));
    return MH_OK;
}

int32_t mh_comm_init(int32_t rank, int32_t world, const void* unique_id128, void** comm_out) {
    MH_REQUIRE(comm_out && world >= 1 && rank >= 0 && rank < world && world <= 64, "mh_comm_init: bad rank / world");
    MH_REQUIRE(world == 1 || unique_id128, "mh_comm_init: the unique id (mh_comm_unique_id on rank 0, broadcast by the caller) is required");
    MhComm* c = new MhComm{nullptr, rank, world};
    if (unique_id128) {  // world == 1 WITH an id: a real one-rank RCCL communicator (RCCL accepts it) instead of the copy shortcut
        RcclApi* R = rccl();
        if (!R) {
            delete c;
            mh_set_error("mh_comm_init: librccl.so not found");
            return MH_ERR_UNSUPPORTED;
        }
        NcclUniqueId id;
        std::memcpy(&id, unique_id128, sizeof(id));
        const int rc = R->CommInitRank(&c->comm, world, id, rank);
        if (rc != 0) {
            delete c;
            mh_set_error("mh_comm_init: ncclCommInitRank failed (%d)", rc);
            return MH_ERR_LAUNCH;
        }
    }
    *comm_out = c;
    return MH_OK;
}

int32_t mh_comm_destroy(void* comm) {
    MhComm* c = static_cast<MhComm*>(comm);
    if (!c) return MH_OK;
    if (c->comm) {
        RcclApi* R = rccl();
        if (R) R->CommDestroy(c->comm);
    }
    delete c;
    return MH_OK;
}

int32_t mh_comm_alltoall(void* comm, const void* send, void* recv, int64_t bytes_per_peer, mh_stream_t stream) {
    MhComm* c = static_cast<MhComm*>(comm);
    MH_REQUIRE(c && send && recv && bytes_per_peer >= 0, "mh_comm_alltoall: bad argument");
    return alltoall_bytes(rccl(), c, send, recv, bytes_per_peer, mh_stream(stream), "mh_comm_alltoall");
}

static int32_t allreduce_issue(RcclApi* R, MhComm* c, float* buf, int64_t n, hipStream_t s);

int32_t mh_allreduce_dense(void* comm, float* buf, int64_t n, mh_stream_t stream) {
    MhComm* c = static_cast<MhComm*>(comm);
    MH_REQUIRE(c && (buf || n == 0) && n >= 0, "mh_allreduce_dense: bad argument");
    if ((c->world == 1 && !c->comm) || n == 0) return MH_OK;
    RcclApi* R = rccl();
    MH_REQUIRE(R, "mh_allreduce_dense: librccl.so not found");
    hipStream_t s = mh_stream(stream);
    if (mh_recording()) mh_record_op([=]() { (void)allreduce_issue(R, c, buf, n, s); });  // see alltoall_bytes
    return allreduce_issue(R, c, buf, n, s);
}

static int32_t allreduce_issue(RcclApi* R, MhComm* c, float* buf, int64_t n, hipStream_t s) {
    // MERLIN_HIP_ALLREDUCE=plain: one ncclAllReduce also for divisible sizes (A/B of the two forms; the only way to reach the
    // plain form on a one-rank communicator, where every n is divisible)
    const char* form = getenv("MERLIN_HIP_ALLREDUCE");  // read per call: a host-side string compare beside a collective
    const bool plain = form && !strcmp(form, "plain");
    if (n % c->world == 0 && !plain) {  // reduce-scatter into this rank's slice (in place), then all-gather the slices
        const size_t per = (size_t)(n / c->world);
        MH_RCCL(R->ReduceScatter(buf, buf + (size_t)c->rank * per, per, NCCL_FLOAT32, NCCL_SUM, c->comm, s), "mh_allreduce_dense");
        MH_RCCL(R->AllGather(buf + (size_t)c->rank * per, buf, per, NCCL_FLOAT32, c->comm, s), "mh_allreduce_dense");
    } else {
        MH_RCCL(R->AllReduce(buf, buf, (size_t)n, NCCL_FLOAT32, NCCL_SUM, c->comm, s), "mh_allreduce_dense");
    }
    return MH_OK;
}

int64_t mh_sharded_lookup_workspace_bytes(int64_t B, int32_t F, int32_t W, int64_t capacity, int32_t D) {
    if (B <= 0 || F <= 0 || W <= 0 || capacity <= 0 || D <= 0) return 0;
    return (int64_t)lookup_ws(B, F, W, capacity, D).total;
}

int32_t mh_sharded_lookup_fwd(void* comm, const void* const* ids, int32_t ids_dtype, int32_t F, int64_t B,
                              int64_t capacity, const float* local_shards, const int64_t* base,
                              const int64_t* shard_rows, int32_t D, float* out, int64_t out_row_stride,
                              const int64_t* out_offset, int32_t* overflow, void* workspace, int64_t workspace_bytes,
                              mh_stream_t stream) {
    MhComm* c = static_cast<MhComm*>(comm);
    MH_REQUIRE(c && ids && local_shards && base && out && out_offset && workspace, "mh_sharded_lookup_fwd: null argument");
    MH_REQUIRE(F >= 1 && F <= MH_MAX_FEATURES && B >= 1 && capacity >= 1 && D >= 4 && D % 4 == 0, "mh_sharded_lookup_fwd: bad shape");
    const int W = c->world;
    const LookupWs L = lookup_ws(B, F, W, capacity, D);
    MH_REQUIRE(workspace_bytes >= (int64_t)L.total, "mh_sharded_lookup_fwd: workspace too small (%lld < %zu)", (long long)workspace_bytes, L.total);
    char* ws = static_cast<char*>(workspace);
    int64_t* send_keys = reinterpret_cast<int64_t*>(ws + L.off_send_keys);
    int64_t* recv_keys = reinterpret_cast<int64_t*>(ws + L.off_recv_keys);
    int64_t* pos_of = reinterpret_cast<int64_t*>(ws + L.off_pos_of);
    int64_t* src_row = reinterpret_cast<int64_t*>(ws + L.off_src_row);
    int64_t* counts = reinterpret_cast<int64_t*>(ws + L.off_counts);
    int64_t* rows = reinterpret_cast<int64_t*>(ws + L.off_rows);
    float* gath = reinterpret_cast<float*>(ws + L.off_gath);
    float* back = reinterpret_cast<float*>(ws + L.off_back);
    const int64_t nw = (int64_t)W * capacity;
    int32_t slots[MH_MAX_FEATURES];
    for (int f = 0; f < F; ++f) slots[f] = f;  // gradient rows come back as a [B, F, D] stack (see _bwd)
    int32_t st = mh_route_build(ids, ids_dtype, F, B, W, slots, F, capacity, send_keys, pos_of, src_row, counts, overflow,
                                ws + L.off_route, (int64_t)L.route_bytes, stream);
    if (st != MH_OK) return st;
    hipStream_t s = mh_stream(stream);
    st = alltoall_bytes(rccl(), c, send_keys, recv_keys, capacity * 8, s, "mh_sharded_lookup_fwd");
    if (st != MH_OK) return st;
    st = mh_route_local_rows(recv_keys, nw, base, shard_rows, F, rows, stream);
    if (st != MH_OK) return st;
    {  // gather the requested rows of the concatenated local shards (row -1 reads as zeros)
        const float* tabs[1] = {local_shards};
        const int64_t trows[1] = {(int64_t)1 << 40};  // the row check was done by mh_route_local_rows
        const void* idp[1] = {rows};
        const int64_t off[1] = {0};
        st = mh_embedding_gather_fwd(tabs, trows, idp, MH_I64, nw, 1, D, gath, D, off, stream);
        if (st != MH_OK) return st;
    }
    st = alltoall_bytes(rccl(), c, gath, back, capacity * (int64_t)D * 4, s, "mh_sharded_lookup_fwd");
    if (st != MH_OK) return st;
    {  // place request (f, b) = back[pos_of[f, b]] into the caller's layout: ONE multi-"table" gather
        const float* tabs[MH_MAX_FEATURES];
        int64_t trows[MH_MAX_FEATURES];
        const void* idp[MH_MAX_FEATURES];
        for (int f = 0; f < F; ++f) {
            tabs[f] = back;
            trows[f] = nw;
            idp[f] = pos_of + (int64_t)f * B;
        }
        st = mh_embedding_gather_fwd(tabs, trows, idp, MH_I64, B, F, D, out, out_row_stride, out_offset, stream);
    }
    return st;
}

int32_t mh_sharded_lookup_bwd(void* comm, int32_t F, int64_t B, int64_t capacity, int32_t D, const float* grad_stack,
                              float* local_shards, float* state, float* state2, int64_t local_rows_total,
                              int32_t optimizer, float lr, float eps, float beta1, float beta2, const float* lr_device,
                              void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MhComm* c = static_cast<MhComm*>(comm);
    MH_REQUIRE(c && grad_stack && local_shards && workspace, "mh_sharded_lookup_bwd: null argument");
    const int W = c->world;
    const LookupWs L = lookup_ws(B, F, W, capacity, D);
    MH_REQUIRE(workspace_bytes >= (int64_t)L.total, "mh_sharded_lookup_bwd: workspace too small");
    char* ws = static_cast<char*>(workspace);
    const int64_t* src_row = reinterpret_cast<const int64_t*>(ws + L.off_src_row);
    const int64_t* rows = reinterpret_cast<const int64_t*>(ws + L.off_rows);
    float* gath = reinterpret_cast<float*>(ws + L.off_gath);
    float* back = reinterpret_cast<float*>(ws + L.off_back);
    const int64_t nw = (int64_t)W * capacity;
    int32_t st;
    {  // gradient rows in owner order: back[p] = grad_stack[b, f, :] with src_row[p] = b * F + f (padding: zero row)
        const float* tabs[1] = {grad_stack};
        const int64_t trows[1] = {B * F};
        const void* idp[1] = {src_row};
        const int64_t off[1] = {0};
        st = mh_embedding_gather_fwd(tabs, trows, idp, MH_I64, nw, 1, D, back, D, off, stream);
        if (st != MH_OK) return st;
    }
    st = alltoall_bytes(rccl(), c, back, gath, capacity * (int64_t)D * 4, mh_stream(stream), "mh_sharded_lookup_bwd");
    if (st != MH_OK) return st;
    float* tabs[1] = {local_shards};
    float* st1[1] = {state};
    float* st2[1] = {state2};
    const int64_t trows[1] = {local_rows_total};
    const void* idp[1] = {rows};
    const int64_t off[1] = {0};
    return mh_embedding_gather_bwd(tabs, state ? st1 : nullptr, trows, idp, MH_I64, nw, 1, D, gath, D, off, optimizer, lr, eps,
                                   state2 ? st2 : nullptr, beta1, beta2, lr_device, ws + L.off_bwd, (int64_t)L.bwd_bytes, stream);
}

}  // extern "C"
