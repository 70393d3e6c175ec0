// Launch recorder: the kernel-launch sequence of one step, kept with by-value arguments and replayed from C.
//
// Reference role: the `tf.function` / Keras `train_function` of BaseModel.fit (merlin/models/tf/models/base.py:1361-1421,
// `make_train_function`): trace the step once, then run it without re-entering Python.  Here the "trace" is not a graph handed to a
// runtime: it is the literal list of HIP launches and event hand-offs the eager step issued, on the streams it issued them on.
// Why not a hipGraph: ROCm 7.2 replays a captured graph on one hardware queue (no overlap between its branches: measured in rounds
// 2-3), and per-stream graph segments pay ~10 us of launch overhead each plus their hand-offs.  What the recorder settled (round 4,
// profiles/r4_notes.md): replayed from C with nothing between the launches, the DLRM step takes 0.957-0.960 ms against 0.945-0.950
// for the same launches issued from Python -- the step is bound by its GPU critical path, not by its host.  The recorder stays as
// the host-free launch mode for hosts slower than the one measured (bench.py --launch recorded, graph.RecordedStep).
// Collectives of the C-ABI communicator (mh_comm.hip) are recorded and replayed like launches.
#include "mh_common.h"

#include <vector>

namespace {

struct Recording {
    std::vector<std::function<void()>> ops;
    std::vector<hipEvent_t> events;
    int launches = 0, hand_offs = 0;
};

Recording* g_rec = nullptr;  // the open recording (the host side is single-threaded while one is open)

}  // namespace

bool mh_recording() { return g_rec != nullptr; }

void mh_record_op(std::function<void()>&& op) {
    g_rec->ops.push_back(std::move(op));
    ++g_rec->launches;
}

extern "C" {

int32_t mh_record_begin(void) {
    MH_REQUIRE(!g_rec, "mh_record_begin: a recording is already open");
    g_rec = new Recording();
    return MH_OK;
}

int32_t mh_record_end(void** handle_out) {
    MH_REQUIRE(g_rec && handle_out, "mh_record_end: no open recording");
    *handle_out = g_rec;
    g_rec = nullptr;
    return MH_OK;
}

int32_t mh_record_abort(void) {  // an exception inside the recorded step: drop what was collected
    if (!g_rec) return MH_OK;
    for (hipEvent_t e : g_rec->events) (void)hipEventDestroy(e);
    delete g_rec;
    g_rec = nullptr;
    return MH_OK;
}

// An event recorded on `stream` NOW; while a recording is open the event belongs to it and the record is replayed.  Outside a
// recording the call is refused (events would leak): the host mirror uses its framework's events then.
int32_t mh_record_event(mh_stream_t stream, int64_t* event_id_out) {
    MH_REQUIRE(g_rec && event_id_out, "mh_record_event: needs an open recording");
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        mh_set_error("mh_record_event: hipEventCreate failed");
        return MH_ERR_LAUNCH;
    }
    g_rec->events.push_back(ev);
    *event_id_out = (int64_t)g_rec->events.size() - 1;
    hipStream_t s = mh_stream(stream);
    g_rec->ops.push_back([=]() { (void)hipEventRecord(ev, s); });
    ++g_rec->hand_offs;
    if (hipEventRecord(ev, s) != hipSuccess) {
        mh_set_error("mh_record_event: hipEventRecord failed");
        return MH_ERR_LAUNCH;
    }
    return MH_OK;
}

int32_t mh_record_wait_event(mh_stream_t stream, int64_t event_id) {
    MH_REQUIRE(g_rec && event_id >= 0 && event_id < (int64_t)g_rec->events.size(), "mh_record_wait_event: bad event / no open recording");
    hipEvent_t ev = g_rec->events[(size_t)event_id];
    hipStream_t s = mh_stream(stream);
    g_rec->ops.push_back([=]() { (void)hipStreamWaitEvent(s, ev, 0); });
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) {
        mh_set_error("mh_record_wait_event: hipStreamWaitEvent failed");
        return MH_ERR_LAUNCH;
    }
    return MH_OK;
}

int32_t mh_record_replay(void* handle) {
    MH_REQUIRE(handle && !g_rec, "mh_record_replay: bad handle, or a recording is open");
    const Recording* r = static_cast<const Recording*>(handle);
    for (const auto& op : r->ops) op();
    MH_CHECK_LAUNCH("mh_record_replay");
    return MH_OK;
}

int32_t mh_record_info(void* handle, int64_t* launches, int64_t* hand_offs) {
    MH_REQUIRE(handle, "mh_record_info: null handle");
    const Recording* r = static_cast<const Recording*>(handle);
    if (launches) *launches = r->launches;
    if (hand_offs) *hand_offs = r->hand_offs;
    return MH_OK;
}

int32_t mh_record_free(void* handle) {
    Recording* r = static_cast<Recording*>(handle);
    if (!r) return MH_OK;
    for (hipEvent_t e : r->events) (void)hipEventDestroy(e);
    delete r;
    return MH_OK;
}

}  // extern "C"
