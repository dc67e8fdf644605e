// side_stream.h — one lazily created helper stream + event pool per device, used by K3/K4 to run the large
// "far" GEMM updates concurrently with the latency-bound diagonal-block / in-block kernels of the next outer
// block (look-ahead). All work is fenced back into the caller's stream before the entry point returns, so the
// C ABI keeps its contract: the call is complete, in stream order, on the stream that was passed in.
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

namespace llmc {

struct SideStream {
    hipStream_t side = nullptr;
    hipEvent_t ev[8] = {};
    int next = 0;
    bool ok = false;

    hipEvent_t event() {
        hipEvent_t e = ev[next];
        next = (next + 1) & 7;
        return e;
    }
};

// One helper stream per (device, caller stream): concurrent entry points on different caller streams (the subsets of
// a block factorised side by side) must not share a helper, or each one's join would wait for the others' work.
// Returns nullptr when helper resources cannot be created (callers then run everything on the main stream).
inline SideStream* side_stream_for(hipStream_t main_st) {
    static std::map<std::pair<int, hipStream_t>, SideStream*> pool;   // `inline`: one pool per process
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, main_st);
    auto it = pool.find(key);
    if (it != pool.end()) return it->second->ok ? it->second : nullptr;
    if (pool.size() >= 64) return nullptr;                             // bounded: callers cycle through few streams
    SideStream* s = new SideStream();
    pool[key] = s;
    // lowest priority: the caller's stream carries the latency-bound critical path and must win the dispatcher
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    if (hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
    for (int i = 0; i < 8; ++i)
        if (hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    s->ok = true;
    return s;
}

// side waits for everything queued on main so far
inline int fork_to_side(SideStream* s, hipStream_t main_st) {
    hipEvent_t e = s->event();
    LLMC_HIP_CHECK(hipEventRecord(e, main_st));
    LLMC_HIP_CHECK(hipStreamWaitEvent(s->side, e, 0));
    return LLMC_OK;
}
// main waits for everything queued on side so far
inline int join_from_side(SideStream* s, hipStream_t main_st) {
    hipEvent_t e = s->event();
    LLMC_HIP_CHECK(hipEventRecord(e, s->side));
    LLMC_HIP_CHECK(hipStreamWaitEvent(main_st, e, 0));
    return LLMC_OK;
}

}  // namespace llmc
