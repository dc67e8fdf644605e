// pipe_streams.h — helper streams of the pipelined K3 / K4 schedules (round 4).
//
// K3 and K4 are a latency-bound CHAIN (diagonal-block factor / in-block column kernel, a few small kernels per step) plus
// throughput-bound BULK products (far updates, triangular-inverse levels) that depend on the chain only block by block.
// Three helper streams per caller stream:
//   fast   medium-sized work the chain needs within a block time (far panel solves, the next block's rows of a far update)
//   bulk   the large far updates
//   inv    the triangular-inverse levels (K3), run behind the factorisation instead of after it
// The helpers are created with a CU MASK (hipExtStreamCreateWithCUMask) that leaves two compute units of every XCD out, so
// the chain's small kernels on the caller's stream always find a free CU instead of queueing behind a far update's
// workgroups (round 3 measured a chain kernel waiting 515 us for a slot: profiles/r03_k3k4_timeline.txt). LLMC_SIDE_CU_MASK=0
// creates plain lowest-priority streams instead. Everything is fenced back into the caller's stream before the entry point
// returns: the C ABI contract (complete, in stream order, on the stream passed in) is unchanged.
#pragma once
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

namespace llmc {

struct PipeStreams {
    hipStream_t fast = nullptr, bulk = nullptr, inv = nullptr;
    hipStream_t chain = nullptr;   // stands in for the caller's stream when that is the NULL stream (see pipe_chain_stream)
    static constexpr int NEV = 512;
    hipEvent_t ev[NEV] = {};
    int next = 0;
    bool ok = false;

    // a fresh event recorded on `s`. The pool is a ring: an event is re-recorded NEV records later; every wait on it is
    // enqueued within a few outer blocks (< 64 records), and a wait captures the record that precedes it.
    int record(hipStream_t s, hipEvent_t* out) {
        hipEvent_t e = ev[next];
        next = (next + 1) % NEV;
        LLMC_HIP_CHECK(hipEventRecord(e, s));
        *out = e;
        return LLMC_OK;
    }
};

inline int pipe_wait(hipStream_t s, hipEvent_t e) {
    if (e) LLMC_HIP_CHECK(hipStreamWaitEvent(s, e, 0));
    return LLMC_OK;
}

int device_cu_count();

inline bool pipe_make_stream(hipStream_t* s, bool masked) {
    if (masked) {
        // bits 33k and 33k + 8 (k = 0..7) cleared: two CUs of every XCD whether the mask enumerates CUs XCD-interleaved
        // (bit % 8 = XCD) or XCD-major (bit / 32 = XCD)
        const int ncu = device_cu_count();
        const int words = (ncu + 31) / 32;
        uint32_t mask[32];
        for (int w = 0; w < 32; ++w) mask[w] = 0xffffffffu;
        if (ncu % 32) mask[words - 1] = (1u << (ncu % 32)) - 1u;
        for (int k = 0; k < 8; ++k)
            for (int o = 0; o <= 8; o += 8) {
                const int bit = 33 * k + o;
                if (bit < ncu) mask[bit >> 5] &= ~(1u << (bit & 31));
            }
        if (words <= 32 && hipExtStreamCreateWithCUMask(s, (uint32_t)words, mask) == hipSuccess) return true;
        (void)hipGetLastError();
    }
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, lo) == hipSuccess;
}

// One set per (device, caller stream), created lazily; nullptr when the resources cannot be created.
inline PipeStreams* pipe_streams_for(hipStream_t main_st) {
    static std::map<std::pair<int, hipStream_t>, PipeStreams*> pool;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(dev, main_st);
    auto it = pool.find(key);
    if (it != pool.end()) return it->second->ok ? it->second : nullptr;
    if (pool.size() >= 32) return nullptr;
    PipeStreams* p = new PipeStreams();
    pool[key] = p;
    const char* e = getenv("LLMC_SIDE_CU_MASK");
    const bool masked = !(e && e[0] == '0');
    if (!pipe_make_stream(&p->fast, masked) || !pipe_make_stream(&p->bulk, masked) || !pipe_make_stream(&p->inv, masked))
        return nullptr;
    // NORMAL priority: a high-priority chain stream made everything slower (K3 32.1 ms against 21.9 with the chain on the
    // caller's own normal-priority stream, gpurun_out/r04c/k3_time_q8.txt) — the far updates' waves are evicted for every
    // small chain kernel
    if (hipStreamCreateWithFlags(&p->chain, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (int i = 0; i < PipeStreams::NEV; ++i)
        if (hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    p->ok = true;
    return p;
}

// The stream the chain runs on. CU-masked streams are created without hipStreamNonBlocking (the API has no flags), i.e.
// they synchronise implicitly with the NULL stream: every launch on the NULL stream would wait for the helpers' queued
// work and the pipeline would collapse into issue order. A caller on the NULL stream (PyTorch's default stream) therefore
// gets the chain on an internal non-blocking stream, fenced in at the start and back out at the end by events.
inline hipStream_t pipe_chain_stream(PipeStreams* ps, hipStream_t st) { return st == nullptr ? ps->chain : st; }

}  // namespace llmc
