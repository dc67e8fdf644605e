// pipe_streams.h — helper streams of the pipelined K4 schedule (round 4).
//
// K4 is a latency-bound CHAIN (the in-block column kernel and the update of its group's columns, per 128-column block) plus a
// throughput-bound BULK product (the far update of everything beyond the next group) that depends on the chain only group
// by group. The bulk product runs on a helper stream of its own, the columns the chain needs next first.
// Measured on MI355X (gpurun_out/r04c, r04e; profiles/r04_stream_experiments.txt):
//   * plain non-blocking helper streams: K4 of down_proj 12.1 -> 10.9-11.4 ms; the same structure for K3 (three helper
//     streams, the inverse behind the factorisation) measured EQUAL to the single-stream schedule (21.9 ms) and was removed:
//     K3's "latency-bound" steps are wide, inefficient kernels that already occupy every CU;
//   * CU-MASKED helpers (hipExtStreamCreateWithCUMask leaving two CUs per XCD to the chain; LLMC_SIDE_CU_MASK=1): the mask
//     works (240 of 256 CUs used; a small kernel beside a saturating one starts in 10 us instead of 21 us) but buys nothing
//     end to end, and masked streams are BLOCKING streams — their mere existence slows every launch on the NULL stream
//     (PyTorch's default stream): K3 called there went from 21 to 32 ms. Off by default;
//   * a HIGH-priority chain stream makes everything slower (21.9 -> 32.1 ms): the bulk product's waves are evicted for
//     every small chain kernel.
// Everything is fenced back into the caller's stream before the entry point returns: the C ABI contract (complete, in stream
// order, on the stream passed in) is unchanged.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
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
    std::atomic<unsigned> next{0};      // two host threads driving the same caller stream must not be handed the same event
    bool ok = false;
    bool masked = false;

    // a fresh event recorded on `s`. The pool is a ring: an event is re-recorded NEV records later; every wait on it is
    // enqueued within a few outer blocks (< 64 records per call: llmc_gptq_quantize records 3 per 512-column group and waits on
    // each within three groups), and a wait captures the record that precedes it.
    int record(hipStream_t s, hipEvent_t* out) {
        hipEvent_t e = ev[next.fetch_add(1u, std::memory_order_relaxed) % NEV];
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
    if (pool.size() >= 32) {
        // bounded: callers cycle through few streams. Past the bound the entry points run their single-stream schedule (same
        // results) — said once, not silently
        static bool told = false;
        if (!told) {
            told = true;
            fprintf(stderr, "[llmc_hip] helper-stream pool is full (32 caller streams): further streams run the single-stream schedule\n");
        }
        return nullptr;
    }
    PipeStreams* p = new PipeStreams();
    pool[key] = p;             // a failed set stays registered with ok = false (no retry storm), its partial resources released below
    const bool masked = opt(OPT_SIDE_CU_MASK) != 0;
    p->masked = masked;
    auto fail = [&]() -> PipeStreams* {
        for (hipStream_t* s : {&p->fast, &p->bulk, &p->inv, &p->chain})
            if (*s) { (void)hipStreamDestroy(*s); *s = nullptr; }
        for (int i = 0; i < PipeStreams::NEV; ++i)
            if (p->ev[i]) { (void)hipEventDestroy(p->ev[i]); p->ev[i] = nullptr; }
        (void)hipGetLastError();
        return nullptr;
    };
    if (!pipe_make_stream(&p->fast, masked) || !pipe_make_stream(&p->bulk, masked) || !pipe_make_stream(&p->inv, masked))
        return fail();
    // NORMAL priority: a high-priority chain stream made everything slower (K3 32.1 ms against 21.9 with the chain on the
    // caller's own normal-priority stream, gpurun_out/r04c/k3_time_q8.txt) — the far updates' waves are evicted for every
    // small chain kernel
    if (hipStreamCreateWithFlags(&p->chain, hipStreamNonBlocking) != hipSuccess) return fail();
    for (int i = 0; i < PipeStreams::NEV; ++i)
        if (hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming) != hipSuccess) return fail();
    p->ok = true;
    return p;
}

// The stream the chain runs on: the caller's. Only with CU-masked helpers (blocking streams, see above) a caller on the NULL
// stream gets the chain on an internal non-blocking stream, fenced in at the start and back out at the end by events —
// otherwise every chain launch would wait for the helpers' queued work and the pipeline would collapse into issue order.
inline hipStream_t pipe_chain_stream(PipeStreams* ps, hipStream_t st) { return (st == nullptr && ps->masked) ? ps->chain : st; }

}  // namespace llmc
