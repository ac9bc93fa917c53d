// capi_host.hip -- the chunked host-queue pipeline (part of the C-ABI of include/opencorr_hip.h; split from capi.hip in round 6, same exported symbols)
#include "capi_internal.h"

namespace ochip_capi {


// ---------------------------------------------------------------------------
// host queues: H2D of the AoS, kernels, D2H (the reference's CUDA module does the same,
// examples/test_2d_dic_gpu_icgn.cpp:136-149) -- but chunk by chunk, so that the copies of one chunk overlap the
// kernels of its neighbours:   H2D(0) K(0) | H2D(1) K(1) D2H(0) | H2D(2) K(2) D2H(1) | ... | D2H(last)
// Kernels run on the engine's stream, copies out on a second stream behind a per-chunk event and from a second host
// thread (copies to / from pageable memory block their thread; PCIe is full duplex); copies in are issued on the
// engine's stream ahead of their kernels.  A POI's result does not depend on the chunk it travels in (tests: split
// queue == whole queue), chunks are large enough for the ICGN2D tile schedule.
// ---------------------------------------------------------------------------
// `chain`: further engines that process the same records right after `e` (oc_hip_compute_chain: FFTCC then ICGN, say) --
// per chunk ONE copy in, every engine's kernels in order on e's stream, ONE copy out, instead of a round trip per engine.
// The callers have moved the chained engines onto e's stream for the duration of the call.
int compute_host(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes,
                 oc_hip_engine* const* chain, int n_chain) {
    const int stride_f = (int)(stride_bytes / 4);
    auto run_all = [&](float* d_pois, size_t n, const float* d_off) -> int {
        OC_TRY(run_compute_device(e, d_pois, stride_f, n, d_off));
        for (int i = 0; i < n_chain; i++) OC_TRY(run_compute_device(chain[i], d_pois, stride_f, n, nullptr));
        return OC_HIP_OK;
    };
    const size_t bytes = count * stride_bytes;
    OC_TRY(e->poi_stage.reserve(bytes));
    if (offsets) OC_TRY(e->off_stage.reserve(count * 2 * sizeof(float)));
    // Chunk schedule.  What a pipeline cannot hide is the copy-in of its FIRST chunk and the copy-out of its LAST one, and
    // every extra chunk costs a launch tail (the ICGN kernels' last workgroups run on a half-empty chip) plus an
    // inter-stream hand-over.  So: small chunks at both ends (half of "host_chunk"), few large ones (three times
    // "host_chunk") in between, whose copies hide behind the neighbours' kernels.  Measured on config B (250 000 POIs,
    // chain of FFTCC2D + ICGN2D1): uniform chunks of 65 536: 4.76 ms, one piece: 5.36 ms, this schedule: see DESIGN 4.4.
    std::vector<std::pair<size_t, size_t>> sched;  // (first POI, POIs)
    {
        const size_t unit = e->host_chunk > 0 ? (size_t)e->host_chunk : count;
        // A lone FFTCC engine (the first call of the reference's unmodified `fftcc->compute(q); icgn->compute(q);`) is
        // TRANSFER bound: 0.44 ms of kernel between 0.5 ms in and 0.5 ms out on config B.  There the copies of the two
        // directions should overlap each other (PCIe is full duplex): uniform chunks of half a unit -- with the edge / middle
        // schedule below the big middle chunk's copy-in, kernel and copy-out run one after the other (round 6: two-call
        // sequence 5.0 -> see DESIGN 4.6).
        const bool transfer_bound = n_chain == 0 && (e->kind == OC_HIP_FFTCC2D || e->kind == OC_HIP_FFTCC3D);
        if (transfer_bound && e->host_chunk > 0 && count >= unit) {
            const size_t piece = std::max<size_t>(unit / 2, 1), np = (count + piece - 1) / piece;
            size_t at = 0;
            for (size_t i = 0; i < np; i++) {
                const size_t n = count / np + (i < count % np ? 1 : 0);
                sched.emplace_back(at, n);
                at += n;
            }
        } else if (count < 2 * unit) {
            sched.emplace_back(0, count);  // not worth a pipeline
        } else {
            const size_t edge = std::max<size_t>(unit / 2, 1), mid_max = 3 * unit;
            sched.emplace_back(0, edge);
            size_t at = edge;
            const size_t mid_total = count - 2 * edge;
            const size_t nmid = (mid_total + mid_max - 1) / mid_max;
            for (size_t i = 0; i < nmid; i++) {
                const size_t n = mid_total / nmid + (i < mid_total % nmid ? 1 : 0);
                sched.emplace_back(at, n);
                at += n;
            }
            sched.emplace_back(at, count - at);
        }
    }
    const size_t nchunk = sched.size();
    if (nchunk > 1) {
        if (!e->copy_stream) OC_HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        if (!e->copy_in_stream) OC_HIP_TRY(hipStreamCreateWithFlags(&e->copy_in_stream, hipStreamNonBlocking));
        // (two loops: a failed creation must not leave the lists at different lengths for the next call)
        while (e->chunk_done.size() < nchunk) {
            hipEvent_t ev = nullptr;
            OC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->chunk_done.push_back(ev);
        }
        while (e->chunk_in.size() < nchunk) {
            hipEvent_t ev = nullptr;
            OC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->chunk_in.push_back(ev);
        }
    }
    char* stage = e->poi_stage.as<char>();
    if (nchunk == 1) {
        OC_HIP_TRY(hipMemcpyAsync(stage, pois, bytes, hipMemcpyHostToDevice, e->stream));
        const float* d_off = nullptr;
        if (offsets) {
            OC_HIP_TRY(hipMemcpyAsync(e->off_stage.p, offsets, count * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
            d_off = e->off_stage.as<float>();
        }
        OC_TRY(run_all(reinterpret_cast<float*>(stage), count, d_off));
        OC_HIP_TRY(hipMemcpyAsync(pois, stage, bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
        return OC_HIP_OK;
    }
    // A copy between pageable host memory and the device blocks the calling thread, so the two directions get a thread
    // each (PCIe is full duplex): this thread feeds chunks in and launches their kernels, the helper waits for each
    // chunk's event and copies its records back.
    const int device = e->device;
    hipError_t out_err = hipSuccess;
    {
        std::lock_guard<std::mutex> hand(e->feed_mu);
        e->chunks_fed = 0;
    }
    auto hand_over = [&](size_t fed) {
        {
            std::lock_guard<std::mutex> hand(e->feed_mu);
            e->chunks_fed = fed;
        }
        e->feed_cv.notify_one();
    };
    auto copy_out = [&] {
        if (hipSetDevice(device) != hipSuccess) {
            out_err = hipErrorInvalidDevice;
            return;
        }
        for (size_t c = 0; c < nchunk && out_err == hipSuccess; c++) {
            const size_t first = sched[c].first, n = sched[c].second;
            // the event is recorded by the feeding thread after chunk c's kernels were enqueued; until then
            // hipStreamWaitEvent would see the event of an earlier call, so sleep until the hand-off
            {
                std::unique_lock<std::mutex> hand(e->feed_mu);
                e->feed_cv.wait(hand, [&] { return e->chunks_fed > c; });
                if (e->chunks_fed == (size_t)-1) break;  // the feeder failed: drain what was issued and stop
            }
            out_err = hipStreamWaitEvent(e->copy_stream, e->chunk_done[c], 0);
            if (out_err == hipSuccess)
                out_err = hipMemcpyAsync(pois + first * stride_bytes, stage + first * stride_bytes, n * stride_bytes, hipMemcpyDeviceToHost,
                                         e->copy_stream);
        }
        if (out_err == hipSuccess) out_err = hipStreamSynchronize(e->copy_stream);
    };
    // std::thread's constructor throws std::system_error when the process is out of threads; nothing may unwind through
    // the extern "C" boundary, so the helper is optional: without it this thread copies out after feeding
    std::thread out_thread;
    bool helper = true;
    try {
        out_thread = std::thread(copy_out);
    } catch (...) {
        helper = false;
    }
    int rc = OC_HIP_OK;
    auto feed = [&]() -> int {
        // the staging buffer may still be read by kernels of an earlier call on the engine's stream
        OC_HIP_TRY(hipEventRecord(e->chunk_in[0], e->stream));
        OC_HIP_TRY(hipStreamWaitEvent(e->copy_in_stream, e->chunk_in[0], 0));
        for (size_t c = 0; c < nchunk; c++) {
            const size_t first = sched[c].first, n = sched[c].second;
            // copies in travel on their own stream: on the kernels' stream a pageable copy would queue behind the
            // previous chunk's kernels and nothing would overlap
            OC_HIP_TRY(hipMemcpyAsync(stage + first * stride_bytes, pois + first * stride_bytes, n * stride_bytes, hipMemcpyHostToDevice,
                                      e->copy_in_stream));
            const float* d_off = nullptr;
            if (offsets) {
                float* o = e->off_stage.as<float>() + 2 * first;
                OC_HIP_TRY(hipMemcpyAsync(o, offsets + 2 * first, n * 2 * sizeof(float), hipMemcpyHostToDevice, e->copy_in_stream));
                d_off = o;
            }
            OC_HIP_TRY(hipEventRecord(e->chunk_in[c], e->copy_in_stream));
            OC_HIP_TRY(hipStreamWaitEvent(e->stream, e->chunk_in[c], 0));
            OC_TRY(run_all(reinterpret_cast<float*>(stage + first * stride_bytes), n, d_off));
            OC_HIP_TRY(hipEventRecord(e->chunk_done[c], e->stream));
            hand_over(c + 1);
        }
        return OC_HIP_OK;
    };
    rc = feed();
    const std::string feed_error = g_last_error;
    if (rc != OC_HIP_OK) hand_over((size_t)-1);
    if (helper) out_thread.join();
    else copy_out();
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    if (rc != OC_HIP_OK) return fail(rc, "%s", feed_error.c_str());
    if (out_err != hipSuccess) return fail(OC_HIP_ERR_HIP, "copying results back failed: %s", hipGetErrorString(out_err));
    return OC_HIP_OK;
}


}  // namespace ochip_capi
