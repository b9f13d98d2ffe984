// extern "C" surface of libwhenet_hip.so (include/whenet_hip.h).  Every entry point catches
// everything: no exception crosses the ABI.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <exception>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "engine.h"
#include "engine_internal.h"

// A handle = one primary engine plus, with option "inflight" > 1, replica engines (own streams,
// activation arena, hipGraphs and a copy of the 8.6-17 MB device weights).  A forward is a chain of
// 51 dependent launches -- about half of a batch-64 forward is launch latency -- so independent
// forwards are spread round-robin over the engines and overlap on the GPU.
constexpr int MAX_INFLIGHT_ENGINES = 4;
struct whenet_ctx {
    whenet::Engine* engine = nullptr;                            // primary: op_*, profile, blocking forward
    std::vector<whenet::Engine*> replicas;                       // engines 1..inflight-1
    std::vector<char> snapshot;
    int device_id = 0, dtype = WHENET_F32;
    int inflight = 1;
    size_t next = 0;                                             // round-robin cursor
    // one large BLOCKING call (whenet_forward_u8 with n >= fanout_min) is cut into fanout_chunk-crop forwards spread over the
    // engines through their pinned-slot pipelines: copies of chunk i+1 overlap the forward of chunk i, results are bitwise
    // those of one forward (the kernels are batch-invariant).  fanout_min = 0 switches it off.
    int fanout_min = 256, fanout_chunk = 128, fanout_stage = 2, fanout_depth = 2;
    // fanout_stage 2 (round 6, default): the caller's array is registered with the runtime for the duration of the call (2 us when the
    // pages are resident, ~0.2 ms for a fresh 77 MB array; profiles/r06/hostreg_probe.txt), so every chunk's H2D copy is asynchronous:
    // ONE host thread enqueues everything, and chunk c's copy starts when chunk c-1's is done (stream-ordered across the engines) --
    // the chunks arrive in order at the link's full rate instead of two engines' copies sharing it (chunk 0 after 0.17 ms, not 0.68).
    // 0 / 1: round 5's forms (pinned staging / the runtime's pageable path, one host thread per engine).
    // -1 = calibrate between 0 and 1: how fast the runtime moves PAGEABLE memory differs from box to box (round 5: the direct form read
    // 126 k / 90 k / 130 k crops/s on three boxes where pinned staging read 110 / 110 / 107 k).  Calls 0 and 1 run one form each UNTIMED
    // (slots, arena growth, graph captures and lane streams are one-time costs of whichever form runs first), calls 2..5 alternate
    // the two forms timed, the best rate of each is kept and the faster form serves every later call.
    int fanout_calib_calls = 0;
    double fanout_calib_rate[2] = {0.0, 0.0};
    int fanout_chosen = -1;                                      // the form the last fan-out call ran (whenet_last_error-free diagnosis: option "fanout_stage" read back)
    // engines of the fan-out beyond primary + replicas: created on the first large call (option "fanout_engines", default 2: measured
    // 133 k crops/s at N = 512 f16 in every process, against 115-141 k from process to process with 3 and 123-128 k with 4), private
    // to it -- "inflight", the round-robin cursor and the chains per forward of everything else are not touched (round 6: the class used
    // to set inflight = 2 behind the caller's back)
    std::vector<whenet::Engine*> fan;
    int fanout_engines = 2;
    std::vector<std::pair<std::string, long>> options;           // replayed on new replicas
    whenet::Engine& at(size_t i) { return i == 0 ? *engine : *replicas[i - 1]; }
    // engine i of a fan-out over `count` engines: primary, replicas, then the private ones
    whenet::Engine& fan_at(size_t i) {
        if (i == 0) return *engine;
        if (i - 1 < replicas.size()) return *replicas[i - 1];
        return *fan[i - 1 - replicas.size()];
    }
    int fan_count() {
        const int want = std::max(inflight, std::min(fanout_engines, 4));
        while (1 + int(replicas.size()) + int(fan.size()) < want) {
            std::unique_ptr<whenet::Engine> r(new whenet::Engine(snapshot.data(), snapshot.size(), device_id, dtype));
            for (const auto& kv : options) r->set_option(kv.first, kv.second);
            fan.push_back(r.release());
        }
        return want;
    }
    whenet::Engine& take() {
        whenet::Engine& e = at(next % size_t(inflight));
        next = (next + 1) % size_t(inflight);
        return e;
    }
};

namespace {

thread_local std::string g_create_error;

// Page ranges this library has registered with the runtime for the duration of a fan-out call (fanout_stage 2).  Two handles driven by
// two threads may be handed adjacent slices of ONE array (whenet_hip/multi.py): the slices share a page, and registering / unregistering
// it from both sides races inside the runtime (segfaults in 3 of 8 runs of tests/test_multi_device.py).  A call whose pages overlap a
// range already held falls back to the runtime's pageable path.
struct HostRegistry {
    std::mutex mu;
    std::vector<std::pair<uintptr_t, uintptr_t>> held;
    bool acquire(const void* p, size_t nbytes, hipError_t* err) {
        const uintptr_t page = 4096, lo = reinterpret_cast<uintptr_t>(p) & ~(page - 1),
                        hi = (reinterpret_cast<uintptr_t>(p) + nbytes + page - 1) & ~(page - 1);
        std::lock_guard<std::mutex> lock(mu);
        for (const auto& r : held)
            if (lo < r.second && r.first < hi) {
                *err = hipErrorUnknown;
                return false;
            }
        *err = hipHostRegister(const_cast<void*>(p), nbytes, hipHostRegisterDefault);
        if (*err != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        held.emplace_back(lo, hi);
        return true;
    }
    void release(const void* p, size_t nbytes) {
        const uintptr_t page = 4096, lo = reinterpret_cast<uintptr_t>(p) & ~(page - 1),
                        hi = (reinterpret_cast<uintptr_t>(p) + nbytes + page - 1) & ~(page - 1);
        std::lock_guard<std::mutex> lock(mu);
        (void)hipHostUnregister(const_cast<void*>(p));
        for (size_t i = 0; i < held.size(); ++i)
            if (held[i].first == lo && held[i].second == hi) {
                held.erase(held.begin() + long(i));
                break;
            }
    }
};
HostRegistry g_host_registry;

template <typename F>
int guarded(whenet_t* h, F&& fn) {
    if (h == nullptr || h->engine == nullptr) return WHENET_EINVAL;
    try {
        fn(*h->engine);
        return WHENET_OK;
    } catch (const whenet::Error& e) {
        h->engine->last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        h->engine->last_error = "out of host memory";
        return WHENET_ENOMEM;
    } catch (const std::exception& e) {
        h->engine->last_error = e.what();
        return WHENET_EHIP;
    } catch (...) {
        h->engine->last_error = "unknown error";
        return WHENET_EHIP;
    }
}

int create_impl(const void* blob, size_t nbytes, int device_id, int dtype, whenet_t** out) {
    if (out == nullptr) return WHENET_EINVAL;
    *out = nullptr;
    try {
        std::unique_ptr<whenet::Engine> e(new whenet::Engine(blob, nbytes, device_id, dtype));
        whenet_t* h = new whenet_t;
        h->engine = e.release();
        h->snapshot.assign(static_cast<const char*>(blob), static_cast<const char*>(blob) + nbytes);
        h->device_id = device_id;
        h->dtype = dtype;
        *out = h;
        return WHENET_OK;
    } catch (const whenet::Error& e) {
        g_create_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_create_error = "out of host memory";
        return WHENET_ENOMEM;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return WHENET_EHIP;
    } catch (...) {
        g_create_error = "unknown error";
        return WHENET_EHIP;
    }
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int whenet_create(const char* snapshot_path, int device_id, int dtype, whenet_t** out) {
    if (snapshot_path == nullptr || out == nullptr) {
        g_create_error = "whenet_create: NULL argument";
        return WHENET_EINVAL;
    }
    std::vector<char> blob;
    try {
        std::ifstream f(snapshot_path, std::ios::binary | std::ios::ate);
        if (!f) {
            g_create_error = std::string("cannot open snapshot '") + snapshot_path + "'";
            return WHENET_ENOENT;
        }
        const std::streamoff sz = f.tellg();
        // a directory or an unseekable path reports -1 (or nonsense): an I/O error, not an allocation
        if (sz < 0 || !f.seekg(0) || uint64_t(sz) > (uint64_t(1) << 32)) {
            g_create_error = std::string("snapshot '") + snapshot_path + "' is not a readable regular file";
            return WHENET_EIO;
        }
        blob.resize(size_t(sz));
        if (sz > 0 && !f.read(blob.data(), std::streamsize(sz))) {
            g_create_error = std::string("cannot read snapshot '") + snapshot_path + "'";
            return WHENET_EIO;
        }
    } catch (const std::bad_alloc&) {
        g_create_error = "out of host memory";
        return WHENET_ENOMEM;
    } catch (const std::exception& e) {
        g_create_error = std::string("cannot read snapshot '") + snapshot_path + "': " + e.what();
        return WHENET_EIO;
    } catch (...) {
        g_create_error = std::string("cannot read snapshot '") + snapshot_path + "'";
        return WHENET_EIO;
    }
    return create_impl(blob.data(), blob.size(), device_id, dtype, out);
}

int whenet_create_from_memory(const void* snapshot, size_t nbytes, int device_id, int dtype, whenet_t** out) {
    if (snapshot == nullptr || out == nullptr) {
        g_create_error = "whenet_create_from_memory: NULL argument";
        return WHENET_EINVAL;
    }
    return create_impl(snapshot, nbytes, device_id, dtype, out);
}

int whenet_create_postproc(int device_id, whenet_t** out) {
    if (out == nullptr) {
        g_create_error = "whenet_create_postproc: NULL argument";
        return WHENET_EINVAL;
    }
    *out = nullptr;
    try {
        std::unique_ptr<whenet::Engine> e(new whenet::Engine(device_id));
        whenet_t* h = new whenet_t;
        h->engine = e.release();
        h->device_id = device_id;
        *out = h;
        return WHENET_OK;
    } catch (const whenet::Error& e) {
        g_create_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_create_error = "out of host memory";
        return WHENET_ENOMEM;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return WHENET_EHIP;
    } catch (...) {
        g_create_error = "unknown error";
        return WHENET_EHIP;
    }
}

void whenet_destroy(whenet_t* h) {
    if (h == nullptr) return;
    try {
        for (whenet::Engine* e : h->replicas) delete e;
        for (whenet::Engine* e : h->fan) delete e;
        delete h->engine;
    } catch (...) {
    }
    delete h;
}

const char* whenet_last_error(const whenet_t* h) {
    if (h == nullptr || h->engine == nullptr) return g_create_error.c_str();
    return h->engine->last_error.c_str();
}

int whenet_get_info(const whenet_t* h, whenet_info_t* out) {
    if (h == nullptr || h->engine == nullptr || out == nullptr) return WHENET_EINVAL;
    h->engine->get_info(out);
    return WHENET_OK;
}

int whenet_set_option(whenet_t* h, const char* key, long value) {
    if (key == nullptr) return WHENET_EINVAL;
    return guarded(h, [&](whenet::Engine& e) {
        const std::string k = key;
        if (k == "inflight") {
            WHENET_REQUIRE(value >= 1 && value <= MAX_INFLIGHT_ENGINES, WHENET_EINVAL, "inflight must be 1..4");
            WHENET_REQUIRE(!h->snapshot.empty() || value == 1, WHENET_EINVAL, "inflight: this handle has no network");
            e.release_aux_streams();
            for (whenet::Engine* r : h->replicas) r->release_aux_streams();
            for (whenet::Engine* r : h->fan) delete r;            // (rebuilt on the next large call)
            h->fan.clear();
            while (int(h->replicas.size()) + 1 > value) {
                delete h->replicas.back();
                h->replicas.pop_back();
            }
            while (int(h->replicas.size()) + 1 < value) {
                std::unique_ptr<whenet::Engine> r(
                    new whenet::Engine(h->snapshot.data(), h->snapshot.size(), h->device_id, h->dtype));
                for (const auto& kv : h->options) r->set_option(kv.first, kv.second);
                h->replicas.push_back(r.release());
            }
            h->inflight = int(value);
            h->next = 0;
            // several forwards in flight: each runs as ONE chain, the concurrency comes from the others (a blocking host
            // forward keeps its own chains: Engine::host_lanes_)
            const long lanes = value > 1 ? 1 : 2;
            e.set_option("device_lanes", lanes);
            for (whenet::Engine* r : h->replicas) r->set_option("device_lanes", lanes);
            // several forwards share the chip: the XCD-grouped placement of the fused kernels pays at every batch (engine.h xcd_grouped)
            e.set_option("concurrent", value > 1);
            for (whenet::Engine* r : h->replicas) r->set_option("concurrent", value > 1);
            return;
        }
        if (k == "fanout_min" || k == "fanout_chunk" || k == "fanout_stage" || k == "fanout_depth" || k == "fanout_engines") {
            if (k == "fanout_engines") {
                WHENET_REQUIRE(value >= 1 && value <= MAX_INFLIGHT_ENGINES, WHENET_EINVAL, "fanout_engines must be 1..4");
                h->fanout_engines = int(value);
                for (whenet::Engine* r : h->fan) delete r;
                h->fan.clear();
                return;
            }
            if (k == "fanout_min") {
                WHENET_REQUIRE(value >= 0, WHENET_EINVAL, "fanout_min must be >= 0 (0 = never)");
                h->fanout_min = int(std::min<long>(value, 1 << 30));
            } else if (k == "fanout_chunk") {
                WHENET_REQUIRE(value >= 1 && value <= 4096, WHENET_EINVAL, "fanout_chunk must be 1..4096");
                h->fanout_chunk = int(value);
            } else if (k == "fanout_stage") {
                WHENET_REQUIRE(value >= -1 && value <= 3, WHENET_EINVAL,
                               "fanout_stage must be -1 (calibrate 0 against 1), 0 (pinned staging), 1 (the runtime's pageable path) or 2 (registered, default)");
                h->fanout_stage = int(value);
                h->fanout_calib_calls = 0;
                h->fanout_calib_rate[0] = h->fanout_calib_rate[1] = 0.0;
            } else {
                WHENET_REQUIRE(value >= 1 && value <= WHENET_MAX_INFLIGHT, WHENET_EINVAL, "fanout_depth must be 1..4");
                h->fanout_depth = int(value);
            }
            return;
        }
        e.set_option(k, value);
        for (whenet::Engine* r : h->replicas) r->set_option(k, value);
        for (whenet::Engine* r : h->fan) r->set_option(k, value);
        h->options.emplace_back(k, value);
    });
}

int whenet_forward_u8(whenet_t* h, const uint8_t* crops, int n, float* ypr, int32_t* argmax, float* logits) {
    return guarded(h, [&](whenet::Engine& e) {
        bool pending = e.has_pending();                 // the caller's own submissions hold slots: leave them alone
        for (whenet::Engine* r : h->replicas) pending = pending || r->has_pending();
        if (h->fanout_min <= 0 || n < h->fanout_min || n <= h->fanout_chunk || pending || h->snapshot.empty()) {
            e.forward_host(crops, n, ypr, argmax, logits);
            return;
        }
        WHENET_REQUIRE(crops != nullptr && ypr != nullptr, WHENET_EINVAL, "crops and ypr must not be NULL");
        // get_angle(np.uint8[N,...]) with a large N (whenet.py:22-27 takes any N): chunk c goes to engine c % E of the fan-out, at most
        // fanout_depth outstanding per engine, results land in the caller's arrays at the chunk's offset.  Round 6 (fanout_stage 2): the
        // array is registered for the call, so this ONE thread enqueues everything and the copies run in order on one stream.  Round 5's
        // forms (stages 0 / 1, further down): every engine is driven by its OWN host thread (the calling thread takes engine 0) --
        // staging a chunk is a 9.6 MB memcpy into pinned memory, ~0.6 ms on one core, more than the 0.42 ms the GPU needs for it, and a
        // copy from pageable memory blocks its caller.  Engines share nothing but the device.
        const int chunk = h->fanout_chunk;
        int stage = h->fanout_stage;
        const bool calibrating = stage < 0;
        if (calibrating)
            stage = h->fanout_calib_calls < 6 ? 1 - (h->fanout_calib_calls & 1) : (h->fanout_calib_rate[1] >= h->fanout_calib_rate[0] ? 1 : 0);
        const auto t_start = std::chrono::steady_clock::now();
        // one engine: the chunk itself supplies the concurrency (two chains); several engines: one chain each
        const int nfan = h->fan_count();
        const int lanes = nfan > 1 ? 1 : 2;
        if (stage == 2 || stage == 3) {
            const bool shared_copy_stream = stage == 2;           // 3 (probe): every engine's own copy stream, chained by events
            whenet::detail::DeviceGuard guard(h->device_id);
            const size_t nbytes = size_t(n) * 150528;
            hipError_t re = hipSuccess;
            const bool registered = g_host_registry.acquire(crops, nbytes, &re);
            if (registered || re == hipErrorHostMemoryAlreadyRegistered) {       // (already registered: the caller pinned it)
                h->fanout_chosen = stage;
                struct Pending { int eng, ticket, off; };
                std::vector<Pending> q;
                size_t head = 0;
                std::vector<int> outstanding(size_t(nfan), 0);
                // every chunk's copy goes through ONE stream (the primary engine's copy stream): in order at the link's full rate, and one
                // stream fewer per engine for the runtime to place on its four hardware queues (with a copy stream per engine the same
                // call read 102 k or 134 k crops/s from process to process: profiles/r06/host_path_probe.txt)
                const hipStream_t copy_on = shared_copy_stream ? h->engine->copy_stream_handle() : nullptr;
                hipEvent_t prev = nullptr;
                auto collect_head = [&] {
                    const Pending& p = q[head++];
                    const size_t o = size_t(p.off);
                    h->fan_at(size_t(p.eng)).collect(p.ticket, ypr + o * 3, argmax ? argmax + o * 3 : nullptr, logits ? logits + o * 252 : nullptr);
                    --outstanding[size_t(p.eng)];
                };
                std::exception_ptr err;
                try {
                    int off = 0, c = 0;
                    while (off < n) {
                        // the first two chunks are half-size: the GPU starts after half a chunk's copy
                        const int want = c < 2 && chunk >= 32 ? chunk / 2 : chunk;
                        const int cnt = std::min(want, n - off), ei = c % nfan;
                        while (outstanding[size_t(ei)] >= h->fanout_depth) collect_head();     // (in submission order: engine ei's oldest is reached)
                        whenet::Engine& eng = h->fan_at(size_t(ei));
                        const int ticket = eng.submit(crops + size_t(off) * 150528, cnt, 1, lanes, copy_on, shared_copy_stream ? nullptr : prev);
                        if (!shared_copy_stream) prev = eng.copied_event(ticket);
                        q.push_back(Pending{ei, ticket, off});
                        ++outstanding[size_t(ei)];
                        off += cnt;
                        ++c;
                    }
                    while (head < q.size()) collect_head();
                } catch (...) {
                    err = std::current_exception();
                    for (int i = 0; i < nfan; ++i) {
                        try { h->fan_at(size_t(i)).abandon_submissions(); } catch (...) {}
                    }
                }
                if (registered) g_host_registry.release(crops, nbytes);
                if (err) std::rethrow_exception(err);
                return;
            }
            stage = 1;                                           // the runtime would not register the array: its pageable path
        }
        h->fanout_chosen = stage;
        const int nthreads = nfan;
        const int nchunks = (n + chunk - 1) / chunk;
        std::vector<std::exception_ptr> errs;
        errs.resize(static_cast<size_t>(nthreads));
        auto drive = [&](int t) {
            whenet::Engine& eng = h->fan_at(size_t(t));
            struct Pending { int ticket, off; };
            std::vector<Pending> q;
            size_t head = 0;
            auto collect_one = [&] {
                const Pending& p = q[head++];
                const size_t o = size_t(p.off);
                eng.collect(p.ticket, ypr + o * 3, argmax ? argmax + o * 3 : nullptr, logits ? logits + o * 252 : nullptr);
            };
            try {
                for (int c = t; c < nchunks; c += nthreads) {
                    if (q.size() - head >= size_t(h->fanout_depth)) collect_one();
                    const int off = c * chunk, cnt = std::min(chunk, n - off);
                    q.push_back(Pending{eng.submit(crops + size_t(off) * 150528, cnt, stage, lanes), off});
                }
                while (head < q.size()) collect_one();
            } catch (...) {
                errs[size_t(t)] = std::current_exception();
                try {                                            // (on a worker thread nothing may escape: std::terminate)
                    eng.abandon_submissions();
                } catch (...) {
                }
            }
        };
        std::vector<std::thread> workers;
        workers.reserve(size_t(nthreads));
        int started = 1;                                         // (engine 0 is the calling thread's)
        try {
            for (int t = 1; t < nthreads; ++t, ++started) workers.emplace_back(drive, t);
        } catch (...) {                                          // no thread to be had: the engines without one are driven from here
        }
        drive(0);
        for (int t = started; t < nthreads; ++t) drive(t);
        for (std::thread& w : workers) w.join();
        for (const std::exception_ptr& ep : errs)
            if (ep) std::rethrow_exception(ep);
        if (calibrating && h->fanout_calib_calls < 6) {
            if (h->fanout_calib_calls >= 2) {                     // (calls 0 and 1 warm both forms up, untimed)
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                h->fanout_calib_rate[stage] = std::max(h->fanout_calib_rate[stage], double(n) / (sec > 0 ? sec : 1e-9));
            }
            ++h->fanout_calib_calls;
        }
    });
}

int whenet_forward_f32(whenet_t* h, const float* image, int n, float* ypr, int32_t* argmax, float* logits) {
    return guarded(h, [&](whenet::Engine& e) { e.forward_host_f32(image, n, ypr, argmax, logits); });
}

int whenet_forward_u8_device(whenet_t* h, const uint8_t* d_crops, int n, float* d_ypr, int32_t* d_argmax,
                             float* d_logits, void* stream) {
    return guarded(h, [&](whenet::Engine& e) {
        // caller's stream: strictly ordered on it (primary engine); handle-owned streams: next engine
        whenet::Engine& eng = (stream != nullptr) ? e : h->take();
        eng.forward_device(d_crops, n, d_ypr, d_argmax, d_logits, static_cast<hipStream_t>(stream));
    });
}

int whenet_sync(whenet_t* h) {
    return guarded(h, [&](whenet::Engine& e) {
        e.sync();
        for (whenet::Engine* r : h->replicas) r->sync();
        for (whenet::Engine* r : h->fan) r->sync();
    });
}

// tickets of the pinned pipeline carry the engine index in their low bits
int whenet_submit_u8(whenet_t* h, const uint8_t* crops, int n, int* ticket) {
    if (ticket == nullptr) return WHENET_EINVAL;
    return guarded(h, [&](whenet::Engine&) {
        const size_t idx = h->next % size_t(h->inflight);
        *ticket = h->take().submit(crops, n) * MAX_INFLIGHT_ENGINES + int(idx);
    });
}

int whenet_frame_rects(int frame_h, int frame_w, const float* bboxes, int k, int32_t* rects) {
    if (frame_h <= 0 || frame_w <= 0 || k < 0 || (k > 0 && (bboxes == nullptr || rects == nullptr))) return WHENET_EINVAL;
    for (int i = 0; i < k; ++i) whenet::frame_box_rect(frame_h, frame_w, bboxes + 4 * i, rects + 4 * i);
    return WHENET_OK;
}

int whenet_normalise_table(float lut[768]) {
    if (lut == nullptr) return WHENET_EINVAL;
    whenet::normalise_table(reinterpret_cast<float (*)[256]>(lut));
    return WHENET_OK;
}

int whenet_submit_frame(whenet_t* h, const uint8_t* frame, int frame_h, int frame_w, int channel_order,
                        const int32_t* rects, int k, int* ticket) {
    if (ticket == nullptr || (channel_order != WHENET_RGB && channel_order != WHENET_BGR)) return WHENET_EINVAL;
    return guarded(h, [&](whenet::Engine&) {
        const size_t idx = h->next % size_t(h->inflight);
        *ticket = h->take().submit_frame(frame, frame_h, frame_w, channel_order == WHENET_BGR, rects, k) *
                      MAX_INFLIGHT_ENGINES + int(idx);
    });
}

int whenet_op_crop_resize(whenet_t* h, const uint8_t* frame, int frame_h, int frame_w, int channel_order,
                          const int32_t* rects, int k, uint8_t* crops) {
    if (channel_order != WHENET_RGB && channel_order != WHENET_BGR) return WHENET_EINVAL;
    return guarded(h, [&](whenet::Engine& e) {
        e.op_crop_resize(frame, frame_h, frame_w, channel_order == WHENET_BGR, rects, k, crops);
    });
}

int whenet_yolo_eval(whenet_t* h, const float* const* feats, const int* grid_h, const int* grid_w, int num_layers,
                     const float* anchors, int num_anchors, int num_classes, float image_h, float image_w,
                     float score_threshold, float iou_threshold, int max_boxes, float* boxes, float* scores,
                     int32_t* classes, int32_t* index, int* count, float* all_boxes, float* all_scores) {
    if (count == nullptr) return WHENET_EINVAL;
    return guarded(h, [&](whenet::Engine& e) {
        *count = e.yolo_eval(feats, grid_h, grid_w, num_layers, anchors, num_anchors, num_classes, image_h, image_w,
                             score_threshold, iou_threshold, max_boxes, boxes, scores, classes, index, all_boxes,
                             all_scores);
    });
}

int whenet_collect(whenet_t* h, int ticket, float* ypr, int32_t* argmax, float* logits) {
    return guarded(h, [&](whenet::Engine&) {
        const int idx = ticket % MAX_INFLIGHT_ENGINES;
        WHENET_REQUIRE(ticket >= 0 && idx < h->inflight, WHENET_EINVAL, "unknown ticket " + std::to_string(ticket));
        h->at(size_t(idx)).collect(ticket / MAX_INFLIGHT_ENGINES, ypr, argmax, logits);
    });
}

int whenet_profile(whenet_t* h, const uint8_t* d_crops, int n, int iters, whenet_launch_stat_t* stats, int cap,
                   int* count) {
    return guarded(h, [&](whenet::Engine& e) {
        // one engine, alone on the GPU: per-launch figures comparable with rocprofv3's kernel trace
        // (which serialises concurrent work); with "inflight" > 1 the timed path overlaps such chains
        for (whenet::Engine* r : h->replicas) r->sync();
        const int c = e.profile(d_crops, n, iters, stats, cap);
        if (count) *count = c;
    });
}

int whenet_op_stem(whenet_t* h, const uint8_t* crops, int n, float* out) {
    return guarded(h, [&](whenet::Engine& e) { e.op_stem(crops, n, out); });
}

int whenet_op_block(whenet_t* h, int index, const float* in, int n, float* expand_out, float* dw_out, float* gate,
                    float* out) {
    return guarded(h, [&](whenet::Engine& e) { e.op_block(index, in, n, expand_out, dw_out, gate, out); });
}

int whenet_op_block_range(whenet_t* h, int first, int last, const float* in, int n, float* out) {
    return guarded(h, [&](whenet::Engine& e) { e.op_block_range(first, last, in, n, out); });
}

int whenet_op_head(whenet_t* h, const float* in, int n, float* feat, float* logits, float* ypr, int32_t* argmax) {
    return guarded(h, [&](whenet::Engine& e) { e.op_head(in, n, feat, logits, ypr, argmax); });
}

int whenet_op_decode(whenet_t* h, const float* logits, int n, float* ypr, int32_t* argmax) {
    return guarded(h, [&](whenet::Engine& e) { e.op_decode(logits, n, ypr, argmax); });
}

int whenet_block_spec(int index, int32_t out[8]) {
    if (out == nullptr) return WHENET_EINVAL;
    try {
        const auto blocks = whenet::make_blocks();
        if (index < 1 || index > int(blocks.size())) return WHENET_EINVAL;
        const whenet::BlockSpec& b = blocks[size_t(index - 1)];
        const int32_t v[8] = {b.k, b.s, b.expand, b.cin, b.cout, b.h_in, b.h_out, b.se_reduced()};
        std::memcpy(out, v, sizeof(v));
        return WHENET_OK;
    } catch (...) {
        return WHENET_ENOMEM;
    }
}

int whenet_dw_plan(int dtype, int index, int32_t out[12]) {
    if (out == nullptr || (dtype != WHENET_F32 && dtype != WHENET_F16)) return WHENET_EINVAL;
    try {
        const auto blocks = whenet::make_blocks();
        if (index < 1 || index > int(blocks.size())) return WHENET_EINVAL;
        const whenet::BlockSpec& b = blocks[size_t(index - 1)];
        const whenet::DwPlan p = whenet::plan_dw(dtype, b.k, b.s, b.h_in, b.h_out, b.cexp());
        const int32_t v[12] = {p.threads, p.CV, p.TH, p.NSX, p.tiles_x, p.tiles_y, p.chunks, p.IH, p.IW,
                               int32_t(p.lds_bytes), b.pad_before(), b.cexp()};
        std::memcpy(out, v, sizeof(v));
        return WHENET_OK;
    } catch (...) {
        return WHENET_EINVAL;
    }
}

int whenet_front_plan(int dtype, int index, int32_t out[12]) {
    if (out == nullptr || (dtype != WHENET_F32 && dtype != WHENET_F16)) return WHENET_EINVAL;
    try {
        const auto blocks = whenet::make_blocks();
        if (index < 2 || index > int(blocks.size())) return WHENET_EINVAL;
        const whenet::BlockSpec& b = blocks[size_t(index - 1)];
        const whenet::FrontPlan p = whenet::plan_front(dtype, b.k, b.s, b.h_in, b.h_out, b.cexp());
        const int32_t v[12] = {p.threads, p.CC, p.TH, p.NSX, p.tiles_x, p.tiles_y, p.chunks, p.EH, p.EW,
                               int32_t(p.lds_bytes), p.w_off, b.cexp()};
        std::memcpy(out, v, sizeof(v));
        return WHENET_OK;
    } catch (...) {
        return WHENET_EINVAL;
    }
}

int whenet_device_alloc(whenet_t* h, size_t nbytes, void** d_ptr) {
    if (d_ptr == nullptr) return WHENET_EINVAL;
    return guarded(h, [&](whenet::Engine& e) { *d_ptr = e.dev_alloc(nbytes); });
}

int whenet_device_free(whenet_t* h, void* d_ptr) {
    return guarded(h, [&](whenet::Engine& e) { e.dev_free(d_ptr); });
}

int whenet_memcpy_h2d(whenet_t* h, void* d_dst, const void* src, size_t nbytes) {
    return guarded(h, [&](whenet::Engine& e) { e.h2d(d_dst, src, nbytes); });
}

int whenet_memcpy_d2h(whenet_t* h, void* dst, const void* d_src, size_t nbytes) {
    return guarded(h, [&](whenet::Engine& e) { e.d2h(dst, d_src, nbytes); });
}

#pragma GCC visibility pop
}  // extern "C"
