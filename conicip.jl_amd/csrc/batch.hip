// Batches of independent problems on one GPU (BASELINE config 5; SURVEY 8b "cip_batch_*", 8e).
// A single KKT system never leaves its GPU and independent problems share nothing, so a batch is an array of
// ordinary handles -- each with its own HIP stream -- plus a host-thread pool that keeps `in_flight` interior-point
// loops running at once: a small system (n ~ 2048) is a chain of tiny dependent launches that leaves most of the
// chip idle, several of them on different streams fill it (measured from Python threads: 481 -> 658 KKT solves/s).
// Per-problem work uses the normal entry points on cip_batch_handle(b, i) (the "leading problem index").
#include "cip_handle.h"
#include "../../include/cipkkt.h"
#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct cip_batch {
    std::vector<cip_handle *> h;
    std::vector<hipStream_t> streams;
};

extern "C" int cip_conicip_many(cip_handle *const *handles, int count, const double *const *c, const double *const *b,
                                const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                                double *const *v, cip_result *res, int in_flight) {
    if (count < 0 || (count > 0 && (!handles || !c || !y || !res))) { cip_set_error("cip_conicip_many: null argument"); return CIP_E_INVALID; }
    if (count == 0) return 0;
    if (in_flight < 1) in_flight = 1;
    if (in_flight > count) in_flight = count;
    std::atomic<int> next(0), first_rc(0);
    std::string first_err;
    std::atomic<bool> have_err(false);
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count) return;
            const int rc = cip_conicip(handles[i], c[i], b ? b[i] : nullptr, d ? d[i] : nullptr, opt, y[i], w ? w[i] : nullptr,
                                       v ? v[i] : nullptr, &res[i], nullptr, 0);
            if (rc != 0) {
                res[i].status = CIP_STATUS_ERROR;
                bool expected = false;
                if (have_err.compare_exchange_strong(expected, true)) { first_rc = rc; first_err = cip_last_error(); }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < in_flight; ++t) pool.emplace_back(worker);
    worker();                                   // the calling thread is the first worker
    for (auto &t : pool) t.join();
    if (have_err) { cip_set_error("problem failed: %s", first_err.c_str()); return first_rc; }
    return 0;
}

// Problems in, solutions out: `in_flight` worker threads, each with ONE handle on its own stream that is re-loaded
// (cip_update_problem: no hipMalloc / hipFree, which synchronise the device) for every problem of the same shape it
// takes from the queue -- so the level-1 upload of one problem overlaps the interior-point loops of the others, and a
// batch of small systems, each a chain of tiny dependent launches, fills the chip.
extern "C" int cip_conicip_problems(int count, const cip_problem *probs, const double *const *c, const double *const *b,
                                    const double *const *d, const cip_options *opt, double *const *y, double *const *w,
                                    double *const *v, cip_result *res, int in_flight) {
    if (count < 0 || (count > 0 && (!probs || !c || !y || !res))) { cip_set_error("cip_conicip_problems: null argument"); return CIP_E_INVALID; }
    if (count == 0) return 0;
    if (in_flight < 1) in_flight = 1;
    if (in_flight > count) in_flight = count;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { cip_set_error("no HIP device"); return CIP_E_NODEVICE; }
    std::atomic<int> next(0), first_rc(0);
    std::string first_err;
    std::atomic<bool> have_err(false);
    auto worker = [&]() {
        (void)hipSetDevice(device);
        cip_handle *h = nullptr;
        hipStream_t st = nullptr;
        auto fail = [&](int i, int rc) {
            res[i].status = CIP_STATUS_ERROR;
            bool expected = false;
            if (have_err.compare_exchange_strong(expected, true)) { first_rc = rc; first_err = cip_last_error(); }
        };
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= count) break;
            int rc = 0;
            if (h && cip_update_problem(h, &probs[i]) != 0) { cip_destroy(h); h = nullptr; }     // other shape: new handle
            if (!h) {
                rc = cip_create_ex(&probs[i], &h);
                if (rc == 0 && st) rc = cip_set_stream(h, st);
            }
            if (rc == 0)
                rc = cip_conicip(h, c[i], b ? b[i] : nullptr, d ? d[i] : nullptr, opt, y[i], w ? w[i] : nullptr,
                                 v ? v[i] : nullptr, &res[i], nullptr, 0);
            if (rc != 0) fail(i, rc);
        }
        if (h) cip_destroy(h);
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < in_flight; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (have_err) { cip_set_error("problem failed: %s", first_err.c_str()); return first_rc; }
    return 0;
}

extern "C" int cip_batch_create(int count, const cip_problem *probs, cip_batch **out) {
    if (count < 0 || (count > 0 && !probs) || !out) { cip_set_error("cip_batch_create: bad argument"); return CIP_E_INVALID; }
    cip_batch *b = new (std::nothrow) cip_batch();
    if (!b) { cip_set_error("out of host memory"); return CIP_E_INVALID; }
    *out = nullptr;
    for (int i = 0; i < count; ++i) {
        cip_handle *h = nullptr;
        int rc = cip_create_ex(&probs[i], &h);
        hipStream_t s = nullptr;
        if (rc == 0 && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { cip_set_error("hipStreamCreate failed"); rc = CIP_E_HIP; }
        if (rc == 0) rc = cip_set_stream(h, s);
        if (rc != 0) {
            if (h) cip_destroy(h);
            if (s) (void)hipStreamDestroy(s);
            const std::string msg = cip_last_error();
            cip_batch_destroy(b);
            cip_set_error("problem %d: %s", i, msg.c_str());
            return rc;
        }
        b->h.push_back(h);
        b->streams.push_back(s);
    }
    *out = b;
    return 0;
}
extern "C" int cip_batch_destroy(cip_batch *b) {
    if (!b) return 0;
    for (size_t i = 0; i < b->h.size(); ++i) {
        (void)hipStreamSynchronize(b->streams[i]);
        cip_destroy(b->h[i]);
        (void)hipStreamDestroy(b->streams[i]);
    }
    delete b;
    return 0;
}
extern "C" int cip_batch_size(const cip_batch *b) { return b ? (int)b->h.size() : 0; }
extern "C" cip_handle *cip_batch_handle(cip_batch *b, int i) { return (b && i >= 0 && i < (int)b->h.size()) ? b->h[i] : nullptr; }
extern "C" int cip_batch_conicip(cip_batch *b, const double *const *c, const double *const *bb, const double *const *d,
                                 const cip_options *opt, double *const *y, double *const *w, double *const *v, cip_result *res,
                                 int in_flight) {
    if (!b) { cip_set_error("cip_batch_conicip: null batch"); return CIP_E_INVALID; }
    return cip_conicip_many(b->h.data(), (int)b->h.size(), c, bb, d, opt, y, w, v, res, in_flight);
}
