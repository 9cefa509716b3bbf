// Native streaming driver for the batch verifier: host threads over the C ABI's split entry points.
//
// What the reference does with rayon over transactions (reference cli/src/commands/start.rs:616-642 sizes the pools;
// VM::check_transactions walks a block in chunks) becomes three stages here: the caller's thread enqueues batches
// (mi_vs_accumulate_begin: derive + both MSMs on a free lane), `fold_threads` threads wait for accumulations and fold
// them on the host (mi_vs_accumulate_end), one finisher exchanges partials (multi-GPU callback), starts the pairing
// products (mi_vs_finalize_async, up to `products` in flight) and publishes verdicts in submission order.  The entry
// points drop the library lock around their waits and folds, so the stages overlap.  Pure host code: no kernels here.
#include <string.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/mi355x_snark.h"

namespace {

struct Stream {
    uint64_t ctx = 0;
    int products = 8, lanes = 12, group = 1;
    mi_vs_exchange_fn exchange = nullptr;
    void *user = nullptr;

    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<uint64_t, uint64_t>> to_fold;  // (sequence number, accumulate handle), submission order
    std::map<uint64_t, mi_vs_partial> folded;           // sequence number -> this rank's partial
    std::deque<int32_t> results;                        // 0 / 1 verdicts, or a negative status
    struct Sub { const void *proofs; size_t n; const void *rand; };
    std::deque<Sub> pending;                            // submitted batches whose verdict has not been taken, oldest first
    uint64_t submitted = 0, open = 0;                   // batches submitted; accumulations begun and not yet folded
    bool finished = false;                              // no more submissions
    bool finisher_done = false;
    int32_t error = MI_OK;                              // first failure inside the pipeline

    std::vector<std::thread> folders;
    std::thread finisher;

    void fail(int32_t st) {  // mu held
        if (error == MI_OK) error = st;
        cv.notify_all();
    }

    void folder_main() {
        for (;;) {
            std::pair<uint64_t, uint64_t> item;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !to_fold.empty() || finished || error != MI_OK; });
                if (to_fold.empty()) return;  // finished (or failed) and nothing left to fold
                item = to_fold.front();
                to_fold.pop_front();
            }
            mi_vs_partial part;
            const int32_t st = mi_vs_accumulate_end(ctx, item.second, &part);  // waits for the GPU, folds
            std::lock_guard<std::mutex> lk(mu);
            open--;
            if (st != MI_OK) fail(st);
            else folded[item.first] = part;
            cv.notify_all();
        }
    }

    void finisher_main() {
        std::deque<uint64_t> tickets;
        std::vector<mi_vs_partial> mine(8), all(8 * 64), one(64);
        uint64_t seq = 0;
        auto publish = [&](int32_t v) {
            std::lock_guard<std::mutex> lk(mu);
            results.push_back(v);
            cv.notify_all();
        };
        auto wait_oldest = [&]() {
            uint8_t ok = 0;
            const int32_t st = mi_vs_finalize_wait(ctx, tickets.front(), &ok);
            tickets.pop_front();
            publish(st == MI_OK ? (int32_t)ok : st);
        };
        bool failed = false;
        while (!failed) {
            // the next group: batches [seq, seq + count), count = group, or what is left of a finished stream
            size_t count = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                auto ready = [&]() -> bool {
                    if (error != MI_OK) return true;
                    const uint64_t want = finished && submitted - seq < (uint64_t)group ? submitted - seq : (uint64_t)group;
                    if (want == 0) return finished;
                    for (uint64_t k = 0; k < want; k++)
                        if (!folded.count(seq + k)) return false;
                    return finished || want == (uint64_t)group;
                };
                cv.wait(lk, ready);
                if (error != MI_OK) break;
                count = (size_t)(finished && submitted - seq < (uint64_t)group ? submitted - seq : (uint64_t)group);
                if (count == 0) break;  // finished and drained
                for (size_t k = 0; k < count; k++) {
                    mine[k] = folded[seq + k];
                    folded.erase(seq + k);
                }
            }
            seq += count;
            size_t world = 1;
            const mi_vs_partial *src = mine.data();
            if (exchange) {
                const int32_t got = exchange(user, mine.data(), count, all.data(), all.size());
                if (got <= 0 || (size_t)got * count > all.size() || (size_t)got > one.size()) {
                    std::lock_guard<std::mutex> lk(mu);
                    fail(got < 0 ? got : MI_ERR_BAD_ARGS);
                    break;
                }
                world = (size_t)got;
                src = all.data();
            }
            for (size_t k = 0; k < count && !failed; k++) {
                for (size_t r = 0; r < world; r++) one[r] = src[r * count + k];
                if ((int)tickets.size() >= products) wait_oldest();
                uint64_t tk = 0;
                int32_t st = mi_vs_finalize_async(ctx, one.data(), world, &tk);
                while (st == MI_ERR_AGAIN && !tickets.empty()) {  // fewer product slots than `products` (MI_SNARK_VS_SLOTS), or another stream holds them
                    wait_oldest();
                    st = mi_vs_finalize_async(ctx, one.data(), world, &tk);
                }
                if (st != MI_OK) {
                    std::lock_guard<std::mutex> lk(mu);
                    fail(st);
                    failed = true;
                    break;
                }
                tickets.push_back(tk);
            }
        }
        while (!tickets.empty()) wait_oldest();  // products already in flight still get their verdicts
        std::lock_guard<std::mutex> lk(mu);
        finisher_done = true;
        cv.notify_all();
    }
};

std::mutex g_mu;
std::map<uint64_t, std::shared_ptr<Stream>> g_streams;
uint64_t g_next = 1;

std::shared_ptr<Stream> find(uint64_t id) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_streams.find(id);
    return it == g_streams.end() ? nullptr : it->second;
}

}  // namespace

extern "C" {

int32_t mi_vs_stream_open(uint64_t ctx, int32_t fold_threads, int32_t products, int32_t lanes, mi_vs_exchange_fn exchange,
                          void *user, int32_t exchange_group, uint64_t *stream_out) {
    if (!stream_out) return MI_ERR_BAD_ARGS;
    auto s = std::make_shared<Stream>();
    s->ctx = ctx;
    s->products = products < 1 ? 1 : (products > MI_VS_MAX_TICKETS ? MI_VS_MAX_TICKETS : products);
    s->lanes = lanes < 1 ? 1 : (lanes > MI_VS_MAX_OPEN_ACCUMULATIONS ? MI_VS_MAX_OPEN_ACCUMULATIONS : lanes);
    s->exchange = exchange;
    s->user = user;
    s->group = exchange_group < 1 ? 1 : (exchange_group > 8 ? 8 : exchange_group);
    if (fold_threads < 1) fold_threads = 1;
    if (fold_threads > 8) fold_threads = 8;
    Stream *raw = s.get();
    for (int i = 0; i < fold_threads; i++) s->folders.emplace_back([raw] { raw->folder_main(); });
    s->finisher = std::thread([raw] { raw->finisher_main(); });
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t id = g_next++;
    g_streams[id] = s;
    *stream_out = id;
    return MI_OK;
}

int32_t mi_vs_stream_submit(uint64_t stream, const void *d_proofs, size_t n, const void *d_randomizers128, void *hip_stream) {
    auto s = find(stream);
    if (!s || !d_proofs || !n || !d_randomizers128) return MI_ERR_BAD_ARGS;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        if (s->finished) return MI_ERR_BAD_ARGS;
        s->cv.wait(lk, [&] { return (int)s->open < s->lanes || s->error != MI_OK; });
        if (s->error != MI_OK) return s->error;
        s->open++;  // reserved: a lane of ours is free
    }
    uint64_t h = 0;
    int32_t st = mi_vs_accumulate_begin(s->ctx, d_proofs, n, d_randomizers128, hip_stream, &h);
    std::lock_guard<std::mutex> lk(s->mu);
    if (st != MI_OK) {
        s->open--;
        s->fail(st);
        return st;
    }
    s->to_fold.emplace_back(s->submitted++, h);
    s->pending.push_back(Stream::Sub{d_proofs, n, d_randomizers128});
    s->cv.notify_all();
    return MI_OK;
}

int32_t mi_vs_stream_finish(uint64_t stream) {
    auto s = find(stream);
    if (!s) return MI_ERR_BAD_ARGS;
    std::lock_guard<std::mutex> lk(s->mu);
    s->finished = true;
    s->cv.notify_all();
    return MI_OK;
}

int32_t mi_vs_stream_next_verdicts(uint64_t stream, int32_t block, uint8_t *all_ok, uint8_t *verdicts, size_t capacity,
                                   size_t *n_out) {
    auto s = find(stream);
    if (!s || !all_ok) return MI_ERR_BAD_ARGS;
    Stream::Sub sub{nullptr, 0, nullptr};
    {
        std::unique_lock<std::mutex> lk(s->mu);
        for (;;) {
            if (!s->results.empty()) {
                const int32_t v = s->results.front();
                s->results.pop_front();
                if (!s->pending.empty()) {
                    sub = s->pending.front();
                    s->pending.pop_front();
                }
                if (v < 0) return v;
                *all_ok = (uint8_t)v;
                break;
            }
            if (s->finisher_done) return s->error != MI_OK ? s->error : MI_ERR_DONE;
            if (!block) return MI_ERR_AGAIN;
            s->cv.wait(lk);
        }
    }
    if (n_out) *n_out = sub.n;
    if (!verdicts) return MI_OK;
    if (capacity < sub.n) return MI_ERR_BAD_ARGS;
    if (*all_ok) {
        memset(verdicts, 1, sub.n);
        return MI_OK;
    }
    // the stream keeps no per-proof state: the failed batch is taken apart from its (still resident) inputs
    return mi_vs_identify_device(s->ctx, sub.proofs, sub.n, sub.rand, verdicts, nullptr);
}

int32_t mi_vs_stream_next(uint64_t stream, int32_t block, uint8_t *all_ok) {
    return mi_vs_stream_next_verdicts(stream, block, all_ok, nullptr, 0, nullptr);
}

int32_t mi_vs_stream_close(uint64_t stream) {
    std::shared_ptr<Stream> s;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_streams.find(stream);
        if (it == g_streams.end()) return MI_ERR_BAD_ARGS;
        s = it->second;
        g_streams.erase(it);
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->finished = true;
        s->cv.notify_all();
    }
    for (auto &t : s->folders) t.join();
    s->finisher.join();
    return MI_OK;
}

}  // extern "C"

namespace mi {
// mi_shutdown: no driver thread may outlive the devices
void stream_release_all() {
    std::vector<uint64_t> ids;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto &kv : g_streams) ids.push_back(kv.first);
    }
    for (uint64_t id : ids) (void)mi_vs_stream_close(id);
}
}  // namespace mi
