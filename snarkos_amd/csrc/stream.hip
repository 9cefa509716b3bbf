// Native streaming driver for the batch verifier: host threads over the C ABI's split entry points.
//
// What the reference does with rayon over transactions (reference cli/src/commands/start.rs:616-642 sizes the pools;
// VM::check_transactions walks a block in chunks) becomes three stages here: the caller's thread enqueues batches
// (mi_vs_accumulate_begin: derive + both MSMs on a free lane), `fold_threads` threads wait for accumulations and fold
// them on the host (mi_vs_accumulate_end), one finisher exchanges partials (multi-GPU callback) and starts the pairing
// products (mi_vs_finalize_async, up to `products` in flight), one publisher waits for them and hands the verdicts out
// in submission order.  Under load two consecutive batches share one lane pass (mi_vs_accumulate_begin_pair).  The
// entry points drop the library lock around their waits and folds, so the stages overlap.  Pure host code: no kernels.
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/mi355x_snark.h"
#include "runtime.hpp"  // the process-wide counters only: the driver itself runs on the public entry points

namespace {

// The driver's threads come from a cache: a stream opens and closes around every block (or every verify_stream call), and
// five thread creations and joins were 0.12 ms of a 4 ms burst.  Idle threads wait for the next stream's job; the cache
// grows when none is idle (several streams at once).  The cache lives as long as the process (its idle threads touch
// nothing of the devices: mi_shutdown closes the streams, which ends their jobs); it is never destroyed, so there is
// no join at exit that a stream nobody closed could hang (std::thread members used to end that case with
// std::terminate).
class ThreadCache {
  public:
    void run(std::function<void()> job) {
        std::lock_guard<std::mutex> lk(mu_);
        q_.push_back(std::move(job));
        if (idle_ < (int)q_.size()) std::thread([this] { main(); }).detach();  // nobody free for it: one more thread
        cv_.notify_one();
    }

  private:
    void main() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            idle_++;
            cv_.wait(lk, [this] { return !q_.empty(); });
            idle_--;
            std::function<void()> job = std::move(q_.front());
            q_.pop_front();
            lk.unlock();
            job();
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    int idle_ = 0;
};
ThreadCache &thread_cache() {
    static ThreadCache *c = new ThreadCache;  // never destroyed (above)
    return *c;
}

constexpr int32_t RECHECK = 2;  // the product over several merged batches failed: this batch is verified on its own when its verdict is taken

struct Stream {
    uint64_t ctx = 0;
    int products = 8, lanes = 12, group = 1;
    mi_vs_exchange_fn exchange = nullptr;
    void *user = nullptr;

    std::mutex mu;
    std::condition_variable cv;
    struct Pass { uint64_t seq, handle; int count; bool merged; };  // a lane pass: batches seq .. seq + count - 1
    std::deque<Pass> to_fold;                           // submission order
    std::map<uint64_t, mi_vs_partial> folded;           // sequence number -> this rank's partial
    // A MERGED pass (no exchange between ranks: mi_vs_accumulate_begin_merged) has ONE partial for its `count` batches:
    // it is stored under the first, cover[first] = count, and the others are marked 0 (covered: no product of their own).
    std::map<uint64_t, int> cover;                      // sequence number -> batches its product decides (absent: 1)
    std::deque<int32_t> results;                        // 0 / 1 verdicts, RECHECK, or a negative status
    struct Sub { const void *proofs; size_t n; const void *rand; };
    std::deque<Sub> pending;                            // submitted batches whose verdict has not been taken, oldest first
    std::vector<size_t> sizes;                          // every submitted batch's size, in order (counters)
    size_t published = 0;                               // verdicts handed to `results` so far
    uint64_t submitted = 0, open = 0;                   // batches submitted; accumulations begun and not yet folded
    // Under load consecutive batches share one lane pass (mi_vs_accumulate_begin_multi): a batch submitted while
    // `pair_at` or more passes are open is HELD until `per_pass` batches are together -- or until a pass ends, the
    // stream is finished or somebody blocks on a verdict, whichever comes first; then what is held runs as one pass.
    // pair_at 0: one batch per pass, always.
    int pair_at = 4;
    int per_pass = 0;                                   // batches per pass for this context's shape (0: not asked yet)
    // MI_SNARK_STREAM_PER_PASS.  Passes that keep a partial per batch (streams with an exchange): 2 measured best for both
    // transcripts.  Merged passes: 4 (tools/merge_sweep.sh: 400 batches 8.8e6 / 9.5e6 / 9.4-10.2e6 proofs/s with 2 / 3 / 4 per
    // pass, 20 batches 4.2-4.3e6 / 4.4e6 / 4.6-4.7e6)
    int per_pass_max = 2;
    bool merge = true;                                  // MI_SNARK_STREAM_MERGE=0: the batches of a pass keep their own aggregates
    std::vector<Sub> held;
    void *held_stream = nullptr;
    bool launching = false;                             // somebody is between taking the held batch and queueing its pass
    bool finished = false;                              // no more submissions
    bool closing = false;                               // mi_vs_stream_close: verdicts not yet started are given up
    bool finisher_done = false;
    int32_t error = MI_OK;                              // first failure inside the pipeline

    int running = 0;  // driver threads (folders, finisher, publisher) that have not returned yet (mu)

    void fail(int32_t st) {  // mu held
        if (error == MI_OK) error = st;
        cv.notify_all();
    }

    void folder_main() {
        for (;;) {
            Pass item;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !to_fold.empty() || (finished && held.empty() && !launching) || error != MI_OK; });
                if (to_fold.empty()) return;  // finished (or failed) and nothing left to fold
                item = to_fold.front();
                to_fold.pop_front();
            }
            mi_vs_partial part[MI_VS_MAX_BATCHES_PER_PASS];
            const int32_t st = item.count > 1 && !item.merged ? mi_vs_accumulate_end_multi(ctx, item.handle, part, item.count)  // waits for the GPU, folds
                                                              : mi_vs_accumulate_end(ctx, item.handle, &part[0]);
            {
                std::lock_guard<std::mutex> lk(mu);
                open--;
                if (st != MI_OK) fail(st);
                else
                    for (int k = 0; k < item.count; k++) {
                        folded[item.seq + k] = part[item.merged ? 0 : k];
                        if (item.merged) cover[item.seq + k] = k == 0 ? item.count : 0;
                    }
                cv.notify_all();
            }
            flush_held(false);  // a pass has ended: a held batch does not wait for company any longer than that
        }
    }

    // Launch the batches as ONE pass when the library takes them together, else as one pass each.  mu NOT held; the
    // caller has reserved one lane (open++) and set `launching`.  Queues the pass(es) for the folders.
    int32_t launch(const std::vector<Sub> &bs, void *st_in) {
        uint64_t h = 0;
        int32_t st = MI_ERR_BAD_ARGS;
        bool together = false, merged = false;
        const int count = (int)bs.size();
        if (count > 1) {
            const void *pr[MI_VS_MAX_BATCHES_PER_PASS], *rn[MI_VS_MAX_BATCHES_PER_PASS];
            size_t ns[MI_VS_MAX_BATCHES_PER_PASS];
            for (int k = 0; k < count; k++) {
                pr[k] = bs[k].proofs;
                rn[k] = bs[k].rand;
                ns[k] = bs[k].n;
            }
            // Without an exchange the batches of a pass become ONE aggregate -- one MSM pair, one fold, one product --
            // and are told apart only if that product fails.  (With an exchange every rank would have to merge the
            // same batches, and what shares a pass depends on each rank's timing.)
            if (!exchange && merge) {
                st = mi_vs_accumulate_begin_merged(ctx, count, pr, ns, rn, st_in, &h);
                merged = together = st == MI_OK;
            }
            if (!together) {
                st = mi_vs_accumulate_begin_multi(ctx, count, pr, ns, rn, st_in, &h);
                together = st == MI_OK;
            }
        }
        if (together || count == 1) {
            if (count == 1) st = mi_vs_accumulate_begin(ctx, bs[0].proofs, bs[0].n, bs[0].rand, st_in, &h);
            std::lock_guard<std::mutex> lk(mu);
            launching = false;
            if (st != MI_OK) {
                open--;
                fail(st);
                return st;
            }
            to_fold.push_back(Pass{submitted, h, count, merged});
            submitted += (uint64_t)count;
            for (const Sub &b : bs) pending.push_back(b);
            cv.notify_all();
            return MI_OK;
        }
        // the shape (or these sizes) does not share a pass: one pass each, in order
        for (int k = 0; k < count; k++) {
            if (k > 0) {
                std::unique_lock<std::mutex> lk(mu);
                per_pass = 1;
                cv.wait(lk, [&] { return (int)open < lanes || error != MI_OK; });
                if (error != MI_OK) {
                    launching = false;
                    cv.notify_all();
                    return error;
                }
                open++;
            }
            st = mi_vs_accumulate_begin(ctx, bs[k].proofs, bs[k].n, bs[k].rand, st_in, &h);
            std::lock_guard<std::mutex> lk(mu);
            if (st != MI_OK) {
                launching = false;
                open--;
                fail(st);
                return st;
            }
            to_fold.push_back(Pass{submitted++, h, 1, false});
            pending.push_back(bs[k]);
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        launching = false;
        cv.notify_all();
        return MI_OK;
    }

    // run what is held as one pass (wait: block for a lane; otherwise only if one is free right now)
    int32_t flush_held(bool wait) {
        std::vector<Sub> bs;
        void *st_in;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (held.empty() || launching || error != MI_OK) return MI_OK;
            if ((int)open >= lanes) {
                if (!wait) return MI_OK;  // every lane is busy: the next pass to end comes back here
                cv.wait(lk, [&] { return (int)open < lanes || error != MI_OK || held.empty() || launching; });
                if (held.empty() || launching || error != MI_OK) return MI_OK;
            }
            bs.swap(held);
            st_in = held_stream;
            launching = true;
            open++;
        }
        return launch(bs, st_in);
    }

    // The finisher starts pairing products as partials arrive; the PUBLISHER waits for them in order and hands the
    // verdicts out.  (One thread doing both published a verdict only when it needed the product's slot back or the
    // stream was finished: a caller trickling batches in would wait for ever.)
    struct Ticket {
        uint64_t id;
        int products;
        int covers[MI_VS_MAX_PRODUCTS_PER_TICKET];  // batches each product decides
    };
    std::deque<Ticket> tickets;    // started, oldest first (mu)
    int inflight = 0;              // started and not yet published (mu)
    bool launches_done = false;    // the finisher has ended (mu)

    void publisher_main() {
        for (;;) {
            Ticket tk;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !tickets.empty() || launches_done; });
                if (tickets.empty()) {  // the finisher has ended and every product has been published
                    finisher_done = true;
                    cv.notify_all();
                    return;
                }
                tk = tickets.front();
                tickets.pop_front();
            }
            uint8_t ok[MI_VS_MAX_PRODUCTS_PER_TICKET] = {};
            const int32_t st = mi_vs_finalize_wait_multi(ctx, tk.id, ok, tk.products);
            std::lock_guard<std::mutex> lk(mu);
            int batches = 0;
            for (int k = 0; k < tk.products; k++)
                for (int c = 0; c < tk.covers[k]; c++) {  // a failed product over several batches decides none of them
                    const int32_t r = st != MI_OK ? st : (ok[k] ? 1 : (tk.covers[k] > 1 ? RECHECK : 0));
                    results.push_back(r);
                    // counters: the batches these verdicts belong to are the oldest unpublished ones (a batch to be
                    // checked again is counted by that check)
                    if (r == 0 || r == 1) {
                        mi::stats().batches_verified++;
                        if (published + (size_t)batches < sizes.size()) mi::stats().proofs_verified += sizes[published + (size_t)batches];
                    }
                    batches++;
                }
            published += (size_t)batches;
            inflight--;
            cv.notify_all();
        }
    }

    void finisher_main() {
        std::vector<mi_vs_partial> mine(8), all(8 * 64), one(64 * MI_VS_MAX_PRODUCTS_PER_TICKET);
        int cov[8];  // per batch of the group: the batches its product decides (0: covered by a merged pass's first batch)
        uint64_t seq = 0;
        bool failed = false;
        while (!failed) {
            // the next group: batches [seq, seq + count), count = group, or what is left of a finished stream
            size_t count = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                auto ready = [&]() -> bool {
                    if (error != MI_OK) return true;
                    const bool fin = finished && held.empty() && !launching;  // every submitted batch has its pass
                    const uint64_t want = fin && submitted - seq < (uint64_t)group ? submitted - seq : (uint64_t)group;
                    if (want == 0) return fin;
                    for (uint64_t k = 0; k < want; k++)
                        if (!folded.count(seq + k)) return false;
                    return fin || want == (uint64_t)group;
                };
                cv.wait(lk, ready);
                if (error != MI_OK) break;
                const bool fin = finished && held.empty() && !launching;
                count = (size_t)(fin && submitted - seq < (uint64_t)group ? submitted - seq : (uint64_t)group);
                if (count == 0) break;  // finished and drained
                // without an exchange nothing ties the group size: whatever is folded already (the other batches of
                // the pass, a burst) goes along, so that it can share a ticket
                if (!exchange)
                    while (count < (size_t)MI_VS_MAX_PRODUCTS_PER_TICKET && folded.count(seq + count)) count++;
                for (size_t k = 0; k < count; k++) {
                    mine[k] = folded[seq + k];
                    folded.erase(seq + k);
                    auto it = cover.find(seq + k);
                    cov[k] = it == cover.end() ? 1 : it->second;
                    if (it != cover.end()) cover.erase(it);
                }
            }
            seq += count;
            // the group's products: one per batch, except that a merged pass's batches have ONE (under its first batch)
            size_t prod_at[8], n_prod = 0;
            for (size_t k = 0; k < count; k++)
                if (cov[k] > 0) prod_at[n_prod++] = k;
            size_t world = 1;
            const mi_vs_partial *src = mine.data();
            if (exchange) {
                const int32_t got = exchange(user, mine.data(), count, all.data(), all.size());
                if (got <= 0 || (size_t)got * count > all.size() || (size_t)got > 64) {
                    std::lock_guard<std::mutex> lk(mu);
                    fail(got < 0 ? got : MI_ERR_BAD_ARGS);
                    break;
                }
                world = (size_t)got;
                src = all.data();
            }
            // the group's batches go out under as few tickets as possible: one product stream serves up to
            // MI_VS_MAX_PRODUCTS_PER_TICKET products that are ready together
            for (size_t k0 = 0; k0 < n_prod && !failed; k0 += MI_VS_MAX_PRODUCTS_PER_TICKET) {
                const int m = (int)(n_prod - k0 < (size_t)MI_VS_MAX_PRODUCTS_PER_TICKET ? n_prod - k0 : (size_t)MI_VS_MAX_PRODUCTS_PER_TICKET);
                for (int j = 0; j < m; j++)
                    for (size_t r = 0; r < world; r++) one[(size_t)j * world + r] = src[r * count + prod_at[k0 + (size_t)j]];
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return inflight < products; });
                }
                uint64_t tk = 0;
                int32_t st = mi_vs_finalize_async_multi(ctx, one.data(), world, m, &tk);
                while (st == MI_ERR_AGAIN) {  // fewer product slots than `products` (MI_SNARK_VS_SLOTS), or other drivers and callers hold them
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (error != MI_OK || closing) break;  // nobody is going to take the verdict: the status stands
                    }
                    (void)mi_vs_wait_product_slot(ctx, 20);  // our publisher's waits free slots too: one wait covers both cases
                    st = mi_vs_finalize_async_multi(ctx, one.data(), world, m, &tk);
                }
                std::lock_guard<std::mutex> lk(mu);
                if (st != MI_OK) {
                    fail(st);
                    failed = true;
                    break;
                }
                Ticket t{tk, m, {0, 0, 0, 0}};
                for (int j = 0; j < m; j++) t.covers[j] = cov[prod_at[k0 + (size_t)j]];
                tickets.push_back(t);
                inflight++;
                cv.notify_all();
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        launches_done = true;  // products already in flight still get their verdicts (publisher)
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
    if (const char *e = getenv("MI_SNARK_STREAM_PAIR_AT")) s->pair_at = atoi(e);  // 0: one batch per pass, always
    if (const char *e = getenv("MI_SNARK_STREAM_MERGE")) s->merge = e[0] != '0';
    if (!exchange && s->merge) s->per_pass_max = MI_VS_MAX_BATCHES_PER_PASS;
    if (const char *e = getenv("MI_SNARK_STREAM_PER_PASS")) {
        const int x = atoi(e);
        if (x >= 1 && x <= MI_VS_MAX_BATCHES_PER_PASS) s->per_pass_max = x;
    }
    // (Streams with an exchange pair as well: the collectives go by batch sequence number, in order, whatever pass a
    // batch rode on; a held batch never waits for a collective, only for a pass to end.)
    // (the jobs hold the stream alive: mi_vs_stream_close waits for them)
    auto job = [s](void (Stream::*main)()) {
        return [s, main] {
            ((*s).*main)();
            std::lock_guard<std::mutex> lk(s->mu);
            s->running--;
            s->cv.notify_all();
        };
    };
    s->running = fold_threads + 2;
    for (int i = 0; i < fold_threads; i++) thread_cache().run(job(&Stream::folder_main));
    thread_cache().run(job(&Stream::finisher_main));
    thread_cache().run(job(&Stream::publisher_main));
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t id = g_next++;
    g_streams[id] = s;
    *stream_out = id;
    return MI_OK;
}

int32_t mi_vs_stream_submit(uint64_t stream, const void *d_proofs, size_t n, const void *d_randomizers128, void *hip_stream) {
    auto s = find(stream);
    if (!s || !d_proofs || !n || !d_randomizers128) return MI_ERR_BAD_ARGS;
    const Stream::Sub mine{d_proofs, n, d_randomizers128};
    bool join, ask;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->finished) return MI_ERR_BAD_ARGS;
        s->sizes.push_back(n);  // one submitting thread per stream: submission order is batch order
        join = !s->held.empty() && s->held_stream == hip_stream && s->held.front().n == n;
        ask = s->per_pass == 0;
    }
    if (ask) {  // how many batches of this size one pass takes (asked once: the shape is the context's)
        const int32_t pp = !s->exchange && s->merge ? mi_vs_batches_per_merged_pass(s->ctx, n) : mi_vs_batches_per_pass(s->ctx, n);
        std::lock_guard<std::mutex> lk(s->mu);
        s->per_pass = pp < 1 ? 1 : (pp > s->per_pass_max ? s->per_pass_max : pp);
    }
    if (!join) {  // a pass is ordered behind ONE caller stream and sized for equal batches: what is held runs without this one
        const int32_t st = s->flush_held(true);
        if (st != MI_OK) return st;
    }
    std::vector<Stream::Sub> bs;
    void *first_stream = hip_stream;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return !s->launching || s->error != MI_OK; });  // a folder is launching what was held
        if (s->error != MI_OK) return s->error;
        if (!s->held.empty() && !(s->held_stream == hip_stream && s->held.front().n == n)) return MI_ERR_AGAIN;  // two submitters raced: retry
        const bool hold = s->pair_at > 0 && s->per_pass > 1 && (int)s->open >= s->pair_at;
        if (!s->held.empty() || hold) {
            s->held.push_back(mine);
            s->held_stream = hip_stream;
            if ((int)s->held.size() < s->per_pass) return MI_OK;  // runs when its pass is full, or when a pass ends / the stream finishes / someone waits
            bs.swap(s->held);
        } else {
            bs.push_back(mine);
        }
        s->launching = true;
        s->cv.wait(lk, [&] { return (int)s->open < s->lanes || s->error != MI_OK; });
        if (s->error != MI_OK) {
            s->launching = false;
            s->cv.notify_all();
            return s->error;
        }
        s->open++;  // reserved: a lane of ours is free
    }
    return s->launch(bs, first_stream);
}

int32_t mi_vs_stream_finish(uint64_t stream) {
    auto s = find(stream);
    if (!s) return MI_ERR_BAD_ARGS;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->finished = true;
        s->cv.notify_all();
    }
    return s->flush_held(true);
}

int32_t mi_vs_stream_next_verdicts(uint64_t stream, int32_t block, uint8_t *all_ok, uint8_t *verdicts, size_t capacity,
                                   size_t *n_out) {
    auto s = find(stream);
    if (!s || !all_ok) return MI_ERR_BAD_ARGS;
    Stream::Sub sub{nullptr, 0, nullptr};
    bool recheck = false;
    if (block) {
        const int32_t st = s->flush_held(true);  // a caller who waits for verdicts is not going to bring the held batch company
        if (st != MI_OK) return st;
    }
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
                recheck = v == RECHECK;
                *all_ok = v == 1 ? 1 : 0;
                break;
            }
            if (s->finisher_done) return s->error != MI_OK ? s->error : MI_ERR_DONE;
            if (!block) return MI_ERR_AGAIN;
            s->cv.wait(lk);
        }
    }
    if (n_out) *n_out = sub.n;
    if (recheck) {
        // its pass was one aggregate over several batches and that product failed: this batch's own verdict comes from
        // its (still resident) inputs -- the whole check, then the per-proof verdicts if it is the one that fails
        if (verdicts && capacity < sub.n) return MI_ERR_BAD_ARGS;
        std::vector<uint8_t> tmp;
        uint8_t *vd = verdicts;
        if (!vd) {
            tmp.resize(sub.n);
            vd = tmp.data();
        }
        return mi_vs_verify_batch_device(s->ctx, sub.proofs, sub.n, sub.rand, vd, all_ok, nullptr);
    }
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
        s->closing = true;
        s->cv.notify_all();
    }
    (void)s->flush_held(true);
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->running == 0; });
    }
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
