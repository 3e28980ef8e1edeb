// batch_core.h -- the coalescing core shared by the host micro-batchers (batcher.hip, batcher_ragged.hip).
//
// Request threads CLAIM a slot (and a run of payload units) in the set being filled with one compare-and-swap, copy
// their payload into pinned memory themselves, COMMIT, and sleep on the set's generation word.  One worker thread per
// batcher closes the set, flushes it (one H2D copy, one launch sequence, one D2H copy -- the owner's callback),
// publishes the generation and wakes every sleeper with ONE futex call.  Two sets: while one is in flight the other
// fills, so the batch size follows the load (about half the threads in flight) and an idle batcher flushes a lone
// request at once.  No mutex is taken on the request path: with hundreds of request threads (the reference runs up
// to 512 requests in flight, src/bin/ucfp.rs:267) a mutex + condition variable turns every flush into a convoy.
#pragma once

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <functional>
#include <thread>

namespace ucfp {

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expected, const timespec* rel = nullptr) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expected, rel, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t>* w, int waiters) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, waiters, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* w) { futex_wake(w, INT_MAX); }
static_assert(sizeof(std::atomic<uint32_t>) == 4, "futex words are plain 32-bit integers");

class BatchCore {
public:
    // flush(set, n, units) runs on the worker thread; returns a UCFP_* code shared by the n submitters of the set.
    using Flush = std::function<int(int, size_t, size_t)>;

    void start(size_t max_batch, size_t max_units, uint32_t max_delay_us, Flush flush) {
        max_batch_ = max_batch;
        max_units_ = max_units;
        max_delay_us_ = max_delay_us;
        flush_ = std::move(flush);
        worker_ = std::thread([this] { loop(); });
    }
    void stop() {
        if (!worker_.joinable()) return;
        stop_.store(true);
        kick();
        room_epoch_.fetch_add(1);
        futex_wake_all(&room_epoch_);
        worker_.join();
    }

    struct Ticket {
        int set = -1;
        size_t slot = 0, at = 0;      // slot in the set, first payload unit
        uint32_t gen = 0;
    };
    // Blocks while neither set has room.  false: the batcher is shutting down.
    bool claim(size_t units, Ticket* t) {
        for (;;) {
            const uint32_t e = room_epoch_.load();
            if (stop_.load()) return false;
            const int s = fill_.load();
            Set& S = sets_[s];
            uint64_t c = S.claim.load();
            bool full = false;
            while (!(c & kClosed) && count(c) < max_batch_) {
                if ((c & kUnitMask) + units > max_units_) {   // no room for this payload: have the set flushed now
                    full = true;
                    break;
                }
                if (S.claim.compare_exchange_weak(c, c + kOne + units)) {
                    t->set = s;
                    t->slot = count(c);
                    t->at = (size_t)(c & kUnitMask);
                    t->gen = S.gen_done.load() + 1;
                    if (t->slot == 0 || t->slot + 1 >= max_batch_) kick();
                    return true;
                }
            }
            if (full && !S.want_close.exchange(true)) kick();
            futex_wait(&room_epoch_, e);
        }
    }
    // The payload of the ticket's slot is in place.
    void commit(const Ticket& t) {
        Set& S = sets_[t.set];
        S.copied.fetch_add(1);
        if (S.claim.load() & kClosed) kick();
    }
    // Sleeps until the ticket's set has been flushed; returns the flush's code.  The result slot stays reserved
    // until release().
    int wait(const Ticket& t) {
        Set& S = sets_[t.set];
        // The sleepers of a set are spread over kWakeWords futex words (by slot).  Waking N threads costs the waker about a
        // microsecond each, one after the other inside the system call -- 65 us of the 163 a set of 57 searches took end to
        // end.  The worker wakes word 0 first; the first sleeper of word 0 to wake then wakes the other words too, while the worker goes on through the words itself: ONE hop of
        // help, never a dependency -- with more runnable threads than cores a woken helper may not run for milliseconds.
        std::atomic<uint32_t>& w = S.wake[t.slot % kWakeWords].v;
        bool slept = false;
        for (;;) {
            const uint32_t g = w.load();
            if ((int32_t)(g - t.gen) >= 0) break;
            futex_wait(&w, g);
            slept = true;
        }
        if (slept && t.slot % kWakeWords == 0 && S.helped.fetch_add(1) == 0)      // (one helper per generation: the system
            for (size_t i = 1; i < kWakeWords; i++) futex_wake_all(&S.wake[i].v);   // calls of many contend with the worker's)
        return S.rc.load();
    }
    void release(const Ticket& t) {
        if (sets_[t.set].readers.fetch_sub(1) == 1) kick();
    }
    void stats(uint64_t* batches, uint64_t* items) const {
        if (batches) *batches = batches_.load();
        if (items) *items = items_.load();
    }

private:
    static constexpr int kSpinUs = 50;
    static constexpr size_t kWakeWords = 4;
    static constexpr uint32_t kGraceUs = 10;
    static constexpr uint64_t kClosed = 1ull << 63, kOne = 1ull << 32, kUnitMask = 0xffffffffull;
    static size_t count(uint64_t c) { return (size_t)((c & ~kClosed) >> 32); }

    struct alignas(64) Set {
        std::atomic<uint64_t> claim{0};        // closed bit | slots handed out | payload units handed out
        std::atomic<uint32_t> copied{0};       // slots whose payload is in place
        std::atomic<uint32_t> readers{0};      // submitters of the last flushed generation still copying out
        std::atomic<uint32_t> gen_done{0};     // last flushed generation
        std::atomic<uint32_t> helped{0};       // word-0 sleepers of the last generation that went on to wake the other words
        struct alignas(64) Word {
            std::atomic<uint32_t> v{0};
        } wake[kWakeWords];                    // futex words: copies of gen_done, the sleepers spread over them (wait())
        std::atomic<int> rc{0};
        std::atomic<bool> want_close{false};
    };

    void kick() {
        kick_.fetch_add(1);
        if (sleeping_.load()) futex_wake_all(&kick_);
    }
    // Worker side: wait until pred() holds (or `until`, if given); pred is re-evaluated after every kick.  The worker
    // polls for kSpinUs before it sleeps: under load the next event is microseconds away and a futex sleep + wake
    // costs more than that on both sides (the request threads then skip the wake system call altogether).
    template <class P>
    void sleep_until(P&& pred, const std::chrono::steady_clock::time_point* until = nullptr) {
        const auto spin_end = std::chrono::steady_clock::now() + std::chrono::microseconds(kSpinUs);
        for (;;) {
            if (pred()) return;
            const auto now = std::chrono::steady_clock::now();
            if (until && now >= *until) return;
            if (now >= spin_end) break;
            __builtin_ia32_pause();
        }
        for (;;) {
            sleeping_.store(true);
            const uint32_t k = kick_.load();
            if (pred()) break;
            if (until) {
                const auto now = std::chrono::steady_clock::now();
                if (now >= *until) break;
                const auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(*until - now).count();
                timespec ts{(time_t)(ns / 1000000000), (long)(ns % 1000000000)};
                futex_wait(&kick_, k, &ts);
            } else {
                futex_wait(&kick_, k);
            }
        }
        sleeping_.store(false);
    }

    void loop() {
        for (;;) {
            const int s = fill_.load();
            Set& S = sets_[s];
            sleep_until([&] { return stop_.load() || count(S.claim.load()) > 0; });
            if (stop_.load() && count(S.claim.load()) == 0) return;
            // linger: let the set fill, but no longer than max_delay_us after its first item.  Behind a flush of several
            // requests there is a GRACE of kGraceUs even with max_delay_us = 0: its submitters were woken a moment ago and
            // are on their way back with their next request, and a set closed on the first of them to arrive flies nearly
            // empty while the rest wait a whole flight for the next one (16 / 64 request threads over a 12.5 M-code shard:
            // 100 k / 336 k searches/s without, 164 k / 410 k with).  The grace ends as soon as as many have arrived as the
            // last flush carried; a lone sequential client never sees it.
            uint32_t delay = max_delay_us_;
            size_t enough = max_batch_;
            if (last_n_ > 1 && delay < kGraceUs) {
                delay = kGraceUs;
                enough = last_n_ < max_batch_ ? last_n_ : max_batch_;
            }
            if (delay) {
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(delay);
                sleep_until([&] { return stop_.load() || S.want_close.load() || count(S.claim.load()) >= enough; }, &deadline);
            }
            // close the set, point later submitters at the other one (open since its own flush ended)
            const uint64_t c = S.claim.fetch_or(kClosed);
            const size_t n = count(c), units = (size_t)(c & kUnitMask);
            fill_.store(s ^ 1);
            room_epoch_.fetch_add(1);
            // as many sleepers as the other set has slots: waking hundreds to compete for a few slots is a stampede.
            // The rest are woken by the closes that follow (every woken claim leads to one).
            futex_wake(&room_epoch_, (int)max_batch_);
            // the payloads of the slots handed out, and the previous generation's results picked up
            sleep_until([&] { return S.copied.load() == n && S.readers.load() == 0; });
            const int rc = flush_(s, n, units);
            S.rc.store(rc);
            S.copied.store(0);
            S.readers.store((uint32_t)n);
            S.want_close.store(false);
            batches_.fetch_add(1);
            items_.fetch_add(n);
            last_n_ = n;
            S.helped.store(0);
            const uint32_t g = S.gen_done.fetch_add(1) + 1;
            for (size_t i = 0; i < kWakeWords; i++) S.wake[i].v.store(g);
            for (size_t i = 0; i < kWakeWords; i++) futex_wake_all(&S.wake[i].v);
            S.claim.store(0);         // reopen (claims see the new generation: gen_done was published first)
        }
    }

    size_t max_batch_ = 0, max_units_ = 0;
    uint32_t max_delay_us_ = 0;
    size_t last_n_ = 0;                         // items of the most recent flush (worker thread only)
    Flush flush_;
    Set sets_[2];
    alignas(64) std::atomic<int> fill_{0};
    std::atomic<uint32_t> room_epoch_{0};       // futex word: bumped when submitters blocked on room should look again
    alignas(64) std::atomic<uint32_t> kick_{0}; // futex word: bumped when the worker should look again
    std::atomic<bool> sleeping_{false};
    std::atomic<bool> stop_{false};
    std::atomic<uint64_t> batches_{0}, items_{0};
    std::thread worker_;
};

}  // namespace ucfp
