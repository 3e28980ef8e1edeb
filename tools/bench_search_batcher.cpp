// bench_search_batcher.cpp -- /v1/query's request shape through the search micro-batcher, from native threads: T request
// threads, ONE query per call (src/server/handlers.rs:143-187), over a Hamming shard of n codes.  Host memory in and out.
// Every answer of a sample of requests is checked against a brute-force scan on the host.
//
//   g++ -O2 -std=c++17 tools/bench_search_batcher.cpp -Iinclude -Lucfp_amd -lucfp_hip -Wl,-rpath,$PWD/ucfp_amd -lpthread -o /tmp/bench_search_batcher
//   /tmp/bench_search_batcher [--n=12500000] [--k=10] [--delay=US] [--batch=256] [threads ...]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ucfp_hip.h"

static uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    size_t n = 12500000, max_batch = 256;
    uint32_t k = 10, delay_us = 0;
    std::vector<int> tcs;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "--n=", 4)) n = (size_t)atoll(argv[i] + 4);
        else if (!strncmp(argv[i], "--k=", 4)) k = (uint32_t)atoi(argv[i] + 4);
        else if (!strncmp(argv[i], "--delay=", 8)) delay_us = (uint32_t)atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--batch=", 8)) max_batch = (size_t)atoll(argv[i] + 8);
        else tcs.push_back(atoi(argv[i]));
    }
    if (tcs.empty()) tcs = {16, 64, 256};
    ucfp_ctx* ctx = nullptr;
    if (ucfp_ctx_create(0, &ctx)) {
        fprintf(stderr, "ctx: %s\n", ucfp_last_error());
        return 1;
    }
    ucfp_index* ix = nullptr;
    if (ucfp_index_create(ctx, UCFP_INDEX_HAMMING64, 0, UCFP_INDEX_APPEND_ONLY, &ix)) {
        fprintf(stderr, "index: %s\n", ucfp_last_error());
        return 1;
    }
    uint64_t seed = 0x5EED;
    std::vector<uint64_t> codes(n), ids(n);
    for (size_t i = 0; i < n; i++) {
        codes[i] = sm64(seed);
        ids[i] = i;
    }
    const size_t nq = 4096;
    std::vector<uint64_t> queries(nq);
    for (size_t j = 0; j < nq; j++) {
        queries[j] = sm64(seed);
        if (j % 2 == 0) codes[(j * 2654435761ull) % n] = queries[j] ^ (1ull << (j % 61)) ^ (1ull << ((j * 3) % 59));     // a planted neighbour
    }
    if (ucfp_index_upsert(ix, 0, ids.data(), codes.data(), n)) {
        fprintf(stderr, "upsert: %s\n", ucfp_last_error());
        return 1;
    }
    // reference answers of a sample of the queries (brute force on the host)
    const size_t ns = 24;
    std::vector<std::vector<std::pair<uint32_t, uint64_t>>> want(ns);
    for (size_t s = 0; s < ns; s++) {
        const uint64_t q = queries[s * 170];
        std::vector<std::pair<uint32_t, uint64_t>> best;
        uint32_t worst = 65;
        for (size_t i = 0; i < n; i++) {
            const uint32_t d = (uint32_t)__builtin_popcountll(q ^ codes[i]);
            if (best.size() < k || d < worst) {
                best.emplace_back(d, ids[i]);
                if (best.size() >= 4 * (size_t)k) {
                    std::sort(best.begin(), best.end());
                    best.resize(k);
                    worst = best.back().first + 1;      // ids ascend with the row: a later row never wins a tie
                }
            }
        }
        std::sort(best.begin(), best.end());
        best.resize(std::min<size_t>(best.size(), k));
        want[s] = best;
    }
    for (int T : tcs) {
        ucfp_search_batcher* b = nullptr;
        if (ucfp_index_search_batcher_create(ix, 0, max_batch, delay_us, &b)) {
            fprintf(stderr, "batcher: %s\n", ucfp_last_error());
            return 1;
        }
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0}, wrong{0};
        auto worker = [&](size_t total) {
            std::vector<uint64_t> o_ids(k);
            std::vector<uint32_t> o_d(k);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= total) return;
                const size_t j = i % nq;
                uint32_t cnt = 0;
                if (ucfp_index_search_batcher_submit(b, &queries[j], k, o_ids.data(), nullptr, o_d.data(), &cnt)) {
                    bad++;
                    continue;
                }
                if (j % 170 == 0 && j / 170 < ns) {
                    const auto& w = want[j / 170];
                    bool ok = cnt == w.size();
                    for (size_t r = 0; ok && r < w.size(); r++) ok = o_d[r] == w[r].first && o_ids[r] == w[r].second;
                    if (!ok) wrong++;
                }
            }
        };
        auto run = [&](size_t total) {
            next = 0;
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back(worker, total);
            for (auto& x : th) x.join();
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        run(4096);
        uint64_t b0 = 0, i0 = 0, b1 = 0, i1 = 0;
        ucfp_index_search_batcher_stats(b, &b0, &i0);
        const size_t total = 200000;
        const double dt = run(total);
        ucfp_index_search_batcher_stats(b, &b1, &i1);
        printf("{\"batcher\": \"search\", \"kind\": \"hamming64\", \"codes\": %zu, \"k\": %u, \"threads\": %d, \"max_batch\": %zu, "
               "\"max_delay_us\": %u, \"queries_per_s\": %.0f, \"avg_batch\": %.1f, \"us_per_request\": %.1f, \"failed\": %d, "
               "\"checked_answers_wrong\": %d}\n",
               n, k, T, max_batch, delay_us, total / dt, (double)(i1 - i0) / (double)(b1 - b0 ? b1 - b0 : 1), dt / total * T * 1e6,
               bad.load(), wrong.load());
        fflush(stdout);
        ucfp_index_search_batcher_destroy(b);
    }
    ucfp_index_destroy(ix);
    ucfp_ctx_destroy(ctx);
    return 0;
}
