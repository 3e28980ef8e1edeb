// Stress of ucfp_amd/csrc/batch_core.h without a GPU: T request threads push items through a BatchCore whose flush
// callback sums each item's payload on the worker thread.  Every submitter must get ITS sum back, no set may exceed
// its limits, and the run must be clean under -fsanitize=thread (tests/test_batch_core.py builds it that way).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ucfp_amd/csrc/batch_core.h"

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 24;
    const size_t per_thread = argc > 2 ? (size_t)atoi(argv[2]) : 400;
    const uint32_t delay_us = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
    const size_t max_batch = 16, max_units = 600;
    std::vector<uint32_t> in[2] = {std::vector<uint32_t>(max_units), std::vector<uint32_t>(max_units)};
    std::vector<uint64_t> off[2] = {std::vector<uint64_t>(max_batch + 1), std::vector<uint64_t>(max_batch + 1)};
    std::vector<uint64_t> out[2] = {std::vector<uint64_t>(max_batch), std::vector<uint64_t>(max_batch)};
    std::atomic<uint64_t> flushed{0}, over{0};
    ucfp::BatchCore core;
    core.start(max_batch, max_units, delay_us, [&](int s, size_t n, size_t units) {
        if (n > max_batch || units > max_units || n == 0) over++;
        off[s][n] = units;
        for (size_t i = 0; i < n; i++) {
            uint64_t sum = 0;
            for (uint64_t j = off[s][i]; j < off[s][i + 1]; j++) sum += in[s][j];
            out[s][i] = sum;
        }
        flushed += n;
        return 0;
    });
    std::atomic<uint64_t> wrong{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            uint64_t x = 88172645463325252ull + (uint64_t)t * 7919;
            for (size_t r = 0; r < per_thread; r++) {
                x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                const size_t units = (size_t)(x % 200);          // 0 .. 199 units: three or more items fill a set
                ucfp::BatchCore::Ticket k;
                if (!core.claim(units, &k)) {
                    wrong++;
                    return;
                }
                off[k.set][k.slot] = k.at;
                uint64_t want = 0;
                for (size_t j = 0; j < units; j++) {
                    in[k.set][k.at + j] = (uint32_t)(x + j * 31 + t);
                    want += (uint32_t)(x + j * 31 + t);
                }
                core.commit(k);
                if (core.wait(k) != 0) wrong++;
                if (out[k.set][k.slot] != want) wrong++;
                core.release(k);
            }
        });
    for (auto& x : th) x.join();
    uint64_t batches = 0, items = 0;
    core.stats(&batches, &items);
    core.stop();
    const uint64_t total = (uint64_t)threads * per_thread;
    printf("items %llu flushed %llu batches %llu wrong %llu over %llu\n", (unsigned long long)items,
           (unsigned long long)flushed.load(), (unsigned long long)batches, (unsigned long long)wrong.load(),
           (unsigned long long)over.load());
    return (items == total && flushed == total && wrong == 0 && over == 0 && batches < total) ? 0 : 1;
}
