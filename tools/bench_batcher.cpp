// bench_batcher.cpp -- the reference's per-request shape through the micro-batchers, from native threads
// (no interpreter lock in the way): T threads, each submits one item per call, like the tokio workers of
// src/server/handlers.rs (up to 512 requests in flight, src/bin/ucfp.rs:267).  Host memory in, host memory out:
// these are PCIe-inclusive, per-request rates -- never bench.py's headline value.
//
//   g++ -O2 -std=c++17 tools/bench_batcher.cpp -Iinclude -Lucfp_amd -lucfp_hip -Wl,-rpath,$PWD/ucfp_amd -lpthread -o /tmp/bench_batcher
//   /tmp/bench_batcher [--delay=US] [threads ...]          (default: no linger; 32 128 512 threads)
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ucfp_hip.h"

static uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <class F>
static double run_threads(int threads, size_t total, F&& one) {
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= total) return;
                if (one(i) != 0) bad++;
            }
        });
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (bad) fprintf(stderr, "%d submissions failed: %s\n", bad.load(), ucfp_last_error());
    return dt;
}

int main(int argc, char** argv) {
    std::vector<int> tcs;
    uint32_t delay_us = 0;            // 0: flush as soon as the previous flush has returned (the batch follows the load)
    const char* png_dir = nullptr;    // --png=DIR: 0.png .. 63.png (python tools/bench_png.py --dump DIR)
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "--delay=", 8)) delay_us = (uint32_t)atoi(argv[i] + 8);
        else if (!strncmp(argv[i], "--png=", 6)) png_dir = argv[i] + 6;
        else tcs.push_back(atoi(argv[i]));
    }
    if (tcs.empty()) tcs = {32, 128, 512};
    ucfp_ctx* ctx = nullptr;
    if (ucfp_ctx_create(0, &ctx)) {
        fprintf(stderr, "ctx: %s\n", ucfp_last_error());
        return 1;
    }
    uint64_t seed = 1;

    // ---- text: 4 KiB ASCII documents (benches/end_to_end.rs:24-38 shape), MinHash k = 5 and SimHash ----
    static const char* words[] = {"the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "pipeline", "inspector",
                                  "fingerprint", "42", "e.g.", "don't", "U.S.A.", "x"};
    std::vector<std::string> docs(512);
    for (auto& d : docs) {
        while (d.size() < 4096) {
            d += words[sm64(seed) & 15];
            d += (sm64(seed) & 7) ? " " : ". ";
        }
        d.resize(4096);
    }
    for (int algo : {UCFP_TEXT_ALGO_MINHASH, UCFP_TEXT_ALGO_SIMHASH}) {
        for (int T : tcs) {
            ucfp_text_batcher* b = nullptr;
            if (ucfp_text_batcher_create(ctx, algo, UCFP_TEXT_RAW_ASCII, 5, 4096, 32u << 20, delay_us, &b)) {
                fprintf(stderr, "text batcher: %s\n", ucfp_last_error());
                return 1;
            }
            auto one = [&](size_t i) {
                uint8_t out[UCFP_MINHASH_BYTES];
                int32_t st = 0;
                const std::string& d = docs[i & 511];
                int rc = ucfp_text_batcher_submit(b, (const uint8_t*)d.data(), d.size(), out, &st);
                return rc ? rc : st;
            };
            run_threads(T, 4096, one);
            uint64_t b0 = 0, i0 = 0, b1 = 0, i1 = 0;
            ucfp_text_batcher_stats(b, &b0, &i0);
            const size_t total = 200000;
            const double dt = run_threads(T, total, one);
            ucfp_text_batcher_stats(b, &b1, &i1);
            printf("{\"batcher\": \"text\", \"algo\": \"%s\", \"doc_bytes\": 4096, \"threads\": %d, \"max_delay_us\": %u, \"docs_per_s\": %.0f, "
                   "\"avg_batch\": %.1f, \"us_per_request\": %.1f}\n",
                   algo == UCFP_TEXT_ALGO_MINHASH ? "minhash-h128" : "simhash-b64", T, delay_us,
                   total / dt, (double)(i1 - i0) / (double)(b1 - b0 ? b1 - b0 : 1), dt / total * T * 1e6);
            fflush(stdout);
            ucfp_text_batcher_destroy(b);
        }
    }

    // ---- image: 256 x 256 gray frames (BASELINE config 1's decoded size), MULTI record ----
    {
        std::vector<uint8_t> frames((size_t)64 * 256 * 256);
        for (size_t i = 0; i < frames.size(); i++) frames[i] = (uint8_t)(((i & 255) + (i >> 8)) ^ (sm64(seed) >> 60));
        const uint32_t algo = 7;
        const size_t rec = ucfp_image_record_bytes(algo);
        for (int T : tcs) {
            ucfp_image_batcher* b = nullptr;
            if (ucfp_image_batcher_create(ctx, algo, 256, 256, UCFP_PIX_GRAY8, nullptr, 1024, delay_us, &b)) {
                fprintf(stderr, "image batcher: %s\n", ucfp_last_error());
                return 1;
            }
            auto one = [&](size_t i) {
                uint8_t out[UCFP_IMAGE_MULTI_BYTES];
                int32_t st = 0;
                int rc = ucfp_image_batcher_submit(b, frames.data() + (i & 63) * 65536, 256, nullptr, out, &st);
                (void)rec;
                return rc ? rc : st;
            };
            run_threads(T, 2048, one);
            uint64_t b0 = 0, i0 = 0, b1 = 0, i1 = 0;
            ucfp_image_batcher_stats(b, &b0, &i0);
            const size_t total = 100000;
            const double dt = run_threads(T, total, one);
            ucfp_image_batcher_stats(b, &b1, &i1);
            printf("{\"batcher\": \"image\", \"frame\": \"256x256 gray\", \"threads\": %d, \"frames_per_s\": %.0f, "
                   "\"avg_batch\": %.1f, \"us_per_request\": %.1f}\n",
                   T, total / dt, (double)(i1 - i0) / (double)(b1 - b0 ? b1 - b0 : 1), dt / total * T * 1e6);
            fflush(stdout);
            ucfp_image_batcher_destroy(b);
        }
    }

    // ---- audio: 4 s clips at 8 kHz (benches/end_to_end.rs:55-75) ----
    {
        const size_t n = 32000;
        std::vector<float> clips(16 * n);
        for (size_t c = 0; c < 16; c++)
            for (size_t i = 0; i < n; i++)
                clips[c * n + i] = 0.3f * sinf(6.2831853f * (300.0f + 40.0f * c) * (float)i / 8000.0f * (1.0f + (float)i / n)) +
                                   0.02f * (float)((int)(sm64(seed) >> 56) - 128) / 128.0f;
        for (int T : tcs) {
            ucfp_audio_batcher* b = nullptr;
            if (ucfp_audio_batcher_create(ctx, 8000, nullptr, 1024, 1024 * n, delay_us, &b)) {
                fprintf(stderr, "audio batcher: %s\n", ucfp_last_error());
                return 1;
            }
            const size_t cap = ucfp_audio_wang_batch_max_hashes(n, 1, 8000, nullptr);
            auto one = [&](size_t i) {
                std::vector<uint8_t> out(cap * 8 + 8);
                size_t got = 0;
                uint8_t* o = out.data() + ((8 - ((uintptr_t)out.data() & 7)) & 7);
                return ucfp_audio_batcher_submit(b, clips.data() + (i & 15) * n, n, o, cap, &got);
            };
            run_threads(T, 1024, one);
            uint64_t b0 = 0, i0 = 0, b1 = 0, i1 = 0;
            ucfp_audio_batcher_stats(b, &b0, &i0);
            const size_t total = 40000;
            const double dt = run_threads(T, total, one);
            ucfp_audio_batcher_stats(b, &b1, &i1);
            printf("{\"batcher\": \"audio\", \"clip\": \"4 s @ 8 kHz\", \"threads\": %d, \"clips_per_s\": %.0f, "
                   "\"avg_batch\": %.1f, \"us_per_request\": %.1f}\n",
                   T, total / dt, (double)(i1 - i0) / (double)(b1 - b0 ? b1 - b0 : 1), dt / total * T * 1e6);
            fflush(stdout);
            ucfp_audio_batcher_destroy(b);
        }
    }
    // ---- encoded uploads: 256 x 256 RGB PNG files (BASELINE config 1), decoded + BLAKE3-hashed + fingerprinted on the device ----
    if (png_dir) {
        std::vector<std::vector<uint8_t>> files;
        for (int i = 0; i < 64; i++) {
            char path[512];
            snprintf(path, sizeof path, "%s/%d.png", png_dir, i);
            FILE* f = fopen(path, "rb");
            if (!f) break;
            std::vector<uint8_t> b;
            uint8_t buf[65536];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + got);
            fclose(f);
            files.push_back(std::move(b));
        }
        if (files.empty()) {
            fprintf(stderr, "no PNG files under %s (python tools/bench_png.py --dump DIR writes them)\n", png_dir);
        } else {
            uint32_t w = 0, h = 0;
            int fmt = 0;
            ucfp_png_probe(files[0].data(), files[0].size(), &w, &h, &fmt);
            for (int T : tcs) {
                ucfp_png_batcher* b = nullptr;
                if (ucfp_png_batcher_create(ctx, 2 /* pHash */, w, h, fmt, nullptr, 1024, 256u << 20, delay_us, &b)) {
                    fprintf(stderr, "png batcher: %s\n", ucfp_last_error());
                    return 1;
                }
                auto one = [&](size_t i) {
                    uint8_t out[UCFP_IMAGE_MULTI_BYTES];
                    int32_t st = 0;
                    const std::vector<uint8_t>& f = files[i % files.size()];
                    int rc = ucfp_png_batcher_submit(b, f.data(), f.size(), out, &st);
                    return rc ? rc : st;
                };
                run_threads(T, 1024, one);
                uint64_t b0 = 0, i0 = 0, b1 = 0, i1 = 0;
                ucfp_png_batcher_stats(b, &b0, &i0);
                const size_t total = 40000;
                const double dt = run_threads(T, total, one);
                ucfp_png_batcher_stats(b, &b1, &i1);
                printf("{\"batcher\": \"png\", \"file\": \"%ux%u PNG, %zu bytes\", \"threads\": %d, \"files_per_s\": %.0f, "
                       "\"avg_batch\": %.1f, \"us_per_request\": %.1f}\n",
                       w, h, files[0].size(), T, total / dt, (double)(i1 - i0) / (double)(b1 - b0 ? b1 - b0 : 1), dt / total * T * 1e6);
                fflush(stdout);
                ucfp_png_batcher_destroy(b);
            }
        }
    }
    ucfp_ctx_destroy(ctx);
    return 0;
}
