/* bench_batcher.c -- per-request image ingest through the micro-batcher from T native threads
 * (what a tokio worker pool behind the FFI would do).  Prints frames/s for several T.
 *   gcc -O2 -std=c11 -pthread -Iinclude tools/bench_batcher.c -o /tmp/bench_batcher -Lucfp_amd -l:libucfp_hip.so \
 *       -Wl,-rpath,$PWD/ucfp_amd -Wl,-rpath,/opt/rocm/lib */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ucfp_hip.h"

enum { W = 512, H = 512 };
static ucfp_image_batcher* g_b;
static ucfp_ctx* g_ctx;
static int g_per_thread, g_direct;

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static void* worker(void* arg) {
    const int id = (int)(size_t)arg;
    uint8_t* frame = (uint8_t*)malloc(W * H);
    for (int i = 0; i < W * H; i++) frame[i] = (uint8_t)((i * 31 + id * 7) >> 3);
    uint8_t rec[UCFP_IMAGE_MULTI_BYTES];
    int32_t st;
    for (int i = 0; i < g_per_thread; i++) {
        frame[i % (W * H)] ^= 1;
        int rc = g_direct ? ucfp_image_hash_batch(g_ctx, UCFP_IMG_MULTI, frame, 1, W, H, W, (size_t)W * H,
                                                  UCFP_PIX_GRAY8, NULL, NULL, rec, &st)
                          : ucfp_image_batcher_submit(g_b, frame, W, NULL, rec, &st);
        if (rc != UCFP_OK || st != 0) {
            fprintf(stderr, "submit failed: %s\n", ucfp_last_error());
            exit(1);
        }
    }
    free(frame);
    return NULL;
}

int main(void) {
    if (ucfp_ctx_create(0, &g_ctx) != UCFP_OK) {
        fprintf(stderr, "%s\n", ucfp_last_error());
        return 1;
    }
    const int threads[] = {1, 8, 32, 128, 512};
    for (int mode = 0; mode < 2; mode++) {
        g_direct = mode == 0;
        for (size_t t = 0; t < sizeof threads / sizeof threads[0]; t++) {
            const int T = threads[t];
            if (g_direct && T > 32) continue;
            if (!g_direct && ucfp_image_batcher_create(g_ctx, UCFP_IMG_MULTI, W, H, UCFP_PIX_GRAY8, NULL, 512, 200, &g_b) != UCFP_OK) {
                fprintf(stderr, "%s\n", ucfp_last_error());
                return 1;
            }
            g_per_thread = T == 1 ? 2000 : 20000 / T + 40;
            pthread_t th[512];
            const double t0 = now();
            for (int i = 0; i < T; i++) pthread_create(&th[i], NULL, worker, (void*)(size_t)i);
            for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
            const double dt = now() - t0;
            uint64_t nb = 0, ni = 0;
            if (!g_direct) ucfp_image_batcher_stats(g_b, &nb, &ni);
            printf("{\"mode\": \"%s\", \"threads\": %d, \"frames_per_s\": %.0f, \"avg_batch\": %.1f}\n",
                   g_direct ? "direct ucfp_image_hash_batch(n=1)" : "micro-batcher", T,
                   (double)T * g_per_thread / dt, nb ? (double)ni / nb : 1.0);
            fflush(stdout);
            if (!g_direct) ucfp_image_batcher_destroy(g_b);
        }
    }
    ucfp_ctx_destroy(g_ctx);
    return 0;
}
