/* abi_smoke.c -- the C ABI used from plain C11, the way a cgo / Rust-FFI / JNI caller would.
 * Built and run by tests/test_c_driver.py.  `host` mode needs no GPU; `gpu` mode hashes one
 * synthetic frame and one document and prints hex for the Python side to compare. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ucfp_hip.h"

static void hex(const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) printf("%02x", p[i]);
    printf("\n");
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "host";
    if (ucfp_abi_version() != UCFP_ABI_VERSION) return 2;
    if (ucfp_image_record_bytes(UCFP_IMG_MULTI) != UCFP_IMAGE_MULTI_BYTES) return 3;
    uint8_t dig[32];
    if (ucfp_blake3((const uint8_t*)"abc", 3, dig) != UCFP_OK) return 4;
    hex(dig, 32);
    if (strcmp(mode, "host") == 0) {
        ucfp_ctx* ctx = NULL;
        int rc = ucfp_ctx_create(0, &ctx);
        printf("ctx_create rc=%d msg=%s\n", rc, ucfp_last_error());
        if (rc == UCFP_OK) ucfp_ctx_destroy(ctx);
        return 0;
    }
    ucfp_ctx* ctx = NULL;
    if (ucfp_ctx_create(0, &ctx) != UCFP_OK) {
        fprintf(stderr, "%s\n", ucfp_last_error());
        return 5;
    }
    /* one 512x512 GRAY8 frame, the reference's ramp: (x + y) & 255 */
    enum { W = 512, H = 512 };
    uint8_t* frame = (uint8_t*)malloc(W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) frame[y * W + x] = (uint8_t)((x + y) & 255);
    uint8_t rec[UCFP_IMAGE_MULTI_BYTES];
    int32_t st = 99;
    ucfp_image_preprocess pre = {8192, 32};
    if (ucfp_image_hash_batch(ctx, UCFP_IMG_MULTI, frame, 1, W, H, W, (size_t)W * H, UCFP_PIX_GRAY8, &pre, dig, rec,
                              &st) != UCFP_OK || st != 0) {
        fprintf(stderr, "image: %s\n", ucfp_last_error());
        return 6;
    }
    hex(rec, sizeof rec);
    const char* doc = "the quick brown fox jumps over the lazy dog";
    uint64_t offs[2] = {0, strlen(doc)};
    uint8_t mh[UCFP_MINHASH_BYTES];
    if (ucfp_text_minhash_batch(ctx, (const uint8_t*)doc, offs, 1, UCFP_TEXT_RAW_ASCII, 5, mh, &st) != UCFP_OK || st != 0)
        return 7;
    hex(mh, 16);
    /* a tiny Hamming index */
    ucfp_index* ix = NULL;
    if (ucfp_index_create(ctx, UCFP_INDEX_HAMMING64, 0, 0, &ix) != UCFP_OK) return 8;
    uint64_t ids[4] = {10, 20, 30, 40}, codes[4] = {0xff, 0xf0, 0x0f, 0x00}, q = 0xf1, out_ids[2];
    float sc[2];
    uint32_t d[2], cnt;
    if (ucfp_index_upsert(ix, 1, ids, codes, 4) != UCFP_OK) return 9;
    if (ucfp_index_search(ix, 1, &q, 1, 2, out_ids, sc, d, &cnt) != UCFP_OK) return 10;
    printf("knn %llu:%u %llu:%u n=%u\n", (unsigned long long)out_ids[0], d[0], (unsigned long long)out_ids[1], d[1], cnt);
    ucfp_index_destroy(ix);
    ucfp_ctx_destroy(ctx);
    free(frame);
    return 0;
}
