/*
 * ucfp_hip.h -- C ABI of the MI355X (gfx950) fingerprint + brute-force ANN core.
 *
 * This is the drop-in boundary for the UCFP hot path.  The reference has no FFI
 * today (it calls three crates.io SDKs in-process); each entry point below names
 * the Rust seam it replaces, as /root/reference/<file>:<line>.  A Rust host binds
 * these with `extern "C"`; this repo's Python host binds them
 * with ctypes (ucfp_amd/_lib.py).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - `*_dev` entry points take DEVICE pointers and a hipStream_t (as void*); they
 *     enqueue work and return without synchronising.  The non-`_dev` variants take
 *     HOST pointers, stage through the context's device workspace and block until
 *     the result is in the caller's buffer (they mirror the per-request reference
 *     call: borrowed input, owned output copy -- src/modality/image.rs:82).
 *   - return value: 0 (UCFP_OK) or a negative ucfp_status.  Per-item failures of a
 *     batch are reported in `status[i]` and do not fail the call.
 *   - a context is thread-safe for concurrent calls on distinct streams; the error
 *     string is per-thread.
 */
#ifndef UCFP_HIP_H
#define UCFP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCFP_ABI_VERSION 1

/* ---- status codes: map 1:1 onto the reference's Error enum (src/error.rs:9-61,
 *      HTTP mapping src/server/error.rs:22-41) ------------------------------------ */
typedef enum ucfp_status {
    UCFP_OK = 0,
    UCFP_E_MODALITY = -1,    /* Error::Modality  -> 400: bad input for the algorithm  */
    UCFP_E_UNSUPPORTED = -2, /* Error::Unsupported -> 501: algorithm/option not built */
    UCFP_E_INDEX = -3,       /* Error::Index -> 500: device / runtime failure         */
    UCFP_E_INVALID = -4,     /* null pointer, bad enum, size overflow (caller bug)    */
    UCFP_E_NOT_FOUND = -5    /* Error::RecordNotFound -> 404                          */
} ucfp_status;

typedef struct ucfp_ctx ucfp_ctx;

/* Create a context bound to HIP device `device_id` (one process per GPU: the caller
 * passes LOCAL_RANK).  Fails with UCFP_E_INDEX when no gfx950 device is usable --
 * there is deliberately no CPU fallback. */
int ucfp_ctx_create(int device_id, ucfp_ctx** out);
void ucfp_ctx_destroy(ucfp_ctx* ctx);
/* Thread-local UTF-8 message of the last failing call (the `String` payload of
 * Error::Modality / Error::Index). Never NULL. */
const char* ucfp_last_error(void);
int ucfp_abi_version(void);

/* =============================== IMAGE ========================================
 * Replaces the arithmetic behind
 *   image::fingerprint_with            src/modality/image.rs:62-88   (multi, 536 B)
 *   image::fingerprint_{p,d,a}hash     src/modality/image.rs:112-194 (single, 168 B)
 * i.e. imgfprint::ImageFingerprinter::fingerprint_with_preprocess and
 * FingerprinterContext::fingerprint_with_algorithm_and_preprocess, AFTER decode.
 * Input is a batch of decoded frames of one geometry.
 */
typedef enum ucfp_pixfmt {
    UCFP_PIX_GRAY8 = 0, /* 1 byte / pixel, luma                                  */
    UCFP_PIX_RGB8 = 1,  /* 3 bytes / pixel, R,G,B                                */
    UCFP_PIX_RGBA8 = 2  /* 4 bytes / pixel, alpha ignored                        */
} ucfp_pixfmt;

typedef enum ucfp_image_algo {
    UCFP_IMG_AHASH = 1, /* ?algorithm=ahash  tag imgfprint-ahash-v1              */
    UCFP_IMG_PHASH = 2, /* ?algorithm=phash  tag imgfprint-phash-v1              */
    UCFP_IMG_DHASH = 4, /* ?algorithm=dhash  tag imgfprint-dhash-v1              */
    UCFP_IMG_MULTI = 7  /* ?algorithm=multi  tag imgfprint-multihash-v1          */
} ucfp_image_algo;

#define UCFP_IMAGE_FP_BYTES 168    /* imgfprint::ImageFingerprint, repr(C)        */
#define UCFP_IMAGE_MULTI_BYTES 536 /* imgfprint::MultiHashFingerprint, repr(C)    */
#define UCFP_IMAGE_NORM 256        /* side of the normalised luma plane           */

/* imgfprint::PreprocessConfig guards (defaults: src/server/algorithms_manifest.rs:446-469;
 * query mapping src/server/handlers.rs:307-319).  max_input_bytes concerns the ENCODED
 * payload and is enforced by the host before decode. */
typedef struct ucfp_image_preprocess {
    uint32_t max_dimension; /* default 8192 */
    uint32_t min_dimension; /* default 32   */
} ucfp_image_preprocess;

/* Bytes of one output record for `algo` (168 for a single algorithm, 536 for MULTI;
 * 0 for an invalid mask). */
size_t ucfp_image_record_bytes(uint32_t algo);

/* Device-resident batch.
 *   frames      n frames; frame i starts at frames + i*frame_stride, row y at
 *               + y*row_stride; pixels packed per `pixfmt`. 16-byte aligned base and
 *               strides take the vectorised path.
 *   exact       n x 32 bytes: BLAKE3 of each ORIGINAL encoded image, computed by the
 *               host that still has those bytes (the reference hashes the upload,
 *               not the pixels); NULL writes zeros.
 *   out         n x ucfp_image_record_bytes(algo), layout exactly the reference's
 *               bytemuck::bytes_of(&fp) (image.rs:82,188).
 *   status      n x int32 (may be NULL): 0 or UCFP_E_MODALITY per frame.
 * Geometry violations of `pre` fail every frame of the batch (all share w,h). */
int ucfp_image_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n,
                              uint32_t width, uint32_t height, size_t row_stride,
                              size_t frame_stride, int pixfmt,
                              const ucfp_image_preprocess* pre, const uint8_t* exact,
                              uint8_t* out, int32_t* status, void* stream);

/* Host-pointer variant (per-request path: n is usually 1). Same arguments, host memory. */
int ucfp_image_hash_batch(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t n,
                          uint32_t width, uint32_t height, size_t row_stride,
                          size_t frame_stride, int pixfmt, const ucfp_image_preprocess* pre,
                          const uint8_t* exact, uint8_t* out, int32_t* status);

/* Synthetic workload of SURVEY 8(d) config 2, generated on device: frame i, pixel (x,y) =
 * ((x + y + 17*i) & 255) ^ (splitmix64((i*h + y)*w + x) >> 60): a ramp with 4 bits of seeded
 * noise. Deterministic; the oracle has the same generator. Bench/test support only. */
int ucfp_image_synth_dev(ucfp_ctx* ctx, uint8_t* frames, size_t n, uint32_t width,
                         uint32_t height, size_t first_index, void* stream);

/* BLAKE3-256 (default hash mode) of a HOST buffer: the `exact` digest the reference stores in
 * ImageFingerprint.exact (BLAKE3 of the uploaded bytes). Host code; no device needed. */
int ucfp_blake3(const uint8_t* data, size_t len, uint8_t out[32]);

#ifdef __cplusplus
}
#endif
#endif /* UCFP_HIP_H */
