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

#define UCFP_ABI_VERSION 2

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

/* 64-bit global hashes of n stored image records -> Hamming codes for ucfp_index_append_dev (byte
 * offset 32 of a 168-byte record, 32 + {32, 200, 368} of the 536-byte bundle).  `algo` names the record type,
 * `which` (AHASH / PHASH / DHASH) the member of a MULTI bundle; ignored otherwise. */
int ucfp_image_record_codes_dev(ucfp_ctx* ctx, const uint8_t* d_records, size_t n, uint32_t algo, uint32_t which,
                                uint64_t* d_codes, void* stream);

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

/* RAGGED batch of decoded frames: every frame has its own geometry -- the reference's image route takes any upload
 * (src/server/handlers.rs:232-302 -> src/modality/image.rs:54-88: "PNG / JPEG / WebP / GIF / BMP", any size), so what a
 * server hands over after decode is a mix of sizes.  Frame i is described by items[i] (a HOST array; the library plans
 * the launch from it): its first pixel sits `offset` bytes into `frames`, rows `row_stride` bytes apart, `pixfmt` pixels.
 * One launch per form of row (up to / beyond 512 pixels) hashes the whole batch: a workgroup takes a frame from source
 * bytes to record with the 256 x 256 normalised plane kept in registers and LDS, never in memory.  Outputs as
 * ucfp_image_hash_batch_dev, indexed like `items`; a frame outside the guards of `pre` (or of zero size) gets
 * status[i] = UCFP_E_MODALITY and a zero record, the others are unaffected.  frames_bytes: the readable bytes behind
 * `frames` (the unaligned loader reads whole dwords, never outside them). */
typedef struct ucfp_image_item {
    uint64_t offset;     /* first pixel of the frame, bytes from `frames`                 */
    uint32_t width, height;
    uint32_t row_stride; /* bytes; >= width x bytes per pixel                              */
    int32_t pixfmt;      /* ucfp_pixfmt                                                    */
} ucfp_image_item;
int ucfp_image_hash_ragged_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_frames, size_t frames_bytes,
                               const ucfp_image_item* items, size_t n, const ucfp_image_preprocess* pre, const uint8_t* d_exact,
                               uint8_t* d_out, int32_t* d_status, void* stream);
/* Host-pointer variant: frames / exact / out / status in host memory; blocks until the records are in `out`. */
int ucfp_image_hash_ragged(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t frames_bytes, const ucfp_image_item* items,
                           size_t n, const ucfp_image_preprocess* pre, const uint8_t* exact, uint8_t* out, int32_t* status);

/* ---- encoded uploads of any size and format in one batch (SURVEY 8f N1 + N4) ----
 * What the reference's route receives (src/server/handlers.rs:232-302) is an encoded file of unknown kind and size, decoded
 * inside the SDK call (src/modality/image.rs:68-70).  ucfp_image_probe tells from the first bytes what it is and what frame
 * it decodes to; the batch entries below take files of ANY mix of kinds and sizes: PNG and baseline JPEG are decoded on
 * the device (scope: the PNG / JPEG front-end sections below), everything else -- WebP, GIF, BMP, the PNG and JPEG kinds the
 * device hands back -- gets status UCFP_IMAGE_NEEDS_HOST and goes to the host's decoder. */
#define UCFP_IMAGE_NEEDS_HOST 1
typedef enum ucfp_upload_format {
    UCFP_UPLOAD_OTHER = 0, /* not PNG / JPEG: the host's decoder decides                    */
    UCFP_UPLOAD_PNG = 1,
    UCFP_UPLOAD_JPEG = 2
} ucfp_upload_format;
typedef struct ucfp_upload_info {
    int32_t format;        /* ucfp_upload_format                                             */
    int32_t status;        /* UCFP_OK: the device decodes it; UCFP_IMAGE_NEEDS_HOST; UCFP_E_MODALITY */
    uint32_t width, height;
    int32_t pixfmt;        /* the frame the file decodes to (a JPEG: its luma plane, GRAY8)   */
    uint32_t reserved;
} ucfp_upload_info;
/* Host-side, on the upload's own bytes (a request thread calls it before submitting): fills *info, returns info->status. */
int ucfp_image_probe(const uint8_t* bytes, size_t len, ucfp_upload_info* info);
/* The same for uploads that are already in device memory (one blob + n + 1 byte offsets, like the text calls): d_info
 * receives n entries.  No synchronisation. */
int ucfp_image_probe_batch_dev(ucfp_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, ucfp_upload_info* d_info,
                               void* stream);
/* Encoded uploads of ANY mix of kinds and sizes -> records.  info: n probe results in HOST memory (the library plans the
 * launches and sizes its workspace from them; every file's own header is checked against its entry on the device, a
 * mismatch is UCFP_IMAGE_NEEDS_HOST) -- or NULL: the files are probed on the device first, which costs one host
 * synchronisation inside the call.  PNG files are inflated and unfiltered, JPEG files Huffman-decoded and inverse-
 * transformed (luma plane) into frames of their own sizes in the context's workspace, BLAKE3 of every file is computed
 * when d_exact is NULL, and ONE ragged hash (ucfp_image_hash_ragged_dev's kernels) makes the records.  status[i]: 0,
 * UCFP_IMAGE_NEEDS_HOST (not PNG / JPEG, or a kind of them the device hands back, or a decode irregularity: the host's
 * decoder decides), UCFP_E_MODALITY (damaged file; geometry outside `pre`); records of files with a non-zero status are
 * zero.  blob_bytes = d_offsets[n]. */
int ucfp_image_upload_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n,
                                     size_t blob_bytes, const ucfp_upload_info* info, const ucfp_image_preprocess* pre,
                                     const uint8_t* d_exact, uint8_t* d_out, int32_t* d_status, void* stream);
/* Decode only: frames into the caller's buffer (16-byte aligned, ucfp_image_upload_frames_bytes(info, n) bytes), laid out
 * by the library; items[i] (HOST, n entries) receives where frame i is and its geometry (width 0: not decoded) -- ready for
 * ucfp_image_hash_ragged_dev. */
size_t ucfp_image_upload_frames_bytes(const ucfp_upload_info* info, size_t n);
int ucfp_image_upload_decode_batch_dev(ucfp_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, size_t blob_bytes,
                                       const ucfp_upload_info* info, uint8_t* d_frames, size_t frames_bytes, ucfp_image_item* items,
                                       int32_t* d_status, void* stream);

/* Host micro-batcher for uploads of ANY kind and size (SURVEY 8f N1 + N4): the per-request shape of handlers::ingest_image
 * (src/server/handlers.rs:232-302: one upload per request thread, up to 512 in flight, src/bin/ucfp.rs:267) with NO geometry
 * or format announced at creation.  submit() is BLOCKING and thread-safe: the calling thread probes its own bytes
 * (ucfp_image_probe); what the device does not decode returns at once with *status = UCFP_IMAGE_NEEDS_HOST / UCFP_E_MODALITY
 * and a zero record; everything else -- PNG and JPEG files of whatever sizes -- is coalesced with the other threads'
 * uploads into ONE H2D copy + ucfp_image_upload_hash_batch_dev (decode, BLAKE3 of the file, ragged hash) + one D2H copy
 * of the records, at most max_batch uploads / max_bytes encoded bytes per flush, flushed no later than max_delay_us after
 * the first pending upload.  Inside, PNG and JPEG uploads coalesce in two lanes (each with max_batch / max_bytes of its own;
 * the JPEG lane on a context the batcher creates for itself), so that a JPEG request -- a few milliseconds of decode -- does
 * not wait for the largest PNG of a shared flush -- tens of milliseconds, its LZ77 pass being one wave's serial work. */
typedef struct ucfp_upload_batcher ucfp_upload_batcher;
int ucfp_upload_batcher_create(ucfp_ctx* ctx, uint32_t algo, const ucfp_image_preprocess* pre, size_t max_batch, size_t max_bytes,
                               uint32_t max_delay_us, ucfp_upload_batcher** out);
void ucfp_upload_batcher_destroy(ucfp_upload_batcher* b);
int ucfp_upload_batcher_submit(ucfp_upload_batcher* b, const uint8_t* bytes, size_t len, uint8_t* out, int32_t* status);
int ucfp_upload_batcher_stats(ucfp_upload_batcher* b, uint64_t* batches, uint64_t* items);

/* ---- PNG front end ----
 * The reference decodes the upload inside the SDK call (src/modality/image.rs:68-70, :176-179: imgfprint ->
 * image::load_from_memory); BASELINE config 1 (1 k 256x256 PNGs) is decode-bound on the CPU.  These entry points take
 * the ENCODED files: one blob + n + 1 byte offsets (like the text calls), all announced with ONE geometry and pixel
 * format -- the host reads those 24 bytes of each upload with ucfp_png_probe and groups by them.  One wave per file:
 * chunk walk, inflate (RFC 1950/1951, speculative parallel Huffman decoding), PNG filter reconstruction.
 * Decoded on the device: 8-bit, non-interlaced -- greyscale and grey + alpha (-> GRAY8, the alpha byte is dropped: luma
 * takes no alpha), RGB and indexed colour (-> RGB8 through the file's PLTE), RGBA (-> RGBA8), each with or without a tRNS
 * chunk (simple transparency only adds an alpha channel: no colour sample changes); files of both layouts of a format may
 * share a batch.  ucfp_png_probe reports the format a file DECODES to.  status[i]:
 *   0                      decoded (and hashed)
 *   UCFP_IMAGE_NEEDS_HOST  a valid PNG of another kind (16-bit, 1/2/4-bit, interlaced) or of another geometry /
 *                          format than announced, or a file whose only fault is a CHECKSUM (the Adler-32 of a stream that
 *                          inflated to the right length, the CRC of an ancillary chunk): decoders differ on those, so
 *                          the host's decoder decides -- decode it there, submit the pixels
 *   UCFP_E_MODALITY        not a PNG / damaged stream / bad CRC on a critical chunk (the reference answers 400)
 * png_bytes = d_offsets[n] (the host knows it; sizes the context's workspace: about png_bytes + n x (2 x frame bytes)). */
/* Host-side: geometry and pixel format of a PNG from its IHDR.  UCFP_OK, UCFP_IMAGE_NEEDS_HOST or UCFP_E_MODALITY. */
int ucfp_png_probe(const uint8_t* png, size_t len, uint32_t* width, uint32_t* height, int* pixfmt);
/* Encoded files -> frames (frame i at d_frames + i*frame_stride, rows row_stride apart). */
int ucfp_image_png_decode_batch_dev(ucfp_ctx* ctx, const uint8_t* d_png, const uint64_t* d_offsets, size_t n,
                                    size_t png_bytes, uint32_t width, uint32_t height, int pixfmt, uint8_t* d_frames,
                                    size_t row_stride, size_t frame_stride, int32_t* d_status, void* stream);
/* Encoded files -> records: decode into the context's workspace, then ucfp_image_hash_batch_dev's kernels.
 * d_exact: n x 32 bytes, BLAKE3 of each file as computed by the host -- or NULL: the files are on the device, so their
 * BLAKE3 is computed there (ucfp_blake3_batch_dev's kernel).  Records of files that did not decode are zero. */
int ucfp_image_png_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_png, const uint64_t* d_offsets, size_t n,
                                  size_t png_bytes, uint32_t width, uint32_t height, int pixfmt,
                                  const ucfp_image_preprocess* pre, const uint8_t* d_exact, uint8_t* d_out,
                                  int32_t* d_status, void* stream);

/* Host micro-batcher for ENCODED uploads (SURVEY 8f N1 + N4): the per-request shape of handlers::ingest_image
 * (src/server/handlers.rs:232-302) with the decode moved to the device.  One batcher per announced geometry and pixel
 * format (ucfp_png_probe tells them from the first 29 bytes); concurrent submit() calls become ONE H2D copy of the
 * encoded bytes + ucfp_image_png_hash_batch_dev (PNG decode, BLAKE3 of the file, hashing) + one D2H copy of the
 * records.  *status as there: UCFP_IMAGE_NEEDS_HOST -> decode that upload on the host and use ucfp_image_batcher_submit. */
typedef struct ucfp_png_batcher ucfp_png_batcher;
int ucfp_png_batcher_create(ucfp_ctx* ctx, uint32_t algo, uint32_t width, uint32_t height, int pixfmt,
                            const ucfp_image_preprocess* pre, size_t max_batch, size_t max_bytes, uint32_t max_delay_us,
                            ucfp_png_batcher** out);
void ucfp_png_batcher_destroy(ucfp_png_batcher* b);
int ucfp_png_batcher_submit(ucfp_png_batcher* b, const uint8_t* png, size_t len, uint8_t* out, int32_t* status);
int ucfp_png_batcher_stats(ucfp_png_batcher* b, uint64_t* batches, uint64_t* items);

/* BLAKE3 (32-byte digests) of n byte strings that are already on the device: one blob + n + 1 byte offsets, like the
 * text calls; blob_bytes = d_offsets[n]; the blob must be readable up to the next multiple of 4 bytes.  This is the
 * records' `exact` field (image.rs:82: BLAKE3 of the upload) for uploads that were copied to the device encoded.
 * Host-side single input: ucfp_blake3. */
int ucfp_blake3_batch_dev(ucfp_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, size_t blob_bytes,
                          uint8_t* d_out, void* stream);

/* Host micro-batcher: the caller side of handlers::ingest_image
 * (src/server/handlers.rs:232-302) hashes one image per request thread, up to 512 in flight
 * (src/bin/ucfp.rs:267).  submit() is BLOCKING and thread-safe: concurrent calls are coalesced
 * into one pinned-memory copy + one launch of at most max_batch frames, flushed no later than
 * max_delay_us after the first pending frame arrived.  One geometry per batcher. */
typedef struct ucfp_image_batcher ucfp_image_batcher;
int ucfp_image_batcher_create(ucfp_ctx* ctx, uint32_t algo, uint32_t width, uint32_t height, int pixfmt,
                              const ucfp_image_preprocess* pre, size_t max_batch, uint32_t max_delay_us,
                              ucfp_image_batcher** out);
void ucfp_image_batcher_destroy(ucfp_image_batcher* b);
int ucfp_image_batcher_submit(ucfp_image_batcher* b, const uint8_t* frame, size_t row_stride,
                              const uint8_t* exact, uint8_t* out, int32_t* status);
int ucfp_image_batcher_stats(ucfp_image_batcher* b, uint64_t* batches, uint64_t* items);

/* Synthetic workload of SURVEY 8(d) config 2, generated on device: frame i, pixel (x,y) =
 * ((x + y + 17*i) & 255) ^ (splitmix64((i*h + y)*w + x) >> 60): a ramp with 4 bits of seeded
 * noise. Deterministic; the oracle has the same generator. Bench/test support only. */
int ucfp_image_synth_dev(ucfp_ctx* ctx, uint8_t* frames, size_t n, uint32_t width,
                         uint32_t height, size_t first_index, void* stream);

/* =============================== AUDIO ========================================
 * Replaces the arithmetic behind
 *   audio::fingerprint_wang_with      src/modality/audio.rs:64-98    ([WangHash], 8 B each)
 *   audio::fingerprint_haitsma_with   src/modality/audio.rs:181-224  (u32 per frame)
 * i.e. audiofp::classical::{Wang, Haitsma}::extract and audiofp::dsp::resample::linear.
 * Input: mono f32 PCM.  Wang requires 8 kHz (audio.rs:422-430) -- other rates are rejected with
 * UCFP_E_MODALITY; resample upstream (ucfp_audio_resample_linear*).  Haitsma resamples to 5 kHz
 * itself, like the reference (audio.rs:194-200).
 */
typedef struct ucfp_wang_config { /* audiofp WangConfig; defaults algorithms_manifest.rs:553-592 */
    uint32_t fan_out;          /* 10  */
    uint32_t target_zone_t;    /* 63 frames */
    uint32_t target_zone_f;    /* 64 bins   */
    uint32_t peaks_per_sec;    /* 30  */
    float min_anchor_mag_db;   /* -50 (dB re full-scale sine) */
} ucfp_wang_config;

typedef struct ucfp_haitsma_config { /* audiofp HaitsmaConfig; defaults manifest :655-672 */
    float fmin; /* 300  */
    float fmax; /* 2000 */
} ucfp_haitsma_config;

#define UCFP_WANG_HASH_BYTES 8 /* u32 LE f_a(9)|f_b(9)|dt(14), u32 LE t_anchor (LandmarkScatter.svelte:4) */

/* Upper bound on the hashes n samples can produce (for sizing `out`). */
size_t ucfp_audio_wang_max_hashes(size_t n_samples, const ucfp_wang_config* cfg);
/* Host buffers; *n_hashes receives the number produced (if it exceeds cap_hashes the output is
 * truncated and UCFP_E_INVALID is returned). cfg NULL = defaults. */
int ucfp_audio_wang(ucfp_ctx* ctx, const float* pcm, size_t n, uint32_t sample_rate, const ucfp_wang_config* cfg,
                    uint8_t* out, size_t cap_hashes, size_t* n_hashes);
/* Device buffers; d_n_hashes is a device u64 (total produced, may exceed cap). No sync. */
int ucfp_audio_wang_dev(ucfp_ctx* ctx, const float* d_pcm, size_t n, uint32_t sample_rate,
                        const ucfp_wang_config* cfg, uint8_t* d_out, size_t cap_hashes, uint64_t* d_n_hashes,
                        void* stream);

/* RAGGED BATCH of clips (SURVEY 8f N1 for audio: the reference ingests one short clip per request,
 * src/server/handlers.rs:704-918; its bench clip is 4 s, benches/end_to_end.rs:55-75): clip i is
 * d_pcm[d_offsets[i] .. d_offsets[i+1]) (n_clips + 1 device u64 offsets, d_offsets[n_clips] <= n_total), all at
 * `sample_rate`.  8000 Hz clips are taken as they are; any other rate is resampled to 8 kHz by
 * audiofp::dsp::resample::linear (A1) INSIDE the kernel that cuts the STFT frames -- HBM sees every source sample
 * once (BASELINE config 3: 44.1 kHz).  One launch sequence covers the whole batch.  Hashes of clip i land in
 * d_out[d_out_offsets[i] .. d_out_offsets[i+1]) (8 bytes each, t_anchor relative to the clip); d_out_offsets has
 * n_clips + 1 device u64 entries; if d_out_offsets[n_clips] > cap_hashes the output was truncated at cap_hashes.
 * Workspace: about 4 KiB per second of audio in the batch, held by the context.  No synchronisation. */
size_t ucfp_audio_wang_batch_max_hashes(size_t n_total, size_t n_clips, uint32_t sample_rate, const ucfp_wang_config* cfg);
int ucfp_audio_wang_batch_dev(ucfp_ctx* ctx, const float* d_pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips,
                              uint32_t sample_rate, const ucfp_wang_config* cfg, uint8_t* d_out, size_t cap_hashes,
                              uint64_t* d_out_offsets, void* stream);

/* Host micro-batcher for clips (audio): one clip per request thread (handlers.rs:704-918); concurrent
 * submit() calls become ONE ucfp_audio_wang_batch_dev call over at most max_batch clips / max_samples samples, flushed
 * no later than max_delay_us after the first pending clip.  All clips at `sample_rate` (resampled to 8 kHz in the
 * kernel).  *n_hashes = hashes of this clip (t_anchor relative to the clip); more than cap_hashes -> UCFP_E_INVALID
 * with the first cap_hashes written (as ucfp_audio_wang). */
typedef struct ucfp_audio_batcher ucfp_audio_batcher;
int ucfp_audio_batcher_create(ucfp_ctx* ctx, uint32_t sample_rate, const ucfp_wang_config* cfg, size_t max_batch,
                              size_t max_samples, uint32_t max_delay_us, ucfp_audio_batcher** out);
void ucfp_audio_batcher_destroy(ucfp_audio_batcher* b);
int ucfp_audio_batcher_submit(ucfp_audio_batcher* b, const float* pcm, size_t n, uint8_t* out, size_t cap_hashes,
                              size_t* n_hashes);
int ucfp_audio_batcher_stats(ucfp_audio_batcher* b, uint64_t* batches, uint64_t* items);

size_t ucfp_audio_haitsma_frames(size_t n_samples, uint32_t sample_rate);
int ucfp_audio_haitsma(ucfp_ctx* ctx, const float* pcm, size_t n, uint32_t sample_rate,
                       const ucfp_haitsma_config* cfg, uint32_t* out, size_t cap_frames, size_t* n_frames);
/* Device buffers; the input must already be at 5 kHz (use ucfp_audio_resample_linear_dev). */
int ucfp_audio_haitsma_dev(ucfp_ctx* ctx, const float* d_pcm5k, size_t n, const ucfp_haitsma_config* cfg,
                           uint32_t* d_out, size_t cap_frames, void* stream);

/* RAGGED BATCH for Haitsma, same shape as ucfp_audio_wang_batch_dev: clip i = d_pcm[d_offsets[i] .. d_offsets[i+1]) at
 * `sample_rate`; clips at another rate than 5 kHz are resampled (A1) into the context's workspace first, like
 * audio.rs:194-200 does per clip.  Sub-fingerprints of clip i land in d_out[d_out_offsets[i] .. d_out_offsets[i+1])
 * (u32 each; a clip's first frame has a zero history); frames past cap_frames are not written (d_out_offsets[n_clips]
 * tells).  One launch sequence for the whole batch. */
size_t ucfp_audio_haitsma_batch_max_frames(size_t n_total, size_t n_clips, uint32_t sample_rate);
int ucfp_audio_haitsma_batch_dev(ucfp_ctx* ctx, const float* d_pcm, const uint64_t* d_offsets, size_t n_total, size_t n_clips,
                                 uint32_t sample_rate, const ucfp_haitsma_config* cfg, uint32_t* d_out, size_t cap_frames,
                                 uint64_t* d_out_offsets, void* stream);

/* audiofp::dsp::resample::linear. Output length = floor(n * sr_out / sr_in). */
size_t ucfp_audio_resample_len(size_t n, uint32_t sr_in, uint32_t sr_out);
int ucfp_audio_resample_linear_dev(ucfp_ctx* ctx, const float* d_in, size_t n, uint32_t sr_in, uint32_t sr_out,
                                   float* d_out, size_t cap, void* stream);

/* =============================== TEXT =========================================
 * Replaces the arithmetic behind
 *   text::fingerprint_minhash_with::<128>   src/modality/text.rs:182-236  (1032-B MinHashSig<128>)
 *   text::fingerprint_simhash_tf / _idf     src/modality/text.rs:328-421  (8-B SimHash64)
 *   text::fingerprint_lsh                   src/modality/text.rs:437-446  (same bytes as minhash)
 * i.e. txtfp::MinHashFingerprinter / SimHashFingerprinter.  Documents are passed as one UTF-8
 * blob plus n+1 byte offsets.
 *   UCFP_TEXT_RAW_ASCII     the GPU canonicalises (ASCII lower-casing; NFKC / case fold / Cf+Bidi
 *                           stripping are the identity on ASCII) and segments (UAX#29 restricted
 *                           to ASCII).  A document holding a byte >= 0x80 gets status
 *                           UCFP_TEXT_NEEDS_HOST: the host canonicalises + tokenises it
 *                           (Unicode tables live there) and resubmits it as
 *   UCFP_TEXT_PRETOKENIZED  tokens already canonical, separated by single spaces.
 * status[i]: 0, UCFP_TEXT_NEEDS_HOST, UCFP_E_MODALITY (no tokens), UCFP_E_UNSUPPORTED (one token,
 * or a run of fewer than k tokens, longer than the ~1.4 KiB LDS batch).
 */
#define UCFP_TEXT_RAW_ASCII 0
#define UCFP_TEXT_PRETOKENIZED 1
#define UCFP_TEXT_NEEDS_HOST 1
#define UCFP_MINHASH_BYTES 1032 /* txtfp::MinHashSig<128>: u16 schema = 1, 6 pad, 128 x u64 LE */
/* COMPATIBILITY: the LAYOUT is txtfp's, the 128 slot VALUES are not -- txtfp 0.2.0's slot derivation could not be
 * recovered offline (the reference's golden slot 0, src/server/tests.rs:1153-1157, is not
 * reproduced).  Records made here must therefore never be compared with upstream `minhash-h128` records: the host
 * stores them with format_version UCFP_MINHASH_FORMAT_VERSION instead of txtfp::FORMAT_VERSION (text.rs:227), so a
 * mixed corpus is rejected as incompatible rather than yielding meaningless Jaccard estimates.  SimHash (XXH3-64 per
 * token, family pinned by tests.rs:1126-1127) carries txtfp's own format_version. */
#define UCFP_MINHASH_FORMAT_VERSION 0x48500001u
#define UCFP_SIMHASH_BYTES 8

int ucfp_text_minhash_batch_dev(ucfp_ctx* ctx, const uint8_t* d_utf8, const uint64_t* d_offsets, size_t n,
                                int mode, uint32_t shingle_k, uint8_t* d_out, int32_t* d_status, void* stream);
int ucfp_text_minhash_batch(ucfp_ctx* ctx, const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode,
                            uint32_t shingle_k, uint8_t* out, int32_t* status);
int ucfp_text_simhash_batch_dev(ucfp_ctx* ctx, const uint8_t* d_utf8, const uint64_t* d_offsets, size_t n,
                                int mode, uint8_t* d_out, int32_t* d_status, void* stream);
int ucfp_text_simhash_batch(ucfp_ctx* ctx, const uint8_t* utf8, const uint64_t* offsets, size_t n, int mode,
                            uint8_t* out, int32_t* status);

/* Host micro-batcher for documents: the caller side of handlers::ingest_text
 * (src/server/handlers.rs:304-460) fingerprints one document per request thread, up to 512 in flight
 * (src/bin/ucfp.rs:267).  submit() is BLOCKING and thread-safe: concurrent calls are packed back to back into one
 * pinned blob + offset table, one H2D copy, one ucfp_text_*_batch_dev launch and one D2H copy -- at most max_batch
 * documents or max_bytes of text per flush, flushed no later than max_delay_us after the first pending document.
 * One (algorithm, mode, k) per batcher: the host keeps one for RAW_ASCII and one for PRETOKENIZED documents.
 * `out` receives UCFP_MINHASH_BYTES or UCFP_SIMHASH_BYTES; *status as in the batch calls. */
#define UCFP_TEXT_ALGO_MINHASH 1
#define UCFP_TEXT_ALGO_SIMHASH 2
typedef struct ucfp_text_batcher ucfp_text_batcher;
int ucfp_text_batcher_create(ucfp_ctx* ctx, uint32_t algo, int mode, uint32_t shingle_k, size_t max_batch, size_t max_bytes,
                             uint32_t max_delay_us, ucfp_text_batcher** out);
void ucfp_text_batcher_destroy(ucfp_text_batcher* b);
int ucfp_text_batcher_submit(ucfp_text_batcher* b, const uint8_t* utf8, size_t len, uint8_t* out, int32_t* status);
int ucfp_text_batcher_stats(ucfp_text_batcher* b, uint64_t* batches, uint64_t* items);

/* ---- banded MinHash LSH (the reference only re-tags the record, text.rs:437-446) ----
 * key_b = FNV-style fold of slots [b*rows, (b+1)*rows) + splitmix64 finaliser; bands*rows <= 128.
 * ucfp_text_lsh_band_keys_dev writes keys band-major: d_keys[b*n + doc]. */
int ucfp_text_lsh_band_keys_dev(ucfp_ctx* ctx, const uint8_t* d_records, size_t n, uint32_t bands, uint32_t rows,
                                uint64_t* d_keys, void* stream);
typedef struct ucfp_lsh ucfp_lsh;
/* cand_per_band: rows examined per band per query (0 = 64). */
int ucfp_lsh_create(ucfp_ctx* ctx, uint32_t bands, uint32_t rows, uint32_t cand_per_band, ucfp_lsh** out);
void ucfp_lsh_destroy(ucfp_lsh* lsh);
/* (Re)build from n device-resident 1032-byte MinHash records and their record ids. */
int ucfp_lsh_build_dev(ucfp_lsh* lsh, const uint64_t* d_ids, const uint8_t* d_records, size_t n, void* stream);
/* For each query record: candidates = rows sharing a band key (first cand_per_band per band, at most
 * 1024 in total); score = equal slots / 128; best k by (score desc, id asc). */
int ucfp_lsh_query_dev(ucfp_lsh* lsh, const uint8_t* d_query_records, size_t nq, uint32_t k, uint64_t* d_out_ids,
                       float* d_out_scores, uint32_t* d_out_counts, void* stream);

/* =============================== INDEX ========================================
 * Replaces `trait IndexBackend` kNN (src/index/mod.rs:29-35) as implemented by
 * EmbeddedBackend::knn (src/index/embedded/mod.rs:268-360): exact brute-force top-k inside
 * one tenant.  COSINE_F32 is the reference's kernel (dot_product :454-472, l2_norm :475-477,
 * insert_topk :484-495); HAMMING64 is the new capability BASELINE config 5 asks for behind
 * /v1/query (the reference has no Hamming search: SURVEY F3).  redb stays the source of
 * truth on the Rust side; this object is the GPU-resident mirror of one shard.
 *
 * Ordering: best first; ties broken by ascending record_id (the reference leaves tie order to
 * rayon's split, i.e. unspecified).  Hamming: distance d = popcount(q ^ x), score = 1 - d/64
 * so that "higher is better" holds (src/core/mod.rs:113-115).  Cosine: score =
 * dot/(|q||v|); zero-norm rows are skipped, a zero-norm query yields no hits (:283-286,:328-330).
 * Every reported cosine score is an f32 dot product; where a batch is steered by f16 matrix-core arithmetic (the chunk
 * minima of 2 .. 64 queries per pass, round 4) that arithmetic only selects WHICH chunks of rows get their exact scores
 * computed, under a proven error bound -- it never reaches an answer.
 */
typedef struct ucfp_index ucfp_index;

typedef enum ucfp_index_kind {
    UCFP_INDEX_HAMMING64 = 1, /* rows are uint64_t                                   */
    UCFP_INDEX_COSINE_F32 = 2 /* rows are float[dim]                                 */
} ucfp_index_kind;

#define UCFP_INDEX_APPEND_ONLY 1u /* no id->row map: upsert appends, delete unsupported   */
#define UCFP_INDEX_MAX_K 128u      /* web caps k at 100 (web/src/routes/api/search/+server.ts:11) */
#define UCFP_INVALID_ID 0xffffffffffffffffull

int ucfp_index_create(ucfp_ctx* ctx, int kind, uint32_t dim, uint32_t flags, ucfp_index** out);
void ucfp_index_destroy(ucfp_index* idx);

/* IndexBackend::upsert (src/index/mod.rs:20-22): rows with a known id are overwritten in
 * place, new ids are appended to the tenant's contiguous range. Host pointers. */
int ucfp_index_upsert(ucfp_index* idx, uint32_t tenant, const uint64_t* ids, const void* rows, size_t n);
/* Bulk append of device-resident rows (APPEND_ONLY indexes; corpus loaders, benches). */
int ucfp_index_append_dev(ucfp_index* idx, uint32_t tenant, const uint64_t* d_ids, const void* d_rows,
                          size_t n, void* stream);
/* IndexBackend::delete (src/index/mod.rs:24-27). `n_removed` may be NULL. */
int ucfp_index_delete(ucfp_index* idx, uint32_t tenant, const uint64_t* ids, size_t n, size_t* n_removed);
int ucfp_index_size(ucfp_index* idx, uint32_t tenant, size_t* out);
/* IndexBackend::flush (src/index/mod.rs:63): waits for queued device work. */
int ucfp_index_flush(ucfp_index* idx);
/* Snapshot of the device mirror (the sidecar flat file GPU shards are rebuilt from at start-up; redb,
 * `src/index/embedded/mod.rs:37-43,104-125`, stays the reference's source of truth).  `load` upserts the
 * snapshot's rows into an index of the same kind / dim. */
int ucfp_index_save(ucfp_index* idx, const char* path);
int ucfp_index_load(ucfp_index* idx, const char* path);

/* IndexBackend::knn for a batch of queries (nq = 1 is the reference's call).
 *   queries     nq rows of the index kind (host memory)
 *   out_ids     nq x k   record ids, UCFP_INVALID_ID past out_counts[q]
 *   out_scores  nq x k   f32 score (higher is better)
 *   out_dist    nq x k   Hamming distance (HAMMING64 only; may be NULL)
 *   out_counts  nq       hits returned for each query (<= k)
 * An unknown tenant or k = 0 returns zero hits, like the reference (:275-277). */
int ucfp_index_search(ucfp_index* idx, uint32_t tenant, const void* queries, size_t nq, uint32_t k,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_dist, uint32_t* out_counts);
/* Same with device pointers, enqueued on `stream` (no synchronisation).  Searches enqueued on different streams are ordered
 * only where they share a workspace: a HAMMING64 index alternates between two, so two batches may be in flight at once
 * (the short staging kernels of one run under the matrix-core scan of the other); cosine searches run one at a time. */
int ucfp_index_search_dev(ucfp_index* idx, uint32_t tenant, const void* d_queries, size_t nq, uint32_t k,
                          uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_dist,
                          uint32_t* d_out_counts, void* stream);

/* Host micro-batcher for the QUERY route: /v1/query is one query per request (src/server/handlers.rs:143-187),
 * up to 512 requests in flight (src/bin/ucfp.rs:267).  submit() is BLOCKING and thread-safe: concurrent calls -- each with
 * ITS OWN query (a u64 hash, or dim floats) and ITS OWN k (QueryRequest.k, src/server/dto.rs:74-87) -- become ONE copy of the
 * queries + ONE ucfp_index_search_dev with the largest k of the batch + one copy of the results; a request gets the first k
 * entries of its row (the order is total: (distance, id) / (score desc, id)).  One batcher per (index, tenant); at most
 * max_batch (<= 4096) queries per flush, flushed no later than max_delay_us after the first pending query.  (All batchers:
 * behind a flush that carried several requests the next one waits up to 10 us -- or until as many requests have arrived
 * as that flush carried -- even with max_delay_us = 0: its submitters are on their way back with their next request.  A
 * lone sequential client is flushed at once.)
 *   out_ids / out_scores / out_dist   k entries each (scores, dist may be NULL); places past *out_count carry
 *                                     UCFP_INVALID_ID / -1 / 2^32 - 1 like ucfp_index_search. */
typedef struct ucfp_search_batcher ucfp_search_batcher;
int ucfp_index_search_batcher_create(ucfp_index* idx, uint32_t tenant, size_t max_batch, uint32_t max_delay_us,
                                     ucfp_search_batcher** out);
void ucfp_index_search_batcher_destroy(ucfp_search_batcher* b);
int ucfp_index_search_batcher_submit(ucfp_search_batcher* b, const void* query, uint32_t k, uint64_t* out_ids, float* out_scores,
                                     uint32_t* out_dist, uint32_t* out_count);
int ucfp_index_search_batcher_stats(ucfp_search_batcher* b, uint64_t* batches, uint64_t* items);

/* Final step of a sharded search: merge `parts` per-shard top-k lists -- the
 * all-gathered [parts][nq][k] ids + keys -- into one. Keys: Hamming distance (kind
 * HAMMING64) or the order-preserving u32 image of -score (COSINE_F32) as produced by
 * ucfp_index_search_dev in d_out_dist. Device pointers. */
int ucfp_topk_merge_dev(ucfp_ctx* ctx, int kind, const uint64_t* d_part_ids, const uint32_t* d_part_keys,
                        uint32_t parts, size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                        uint32_t* d_out_keys, uint32_t* d_out_counts, void* stream);

/* ---- the packed wire format of a sharded search: 16-byte entries {u64 id, u32 key, u32 0}, [nq][k] per shard.
 * For hosts that move the per-shard lists with their own transport (MPI, gloo, host TCP) instead of RCCL:
 * pack -> (their all-gather into [parts][nq][k] entries) -> merge_packed. */
int ucfp_topk_pack_dev(ucfp_ctx* ctx, const uint64_t* d_ids, const uint32_t* d_keys, size_t nq, uint32_t k,
                       void* d_entries, void* stream);
int ucfp_topk_merge_packed_dev(ucfp_ctx* ctx, int kind, const void* d_entries, uint32_t parts, size_t nq, uint32_t k,
                               uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_keys, uint32_t* d_out_counts,
                               void* stream);
/* The same with the missing-shard mask: a shard that could not scan sends nq x k entries of 0xff bytes (id 2^64-1, key
 * 2^32-1 like an empty list, and the pad word 0xffffffff where ucfp_topk_pack_dev writes 0); bit p of *d_missing (device
 * u64, may be NULL; parts <= 64) is set for every such part. */
int ucfp_topk_merge_packed_ex_dev(ucfp_ctx* ctx, int kind, const void* d_entries, uint32_t parts, size_t nq, uint32_t k,
                                  uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_keys, uint32_t* d_out_counts,
                                  uint64_t* d_missing, void* stream);

/* ============================ SHARDED SEARCH (multi-GPU) ==============================
 * "Multi-GPU variant takes a device list; shards are internal" of IndexBackend::knn
 * (src/index/mod.rs:29-35).  Model: ONE PROCESS PER GPU (the "device list" is the job's ranks: rank r owns GPU
 * LOCAL_RANK and one ucfp_index holding its range of the corpus, ucfp_shard_range).  Queries are replicated; each
 * rank scans its shard; the only data-path exchange is ONE ncclAllGather (RCCL over xGMI) of the per-shard top-k
 * as packed 16-byte entries -- nq x k x 16 B per rank, latency-bound -- then every rank runs the same merge
 * ((key asc, id asc)) and holds the full answer.  The reference has no multi-device search.
 *
 *   rank 0:      ucfp_shard_unique_id(uid)            and ships the 128 bytes to the other ranks (the host's own
 *                                                     control channel: env, file, TCP -- like ncclUniqueId)
 *   every rank:  ucfp_shard_comm_create(ctx, uid, rank, world, &comm)     (collective: all ranks call it)
 *                ucfp_index_search_sharded_dev(idx, comm, ...)            (collective, same nq / k everywhere)
 * world = 1 needs no uid and never loads RCCL.  RCCL is dlopen()ed (librccl.so.1) when world > 1.
 *
 * ucfp_shard_comm_create_ex(..., flags, ...): UCFP_SHARD_FORCE_RCCL (or UCFP_SHARD_FORCE_RCCL=1 in the environment)
 * builds a real RCCL communicator even at world = 1 (uid required) -- ncclCommInitRank(nranks = 1), one ncclAllGather
 * per batch on the exchange stream, the merge over the gathered buffer -- so that a single-GPU host executes, and can
 * test, exactly the code path a multi-GPU job runs.  ucfp_shard_comm_uses_rccl tells which branch a communicator takes.
 *
 * Failure of one rank: if this rank's own shard scan cannot be enqueued (or any later local step fails),
 * ucfp_index_search_sharded_submit still joins the all-gather with an empty list (so the other ranks do not block),
 * completes the ticket and returns the error; the answer every rank then holds lacks that shard, and every rank can tell:
 * the list carries a mark in the entries' pad word that the merge turns into ucfp_index_search_sharded_missing's mask. */
#define UCFP_SHARD_UID_BYTES 128
#define UCFP_SHARD_FORCE_RCCL 1u
typedef struct ucfp_shard_comm ucfp_shard_comm;
int ucfp_shard_unique_id(uint8_t uid[UCFP_SHARD_UID_BYTES]);
int ucfp_shard_comm_create(ucfp_ctx* ctx, const uint8_t uid[UCFP_SHARD_UID_BYTES], int rank, int world,
                           ucfp_shard_comm** out);
int ucfp_shard_comm_create_ex(ucfp_ctx* ctx, const uint8_t uid[UCFP_SHARD_UID_BYTES], int rank, int world,
                              uint32_t flags, ucfp_shard_comm** out);
int ucfp_shard_comm_uses_rccl(ucfp_shard_comm* comm);   /* 1: batches go through ncclAllGather; 0: local short cut */
void ucfp_shard_comm_destroy(ucfp_shard_comm* comm);
/* rank / world / number of all-gathers issued so far (each may be NULL) */
int ucfp_shard_comm_info(ucfp_shard_comm* comm, int* rank, int* world, uint64_t* exchanges);
/* [start, end) of a corpus of n_total rows (global insertion order) owned by `rank`: contiguous ranges, the first
 * n_total % world ranks hold one extra row. */
void ucfp_shard_range(uint64_t n_total, int rank, int world, uint64_t* start, uint64_t* end);

/* Sharded IndexBackend::knn for a batch (device pointers; queries identical on every rank).
 * submit: the shard scan is enqueued BEHIND `stream` (whatever wrote the queries there has finished) on the scan stream of
 *         one of the communicator's two buffer sets, the all-gather + merge on its exchange stream: the scans of two
 *         successive batches run side by side (one's short staging kernels fill the gaps of the other's matrix-core
 *         scan) and the exchange of a batch overlaps the scan of the next; at most two batches in flight.
 *         Outputs (as ucfp_index_search_dev; d_out_scores / d_out_keys may be NULL) are written by those streams:
 *         they and the queries must stay untouched until the ticket is collected.
 * collect: makes `stream` wait for that batch's results (no host synchronisation).
 * ucfp_index_search_sharded_dev = submit + collect on the same stream. */
int ucfp_index_search_sharded_submit(ucfp_index* idx, ucfp_shard_comm* comm, uint32_t tenant, const void* d_queries,
                                     size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                                     uint32_t* d_out_keys, uint32_t* d_out_counts, void* stream, uint64_t* ticket);
int ucfp_index_search_sharded_collect(ucfp_shard_comm* comm, uint64_t ticket, void* stream);
/* Which shards are ABSENT from a ticket's answer: bit r of *missing_mask = rank r joined the all-gather with the empty list of
 * a failed scan (see "Failure of one rank" above) -- the same mask on every rank, so a host can report a partial result
 * wherever it reads the answer, not only on the rank that failed.  Waits for the batch (host synchronisation); valid while
 * the ticket's buffer set has not been handed to a later batch (i.e. before the second submit after it). */
int ucfp_index_search_sharded_missing(ucfp_shard_comm* comm, uint64_t ticket, uint64_t* missing_mask);
int ucfp_index_search_sharded_dev(ucfp_index* idx, ucfp_shard_comm* comm, uint32_t tenant, const void* d_queries,
                                  size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                                  uint32_t* d_out_keys, uint32_t* d_out_counts, void* stream);

/* ---- JPEG front end ----
 * The reference's image route also takes JPEG uploads (src/modality/image.rs:54 "PNG / JPEG / WebP / GIF / BMP", decoders at
 * Cargo.toml:143) and decodes them inside the same SDK call (image.rs:68-70, :176-179).  Same call shape as the PNG entry
 * points: one blob + n + 1 byte offsets, ONE announced geometry (ucfp_jpeg_probe reads it from the frame header).  What
 * is decoded of a JPEG is its LUMA component (DESIGN J1: the Y plane of a YCbCr file, the plane of a greyscale one; chroma
 * is parsed, never transformed): frames are GRAY8, and what libjpeg returns for out_color_space = JCS_GRAYSCALE with the
 * accurate integer IDCT, bit for bit.  One wave per file, one lane per restart interval (Huffman decoding is serial
 * within one), one thread per 8x8 block for the inverse DCT.  status[i]:
 *   0                      decoded (and hashed)
 *   UCFP_IMAGE_NEEDS_HOST  a JPEG this path does not decode (progressive, arithmetic-coded, 12-bit, CMYK / RGB-coded,
 *                          several scans, luma below the MCU's resolution, 16-bit quantisation tables), another geometry
 *                          than announced, or ANY irregularity of the entropy-coded data: the host's decoder decides
 *   UCFP_E_MODALITY        no SOI marker: not a JPEG (the reference answers 400)
 * jpg_bytes = d_offsets[n]; workspace about jpg_bytes + n x 3 x width x height bytes. */
int ucfp_jpeg_probe(const uint8_t* jpg, size_t len, uint32_t* width, uint32_t* height);
int ucfp_image_jpeg_decode_batch_dev(ucfp_ctx* ctx, const uint8_t* d_jpg, const uint64_t* d_offsets, size_t n,
                                     size_t jpg_bytes, uint32_t width, uint32_t height, uint8_t* d_frames,
                                     size_t row_stride, size_t frame_stride, int32_t* d_status, void* stream);
/* Encoded files -> records (as ucfp_image_png_hash_batch_dev; d_exact NULL: BLAKE3 of every file on the device). */
int ucfp_image_jpeg_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_jpg, const uint64_t* d_offsets, size_t n,
                                   size_t jpg_bytes, uint32_t width, uint32_t height, const ucfp_image_preprocess* pre,
                                   const uint8_t* d_exact, uint8_t* d_out, int32_t* d_status, void* stream);

/* Micro-batcher for JPEG uploads: the object of ucfp_png_batcher_create with the JPEG front end behind it (records of the
 * files' luma planes).  Submit / stats / destroy with the ucfp_png_batcher_* calls. */
int ucfp_jpeg_batcher_create(ucfp_ctx* ctx, uint32_t algo, uint32_t width, uint32_t height, const ucfp_image_preprocess* pre,
                             size_t max_batch, size_t max_bytes, uint32_t max_delay_us, ucfp_png_batcher** out);

/* BLAKE3-256 (default hash mode) of a HOST buffer: the `exact` digest the reference stores in
 * ImageFingerprint.exact (BLAKE3 of the uploaded bytes). Host code; no device needed. */
int ucfp_blake3(const uint8_t* data, size_t len, uint8_t out[32]);

/* =============================== STORED TABLES ====================
 * The reference's source of truth is one redb file: tables ucfp/fingerprints/v1, ucfp/vectors/v1, ucfp/catalog/v2,
 * all keyed (tenant_id, record_id) and written in one transaction per upsert (src/index/embedded/mod.rs:37-43,
 * :157-227); EmbeddedBackend::open (:104-125) is where a device mirror has to be rebuilt.  redb's page format lives in
 * a crate outside the tree, so the drop-in keeps a SIDECAR: an append-only log of exactly those rows (fingerprint
 * bytes, embedding, the catalog row's serde_json text) that the host appends to right after its redb transaction
 * commits, and replays at start-up.  Last entry of a key wins; a torn tail is cut on the next open (CRC per entry).
 * Host-only calls (no GPU involved). */
typedef struct ucfp_sidecar ucfp_sidecar;
int ucfp_sidecar_open(const char* path, ucfp_sidecar** out);      /* creates the log if missing, validates it otherwise */
void ucfp_sidecar_close(ucfp_sidecar* sc);
/* One call per record of IndexBackend::upsert (mod.rs:176-208): embedding NULL / dim 0 = "no vector" (:184-191). */
int ucfp_sidecar_append_upsert(ucfp_sidecar* sc, uint32_t tenant, uint64_t record_id, const uint8_t* fingerprint,
                               uint32_t fp_len, const float* embedding, uint32_t dim, const char* catalog_json,
                               uint32_t json_len);
int ucfp_sidecar_append_delete(ucfp_sidecar* sc, uint32_t tenant, uint64_t record_id);   /* IndexBackend::delete :229-266 */
int ucfp_sidecar_sync(ucfp_sidecar* sc);                           /* fdatasync: call where the host fsyncs redb */

/* Read side: the live rows of a log (after replaying overwrites and deletes), in ascending (tenant, record_id) order
 * -- the order of the reference's range scans.  Pointers returned by _row point into the mapped file and live until
 * _close; embedding bytes are not necessarily 4-byte aligned. */
typedef struct ucfp_sidecar_snapshot ucfp_sidecar_snapshot;
int ucfp_sidecar_snapshot_open(const char* path, ucfp_sidecar_snapshot** out, uint64_t* live_rows, uint64_t* log_entries,
                               uint64_t* torn_bytes);
void ucfp_sidecar_snapshot_close(ucfp_sidecar_snapshot* s);
int ucfp_sidecar_snapshot_row(ucfp_sidecar_snapshot* s, uint64_t i, uint32_t* tenant, uint64_t* record_id,
                              const uint8_t** fingerprint, uint32_t* fp_len, const uint8_t** embedding_bytes, uint32_t* dim,
                              const char** catalog_json, uint32_t* json_len);
/* Bulk gathers for the rebuild: fingerprints of the rows whose catalog `algorithm` is `algorithm` and whose blob is
 * fp_len bytes ("only comparable hashes share an index"), or the embeddings of one dimension, packed into caller arrays
 * (any of which may be NULL).  *n = matching rows; if it exceeds cap only the first cap were written. */
int ucfp_sidecar_snapshot_gather_fingerprints(ucfp_sidecar_snapshot* s, const char* algorithm, uint32_t fp_len,
                                              uint32_t* tenants, uint64_t* ids, uint8_t* fingerprints, uint64_t cap,
                                              uint64_t* n);
int ucfp_sidecar_snapshot_gather_vectors(ucfp_sidecar_snapshot* s, uint32_t dim, uint32_t* tenants, uint64_t* ids,
                                         float* rows, uint64_t cap, uint64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* UCFP_HIP_H */
