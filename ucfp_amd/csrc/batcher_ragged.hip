// batcher_ragged.hip -- host micro-batchers for the variable-length modalities (text, audio, PNG uploads).
//
// The reference fingerprints one document / one clip per HTTP request on a tokio worker
// (src/server/handlers.rs:304-460 text, :704-918 audio), up to 512 requests in flight (src/bin/ucfp.rs:267).
// Same contract as the image batcher (batcher.hip): submit() is BLOCKING and thread-safe; a worker thread packs
// whatever is pending -- at most `max_batch` items or `max_units` payload units, or whatever arrived within
// `max_delay_us` of the first item -- back to back into one pinned blob plus an offset table, issues ONE H2D copy,
// ONE call of the ragged-batch entry point (ucfp_text_*_batch_dev / ucfp_audio_wang_batch_dev) and ONE D2H copy,
// then wakes the submitters.  Every submitter copies its own payload into the pinned blob, outside the lock.

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "../../include/ucfp_hip.h"
#include "batch_core.h"
#include "common.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
int ctx_device(const ucfp_ctx* ctx);
}  // namespace ucfp
using ucfp::capi_fail;

namespace {

enum Kind { kTextMinhash, kTextSimhash, kAudioWang, kPngHash, kJpegHash, kUploadHash };

// Pinned / device staging of one set.  Input: [max_batch + 1 payload offsets | payload], one H2D copy.
// Result (text): [max_batch status words | n records], one D2H copy.  Result (audio): [max_batch + 1 hash offsets],
// then the hashes produced.
struct Ragged {
    ucfp_ctx* ctx = nullptr;
    int device = 0;
    Kind kind = kTextMinhash;
    int mode = 0;                 // text
    uint32_t shingle_k = 0;       // text / minhash
    uint32_t sample_rate = 0;     // audio
    ucfp_wang_config wang{};      // audio
    uint32_t algo = 0, width = 0, height = 0;   // png
    int pixfmt = 0;
    ucfp_image_preprocess pre{8192, 32};
    size_t unit = 1;              // payload unit in bytes: 1 (UTF-8) or 4 (f32 sample)
    size_t rec = 0;               // fixed result bytes per item (text); 0 = ragged results (audio)
    size_t max_batch = 0, max_units = 0, out_cap = 0;   // out_cap: ragged results, 8-byte hashes per flush
    size_t in_head = 0, out_head = 0;                   // bytes of the offset table / status table in front

    uint8_t* h_in[2] = {nullptr, nullptr};
    uint8_t* h_out[2] = {nullptr, nullptr};
    ucfp_upload_info* h_info[2] = {nullptr, nullptr};   // uploads of any kind and size: each slot's probe result
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    uint8_t* d_hashes = nullptr;  // audio
    hipStream_t stream = nullptr;
    ucfp::BatchCore core;

    uint64_t* offsets(int s) { return reinterpret_cast<uint64_t*>(h_in[s]); }
    uint8_t* payload(int s) { return h_in[s] + in_head; }
};

int run_set(Ragged* b, int s, size_t n, size_t units) {
    b->offsets(s)[n] = units;
    hipError_t e = hipMemcpyAsync(b->d_in, b->h_in[s], b->in_head + units * b->unit, hipMemcpyHostToDevice, b->stream);
    if (e != hipSuccess) return UCFP_E_INDEX;
    const uint64_t* d_off = reinterpret_cast<const uint64_t*>(b->d_in);
    const uint8_t* d_pay = b->d_in + b->in_head;
    int rc = UCFP_OK;
    switch (b->kind) {
        case kTextMinhash:
            rc = ucfp_text_minhash_batch_dev(b->ctx, d_pay, d_off, n, b->mode, b->shingle_k, b->d_out + b->out_head,
                                             reinterpret_cast<int32_t*>(b->d_out), b->stream);
            break;
        case kTextSimhash:
            rc = ucfp_text_simhash_batch_dev(b->ctx, d_pay, d_off, n, b->mode, b->d_out + b->out_head,
                                             reinterpret_cast<int32_t*>(b->d_out), b->stream);
            break;
        case kPngHash:
            // encoded uploads: decode, BLAKE3 (exact = NULL) and hash on the device
            rc = ucfp_image_png_hash_batch_dev(b->ctx, b->algo, d_pay, d_off, n, units, b->width, b->height, b->pixfmt, &b->pre,
                                               nullptr, b->d_out + b->out_head, reinterpret_cast<int32_t*>(b->d_out), b->stream);
            break;
        case kJpegHash:
            // JPEG uploads: luma plane decoded on the device (DESIGN J1), BLAKE3, hash
            rc = ucfp_image_jpeg_hash_batch_dev(b->ctx, b->algo, d_pay, d_off, n, units, b->width, b->height, &b->pre, nullptr,
                                                b->d_out + b->out_head, reinterpret_cast<int32_t*>(b->d_out), b->stream);
            break;
        case kUploadHash:
            // uploads of any kind and size: PNG and JPEG decoded on the device, each to its own geometry, one ragged hash
            rc = ucfp_image_upload_hash_batch_dev(b->ctx, b->algo, d_pay, d_off, n, units, b->h_info[s], &b->pre, nullptr,
                                                  b->d_out + b->out_head, reinterpret_cast<int32_t*>(b->d_out), b->stream);
            break;
        case kAudioWang:
            rc = ucfp_audio_wang_batch_dev(b->ctx, reinterpret_cast<const float*>(d_pay), d_off, units, n, b->sample_rate,
                                           &b->wang, b->d_hashes, b->out_cap, reinterpret_cast<uint64_t*>(b->d_out), b->stream);
            break;
    }
    if (rc) return rc;
    if (b->rec) {
        e = hipMemcpyAsync(b->h_out[s], b->d_out, b->out_head + n * b->rec, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    } else {
        // ragged results: the offsets first, then exactly the hashes produced
        e = hipMemcpyAsync(b->h_out[s], b->d_out, (n + 1) * 8, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e == hipSuccess) {
            const uint64_t made = reinterpret_cast<const uint64_t*>(b->h_out[s])[n];
            const uint64_t total = made < b->out_cap ? made : b->out_cap;
            if (total) e = hipMemcpyAsync(b->h_out[s] + b->out_head, b->d_hashes, total * 8, hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        }
    }
    return e == hipSuccess ? UCFP_OK : UCFP_E_INDEX;
}

// Stops the worker and frees the staging (the owning struct is deleted by its caller).
void teardown(Ragged* b) {
    b->core.stop();
    (void)hipSetDevice(b->device);
    for (int s = 0; s < 2; s++) {
        if (b->h_in[s]) (void)hipHostFree(b->h_in[s]);
        if (b->h_out[s]) (void)hipHostFree(b->h_out[s]);
        delete[] b->h_info[s];
        b->h_info[s] = nullptr;
    }
    if (b->d_in) (void)hipFree(b->d_in);
    if (b->d_out) (void)hipFree(b->d_out);
    if (b->d_hashes) (void)hipFree(b->d_hashes);
    if (b->stream) (void)hipStreamDestroy(b->stream);
}

// Allocates the staging of a configured batcher and starts its worker.
int start(Ragged* b, uint32_t max_delay_us) {
    b->in_head = ((b->max_batch + 1) * 8 + 255) & ~(size_t)255;
    b->out_head = b->rec ? (b->max_batch * 4 + 255) & ~(size_t)255 : ((b->max_batch + 1) * 8 + 255) & ~(size_t)255;
    const size_t in_bytes = b->in_head + b->max_units * b->unit + 64;      // the text kernel reads whole 16-byte pieces
    const size_t hashes = (b->out_cap ? b->out_cap : 1) * 8;
    const size_t out_bytes = b->out_head + (b->rec ? b->max_batch * b->rec : hashes);
    hipError_t e = hipSetDevice(b->device);
    for (int s = 0; s < 2 && e == hipSuccess; s++) {
        e = hipHostMalloc((void**)&b->h_in[s], in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_out[s], out_bytes, hipHostMallocDefault);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_in, in_bytes);
    if (e == hipSuccess) e = hipMemset(b->d_in, 0, in_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&b->d_out, b->rec ? out_bytes : b->out_head);
    if (e == hipSuccess && !b->rec) e = hipMalloc((void**)&b->d_hashes, hashes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        teardown(b);
        return capi_fail(UCFP_E_INDEX, "batcher allocation failed: %s", hipGetErrorString(e));
    }
    b->core.start(b->max_batch, b->max_units, max_delay_us, [b](int s, size_t n, size_t units) {
        (void)hipSetDevice(b->device);
        return run_set(b, s, n, units);
    });
    return UCFP_OK;
}

// Takes a slot for `units` payload units, copies the payload in, waits for the flush.  t->set < 0: no slot was taken
// (the return value says why).  Otherwise the pinned result slot stays reserved: the caller copies its result out
// (if the flush succeeded) and then calls core.release(t).
int submit(Ragged* b, const void* payload, size_t units, ucfp::BatchCore::Ticket* t) {
    t->set = -1;
    if (units > b->max_units)
        return capi_fail(UCFP_E_INVALID, "item of %zu units exceeds the batcher's capacity of %zu", units, b->max_units);
    if (!b->core.claim(units, t)) {
        t->set = -1;
        return capi_fail(UCFP_E_INDEX, "batcher is shutting down");
    }
    b->offsets(t->set)[t->slot] = t->at;
    if (units) memcpy(b->payload(t->set) + t->at * b->unit, payload, units * b->unit);
    b->core.commit(*t);
    return b->core.wait(*t);
}

}  // namespace

struct ucfp_text_batcher {
    Ragged r;
};
struct ucfp_audio_batcher {
    Ragged r;
};
struct ucfp_png_batcher {
    Ragged r;
};
// Two lanes behind one handle: PNG uploads coalesce in `r` (the caller's context), JPEG uploads in `rj` on a context of the
// batcher's own (its own decode workspace and streams), so the two kinds' flushes run side by side.  A PNG is one wave's tens
// of milliseconds of serial inflate whatever the batch; a JPEG flush takes a few -- in one lane every JPEG request waited
// for the largest PNG that happened to share its flush.
struct ucfp_upload_batcher {
    Ragged r, rj;
    ucfp_ctx* jctx = nullptr;
};

extern "C" {

int ucfp_text_batcher_create(ucfp_ctx* ctx, uint32_t algo, int mode, uint32_t shingle_k, size_t max_batch, size_t max_bytes,
                             uint32_t max_delay_us, ucfp_text_batcher** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (algo != UCFP_TEXT_ALGO_MINHASH && algo != UCFP_TEXT_ALGO_SIMHASH)
        return capi_fail(UCFP_E_UNSUPPORTED, "text batcher algo %u (one of UCFP_TEXT_ALGO_*)", algo);
    if (mode != UCFP_TEXT_RAW_ASCII && mode != UCFP_TEXT_PRETOKENIZED) return capi_fail(UCFP_E_INVALID, "unknown text mode %d", mode);
    if (algo == UCFP_TEXT_ALGO_MINHASH && (shingle_k == 0 || shingle_k > 64))
        return capi_fail(UCFP_E_MODALITY, "shingle k must be in [1, 64] (got %u)", shingle_k);
    if (max_batch == 0 || max_batch > (1u << 20) || max_bytes == 0 || max_bytes >= ((size_t)1 << 32))
        return capi_fail(UCFP_E_INVALID, "batcher needs 1 <= max_batch <= 2^20 and 1 <= max_bytes < 2^32");
    ucfp_text_batcher* b = new (std::nothrow) ucfp_text_batcher();
    if (!b) return capi_fail(UCFP_E_INDEX, "out of host memory");
    Ragged& r = b->r;
    r.ctx = ctx;
    r.device = ucfp::ctx_device(ctx);
    r.kind = algo == UCFP_TEXT_ALGO_MINHASH ? kTextMinhash : kTextSimhash;
    r.mode = mode;
    r.shingle_k = shingle_k;
    r.unit = 1;
    r.rec = algo == UCFP_TEXT_ALGO_MINHASH ? UCFP_MINHASH_BYTES : UCFP_SIMHASH_BYTES;
    r.max_batch = max_batch;
    r.max_units = max_bytes;
    Ragged* raw = &b->r;
    const int rc = start(raw, max_delay_us);
    if (rc) {
        delete b;
        return rc;
    }
    *out = b;
    return UCFP_OK;
}

void ucfp_text_batcher_destroy(ucfp_text_batcher* b) {
    if (!b) return;
    teardown(&b->r);
    delete b;
}

int ucfp_text_batcher_submit(ucfp_text_batcher* b, const uint8_t* utf8, size_t len, uint8_t* out, int32_t* status) {
    if (!b || !out || (len && !utf8)) return capi_fail(UCFP_E_INVALID, "batcher/utf8/out is NULL");
    Ragged& r = b->r;
    ucfp::BatchCore::Ticket t;
    const int rc = submit(&r, utf8, len, &t);
    if (t.set < 0) return rc;
    if (rc == UCFP_OK) {
        memcpy(out, r.h_out[t.set] + r.out_head + t.slot * r.rec, r.rec);
        if (status) *status = reinterpret_cast<const int32_t*>(r.h_out[t.set])[t.slot];
    }
    r.core.release(t);
    if (rc != UCFP_OK) return capi_fail(rc, "batched text launch failed");
    return UCFP_OK;
}

int ucfp_text_batcher_stats(ucfp_text_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    b->r.core.stats(batches, items);
    return UCFP_OK;
}

int ucfp_png_batcher_create(ucfp_ctx* ctx, uint32_t algo, uint32_t width, uint32_t height, int pixfmt,
                            const ucfp_image_preprocess* pre, size_t max_batch, size_t max_bytes, uint32_t max_delay_us,
                            ucfp_png_batcher** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    const size_t rec = ucfp_image_record_bytes(algo);
    if (!rec) return capi_fail(UCFP_E_UNSUPPORTED, "image algo mask %u", algo);
    if (pixfmt < UCFP_PIX_GRAY8 || pixfmt > UCFP_PIX_RGBA8) return capi_fail(UCFP_E_INVALID, "unknown pixfmt %d", pixfmt);
    if (!width || !height || max_batch == 0 || max_batch > 65536 || max_bytes == 0 || max_bytes >= ((size_t)1 << 32))
        return capi_fail(UCFP_E_INVALID, "batcher needs a geometry, 1 <= max_batch <= 65536 and 1 <= max_bytes < 2^32");
    ucfp_png_batcher* b = new (std::nothrow) ucfp_png_batcher();
    if (!b) return capi_fail(UCFP_E_INDEX, "out of host memory");
    Ragged& r = b->r;
    r.ctx = ctx;
    r.device = ucfp::ctx_device(ctx);
    r.kind = kPngHash;
    r.algo = algo;
    r.width = width;
    r.height = height;
    r.pixfmt = pixfmt;
    if (pre) r.pre = *pre;
    r.unit = 1;
    r.rec = rec;
    r.max_batch = max_batch;
    r.max_units = max_bytes;
    const int rc = start(&b->r, max_delay_us);
    if (rc) {
        delete b;
        return rc;
    }
    *out = b;
    return UCFP_OK;
}

int ucfp_jpeg_batcher_create(ucfp_ctx* ctx, uint32_t algo, uint32_t width, uint32_t height, const ucfp_image_preprocess* pre,
                             size_t max_batch, size_t max_bytes, uint32_t max_delay_us, ucfp_png_batcher** out) {
    // the same batcher object with the JPEG front end behind it: submit / stats / destroy through ucfp_png_batcher_*
    const int rc = ucfp_png_batcher_create(ctx, algo, width, height, UCFP_PIX_GRAY8, pre, max_batch, max_bytes, max_delay_us, out);
    if (rc) return rc;
    // (the worker is idle until the first submit: switching the kind here is ordered before it by the caller's own
    // hand-over of the handle to its request threads)
    (*out)->r.kind = kJpegHash;
    return UCFP_OK;
}

void ucfp_png_batcher_destroy(ucfp_png_batcher* b) {
    if (!b) return;
    teardown(&b->r);
    delete b;
}

int ucfp_png_batcher_submit(ucfp_png_batcher* b, const uint8_t* png, size_t len, uint8_t* out, int32_t* status) {
    if (!b || !out || !png || len == 0) return capi_fail(UCFP_E_INVALID, "batcher/png/out is NULL or empty");
    Ragged& r = b->r;
    ucfp::BatchCore::Ticket t;
    const int rc = submit(&r, png, len, &t);
    if (t.set < 0) return rc;
    if (rc == UCFP_OK) {
        memcpy(out, r.h_out[t.set] + r.out_head + t.slot * r.rec, r.rec);
        if (status) *status = reinterpret_cast<const int32_t*>(r.h_out[t.set])[t.slot];
    }
    r.core.release(t);
    if (rc != UCFP_OK) return capi_fail(rc, "batched PNG launch failed");
    return UCFP_OK;
}

int ucfp_png_batcher_stats(ucfp_png_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    b->r.core.stats(batches, items);
    return UCFP_OK;
}

// ---- uploads of ANY kind and size (SURVEY 8f N1 + N4): no geometry at creation ----
int ucfp_upload_batcher_create(ucfp_ctx* ctx, uint32_t algo, const ucfp_image_preprocess* pre, size_t max_batch, size_t max_bytes,
                               uint32_t max_delay_us, ucfp_upload_batcher** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    const size_t rec = ucfp_image_record_bytes(algo);
    if (!rec) return capi_fail(UCFP_E_UNSUPPORTED, "image algo mask %u", algo);
    if (max_batch == 0 || max_batch > 65536 || max_bytes == 0 || max_bytes >= ((size_t)1 << 32))
        return capi_fail(UCFP_E_INVALID, "batcher needs 1 <= max_batch <= 65536 and 1 <= max_bytes < 2^32");
    ucfp_upload_batcher* b = new (std::nothrow) ucfp_upload_batcher();
    if (!b) return capi_fail(UCFP_E_INDEX, "out of host memory");
    int rc = ucfp_ctx_create(ucfp::ctx_device(ctx), &b->jctx);
    if (rc) {
        delete b;
        return rc;
    }
    bool started[2] = {false, false};
    for (int lane = 0; lane < 2 && rc == UCFP_OK; lane++) {
        Ragged& r = lane ? b->rj : b->r;
        r.ctx = lane ? b->jctx : ctx;
        r.device = ucfp::ctx_device(ctx);
        r.kind = kUploadHash;
        r.algo = algo;
        if (pre) r.pre = *pre;
        r.unit = 1;
        r.rec = rec;
        r.max_batch = max_batch;
        r.max_units = max_bytes;
        for (int s = 0; s < 2; s++) {
            r.h_info[s] = new (std::nothrow) ucfp_upload_info[max_batch];
            if (!r.h_info[s]) rc = capi_fail(UCFP_E_INDEX, "out of host memory");
        }
        if (rc == UCFP_OK) {
            rc = start(&r, max_delay_us);       // (tears the lane down itself when it fails)
            started[lane] = rc == UCFP_OK;
        }
    }
    if (rc) {
        if (started[0]) teardown(&b->r);
        for (int s = 0; s < 2; s++) {
            if (!started[0]) delete[] b->r.h_info[s], b->r.h_info[s] = nullptr;
            if (!started[1]) delete[] b->rj.h_info[s], b->rj.h_info[s] = nullptr;
        }
        ucfp_ctx_destroy(b->jctx);
        delete b;
        return rc;
    }
    *out = b;
    return UCFP_OK;
}

void ucfp_upload_batcher_destroy(ucfp_upload_batcher* b) {
    if (!b) return;
    teardown(&b->r);
    teardown(&b->rj);
    ucfp_ctx_destroy(b->jctx);
    delete b;
}

int ucfp_upload_batcher_submit(ucfp_upload_batcher* b, const uint8_t* bytes, size_t len, uint8_t* out, int32_t* status) {
    if (!b || !out || (len && !bytes)) return capi_fail(UCFP_E_INVALID, "batcher/bytes/out is NULL");
    // the request thread looks at its own upload: what the device does not decode never takes a slot
    ucfp_upload_info info;
    const int pst = ucfp_image_probe(bytes, len, &info);
    Ragged& r = info.format == UCFP_UPLOAD_JPEG ? b->rj : b->r;
    if (pst != UCFP_OK || len > r.max_units) {
        memset(out, 0, r.rec);
        if (status) *status = pst != UCFP_OK ? pst : UCFP_IMAGE_NEEDS_HOST;     // (larger than a whole batch: the host path)
        return UCFP_OK;
    }
    ucfp::BatchCore::Ticket t;
    t.set = -1;
    if (!r.core.claim(len, &t)) return capi_fail(UCFP_E_INDEX, "batcher is shutting down");
    r.offsets(t.set)[t.slot] = t.at;
    r.h_info[t.set][t.slot] = info;
    memcpy(r.payload(t.set) + t.at, bytes, len);
    r.core.commit(t);
    const int rc = r.core.wait(t);
    if (rc == UCFP_OK) {
        memcpy(out, r.h_out[t.set] + r.out_head + t.slot * r.rec, r.rec);
        if (status) *status = reinterpret_cast<const int32_t*>(r.h_out[t.set])[t.slot];
    }
    r.core.release(t);
    if (rc != UCFP_OK) return capi_fail(rc, "batched upload launch failed");
    return UCFP_OK;
}

int ucfp_upload_batcher_stats(ucfp_upload_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    uint64_t bp = 0, ip = 0, bj = 0, ij = 0;
    b->r.core.stats(&bp, &ip);
    b->rj.core.stats(&bj, &ij);
    if (batches) *batches = bp + bj;
    if (items) *items = ip + ij;
    return UCFP_OK;
}

int ucfp_audio_batcher_create(ucfp_ctx* ctx, uint32_t sample_rate, const ucfp_wang_config* cfg, size_t max_batch,
                              size_t max_samples, uint32_t max_delay_us, ucfp_audio_batcher** out) {
    if (!ctx || !out) return capi_fail(UCFP_E_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (sample_rate < 1000 || sample_rate > 384000)
        return capi_fail(UCFP_E_MODALITY, "invalid sample rate %u (1 000 .. 384 000 Hz)", sample_rate);
    if (max_batch == 0 || max_batch > (1u << 20) || max_samples == 0 || max_samples >= ((size_t)1 << 30))
        return capi_fail(UCFP_E_INVALID, "batcher needs 1 <= max_batch <= 2^20 and 1 <= max_samples < 2^30");
    ucfp_audio_batcher* b = new (std::nothrow) ucfp_audio_batcher();
    if (!b) return capi_fail(UCFP_E_INDEX, "out of host memory");
    Ragged& r = b->r;
    r.ctx = ctx;
    r.device = ucfp::ctx_device(ctx);
    r.kind = kAudioWang;
    r.sample_rate = sample_rate;
    if (cfg) r.wang = *cfg;
    else r.wang = ucfp_wang_config{10, 63, 64, 30, -50.0f};
    r.unit = 4;
    r.rec = 0;
    r.max_batch = max_batch;
    r.max_units = max_samples;
    r.out_cap = ucfp_audio_wang_batch_max_hashes(max_samples, max_batch, sample_rate, &r.wang);
    const int rc = start(&b->r, max_delay_us);
    if (rc) {
        delete b;
        return rc;
    }
    *out = b;
    return UCFP_OK;
}

void ucfp_audio_batcher_destroy(ucfp_audio_batcher* b) {
    if (!b) return;
    teardown(&b->r);
    delete b;
}

int ucfp_audio_batcher_submit(ucfp_audio_batcher* b, const float* pcm, size_t n, uint8_t* out, size_t cap_hashes,
                              size_t* n_hashes) {
    if (!b || !n_hashes || (n && !pcm) || (cap_hashes && !out)) return capi_fail(UCFP_E_INVALID, "NULL argument");
    *n_hashes = 0;
    Ragged& r = b->r;
    ucfp::BatchCore::Ticket t;
    const int rc = submit(&r, pcm, n, &t);
    if (t.set < 0) return rc;
    uint64_t cnt = 0;
    if (rc == UCFP_OK) {
        const uint64_t* oo = reinterpret_cast<const uint64_t*>(r.h_out[t.set]);
        const uint64_t lo = oo[t.slot], hi = oo[t.slot + 1];
        cnt = hi - lo;
        const uint64_t have = hi <= r.out_cap ? cnt : (lo < r.out_cap ? r.out_cap - lo : 0);   // out_cap is an upper bound: have == cnt
        const size_t m = (size_t)(have < cap_hashes ? have : cap_hashes);
        if (m) memcpy(out, r.h_out[t.set] + r.out_head + lo * 8, m * 8);
    }
    r.core.release(t);
    if (rc != UCFP_OK) return capi_fail(rc, "batched audio launch failed");
    *n_hashes = (size_t)cnt;
    if (cnt > cap_hashes)
        return capi_fail(UCFP_E_INVALID, "output holds %zu hashes, %llu produced", cap_hashes, (unsigned long long)cnt);
    return UCFP_OK;
}

int ucfp_audio_batcher_stats(ucfp_audio_batcher* b, uint64_t* batches, uint64_t* items) {
    if (!b) return capi_fail(UCFP_E_INVALID, "batcher is NULL");
    b->r.core.stats(batches, items);
    return UCFP_OK;
}

}  // extern "C"
