// upload.hip -- ragged image batches behind the C ABI: frames (and encoded uploads) of ANY mix of geometries in one call.
//
// The reference's image route takes whatever a client uploads -- src/server/handlers.rs:232-302 hands the body to
// src/modality/image.rs:54-88 ("PNG / JPEG / WebP / GIF / BMP", any size) -- so a server in front of this library holds,
// at any moment, uploads of many sizes and two or three formats.  The round-2/3 entry points wanted ONE announced
// geometry per call and per batcher: nothing coalesces then.  Here:
//   ucfp_image_hash_ragged[_dev]        decoded frames, each with its own (offset, width, height, stride, pixfmt);
//   ucfp_image_probe[_batch_dev]        what an upload is: PNG / JPEG / something else, and the frame it decodes to;
//   ucfp_image_upload_decode_batch_dev  PNG and JPEG files of any sizes -> frames (the library lays them out);
//   ucfp_image_upload_hash_batch_dev    ... -> records: decode (png.hip, jpeg.hip on per-file tables), BLAKE3, hash.
// The host plans every launch from per-item descriptors (a few dozen nanoseconds each): which kernel group a frame
// belongs to, where its scanlines / coefficients / pixels go in the workspace; the tables travel to the device through
// one of two pinned staging pairs owned by the context.

#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/ucfp_hip.h"
#include "common.h"
#include "ctx.h"
#include "upload_probe.h"

using ucfp::capi_fail;
using ucfp::grow;

namespace {

// Tables of one call, packed into one blob (256-byte aligned pieces), uploaded with one copy.
struct Stager {
    std::vector<uint8_t> buf;
    size_t add(const void* p, size_t bytes) {
        const size_t at = (buf.size() + 255) & ~(size_t)255;
        buf.resize(at + bytes);
        if (bytes) memcpy(buf.data() + at, p, bytes);
        return at;
    }
    size_t reserve(size_t bytes) {
        const size_t at = (buf.size() + 255) & ~(size_t)255;
        buf.resize(at + bytes);
        return at;
    }
};

// blob -> device through one of the context's two pinned / device pairs; *pair tells which (record ctx->item_used[pair] on
// the stream behind the kernels that read the tables).  No tables: *d = nullptr, *pair = -1.
int stage_tables(ucfp_ctx* ctx, const Stager& sg, hipStream_t st, const uint8_t** d, int* pair) {
    *d = nullptr;
    *pair = -1;
    const size_t bytes = sg.buf.size();
    if (!bytes) return UCFP_OK;
    std::lock_guard<std::mutex> lk(ctx->item_mu);
    const int p = ctx->item_next;
    ctx->item_next ^= 1;
    hipError_t e = hipEventSynchronize(ctx->item_used[p]);          // the launches that read this pair last are done
    if (e == hipSuccess && ctx->item_cap[p] < bytes) {
        if (ctx->item_h[p]) (void)hipHostFree(ctx->item_h[p]);
        if (ctx->item_d[p]) (void)hipFree(ctx->item_d[p]);
        ctx->item_h[p] = ctx->item_d[p] = nullptr;
        ctx->item_cap[p] = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        e = hipHostMalloc((void**)&ctx->item_h[p], want, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->item_d[p], want);
        if (e == hipSuccess) ctx->item_cap[p] = want;
    }
    if (e != hipSuccess) return capi_fail(UCFP_E_INDEX, "item table staging failed: %s", hipGetErrorString(e));
    memcpy(ctx->item_h[p], sg.buf.data(), bytes);
    e = hipMemcpyAsync(ctx->item_d[p], ctx->item_h[p], bytes, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return capi_fail(UCFP_E_INDEX, "item table copy failed: %s", hipGetErrorString(e));
    *d = ctx->item_d[p];
    *pair = p;
    return UCFP_OK;
}

// ---- ragged hashing of decoded frames ----
struct HashPlan {
    size_t off[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};   // ImgItem tables of the three width groups (a launch each)
    size_t off_preset = 0, n_preset = 0;            // (slot, status) of the frames that are not hashed
    std::vector<uint32_t> big;                      // items for the many-waves-per-frame path, one call each
};

// items[k] fills slot slots[k] (k itself without `slots`).  Validates, sorts the frames into kernel groups, appends the
// tables to `sg`.  Frames outside the guards get status UCFP_E_MODALITY (a zero record), the batch goes on.
int plan_hash(ucfp_ctx* ctx, const uint8_t* base, size_t frames_bytes, const ucfp_image_item* items, const uint32_t* slots, size_t n,
              uint32_t min_dim, uint32_t max_dim, std::vector<uint32_t>* preset, Stager* sg, HashPlan* hp) {
    const size_t isz = ucfp::image_any_item_bytes();
    std::vector<uint8_t> tab[3];
    for (size_t k = 0; k < n; k++) {
        const ucfp_image_item& it = items[k];
        const uint32_t slot = slots ? slots[k] : (uint32_t)k;
        if (it.pixfmt < UCFP_PIX_GRAY8 || it.pixfmt > UCFP_PIX_RGBA8) return capi_fail(UCFP_E_INVALID, "item %zu: unknown pixfmt %d", k, it.pixfmt);
        const size_t bpp = it.pixfmt == UCFP_PIX_GRAY8 ? 1 : it.pixfmt == UCFP_PIX_RGB8 ? 3 : 4;
        if (it.width == 0 || it.height == 0 || it.width < min_dim || it.height < min_dim || it.width > max_dim || it.height > max_dim) {
            preset->push_back(slot);                  // Error::Modality for this frame (image.rs:70), the others go on
            preset->push_back((uint32_t)UCFP_E_MODALITY);
            continue;
        }
        if (it.row_stride < (size_t)it.width * bpp) return capi_fail(UCFP_E_INVALID, "item %zu: row_stride %u < width * bpp", k, it.row_stride);
        const uint64_t end = it.offset + (uint64_t)(it.height - 1) * it.row_stride + (uint64_t)it.width * bpp;
        if (end > frames_bytes || end < it.offset) return capi_fail(UCFP_E_INVALID, "item %zu reaches beyond the %zu bytes of frames", k, frames_bytes);
        uint32_t cls = 0, magic = 0, shift = 0;
        if ((size_t)it.width * it.height > ctx->any_max_pixels ||
            !ucfp::image_any_plan(base, it.offset, it.width, it.height, it.row_stride, it.pixfmt, &cls, &magic, &shift)) {
            hp->big.push_back((uint32_t)k);
            continue;
        }
        if (const int ge = ucfp::image_any_geometry_ready(ctx, it.width))
            return capi_fail(UCFP_E_INDEX, "column table of width %u: %s", it.width, hipGetErrorString((hipError_t)ge));
        const int g = ucfp::image_any_group(cls);
        tab[g].resize((hp->cnt[g] + 1) * isz);
        ucfp::image_any_item_write(tab[g].data(), hp->cnt[g]++, it.offset, it.width, it.height, it.row_stride, slot, cls, magic, shift);
    }
    for (int g = 0; g < 3; g++) hp->off[g] = sg->add(tab[g].data(), tab[g].size());
    return UCFP_OK;
}

int launch_hash(ucfp_ctx* ctx, const HashPlan& hp, const uint8_t* d_tab, uint32_t algo, const uint8_t* base, size_t frames_bytes,
                const ucfp_image_item* items, const uint32_t* slots, uint32_t min_dim, uint32_t max_dim, const uint8_t* exact, uint8_t* out,
                int32_t* status, hipStream_t st) {
    const size_t rec = algo == 7u ? 536 : 168;
    for (int g = 0; g < 3; g++)
        if (hp.cnt[g])
            ucfp::launch_image_hash_any(algo, base, d_tab + hp.off[g], hp.cnt[g], g, 0, 0, 0, 0, 0, 0, 0, base, base + frames_bytes, exact, out,
                                        status, ctx->geo, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return capi_fail(UCFP_E_INDEX, "ragged image launch failed: %s", hipGetErrorString(e));
    for (uint32_t k : hp.big) {
        const ucfp_image_item& it = items[k];
        const size_t slot = slots ? slots[k] : k;
        e = (hipError_t)ucfp::image_hash_ordered(ctx, algo, base + it.offset, 1, it.width, it.height, it.row_stride,
                                                 (size_t)it.row_stride * it.height, it.pixfmt, min_dim, max_dim,
                                                 exact ? exact + 32 * slot : nullptr, out + slot * rec, status ? status + slot : nullptr, st);
        if (e != hipSuccess) return capi_fail(UCFP_E_INDEX, "image launch failed: %s", hipGetErrorString(e));
    }
    return UCFP_OK;
}

int hash_ragged(ucfp_ctx* ctx, uint32_t algo, const uint8_t* base, size_t frames_bytes, const ucfp_image_item* items, size_t n,
                uint32_t min_dim, uint32_t max_dim, const uint8_t* exact, uint8_t* out, int32_t* status, hipStream_t st) {
    const size_t rec = algo == 7u ? 536 : 168;
    Stager sg;
    HashPlan hp;
    std::vector<uint32_t> preset;
    int rc = plan_hash(ctx, base, frames_bytes, items, nullptr, n, min_dim, max_dim, &preset, &sg, &hp);
    if (rc) return rc;
    const size_t o_pre = sg.add(preset.data(), preset.size() * 4);
    const uint8_t* d_tab = nullptr;
    int pair = -1;
    rc = stage_tables(ctx, sg, st, &d_tab, &pair);
    if (rc) return rc;
    rc = launch_hash(ctx, hp, d_tab, algo, base, frames_bytes, items, nullptr, min_dim, max_dim, exact, out, status, st);
    if (rc == UCFP_OK && !preset.empty()) {
        ucfp::launch_image_preset_list(reinterpret_cast<const uint32_t*>(d_tab + o_pre), preset.size() / 2, out, (uint32_t)rec, status, st);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = capi_fail(UCFP_E_INDEX, "ragged image launch failed: %s", hipGetErrorString(e));
    }
    if (pair >= 0) (void)hipEventRecord(ctx->item_used[pair], st);
    return rc;
}

// ---- encoded uploads ----
// Where everything of a batch of uploads goes: the decoders' tables (UpItem rows, the JPEG inverse DCT's workgroup map),
// the frames they decode to, the hash kernels' tables for those frames, and the files that are not decoded at all.
struct UploadPlan {
    std::vector<ucfp::UpItem> png, jpg;
    std::vector<uint32_t> first;                 // JPEG: first inverse-DCT workgroup of every entry (+ the total)
    std::vector<ucfp_image_item> frames;         // decoded frames (offsets into the frame area) ...
    std::vector<uint32_t> slots;                 // ... and the upload each belongs to
    std::vector<uint32_t> preset;                // (upload, status) of the uploads the device does not decode
    size_t frame_bytes = 0, raw_total = 0, seg_words = 0, coef_words = 0;
    uint32_t png_max_w[3] = {0, 0, 0};
};

// guards: hash entries pass the preprocess window (an upload outside it is Error::Modality without being decoded); the
// decode-only entry passes 1 .. 2^32 - 1
void plan_uploads(const ucfp_upload_info* info, size_t n, uint32_t min_dim, uint32_t max_dim, UploadPlan* up) {
    up->first.push_back(0);
    for (size_t i = 0; i < n; i++) {
        const ucfp_upload_info& u = info[i];
        int st = u.status;
        if (st == UCFP_OK && (u.format != UCFP_UPLOAD_PNG && u.format != UCFP_UPLOAD_JPEG)) st = UCFP_IMAGE_NEEDS_HOST;
        if (st == UCFP_OK && (u.pixfmt < UCFP_PIX_GRAY8 || u.pixfmt > UCFP_PIX_RGBA8)) st = UCFP_IMAGE_NEEDS_HOST;
        if (st == UCFP_OK && !ucfp::upload_device_decodes(u.format, u.width, u.height, u.pixfmt)) st = UCFP_IMAGE_NEEDS_HOST;
        if (st == UCFP_OK && (u.width < min_dim || u.height < min_dim || u.width > max_dim || u.height > max_dim)) st = UCFP_E_MODALITY;
        if (st != UCFP_OK) {
            up->preset.push_back((uint32_t)i);
            up->preset.push_back((uint32_t)st);
            continue;
        }
        const int fmt = u.format == UCFP_UPLOAD_JPEG ? UCFP_PIX_GRAY8 : u.pixfmt;
        const size_t bpp = fmt == UCFP_PIX_GRAY8 ? 1 : fmt == UCFP_PIX_RGB8 ? 3 : 4;
        const uint32_t row = (uint32_t)(((size_t)u.width * bpp + 15) & ~(size_t)15);      // 16-byte rows: the aligned loaders
        ucfp::UpItem it{};
        it.file = (uint32_t)i;
        it.w = u.width;
        it.h = u.height;
        it.pixfmt = fmt;
        it.row_stride = row;
        it.frame_off = up->frame_bytes;
        if (u.format == UCFP_UPLOAD_PNG) {
            it.aux_off = up->raw_total;
            up->raw_total += ucfp::png_raw_bytes(u.width, u.height, fmt);
            if (u.width > up->png_max_w[fmt]) up->png_max_w[fmt] = u.width;
            up->png.push_back(it);
        } else {
            ucfp::jpeg_plane_geometry(u.width, u.height, &it.bxp, &it.byp, &it.max_seg);
            it.aux_off = up->coef_words;
            it.seg_off = up->seg_words;
            up->coef_words += (size_t)it.bxp * it.byp * 64;
            up->seg_words += (size_t)it.max_seg + 2;
            up->first.push_back(up->first.back() + (uint32_t)(((size_t)it.bxp * it.byp + 255) / 256));
            up->jpg.push_back(it);
        }
        up->frames.push_back(ucfp_image_item{up->frame_bytes, u.width, u.height, row, fmt});
        up->slots.push_back((uint32_t)i);
        up->frame_bytes += (size_t)row * u.height;
    }
}

// uploads that are already on the device: one thread per file reads the header
__global__ void upload_probe_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offsets, size_t n,
                                    ucfp_upload_info* __restrict__ info) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ucfp_upload_info r;
    ucfp::upload_probe_bytes(blob + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), &r);
    info[i] = r;
}

// Decode (and, with `algo`, hash) a batch of uploads.  frames_out == nullptr: frames go to the context's workspace.
int upload_run(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, size_t blob_bytes,
               const ucfp_upload_info* info, uint32_t min_dim, uint32_t max_dim, uint8_t* frames_out, size_t frames_cap,
               ucfp_image_item* items_out, const uint8_t* d_exact, uint8_t* d_out, int32_t* d_status, hipStream_t st) {
    const size_t rec = algo ? (algo == 7u ? 536 : 168) : 0;
    std::vector<ucfp_upload_info> probed;
    // callers hold ctx->mu
    if (!info) {
        // the uploads are on the device only: probe them there, bring the 24 bytes per file back (the one synchronisation)
        probed.resize(n);
        int rc = grow(&ctx->b3_ws, &ctx->b3_ws_cap, n * sizeof(ucfp_upload_info));
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(st, ctx->png_done, 0));
        hipLaunchKernelGGL(upload_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_blob, d_offsets, n,
                           reinterpret_cast<ucfp_upload_info*>(ctx->b3_ws));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(probed.data(), ctx->b3_ws, n * sizeof(ucfp_upload_info), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        info = probed.data();
    }
    UploadPlan up;
    plan_uploads(info, n, min_dim, max_dim, &up);
    if (items_out)
        for (size_t i = 0; i < n; i++) items_out[i] = ucfp_image_item{0, 0, 0, 0, 0};
    if (items_out)
        for (size_t k = 0; k < up.frames.size(); k++) items_out[up.slots[k]] = up.frames[k];
    if (frames_out && up.frame_bytes > frames_cap)
        return capi_fail(UCFP_E_INVALID, "the decoded frames need %zu bytes, the buffer holds %zu", up.frame_bytes, frames_cap);
    // workspace: [PNG: gathered streams | info | scanlines] [JPEG: clean bytes | info | segments | tables | coefficients] [frames]
    ucfp::PngWs lp;
    ucfp::JpegWs lj;
    const size_t png_bytes = up.png.empty() ? 0 : ucfp::png_ragged_ws_bytes(up.png.size(), blob_bytes, up.raw_total, &lp);
    const size_t jpg_bytes = up.jpg.empty() ? 0 : ucfp::jpeg_ragged_ws_bytes(up.jpg.size(), blob_bytes, up.seg_words, up.coef_words, &lj);
    const size_t o_jpg = (png_bytes + 255) & ~(size_t)255, o_frames = (o_jpg + jpg_bytes + 255) & ~(size_t)255;
    int rc = grow(&ctx->png_ws, &ctx->png_ws_cap, o_frames + (frames_out ? 0 : up.frame_bytes) + 64);
    if (rc) return rc;
    uint8_t* frames = frames_out ? frames_out : ctx->png_ws + o_frames;
    // tables
    Stager sg;
    const size_t o_png = sg.add(up.png.data(), up.png.size() * sizeof(ucfp::UpItem));
    const size_t o_jtab = sg.add(up.jpg.data(), up.jpg.size() * sizeof(ucfp::UpItem));
    const size_t o_first = sg.add(up.first.data(), up.first.size() * 4);
    HashPlan hp;
    if (algo) {
        rc = plan_hash(ctx, frames, up.frame_bytes + 64, up.frames.data(), up.slots.data(), up.frames.size(), min_dim, max_dim, &up.preset, &sg,
                       &hp);
        if (rc) return rc;
    }
    const size_t o_pre = sg.add(up.preset.data(), up.preset.size() * 4);
    const uint8_t* d_tab = nullptr;
    int pair = -1;
    rc = stage_tables(ctx, sg, st, &d_tab, &pair);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(st, ctx->png_done, 0));
    // decode.  With records to make, the decoders' verdicts are merged at the end; decode-only callers get them right away.
    int32_t* dec_status = algo ? nullptr : d_status;
    // The PNG chain (scan, inflate, unfilter) is a few hundred one-wave decodes that run for tens of milliseconds on a chip
    // with 1024 SIMDs; the JPEG chain and the files' BLAKE3 are independent of it (own workspace regions, own frames), so with
    // both kinds in the batch they run on the context's side stream beside it (400 mixed uploads: 54.2 -> 35.1 ms).
    // (every allocation of the call happens before the fork: nothing between fork and join can fail and return)
    const size_t cvb = (ucfp::blake3_ws_bytes(n, blob_bytes) + 255) & ~(size_t)255;
    if (algo && !d_exact) {
        rc = grow(&ctx->b3_ws, &ctx->b3_ws_cap, cvb + n * 32);
        if (rc) return rc;
    }
    static const bool no_side = getenv("UCFP_UPLOAD_NO_SIDE_STREAM") != nullptr;   // (A/B)
    const bool fork = !up.png.empty() && !up.jpg.empty() && !no_side;
    hipStream_t sj = fork ? ctx->side : st;
    if (fork) {
        HIP_TRY(hipEventRecord(ctx->side_fork, st));
        HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->side_fork, 0));
    }
    if (!up.png.empty())
        ucfp::launch_png_decode_ragged(d_blob, d_offsets, reinterpret_cast<const ucfp::UpItem*>(d_tab + o_png), up.png.size(), up.png_max_w,
                                       ctx->png_ws, lp, frames, dec_status, st);
    if (!up.jpg.empty())
        ucfp::launch_jpeg_decode_ragged(d_blob, d_offsets, reinterpret_cast<const ucfp::UpItem*>(d_tab + o_jtab), up.jpg.size(),
                                        reinterpret_cast<const uint32_t*>(d_tab + o_first), up.first.back(), ctx->png_ws + o_jpg, lj, frames,
                                        dec_status, sj);
    if (algo && !d_exact) {
        // the files are here: their BLAKE3 (the records' `exact` field, image.rs:82) is computed on the device too
        ucfp::launch_blake3_batch(d_blob, d_offsets, n, ctx->b3_ws, ctx->b3_ws + cvb, sj);
        d_exact = ctx->b3_ws + cvb;
    }
    if (fork) {
        (void)hipEventRecord(ctx->side_join, ctx->side);
        (void)hipStreamWaitEvent(st, ctx->side_join, 0);
    }
    HIP_TRY(hipGetLastError());
    if (algo) {
        rc = launch_hash(ctx, hp, d_tab, algo, frames, up.frame_bytes + 64, up.frames.data(), up.slots.data(), min_dim, max_dim, d_exact, d_out,
                         d_status, st);
        if (rc) return rc;
        if (!up.png.empty())
            ucfp::launch_png_merge_status(ctx->png_ws, lp, up.png.size(), d_out, (uint32_t)rec, d_status, st,
                                          reinterpret_cast<const ucfp::UpItem*>(d_tab + o_png));
        if (!up.jpg.empty())
            ucfp::launch_jpeg_merge_status(ctx->png_ws + o_jpg, lj, up.jpg.size(), d_out, (uint32_t)rec, d_status, st,
                                           reinterpret_cast<const ucfp::UpItem*>(d_tab + o_jtab));
    }
    if (!up.preset.empty())
        ucfp::launch_image_preset_list(reinterpret_cast<const uint32_t*>(d_tab + o_pre), up.preset.size() / 2, algo ? d_out : nullptr,
                                       (uint32_t)rec, d_status, st);
    HIP_TRY(hipGetLastError());
    if (pair >= 0) HIP_TRY(hipEventRecord(ctx->item_used[pair], st));
    HIP_TRY(hipEventRecord(ctx->png_done, st));
    return UCFP_OK;
}

int upload_check(ucfp_ctx* ctx, const void* d_blob, const void* d_offsets, size_t n, size_t blob_bytes) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    if (n && (!d_blob || !d_offsets)) return capi_fail(UCFP_E_INVALID, "blob/offsets is NULL");
    if (n > 0x7fffffffu || blob_bytes >= ((size_t)1 << 32)) return capi_fail(UCFP_E_INVALID, "upload batch too large for one call");
    return UCFP_OK;
}

}  // namespace

extern "C" {

int ucfp_image_hash_ragged_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_frames, size_t frames_bytes,
                               const ucfp_image_item* items, size_t n, const ucfp_image_preprocess* pre, const uint8_t* d_exact,
                               uint8_t* d_out, int32_t* d_status, void* stream) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    if (ucfp_image_record_bytes(algo) == 0)
        return capi_fail(UCFP_E_UNSUPPORTED, "image algo mask %u is not one of ahash|phash|dhash|multi", algo);
    if (n == 0) return UCFP_OK;
    if (!items || !d_frames || !d_out) return capi_fail(UCFP_E_INVALID, "items/frames/out is NULL");
    if (n > 0x7fffffffu) return capi_fail(UCFP_E_INVALID, "batch of %zu frames exceeds one launch", n);
    const uint32_t min_dim = pre ? pre->min_dimension : 32u, max_dim = pre ? pre->max_dimension : 8192u;
    HIP_TRY(hipSetDevice(ctx->device));
    return hash_ragged(ctx, algo, d_frames, frames_bytes, items, n, min_dim, max_dim, d_exact, d_out, d_status, (hipStream_t)stream);
}

int ucfp_image_hash_ragged(ucfp_ctx* ctx, uint32_t algo, const uint8_t* frames, size_t frames_bytes, const ucfp_image_item* items,
                           size_t n, const ucfp_image_preprocess* pre, const uint8_t* exact, uint8_t* out, int32_t* status) {
    if (!ctx) return capi_fail(UCFP_E_INVALID, "ctx is NULL");
    const size_t rec = ucfp_image_record_bytes(algo);
    if (rec == 0) return capi_fail(UCFP_E_UNSUPPORTED, "image algo mask %u is not one of ahash|phash|dhash|multi", algo);
    if (n == 0) return UCFP_OK;
    if (!items || !frames || !out) return capi_fail(UCFP_E_INVALID, "items/frames/out is NULL");
    if (n > 0x7fffffffu) return capi_fail(UCFP_E_INVALID, "batch of %zu frames exceeds one launch", n);
    const uint32_t min_dim = pre ? pre->min_dimension : 32u, max_dim = pre ? pre->max_dimension : 8192u;
    const size_t in_bytes = (frames_bytes + 64 + 255) & ~(size_t)255;
    const size_t o_ex = (n * rec + 255) & ~(size_t)255, o_st = o_ex + ((n * 32 + 255) & ~(size_t)255);
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = grow(&ctx->stage_in, &ctx->stage_in_cap, in_bytes);
    if (rc) return rc;
    rc = grow(&ctx->stage_out, &ctx->stage_out_cap, o_st + n * 4);
    if (rc) return rc;
    hipStream_t st = ctx->host_stream;
    HIP_TRY(hipMemcpyAsync(ctx->stage_in, frames, frames_bytes, hipMemcpyHostToDevice, st));
    uint8_t* d_exact = ctx->stage_out + o_ex;
    int32_t* d_status = reinterpret_cast<int32_t*>(ctx->stage_out + o_st);
    if (exact) HIP_TRY(hipMemcpyAsync(d_exact, exact, n * 32, hipMemcpyHostToDevice, st));
    rc = hash_ragged(ctx, algo, ctx->stage_in, frames_bytes, items, n, min_dim, max_dim, exact ? d_exact : nullptr, ctx->stage_out,
                     d_status, st);
    if (rc) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, ctx->stage_out, n * rec, hipMemcpyDeviceToHost, st));
    if (status) HIP_TRY(hipMemcpyAsync(status, d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return UCFP_OK;
}

// ---- uploads ----
int ucfp_image_probe(const uint8_t* bytes, size_t len, ucfp_upload_info* info) {
    if (!info || (len && !bytes)) return capi_fail(UCFP_E_INVALID, "NULL argument");
    const int rc = ucfp::upload_probe_bytes(bytes, len, info);
    if (rc < 0) return capi_fail(rc, "not an image this library knows (empty, or a damaged PNG / JPEG header)");
    return rc;
}

int ucfp_image_probe_batch_dev(ucfp_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, ucfp_upload_info* d_info,
                               void* stream) {
    int rc = upload_check(ctx, d_blob, d_offsets, n, 0);
    if (rc) return rc;
    if (n == 0) return UCFP_OK;
    if (!d_info) return capi_fail(UCFP_E_INVALID, "info is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(upload_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_blob, d_offsets, n, d_info);
    HIP_TRY(hipGetLastError());
    return UCFP_OK;
}

size_t ucfp_image_upload_frames_bytes(const ucfp_upload_info* info, size_t n) {
    if (!info) return 0;
    UploadPlan up;
    plan_uploads(info, n, 1u, 0xffffffffu, &up);
    return up.frame_bytes + 64;
}

int ucfp_image_upload_decode_batch_dev(ucfp_ctx* ctx, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n, size_t blob_bytes,
                                       const ucfp_upload_info* info, uint8_t* d_frames, size_t frames_bytes, ucfp_image_item* items,
                                       int32_t* d_status, void* stream) {
    int rc = upload_check(ctx, d_blob, d_offsets, n, blob_bytes);
    if (rc) return rc;
    if (n == 0) return UCFP_OK;
    if (!d_frames || !items) return capi_fail(UCFP_E_INVALID, "frames/items is NULL");
    if ((uintptr_t)d_frames & 15u) return capi_fail(UCFP_E_INVALID, "the frame buffer must be 16-byte aligned");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    return upload_run(ctx, 0, d_blob, d_offsets, n, blob_bytes, info, 1u, 0xffffffffu, d_frames, frames_bytes, items, nullptr, nullptr, d_status,
                      (hipStream_t)stream);
}

int ucfp_image_upload_hash_batch_dev(ucfp_ctx* ctx, uint32_t algo, const uint8_t* d_blob, const uint64_t* d_offsets, size_t n,
                                     size_t blob_bytes, const ucfp_upload_info* info, const ucfp_image_preprocess* pre,
                                     const uint8_t* d_exact, uint8_t* d_out, int32_t* d_status, void* stream) {
    int rc = upload_check(ctx, d_blob, d_offsets, n, blob_bytes);
    if (rc) return rc;
    if (ucfp_image_record_bytes(algo) == 0)
        return capi_fail(UCFP_E_UNSUPPORTED, "image algo mask %u is not one of ahash|phash|dhash|multi", algo);
    if (n == 0) return UCFP_OK;
    if (!d_out) return capi_fail(UCFP_E_INVALID, "out is NULL");
    const uint32_t min_dim = pre ? pre->min_dimension : 32u, max_dim = pre ? pre->max_dimension : 8192u;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    return upload_run(ctx, algo, d_blob, d_offsets, n, blob_bytes, info, min_dim, max_dim, nullptr, 0, nullptr, d_exact, d_out, d_status,
                      (hipStream_t)stream);
}

}  // extern "C"
