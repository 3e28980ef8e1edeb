// upload_probe.h -- what an encoded upload is, from its header bytes: one statement for the host (ucfp_png_probe,
// ucfp_jpeg_probe, ucfp_image_probe) and the device (upload_probe_kernel, for uploads that are already in device memory).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/ucfp_hip.h"

namespace ucfp {

// IHDR of a PNG -> geometry and the pixel format the file DECODES to.  UCFP_OK, UCFP_IMAGE_NEEDS_HOST (a valid PNG of a kind
// the device decoder hands back: 16-bit, 1/2/4-bit, interlaced) or UCFP_E_MODALITY (not a PNG / damaged IHDR).
__host__ __device__ inline int png_probe_bytes(const uint8_t* png, size_t len, uint32_t* width, uint32_t* height, int* pixfmt) {
    const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (len < 8 + 25) return UCFP_E_MODALITY;
    for (int i = 0; i < 8; i++)
        if (png[i] != sig[i]) return UCFP_E_MODALITY;
    auto be = [&](size_t o) { return (uint32_t)png[o] << 24 | (uint32_t)png[o + 1] << 16 | (uint32_t)png[o + 2] << 8 | png[o + 3]; };
    if (be(8) != 13 || png[12] != 'I' || png[13] != 'H' || png[14] != 'D' || png[15] != 'R') return UCFP_E_MODALITY;
    *width = be(16);
    *height = be(20);
    const int depth = png[24], ctype = png[25], comp = png[26], filt = png[27], lace = png[28];
    if (*width == 0 || *height == 0 || comp != 0 || filt != 0 || lace > 1) return UCFP_E_MODALITY;
    if (depth != 8 || lace != 0) return UCFP_IMAGE_NEEDS_HOST;
    // the format the file DECODES to: indexed colour -> RGB8 through its palette, grey + alpha -> GRAY8 (alpha dropped)
    if (ctype == 0 || ctype == 4) *pixfmt = UCFP_PIX_GRAY8;
    else if (ctype == 2 || ctype == 3) *pixfmt = UCFP_PIX_RGB8;
    else if (ctype == 6) *pixfmt = UCFP_PIX_RGBA8;
    else return UCFP_IMAGE_NEEDS_HOST;
    return UCFP_OK;
}

// Frame header of a JPEG -> geometry.  UCFP_OK (SOF0 / SOF1, 8-bit, 1 or 3 components: the scan kernel decides the rest),
// UCFP_IMAGE_NEEDS_HOST (progressive, arithmetic, 12-bit ..., or no frame header found) or UCFP_E_MODALITY (no SOI).
__host__ __device__ inline int jpeg_probe_bytes(const uint8_t* jpg, size_t len, uint32_t* width, uint32_t* height) {
    *width = *height = 0;
    if (len < 4 || jpg[0] != 0xFF || jpg[1] != 0xD8) return UCFP_E_MODALITY;
    size_t pos = 2;
    for (;;) {
        if (pos + 4 > len || jpg[pos] != 0xFF) return UCFP_IMAGE_NEEDS_HOST;
        while (pos < len && jpg[pos] == 0xFF) pos++;
        if (pos >= len) return UCFP_IMAGE_NEEDS_HOST;
        const int m = jpg[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9 || m == 0xDA || pos + 2 > len) return UCFP_IMAGE_NEEDS_HOST;      // a scan before any frame header
        const size_t l = (size_t)jpg[pos] << 8 | jpg[pos + 1];
        if (l < 2 || pos + l > len) return UCFP_IMAGE_NEEDS_HOST;
        if (m == 0xC0 || m == 0xC1) {
            if (l < 8 || jpg[pos + 2] != 8) return UCFP_IMAGE_NEEDS_HOST;
            *height = (uint32_t)jpg[pos + 3] << 8 | jpg[pos + 4];
            *width = (uint32_t)jpg[pos + 5] << 8 | jpg[pos + 6];
            const int nc = jpg[pos + 7];
            if (*width == 0 || *height == 0 || (nc != 1 && nc != 3)) return UCFP_IMAGE_NEEDS_HOST;
            return UCFP_OK;
        }
        if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8) return UCFP_IMAGE_NEEDS_HOST;   // progressive, arithmetic ...
        pos += l;
    }
}

// A PNG is inflated by ONE wave, serially in its bit stream (png.hip): ~33 MB/s of scanlines per file, whatever the batch --
// a batch of uploads takes as long as its largest PNG.  Up to this many scanline bytes (about 590 x 590 RGB: ~30 ms) the
// device's aggregate rate (150-220 k files/s over a full batch) is worth that latency; a larger PNG goes to the host, whose
// zlib inflates it ten times faster than one wave does.  (The uniform-geometry PNG entry points have no such limit.)
constexpr size_t kUploadPngMaxRawBytes = (size_t)1 << 20;

// what the device decoders take of an upload of this geometry (beyond it: the host's decoder)
__host__ __device__ inline bool upload_device_decodes(int format, uint32_t w, uint32_t h, int pixfmt) {
    if (w == 0 || h == 0 || w > 16384 || h > 16384) return false;
    if (format == UCFP_UPLOAD_PNG) {
        const size_t bpp = pixfmt == UCFP_PIX_GRAY8 ? 2 : pixfmt == UCFP_PIX_RGB8 ? 3 : 4;      // (a grey + alpha file: 2 bytes per pixel)
        if ((size_t)h * ((size_t)w * bpp + 1) > kUploadPngMaxRawBytes) return false;
        if ((size_t)w * bpp > 60000) return false;                                              // the unfilter kernel's row buffer
    }
    return true;
}

__host__ __device__ inline int upload_probe_bytes(const uint8_t* bytes, size_t len, ucfp_upload_info* info) {
    ucfp_upload_info r;
    r.format = UCFP_UPLOAD_OTHER;
    r.status = UCFP_IMAGE_NEEDS_HOST;
    r.width = r.height = 0;
    r.pixfmt = UCFP_PIX_GRAY8;
    r.reserved = 0;
    if (len == 0) {
        r.status = UCFP_E_MODALITY;                  // an empty body is no image in any format (image.rs:70 -> 400)
    } else if (len >= 8 && bytes[0] == 137 && bytes[1] == 80 && bytes[2] == 78 && bytes[3] == 71) {
        r.format = UCFP_UPLOAD_PNG;
        int fmt = 0;
        r.status = png_probe_bytes(bytes, len, &r.width, &r.height, &fmt);
        r.pixfmt = fmt;
    } else if (len >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
        r.format = UCFP_UPLOAD_JPEG;
        r.status = jpeg_probe_bytes(bytes, len, &r.width, &r.height);
        r.pixfmt = UCFP_PIX_GRAY8;                   // what is decoded of a JPEG is its luma plane (DESIGN J1)
    }
    // (WebP, GIF, BMP ... and anything unrecognised: the host's decoder decides -- the reference's `image` crate sniffs more
    // formats than this library decodes)
    if (r.status == UCFP_OK && !upload_device_decodes(r.format, r.width, r.height, r.pixfmt)) r.status = UCFP_IMAGE_NEEDS_HOST;
    *info = r;
    return r.status;
}

}  // namespace ucfp
