/*
 * ucfp_oracle_jpeg.c -- CPU restatement of the JPEG front end.  TEST INFRASTRUCTURE ONLY: nothing under
 * ucfp_amd/ or bench.py's timed regions may call this (tests, smoke() and the cpu_baseline leg only).
 *
 * The reference accepts JPEG uploads (src/modality/image.rs:54 "PNG / JPEG / WebP / GIF / BMP"; decoders enabled at
 * Cargo.toml:143) and decodes them inside the SDK call (image.rs:68-70: imgfprint -> image::load_from_memory, i.e. the
 * `image` crate's JPEG decoder).  That crate is NOT in the tree (un-vendored), so this file restates the
 * PUBLISHED algorithms -- ITU-T T.81 (baseline sequential DCT, Huffman coding, restart intervals) and the Independent
 * JPEG Group's accurate integer inverse DCT (jidctint.c "islow": the LL&M 13-bit fixed-point factorisation, the default of
 * libjpeg / libjpeg-turbo) -- and is PINNED against libjpeg itself: tests/test_oracle_jpeg.py compares every pixel with
 * Pillow's decode of the same file in draft("L") mode (libjpeg-turbo, out_color_space = JCS_GRAYSCALE, JDCT_ISLOW).
 * Parity with the reference's own decoder stays unpinned (its IDCT rounding is its own); stated in DESIGN.md.
 *
 * J1 (ours): what is hashed of a JPEG is its LUMA COMPONENT as coded -- the Y plane of a YCbCr file, the only plane of a
 * greyscale one -- decoded at full resolution; chroma blocks are parsed (the entropy stream interleaves them) and never
 * transformed.  JFIF defines Y with the weights DESIGN I1 uses for RGB input (0.299 / 0.587 / 0.114), so this is the same
 * luma without the detour through upsampled RGB.  The host path (ucfp_amd/image.py) decodes JPEG the same way.
 *
 * Decoded here (everything else that parses as a JPEG -> JPG_NEEDS_HOST: the host's decoder takes it):
 *   SOF0 / SOF1 (sequential Huffman), 8-bit samples, 1 component or 3 components coded as YCbCr, ONE interleaved scan,
 *   the luma component sampled at the MCU's full resolution (4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1 ...), 8-bit quantisation
 *   tables, optional restart intervals.  Any irregularity of the entropy-coded data (a code that is not in the table, a
 *   run past coefficient 63, data that ends early, a stray marker) is NEEDS_HOST as well -- decoders differ in what they
 *   forgive, so the host's decides.  Not a JPEG at all (no SOI) -> JPG_CORRUPT.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { JPG_OK = 0, JPG_NEEDS_HOST = 1, JPG_CORRUPT = -1 };

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    int present;
    uint8_t bits[17];      /* bits[l] = codes of length l */
    uint8_t vals[256];
    int32_t maxcode[18];   /* largest code of length l (-1: none), T.81 F.2.2.3 */
    int32_t valptr[17];
    int32_t mincode[17];
} huff_t;

typedef struct {
    uint32_t w, h;
    int ncomp;
    int cid[3], hs[3], vs[3], tq[3], td[3], ta[3];
    int hmax, vmax;
    uint16_t qt[4][64];    /* zigzag order, as in the DQT segment */
    int qt_present[4];
    huff_t dc[4], ac[4];
    uint32_t restart;      /* MCUs per restart interval (0: none) */
    size_t scan;           /* offset of the entropy-coded data */
} jpg_t;

static void huff_derive(huff_t* h) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        k += h->bits[l];
        code += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
}

/* Header walk: everything up to and including SOS.  JPG_OK / JPG_NEEDS_HOST / JPG_CORRUPT. */
static int jpg_parse(const uint8_t* p, size_t n, jpg_t* J) {
    memset(J, 0, sizeof *J);
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return JPG_CORRUPT;
    size_t pos = 2;
    int have_sof = 0, adobe_transform = -1, jfif = 0;
    for (;;) {
        if (pos + 4 > n) return have_sof ? JPG_NEEDS_HOST : JPG_CORRUPT;
        if (p[pos] != 0xFF) return JPG_NEEDS_HOST;
        while (pos < n && p[pos] == 0xFF) pos++;   /* fill bytes */
        if (pos >= n) return JPG_NEEDS_HOST;
        const int m = p[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;   /* parameterless */
        if (m == 0xD9) return JPG_NEEDS_HOST;                                 /* EOI before a scan */
        if (pos + 2 > n) return JPG_NEEDS_HOST;
        const size_t len = (size_t)p[pos] << 8 | p[pos + 1];
        if (len < 2 || pos + len > n) return JPG_NEEDS_HOST;
        const uint8_t* s = p + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (have_sof || sl < 6) return JPG_NEEDS_HOST;
            have_sof = 1;
            if (s[0] != 8) return JPG_NEEDS_HOST;
            J->h = (uint32_t)s[1] << 8 | s[2];
            J->w = (uint32_t)s[3] << 8 | s[4];
            J->ncomp = s[5];
            if (J->w == 0 || J->h == 0) return JPG_NEEDS_HOST;    /* h = 0: DNL marker */
            if (J->ncomp != 1 && J->ncomp != 3) return JPG_NEEDS_HOST;
            if (sl < 6 + 3 * (size_t)J->ncomp) return JPG_NEEDS_HOST;
            for (int c = 0; c < J->ncomp; c++) {
                J->cid[c] = s[6 + 3 * c];
                J->hs[c] = s[7 + 3 * c] >> 4;
                J->vs[c] = s[7 + 3 * c] & 15;
                J->tq[c] = s[8 + 3 * c];
                if (J->hs[c] < 1 || J->hs[c] > 4 || J->vs[c] < 1 || J->vs[c] > 4 || J->tq[c] > 3) return JPG_NEEDS_HOST;
                if (J->hs[c] > J->hmax) J->hmax = J->hs[c];
                if (J->vs[c] > J->vmax) J->vmax = J->vs[c];
            }
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JPG_NEEDS_HOST;                                /* progressive, lossless, arithmetic, hierarchical */
        } else if (m == 0xCC) {
            return JPG_NEEDS_HOST;
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o < sl) {
                if (o + 17 > sl) return JPG_NEEDS_HOST;
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return JPG_NEEDS_HOST;
                huff_t* h = tc ? &J->ac[th] : &J->dc[th];
                memset(h, 0, sizeof *h);
                int cnt = 0;
                for (int l = 1; l <= 16; l++) cnt += (h->bits[l] = s[o + l]);
                if (cnt > 256 || o + 17 + (size_t)cnt > sl) return JPG_NEEDS_HOST;
                memcpy(h->vals, s + o + 17, (size_t)cnt);
                /* an over-subscribed code is not a prefix code */
                int code = 0;
                for (int l = 1; l <= 16; l++) {
                    code += h->bits[l];
                    if (code > (1 << l)) return JPG_NEEDS_HOST;
                    code <<= 1;
                }
                h->present = 1;
                huff_derive(h);
                o += 17 + (size_t)cnt;
            }
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (pq != 0 || tq > 3 || o + 65 > sl) return JPG_NEEDS_HOST;   /* 16-bit tables: host */
                for (int i = 0; i < 64; i++) J->qt[tq][i] = s[o + 1 + i];
                J->qt_present[tq] = 1;
                o += 65;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return JPG_NEEDS_HOST;
            J->restart = (uint32_t)s[0] << 8 | s[1];
        } else if (m == 0xE0) {
            if (sl >= 5 && memcmp(s, "JFIF", 5) == 0) jfif = 1;
        } else if (m == 0xEE) {
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) return JPG_NEEDS_HOST;
            const int ns = s[0];
            if (ns != J->ncomp || sl < 1 + 2 * (size_t)ns + 3) return JPG_NEEDS_HOST;   /* one interleaved scan */
            for (int c = 0; c < ns; c++) {
                if (s[1 + 2 * c] != J->cid[c]) return JPG_NEEDS_HOST;                  /* components in frame order */
                J->td[c] = s[2 + 2 * c] >> 4;
                J->ta[c] = s[2 + 2 * c] & 15;
                if (J->td[c] > 3 || J->ta[c] > 3 || !J->dc[J->td[c]].present || !J->ac[J->ta[c]].present)
                    return JPG_NEEDS_HOST;
                if (!J->qt_present[J->tq[c]]) return JPG_NEEDS_HOST;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return JPG_NEEDS_HOST;
            J->scan = pos + len;
            break;
        }
        pos += len;
    }
    /* colour space of a 3-component file, libjpeg's rule (jdapimin.c default_decompress_parms): JFIF -> YCbCr; Adobe ->
     * transform 1 = YCbCr, anything else = not YCbCr; neither -> by the component ids (1 2 3 -> YCbCr, 'R' 'G' 'B' -> RGB,
     * otherwise assumed YCbCr).  Only YCbCr has a luma component to take. */
    if (J->ncomp == 3) {
        int ycc = 1;
        if (jfif) ycc = 1;
        else if (adobe_transform >= 0) ycc = adobe_transform == 1;
        else if (J->cid[0] == 'R' && J->cid[1] == 'G' && J->cid[2] == 'B') ycc = 0;
        if (!ycc) return JPG_NEEDS_HOST;
        if (J->hs[0] != J->hmax || J->vs[0] != J->vmax) return JPG_NEEDS_HOST;   /* luma below the MCU's resolution */
    } else {
        J->hmax = J->hs[0] = 1;      /* a single-component scan is not interleaved: one block per MCU (T.81 A.2.2) */
        J->vmax = J->vs[0] = 1;
    }
    return JPG_OK;
}

int ucfp_oracle_jpeg_probe(const uint8_t* jpg, size_t n, uint32_t* w, uint32_t* h) {
    jpg_t J;
    const int rc = jpg_parse(jpg, n, &J);
    *w = J.w;
    *h = J.h;
    return rc;
}

/* ---- entropy-coded data: T.81 B.1.1.5 (byte stuffing), E.1.4 / F.2.2 (restart intervals, Huffman decoding) ----
 * Step 1 (what the device's scan kernel does too): the scan's bytes up to the first marker that is not RSTn, with the
 * stuffed zero bytes removed, cut at the RSTn markers into SEGMENTS -- one per restart interval, each starting on a byte. */
typedef struct {
    uint8_t* clean;
    size_t* seg;      /* nseg + 1 offsets into clean */
    size_t nseg;
} scan_t;

static int unstuff(const uint8_t* p, size_t n, size_t pos, scan_t* S) {
    S->clean = (uint8_t*)malloc(n - pos + 8);
    S->seg = (size_t*)malloc((n - pos + 2) / 2 * sizeof(size_t) + 4 * sizeof(size_t));
    size_t o = 0;
    S->nseg = 0;
    S->seg[0] = 0;
    int expect = 0;
    while (pos < n) {
        const uint8_t b = p[pos];
        if (b != 0xFF) {
            S->clean[o++] = b;
            pos++;
            continue;
        }
        if (pos + 1 >= n) break;                       /* a lone FF at the end of the file: data ends here */
        const uint8_t m = p[pos + 1];
        if (m == 0x00) {
            S->clean[o++] = 0xFF;
            pos += 2;
        } else if (m >= 0xD0 && m <= 0xD7) {
            if (m != (0xD0 | expect)) return JPG_NEEDS_HOST;      /* restart markers count modulo 8 (E.1.4) */
            expect = (expect + 1) & 7;
            S->seg[++S->nseg] = o;
            pos += 2;
        } else if (m == 0xFF) {
            return JPG_NEEDS_HOST;                     /* fill bytes inside entropy-coded data: host */
        } else {
            break;                                     /* EOI or any other marker ends the scan */
        }
    }
    S->seg[++S->nseg] = o;
    memset(S->clean + o, 0, 8);
    return JPG_OK;
}

/* Step 2: a bit reader over one segment.  Bits past the segment's end read as zero; a block that needed them is an
 * irregularity (checked once per block: used > avail). */
typedef struct {
    const uint8_t* p;
    size_t nbits, used;
} bits_t;

static uint32_t peek16(const bits_t* b) {
    uint32_t v = 0;
    for (int i = 0; i < 3; i++) {
        const size_t byte = b->used / 8 + (size_t)i;
        v = v << 8 | (byte * 8 < b->nbits ? b->p[byte] : 0);
    }
    return (v >> (8 - b->used % 8)) & 0xffff;
}

static int get_bits(bits_t* b, int n) {
    if (n == 0) return 0;
    const int v = (int)(peek16(b) >> (16 - n));
    b->used += (size_t)n;
    return v;
}

static int decode_sym(bits_t* b, const huff_t* h) {
    const uint32_t look = peek16(b);
    for (int l = 1; l <= 16; l++) {
        const int code = (int)(look >> (16 - l));
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) {
            b->used += (size_t)l;
            return h->vals[h->valptr[l] + code - h->mincode[l]];
        }
    }
    return -1;
}

static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

/* ---- jidctint.c (islow): 13-bit fixed point LL&M, dequantisation folded in; clamp as libjpeg-turbo's SIMD form ---- */
#define CONST_BITS 13
#define PASS1_BITS 2
#define DESCALE(x, n) (((x) + ((int32_t)1 << ((n)-1))) >> (n))
/* Range guards (J4): the 32-bit butterflies below are exact -- and every implementation of this IDCT gives the same pixels
 * (libjpeg's C code with its wrapping range table, libjpeg-turbo's 16-bit SIMD form with saturating packs, this one) -- only
 * while (a) every dequantised coefficient is within +-16383 (8 x 16383 x 11363 < 2^31 for the column pass; also what the
 * SIMD form's 16-bit multiply holds), (b) every column-pass result is within +-23000 (8 x 23000 x 11363 < 2^31 for the row
 * pass) and (c) every sample before the range limit is within -512 .. 511 of the centre.  No stream made by an encoder
 * comes near these; a crafted one that crosses them returns 1 and the file is JPG_NEEDS_HOST (the host's decoder decides). */
#define IDCT_MAX_COEF 16383
#define IDCT_MAX_PASS1 23000
static int idct_islow(const int16_t* coef, const uint16_t* qt_natural, uint8_t* out /* 8 x 8, stride 8 */) {
    int32_t ws[64];
    for (int i = 0; i < 64; i++) {
        const int32_t v = (int32_t)coef[i] * (int32_t)qt_natural[i];
        if (v > IDCT_MAX_COEF || v < -IDCT_MAX_COEF) return 1;
    }
    for (int c = 0; c < 8; c++) {
        int32_t in[8];
        for (int r = 0; r < 8; r++) in[r] = (int32_t)coef[8 * r + c] * (int32_t)qt_natural[8 * r + c];
        int32_t z2 = in[2], z3 = in[6];
        int32_t z1 = (z2 + z3) * 4433;
        int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        z2 = in[0];
        z3 = in[4];
        int32_t tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[7];
        tmp1 = in[5];
        tmp2 = in[3];
        tmp3 = in[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * 9633;
        tmp0 *= 2446;
        tmp1 *= 16819;
        tmp2 *= 25172;
        tmp3 *= 12299;
        z1 *= -7373;
        z2 *= -20995;
        z3 *= -16069;
        z4 *= -3196;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        ws[8 * 0 + c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        ws[8 * 7 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[8 * 1 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        ws[8 * 6 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[8 * 2 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        ws[8 * 5 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[8 * 3 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        ws[8 * 4 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int i = 0; i < 64; i++)
        if (ws[i] > IDCT_MAX_PASS1 || ws[i] < -IDCT_MAX_PASS1) return 1;
    int wild = 0;
    for (int r = 0; r < 8; r++) {
        const int32_t* w = ws + 8 * r;
        int32_t z2 = w[2], z3 = w[6];
        int32_t z1 = (z2 + z3) * 4433;
        int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        int32_t tmp0 = (w[0] + w[4]) * 8192, tmp1 = (w[0] - w[4]) * 8192;
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7];
        tmp1 = w[5];
        tmp2 = w[3];
        tmp3 = w[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * 9633;
        tmp0 *= 2446;
        tmp1 *= 16819;
        tmp2 *= 25172;
        tmp3 *= 12299;
        z1 *= -7373;
        z2 *= -20995;
        z3 *= -16069;
        z4 *= -3196;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        const int32_t o[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0,
                              tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
        for (int c = 0; c < 8; c++) {
            int32_t v = DESCALE(o[c], CONST_BITS + PASS1_BITS + 3) + 128;
            if (v < -512 + 128 || v > 511 + 128) wild = 1;
            out[8 * r + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    return wild;
}

/* Whole file -> luma plane (row stride w).  cap = bytes available in `luma`. */
int ucfp_oracle_jpeg_decode_luma(const uint8_t* jpg, size_t n, uint8_t* luma, size_t cap) {
    jpg_t J;
    int rc = jpg_parse(jpg, n, &J);
    if (rc) return rc;
    if ((size_t)J.w * J.h > cap) return JPG_CORRUPT;
    const uint32_t mcu_w = 8u * (uint32_t)J.hmax, mcu_h = 8u * (uint32_t)J.vmax;
    const uint32_t mx = (J.w + mcu_w - 1) / mcu_w, my = (J.h + mcu_h - 1) / mcu_h;
    uint16_t qn[64];
    for (int i = 0; i < 64; i++) qn[kZigzag[i]] = J.qt[J.tq[0]][i];
    scan_t S;
    rc = unstuff(jpg, n, J.scan, &S);
    const uint32_t total = mx * my;
    const uint32_t per = J.restart ? J.restart : total;
    if (rc == JPG_OK && S.nseg != (total + per - 1) / per) rc = JPG_NEEDS_HOST;     /* one segment per restart interval */
    for (uint32_t sg = 0; rc == JPG_OK && sg < S.nseg; sg++) {
        bits_t b = {S.clean + S.seg[sg], (S.seg[sg + 1] - S.seg[sg]) * 8, 0};
        int pred[3] = {0, 0, 0};
        const uint32_t m1 = (sg + 1) * per < total ? (sg + 1) * per : total;
        for (uint32_t m = sg * per; m < m1 && rc == JPG_OK; m++)
            for (int c = 0; c < J.ncomp && rc == JPG_OK; c++) {
                const int nb = J.ncomp == 1 ? 1 : J.hs[c] * J.vs[c];
                for (int bi = 0; bi < nb && rc == JPG_OK; bi++) {
                    int16_t coef[64];
                    memset(coef, 0, sizeof coef);
                    const int s = decode_sym(&b, &J.dc[J.td[c]]);
                    if (s < 0 || s > 11) { rc = JPG_NEEDS_HOST; break; }
                    pred[c] += s ? extend(get_bits(&b, s), s) : 0;
                    coef[0] = (int16_t)pred[c];
                    for (int k = 1; k < 64;) {
                        const int rs = decode_sym(&b, &J.ac[J.ta[c]]);
                        if (rs < 0) { rc = JPG_NEEDS_HOST; break; }
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r != 15) break;        /* EOB */
                            k += 16;
                            if (k > 64) rc = JPG_NEEDS_HOST;
                            continue;
                        }
                        k += r;
                        if (k > 63 || sz > 10) { rc = JPG_NEEDS_HOST; break; }
                        coef[kZigzag[k]] = (int16_t)extend(get_bits(&b, sz), sz);
                        k++;
                    }
                    if (rc == JPG_OK && b.used > b.nbits) rc = JPG_NEEDS_HOST;          /* the block read past its segment */
                    if (rc) break;
                    if (c == 0) {
                        const uint32_t bx = (m % mx) * (uint32_t)J.hmax + (uint32_t)(bi % J.hs[0]);
                        const uint32_t by = (m / mx) * (uint32_t)J.vmax + (uint32_t)(bi / J.hs[0]);
                        uint8_t px[64];
                        if (by * 8 >= J.h || bx * 8 >= J.w) continue;     /* MCU padding: decoded, never transformed */
                        if (idct_islow(coef, qn, px)) { rc = JPG_NEEDS_HOST; break; }
                        for (uint32_t y = 0; y < 8; y++)
                            for (uint32_t x = 0; x < 8; x++)
                                if (by * 8 + y < J.h && bx * 8 + x < J.w)
                                    luma[(size_t)(by * 8 + y) * J.w + bx * 8 + x] = px[8 * y + x];
                    }
                }
            }
    }
    free(S.clean);
    free(S.seg);
    return rc;
}
