/* ucfp_oracle_png.c -- CPU restatement of the PNG front end.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference decodes inside the SDK call: imgfprint::ImageFingerprinter::fingerprint_with_preprocess(bytes, ..)
 * (src/modality/image.rs:68-70, :176-179) hands the upload to the `image` crate (image::load_from_memory ->
 * png 0.17 decoder -> DynamicImage).  Neither crate is vendored under /root/reference, so this file restates the
 * PUBLISHED formats they implement: PNG (W3C PNG 2nd ed. / RFC 2083: chunk layout 5.3, IHDR 11.2.2, filter types
 * 9.2, Paeth 9.4), zlib (RFC 1950) and deflate (RFC 1951: 3.2.3 block types, 3.2.5 length/distance codes, 3.2.6
 * fixed codes, 3.2.7 dynamic codes).
 *
 * PINNED: tests/test_oracle_png.py checks ucfp_oracle_inflate against zlib.decompress and ucfp_oracle_png_decode
 * against Pillow's decoded pixels (both libraries wrap the reference implementations of these formats) on the
 * synthetic config-1 set, on every filter type, colour type and compression level, and on stored / fixed / dynamic
 * blocks.
 *
 * Scope (mirrors the HIP path): 8-bit greyscale (0), RGB (2), RGBA (6), non-interlaced, with or without a well-formed tRNS
 * chunk (it only adds an alpha channel: luma takes none).  Everything else
 * returns UCFP_PNG_NEEDS_HOST: the host's own decoder (the `image` crate) takes those.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PNG_OK 0
#define PNG_NEEDS_HOST 1
#define PNG_CORRUPT (-1)

typedef struct {
    const uint8_t* in;
    size_t n;
    size_t pos;      /* next byte */
    uint32_t bitbuf; /* LSB-first */
    int bitcnt;
    int err;
} Bits;

static uint32_t need(Bits* b, int k) { /* peek k <= 16 bits, LSB first (RFC 1951 3.1.1) */
    while (b->bitcnt < k) {
        uint32_t byte = 0;
        if (b->pos < b->n) byte = b->in[b->pos];
        else b->err = 1;
        b->pos++;
        b->bitbuf |= byte << b->bitcnt;
        b->bitcnt += 8;
    }
    return b->bitbuf & ((1u << k) - 1u);
}
static uint32_t take(Bits* b, int k) {
    if (k == 0) return 0;
    uint32_t v = need(b, k);
    b->bitbuf >>= k;
    b->bitcnt -= k;
    return v;
}

typedef struct {
    uint16_t count[16];   /* codes of each length */
    uint16_t symbol[320]; /* symbols ordered by (length, value) */
} Huff;

/* Canonical code from lengths (RFC 1951 3.2.2).  Returns 0 complete, 1 incomplete, -1 over-subscribed. */
static int huff_build(Huff* h, const uint8_t* len, int n) {
    uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int s = 0; s < n; s++) h->count[len[s]]++;
    if (h->count[0] == n) return 1;
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= h->count[l];
        if (left < 0) return -1;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h->count[l];
    for (int s = 0; s < n; s++)
        if (len[s]) h->symbol[offs[len[s]]++] = (uint16_t)s;
    return left > 0 ? 1 : 0;
}

static int huff_decode(Bits* b, const Huff* h) { /* bit by bit, codes are packed MSB first (3.1.1) */
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        code |= (int)take(b, 1);
        int cnt = h->count[l];
        if (code - cnt < first) return h->symbol[index + (code - first)];
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static int inflate_codes(Bits* b, const Huff* ll, const Huff* dd, uint8_t* out, size_t cap, size_t* op) {
    for (;;) {
        int sym = huff_decode(b, ll);
        if (sym < 0 || b->err) return PNG_CORRUPT;
        if (sym < 256) {
            if (*op >= cap) return PNG_CORRUPT;
            out[(*op)++] = (uint8_t)sym;
        } else if (sym == 256) {
            return PNG_OK;
        } else {
            sym -= 257;
            if (sym >= 29) return PNG_CORRUPT;
            size_t len = kLenBase[sym] + take(b, kLenExtra[sym]);
            int ds = huff_decode(b, dd);
            if (ds < 0 || ds >= 30) return PNG_CORRUPT;
            size_t dist = kDistBase[ds] + take(b, kDistExtra[ds]);
            if (b->err || dist > *op || *op + len > cap) return PNG_CORRUPT;
            for (size_t i = 0; i < len; i++, (*op)++) out[*op] = out[*op - dist];
        }
    }
}

/* Raw deflate stream (RFC 1951).  *produced = bytes written. */
static int inflate_raw(Bits* b, uint8_t* out, size_t cap, size_t* produced) {
    size_t op = 0;
    int last;
    do {
        last = (int)take(b, 1);
        int type = (int)take(b, 2);
        if (b->err) return PNG_CORRUPT;
        if (type == 0) { /* stored: skip to the byte boundary, LEN, NLEN, bytes */
            b->bitbuf = 0;
            b->bitcnt = 0;
            if (b->pos + 4 > b->n) return PNG_CORRUPT;
            uint32_t len = b->in[b->pos] | (uint32_t)b->in[b->pos + 1] << 8;
            uint32_t nlen = b->in[b->pos + 2] | (uint32_t)b->in[b->pos + 3] << 8;
            b->pos += 4;
            if ((len ^ 0xffffu) != nlen || b->pos + len > b->n || op + len > cap) return PNG_CORRUPT;
            memcpy(out + op, b->in + b->pos, len);
            b->pos += len;
            op += len;
        } else if (type == 1) { /* fixed codes, 3.2.6 */
            uint8_t len[320];
            Huff ll, dd;
            int s = 0;
            for (; s < 144; s++) len[s] = 8;
            for (; s < 256; s++) len[s] = 9;
            for (; s < 280; s++) len[s] = 7;
            for (; s < 288; s++) len[s] = 8;
            huff_build(&ll, len, 288);
            for (s = 0; s < 30; s++) len[s] = 5;
            huff_build(&dd, len, 30);
            int rc = inflate_codes(b, &ll, &dd, out, cap, &op);
            if (rc) return rc;
        } else if (type == 2) { /* dynamic codes, 3.2.7 */
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int nlen = (int)take(b, 5) + 257, ndist = (int)take(b, 5) + 1, ncode = (int)take(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return PNG_CORRUPT;
            uint8_t len[320];
            Huff cl, ll, dd;
            memset(len, 0, sizeof len);
            for (int i = 0; i < ncode; i++) len[order[i]] = (uint8_t)take(b, 3);
            if (huff_build(&cl, len, 19) != 0) return PNG_CORRUPT;   /* the code-length code must be complete */
            uint8_t lens[320];
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = huff_decode(b, &cl);
                if (sym < 0 || b->err) return PNG_CORRUPT;
                if (sym < 16) {
                    lens[idx++] = (uint8_t)sym;
                } else {
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return PNG_CORRUPT;
                        prev = lens[idx - 1];
                        rep = 3 + (int)take(b, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)take(b, 3);
                    } else {
                        rep = 11 + (int)take(b, 7);
                    }
                    if (idx + rep > nlen + ndist) return PNG_CORRUPT;
                    while (rep--) lens[idx++] = (uint8_t)prev;
                }
            }
            if (lens[256] == 0) return PNG_CORRUPT;
            int r = huff_build(&ll, lens, nlen);
            if (r < 0 || (r > 0 && nlen - ll.count[0] != 1)) return PNG_CORRUPT;   /* incomplete only with one code */
            r = huff_build(&dd, lens + nlen, ndist);
            if (r < 0 || (r > 0 && ndist - dd.count[0] > 1)) return PNG_CORRUPT;   /* no distance code at all: a block of literals */
            int rc = inflate_codes(b, &ll, &dd, out, cap, &op);
            if (rc) return rc;
        } else {
            return PNG_CORRUPT;
        }
    } while (!last);
    *produced = op;
    return PNG_OK;
}

static uint32_t adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) {
        a = (a + p[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    return b << 16 | a;
}

/* zlib stream (RFC 1950): CMF/FLG, deflate data, Adler-32 of the output.  *checksum_ok = 0 when the data inflated
 * but the 4-byte trailer is absent or differs. */
static int inflate_zlib(const uint8_t* z, size_t n, uint8_t* out, size_t cap, size_t* produced, int* checksum_ok) {
    *produced = 0;
    *checksum_ok = 0;
    if (n < 6) return PNG_CORRUPT;
    if ((z[0] & 15) != 8 || (z[0] >> 4) > 7 || ((z[0] << 8 | z[1]) % 31) != 0 || (z[1] & 0x20)) return PNG_CORRUPT;
    Bits b = {z, n, 2, 0, 0, 0};
    int rc = inflate_raw(&b, out, cap, produced);
    if (rc) return rc;
    /* the deflate data ends inside byte pos-1 (bits already pulled into the buffer belong to whole bytes) */
    size_t end = b.pos - (size_t)(b.bitcnt / 8);
    if (end + 4 > n) return PNG_OK;
    uint32_t want = (uint32_t)z[end] << 24 | (uint32_t)z[end + 1] << 16 | (uint32_t)z[end + 2] << 8 | z[end + 3];
    *checksum_ok = adler32(out, *produced) == want;
    return PNG_OK;
}

/* As zlib's uncompress(): a missing or wrong Adler-32 is an error. */
int ucfp_oracle_inflate(const uint8_t* z, size_t n, uint8_t* out, size_t cap, size_t* produced) {
    int ok = 0;
    int rc = inflate_zlib(z, n, out, cap, produced, &ok);
    return rc ? rc : (ok ? PNG_OK : PNG_CORRUPT);
}

static uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

static uint32_t crc32_png(const uint8_t* p, size_t n) { /* PNG 5.5 / annex D */
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    }
    return c ^ 0xffffffffu;
}

/* IHDR of a PNG: geometry and the pixel format of include/ucfp_hip.h (0 GRAY8, 1 RGB8, 2 RGBA8).
 * PNG_NEEDS_HOST for anything the HIP path hands back to the host decoder. */
int ucfp_oracle_png_probe(const uint8_t* png, size_t n, uint32_t* w, uint32_t* h, int* pixfmt) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (n < 8 + 25 || memcmp(png, sig, 8) != 0) return PNG_CORRUPT;
    if (be32(png + 8) != 13 || memcmp(png + 12, "IHDR", 4) != 0) return PNG_CORRUPT;
    *w = be32(png + 16);
    *h = be32(png + 20);
    int depth = png[24], ctype = png[25], comp = png[26], filt = png[27], lace = png[28];
    if (*w == 0 || *h == 0 || comp != 0 || filt != 0 || lace > 1) return PNG_CORRUPT;
    if (depth != 8 || lace != 0) return PNG_NEEDS_HOST;
    /* the format the file DECODES to: indexed colour (3) -> RGB8 through PLTE, grey + alpha (4) -> GRAY8 (alpha dropped:
     * luma takes no alpha, DESIGN I1; the host path's Pillow convert gives the same luma) */
    if (ctype == 0 || ctype == 4) *pixfmt = 0;
    else if (ctype == 2 || ctype == 3) *pixfmt = 1;
    else if (ctype == 6) *pixfmt = 2;
    else return PNG_NEEDS_HOST;
    return PNG_OK;
}

static int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

/* Whole file -> packed pixels (row stride w * bpp).  cap = bytes available in `pixels`. */
int ucfp_oracle_png_decode(const uint8_t* png, size_t n, uint8_t* pixels, size_t cap) {
    uint32_t w, h;
    int fmt;
    int rc = ucfp_oracle_png_probe(png, n, &w, &h, &fmt);
    if (rc) return rc;
    const int ctype = png[25];
    const size_t obpp = fmt == 0 ? 1 : fmt == 1 ? 3 : 4;                       /* bytes per pixel that leave */
    const size_t bpp = ctype == 3 ? 1 : ctype == 4 ? 2 : obpp, row = (size_t)w * bpp;   /* ... and in the file */
    if ((size_t)w * obpp * h > cap) return PNG_CORRUPT;
    uint8_t plte[768];
    size_t plte_n = 0;
    memset(plte, 0, sizeof plte);                                              /* entries the file lacks are black */
    /* walk the chunks (5.3): length, type, data, CRC; IDATs must be consecutive (5.6) */
    uint8_t* z = (uint8_t*)malloc(n);
    size_t zn = 0, pos = 8;
    int seen_idat = 0, idat_done = 0, seen_end = 0, needs_host = 0;
    rc = PNG_OK;
    while (pos + 12 <= n) {
        uint32_t len = be32(png + pos);
        const uint8_t* type = png + pos + 4;
        if (len > 0x7fffffffu || pos + 12 + (size_t)len > n) { rc = PNG_CORRUPT; break; }
        if (crc32_png(png + pos + 4, 4 + (size_t)len) != be32(png + pos + 8 + len)) {
            /* P5: a critical chunk with a bad CRC is damage; an ancillary one is a checksum-only failure of data the
             * pixels do not depend on -- decoders differ, the host's decoder decides */
            rc = (type[0] & 0x20) ? PNG_NEEDS_HOST : PNG_CORRUPT;
            break;
        }
        if (memcmp(type, "IDAT", 4) == 0) {
            if (idat_done) { rc = PNG_CORRUPT; break; }
            seen_idat = 1;
            memcpy(z + zn, png + pos + 8, len);
            zn += len;
        } else {
            if (seen_idat) idat_done = 1;
            if (memcmp(type, "IEND", 4) == 0) { seen_end = 1; break; }
            if (memcmp(type, "tRNS", 4) == 0) {
                /* PNG 11.3.2.1: simple transparency adds an alpha channel and changes no colour sample; luma takes no alpha
                 * (I1), so a well-formed chunk -- grey: 2 bytes, RGB: 6, indexed: 1 .. palette entries, behind PLTE, all in
                 * front of IDAT -- is skipped; anything else is the host decoder's to judge */
                const int fine = !seen_idat && ((ctype == 0 && len == 2) || (ctype == 2 && len == 6) ||
                                                (ctype == 3 && plte_n && len >= 1 && len <= plte_n));
                if (!fine) needs_host = 1;
            }
            if (memcmp(type, "PLTE", 4) == 0) {                      /* PNG 11.2.3 */
                if (len == 0 || len % 3 != 0 || len > 768 || seen_idat || plte_n) { rc = PNG_CORRUPT; break; }
                memcpy(plte, png + pos + 8, len);
                plte_n = len / 3;
            }
            if (!(type[0] & 0x20) && memcmp(type, "IHDR", 4) != 0 && memcmp(type, "PLTE", 4) != 0) { rc = PNG_CORRUPT; break; }   /* unknown critical chunk */
        }
        pos += 12 + (size_t)len;
    }
    if (rc == PNG_NEEDS_HOST) { free(z); return rc; }
    if (rc == PNG_OK && (!seen_idat || !seen_end)) rc = PNG_CORRUPT;
    if (rc == PNG_OK && ctype == 3 && plte_n == 0) rc = PNG_CORRUPT;
    if (rc == PNG_OK && needs_host) rc = PNG_NEEDS_HOST;
    if (rc) { free(z); return rc; }
    const size_t raw_n = (row + 1) * h;
    uint8_t* raw = (uint8_t*)malloc(raw_n);
    size_t got = 0;
    int checksum_ok = 0;
    rc = inflate_zlib(z, zn, raw, raw_n, &got, &checksum_ok);
    free(z);
    if (rc == PNG_OK && got != raw_n) rc = PNG_CORRUPT;
    /* P5: the right number of bytes but no / a wrong Adler-32: checksum-only, the host's decoder decides */
    if (rc == PNG_OK && !checksum_ok) rc = PNG_NEEDS_HOST;
    /* unfilter (9.2): x = filtered byte, a = left pixel's byte, b = above, c = above-left */
    uint8_t* planes = (ctype == 3 || ctype == 4) ? (uint8_t*)malloc(row * h ? row * h : 1) : pixels;
    for (uint32_t y = 0; rc == PNG_OK && y < h; y++) {
        const uint8_t* src = raw + (row + 1) * y;
        uint8_t* dst = planes + row * y;
        const uint8_t* up = y ? dst - row : NULL;
        int ft = src[0];
        if (ft > 4) { rc = PNG_CORRUPT; break; }
        for (size_t x = 0; x < row; x++) {
            int a = x >= bpp ? dst[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[1 + x];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            dst[x] = (uint8_t)v;
        }
    }
    if (planes != pixels) {
        for (size_t i = 0; rc == PNG_OK && i < (size_t)w * h; i++) {
            if (ctype == 3) memcpy(pixels + 3 * i, plte + 3 * (size_t)planes[i], 3);
            else pixels[i] = planes[2 * i];
        }
        free(planes);
    }
    free(raw);
    return rc;
}
