"""Pins the oracle's PNG front end (oracle/ucfp_oracle_png.c: RFC 1950/1951 inflate + PNG unfilter) against the
reference implementations of those formats that ARE importable here: zlib (inflate) and Pillow/libpng (pixels)."""
import io
import zlib

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def _png(arr, mode, **kw):
    b = io.BytesIO()
    PIL.fromarray(arr, mode).save(b, "PNG", **kw)
    return b.getvalue()


def config1_png(i, side=256, level=1):
    """bench.py's config-1 image i: the colour ramp of benches/end_to_end.rs:77-85 xor per-image noise."""
    yy, xx = np.mgrid[0:side, 0:side]
    rng = np.random.default_rng(0xC0F1 + i)
    base = np.stack([(xx + i) & 255, (yy + 2 * i) & 255, (xx + yy) & 255], -1).astype(np.uint8)
    img = base ^ rng.integers(0, 8, (side, side, 3), dtype=np.uint8)
    return _png(img, "RGB", compress_level=level), img


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_inflate_matches_zlib(oracle, level):
    rng = np.random.default_rng(level)
    cases = [b"", b"a", b"abc" * 1000, bytes(rng.integers(0, 256, 70_000, dtype=np.uint8)),
             bytes(rng.integers(0, 4, 200_000, dtype=np.uint8)), b"\0" * 300_000,
             bytes((np.arange(100_000) % 251).astype(np.uint8))]
    for raw in cases:
        z = zlib.compress(raw, level)
        rc, out = oracle.inflate(z, len(raw))
        assert rc == 0 and out == raw, (level, len(raw))
    # fixed-code blocks (Z_FIXED) and a damaged stream
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    z = co.compress(cases[2]) + co.flush()
    assert oracle.inflate(z, len(cases[2])) == (0, cases[2])
    bad = bytearray(zlib.compress(cases[3], 6))
    bad[len(bad) // 2] ^= 0x55
    assert oracle.inflate(bytes(bad), len(cases[3]))[0] != 0


def test_config1_pngs_decode_to_pillow_pixels(oracle):
    for i in (0, 1, 2, 500, 999):
        for level in (1, 6):
            png, img = config1_png(i, level=level)
            rc, px = oracle.png_decode(png)
            assert rc == 0 and np.array_equal(px, img)
            assert np.array_equal(px, np.asarray(PIL.open(io.BytesIO(png)).convert("RGB")))


@pytest.mark.parametrize("mode,shape", [("L", (97, 131)), ("RGB", (64, 50, 3)), ("RGBA", (33, 77, 4))])
def test_colour_types_and_every_filter(oracle, mode, shape):
    rng = np.random.default_rng(len(mode))
    smooth = (np.add.outer(np.arange(shape[0]) * 3, np.arange(shape[1]) * 2)[..., None] +
              np.arange(shape[2] if len(shape) == 3 else 1) * 40) & 255
    smooth = smooth.reshape(shape).astype(np.uint8)
    noisy = rng.integers(0, 256, shape, dtype=np.uint8)
    for arr in (smooth, noisy, smooth ^ (noisy & 3)):
        for kw in ({"compress_level": 1}, {"compress_level": 9, "optimize": True}, {"compress_level": 0}):
            png = _png(arr, mode, **kw)
            rc, w, h, fmt = oracle.png_probe(png)
            assert (rc, w, h, fmt) == (0, shape[1], shape[0], {"L": 0, "RGB": 1, "RGBA": 2}[mode])
            rc, px = oracle.png_decode(png)
            assert rc == 0 and np.array_equal(px, arr), (mode, kw)
    # every filter type, forced per row by hand-built scanlines
    h, w = shape[0], shape[1]
    bpp = 1 if len(shape) == 2 else shape[2]
    img = noisy.reshape(h, w * bpp).astype(np.int32)
    rows = []
    for y in range(h):
        ft = y % 5
        cur, up = img[y], (img[y - 1] if y else np.zeros_like(img[0]))
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (a + up) >> 1
        else:
            p = a + up - c
            pa, pb, pc = abs(p - a), abs(p - up), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
        rows.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
    import struct

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    ctype = {1: 0, 3: 2, 4: 6}[bpp]
    z = zlib.compress(b"".join(rows), 6)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
           chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", z[:100]) + chunk(b"IDAT", z[100:]) + chunk(b"IEND", b""))
    rc, px = oracle.png_decode(png)
    assert rc == 0 and np.array_equal(px.reshape(h, w * bpp), img.astype(np.uint8))
    assert np.array_equal(px, np.asarray(PIL.open(io.BytesIO(png))))


def test_formats_handed_back_to_the_host(oracle):
    rng = np.random.default_rng(5)
    pal = PIL.fromarray(rng.integers(0, 256, (40, 40), dtype=np.uint8), "L").convert("P")
    b = io.BytesIO()
    pal.save(b, "PNG", transparency=3)
    rc, px = oracle.png_decode(b.getvalue())                                      # palette with tRNS: the alpha it adds is
    assert rc == oracle.PNG_OK and np.array_equal(px, np.asarray(pal.convert("RGB")))   # dropped (I1), no colour changes
    f = b.getvalue()
    i = f.index(b"tRNS")
    ln = int.from_bytes(f[i - 4:i], "big")
    import struct
    import zlib
    wrong = f[:i - 4] + struct.pack(">I", 300) + b"tRNS" + bytes(300) + struct.pack(">I", zlib.crc32(b"tRNS" + bytes(300))) + f[i + 8 + ln:]
    assert oracle.png_decode(wrong)[0] == oracle.PNG_NEEDS_HOST                   # more alphas than palette entries: the host's
    b = io.BytesIO()
    PIL.fromarray(rng.integers(0, 16, (40, 40), dtype=np.uint8), "L").convert("P").save(b, "PNG", bits=4)
    assert oracle.png_decode(b.getvalue())[0] == oracle.PNG_NEEDS_HOST            # 4-bit palette
    b = io.BytesIO()
    PIL.fromarray(rng.integers(0, 65535, (40, 40), dtype=np.uint16)).save(b, "PNG")
    assert oracle.png_decode(b.getvalue())[0] == oracle.PNG_NEEDS_HOST            # 16 bit
    png, _ = config1_png(0, side=64)
    assert oracle.png_decode(png[:200])[0] == oracle.PNG_CORRUPT                  # truncated
    bad = bytearray(png)
    bad[len(bad) // 2] ^= 1
    assert oracle.png_decode(bytes(bad))[0] == oracle.PNG_CORRUPT                 # chunk CRC
    assert oracle.png_decode(b"GIF89a" + bytes(60))[0] == oracle.PNG_CORRUPT


def palette_png(rng, h, w, colours=256, **kw):
    """An 8-bit indexed-colour PNG with `colours` palette entries (PLTE shorter than 256 entries when colours < 256)."""
    idx = rng.integers(0, colours, (h, w), dtype=np.uint8)
    im = PIL.fromarray(idx, "P")
    im.putpalette(rng.integers(0, 256, colours * 3, dtype=np.uint8).tobytes())
    b = io.BytesIO()
    im.save(b, "PNG", **kw)
    return b.getvalue(), np.asarray(im.convert("RGB"))


def grey_alpha_png(rng, h, w, **kw):
    ga = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    ga[..., 0] = (np.add.outer(np.arange(h) * 2, np.arange(w) * 3) & 255) ^ (ga[..., 0] & 7)
    b = io.BytesIO()
    PIL.fromarray(ga, "LA").save(b, "PNG", **kw)
    return b.getvalue(), ga[..., 0].copy()


def test_palette_and_grey_alpha_decode_like_pillow(oracle):
    """Round 3: indexed colour (8-bit) decodes to RGB8 through PLTE, grey + alpha (8-bit) to GRAY8 with the alpha dropped
    -- what Pillow's convert("RGB") / the L band give, and what the host path feeds the hash with."""
    rng = np.random.default_rng(11)
    for colours, h, w, kw in ((256, 40, 53, {}), (7, 33, 64, {}), (200, 1, 1, {}), (256, 70, 129, {"compress_level": 9}),
                              (31, 65, 200, {"optimize": True})):
        png, rgb = palette_png(rng, h, w, colours, **kw)
        if png[24] != 8:
            continue                                                              # Pillow packed it below 8 bits: host
        rc, px = oracle.png_decode(png)
        assert rc == 0 and oracle.png_probe(png)[3] == 1 and np.array_equal(px, rgb), (colours, h, w)
        assert np.array_equal(px, np.asarray(PIL.open(io.BytesIO(png)).convert("RGB")))
    for h, w in ((40, 53), (1, 1), (64, 64), (129, 70)):
        png, g = grey_alpha_png(rng, h, w)
        rc, px = oracle.png_decode(png)
        assert rc == 0 and oracle.png_probe(png)[3] == 0 and np.array_equal(px, g)
        assert np.array_equal(px, np.asarray(PIL.open(io.BytesIO(png)).getchannel("L")))
    # an index beyond the file's PLTE is black; an indexed file without PLTE is damaged
    png, _ = palette_png(rng, 8, 8, 4)
    assert png[24] in (2, 8)
