"""GPU parity of the PNG front end (ucfp_image_png_*): decoded pixels bit-equal to Pillow's and to the oracle's,
records equal to the oracle's records of those pixels."""
import io
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")

from test_oracle_png import _png, config1_png, grey_alpha_png, palette_png   # noqa: E402


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))


def _raw_png(arr, idat_split=None, comp=None, extra=b""):
    """PNG with filter type 0 everywhere, a chosen zlib stream maker and IDAT chunking."""
    h = arr.shape[0]
    bpp = 1 if arr.ndim == 2 else arr.shape[2]
    w = arr.shape[1]
    rows = b"".join(b"\0" + arr[y].tobytes() for y in range(h))
    z = (comp or (lambda d: zlib.compress(d, 6)))(rows)
    parts = [z] if not idat_split else [z[i:i + idat_split] for i in range(0, len(z), idat_split)]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[bpp]
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + extra +
            b"".join(_chunk(b"IDAT", p) for p in parts) + _chunk(b"IEND", b""))


def test_config1_set_decodes_to_pillow_pixels_and_oracle_records(gpu_ctx, oracle):
    """BASELINE config 1: 256x256 RGB ramps xor noise (bench.py's generator), compress levels 1 and 6."""
    from ucfp_amd import image
    pngs, imgs = [], []
    for i in range(96):
        p, im = config1_png(i, level=1 if i % 2 else 6)
        pngs.append(p)
        imgs.append(im)
    fr, st = image.decode_pngs(pngs, 256, 256, image.PIX_RGB8, ctx=gpu_ctx)
    assert not st.any(), st
    for i in range(len(pngs)):
        assert np.array_equal(fr[i], imgs[i]), i
        assert np.array_equal(fr[i], np.asarray(PIL.open(io.BytesIO(pngs[i])).convert("RGB")))
    rng = np.random.default_rng(1)
    ex = rng.integers(0, 256, (len(pngs), 32), dtype=np.uint8)
    for algo in (image.PHASH, image.MULTI):
        rec, st = image.fingerprint_pngs(pngs, 256, 256, image.PIX_RGB8, algo=algo, exact=ex, ctx=gpu_ctx)
        ref, _ = oracle.image_hash_batch(np.stack(imgs), algo, pixfmt=1, exact=ex)
        assert not st.any() and np.array_equal(rec, ref)


@pytest.mark.parametrize("mode,shape,fmt", [("L", (97, 131), 0), ("RGB", (64, 50, 3), 1), ("RGBA", (33, 77, 4), 2),
                                            ("RGB", (300, 1021, 3), 1), ("L", (1, 1), 0), ("RGBA", (2, 3, 4), 2)])
def test_colour_types_filters_and_levels(gpu_ctx, oracle, mode, shape, fmt):
    from ucfp_amd import image
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    base = np.add.outer(np.arange(shape[0]) * 3, np.arange(shape[1]) * 2)
    if len(shape) == 3:
        base = base[..., None] + np.arange(shape[2]) * 40
    smooth = (base & 255).astype(np.uint8)
    noisy = rng.integers(0, 256, shape, dtype=np.uint8)
    flat = np.full(shape, 77, np.uint8)
    pngs, want = [], []
    for arr in (smooth, noisy, smooth ^ (noisy & 3), flat, (noisy & 0xF0)):
        for kw in ({"compress_level": 1}, {"compress_level": 9, "optimize": True}, {"compress_level": 0},
                   {"compress_level": 6}):
            pngs.append(_png(arr, mode, **kw))
            want.append(arr)
    fr, st = image.decode_pngs(pngs, shape[1], shape[0], fmt, ctx=gpu_ctx)
    assert not st.any(), st
    for i, wv in enumerate(want):
        assert np.array_equal(fr[i], wv), (mode, i)
        rc, px = oracle.png_decode(pngs[i])
        assert rc == 0 and np.array_equal(px, fr[i])


def test_every_filter_type_and_block_type(gpu_ctx, oracle):
    """Hand-built streams: filter types 0-4 row by row; stored, fixed-code and dynamic blocks; IDAT split into 1-byte,
    100-byte and 8 KiB chunks (libpng's default); ancillary chunks in between."""
    from ucfp_amd import image
    rng = np.random.default_rng(9)
    h, w, bpp = 70, 53, 3
    img = rng.integers(0, 256, (h, w * bpp), dtype=np.uint8).astype(np.int32)
    img[20:40] = (np.arange(w * bpp) * 3) & 255                      # compressible rows
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
    raw = b"".join(rows)

    def fixed(d):
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        return co.compress(d) + co.flush()

    def mixed(d):   # stored + dynamic + fixed blocks in one stream
        co = zlib.compressobj(0)
        out = co.compress(d[:3000]) + co.flush(zlib.Z_FULL_FLUSH)
        co2 = zlib.compressobj(9, zlib.DEFLATED, -15)
        body = co2.compress(d[3000:9000]) + co2.flush(zlib.Z_SYNC_FLUSH)
        co3 = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
        tail = co3.compress(d[9000:]) + co3.flush()
        # splice raw deflate pieces after the level-0 stream's header: drop its final block + adler, append
        co0 = zlib.compressobj(0, zlib.DEFLATED, -15)
        first = co0.compress(d[:3000]) + co0.flush(zlib.Z_SYNC_FLUSH)
        return b"\x78\x01" + first + body + tail + struct.pack(">I", zlib.adler32(d))
    pngs = []
    ihdr = _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
    for comp in (lambda d: zlib.compress(d, 0), lambda d: zlib.compress(d, 1), lambda d: zlib.compress(d, 9), fixed, mixed):
        z = comp(raw)
        assert zlib.decompress(z) == raw
        for split in (None, 1, 100, 8192):
            parts = [z] if not split else [z[i:i + split] for i in range(0, len(z), split)]
            pngs.append(b"\x89PNG\r\n\x1a\n" + ihdr + _chunk(b"tEXt", b"k\0v") + _chunk(b"pHYs", bytes(9)) +
                        b"".join(_chunk(b"IDAT", p) for p in parts) + _chunk(b"tIME", bytes(7)) + _chunk(b"IEND", b""))
    fr, st = image.decode_pngs(pngs, w, h, image.PIX_RGB8, ctx=gpu_ctx)
    assert not st.any(), st
    want = img.astype(np.uint8).reshape(h, w, bpp)
    for i in range(len(pngs)):
        assert np.array_equal(fr[i], want), i
        assert np.array_equal(np.asarray(PIL.open(io.BytesIO(pngs[i]))), want)


def test_dense_matches_and_long_distances(gpu_ctx, oracle):
    """Streams made almost entirely of matches (flat and periodic images: dist 1, dist = row, dist up to 32 KiB), where
    one subsequence of the speculative decoder expands to kilobytes."""
    from ucfp_amd import image
    rng = np.random.default_rng(21)
    h, w = 256, 1024
    tile = rng.integers(0, 256, (8, w), dtype=np.uint8)
    cases = [np.zeros((h, w), np.uint8), np.tile(tile, (h // 8, 1)), np.tile(rng.integers(0, 256, (h, 4), dtype=np.uint8), (1, w // 4)),
             np.tile(rng.integers(0, 256, (31, w), dtype=np.uint8), (9, 1))[:h]]   # period 31 rows = 31 775 bytes back
    pngs = [_raw_png(c, comp=lambda d: zlib.compress(d, 9)) for c in cases] + [_png(c, "L", compress_level=9) for c in cases]
    fr, st = image.decode_pngs(pngs, w, h, image.PIX_GRAY8, ctx=gpu_ctx)
    assert not st.any(), st
    for i in range(len(pngs)):
        assert np.array_equal(fr[i], cases[i % len(cases)]), i


def test_needs_host_and_damaged_files(gpu_ctx, oracle):
    from ucfp_amd import image
    rng = np.random.default_rng(5)
    good, img = config1_png(7, side=64)
    pal = io.BytesIO()             # a grey + alpha file whose tRNS chunk has no business there (PNG 11.3.2.1): the host decides
    pal.write(_raw_png(rng.integers(0, 256, (64, 64, 2), dtype=np.uint8), extra=_chunk(b"tRNS", bytes(2))))
    deep = io.BytesIO()
    PIL.fromarray(rng.integers(0, 65535, (64, 64), dtype=np.uint16)).save(deep, "PNG")
    other_geom, _ = config1_png(8, side=32)
    gray = _png(rng.integers(0, 256, (64, 64), dtype=np.uint8), "L")
    truncated = good[: len(good) // 2]
    z = bytearray(good)
    z[len(z) // 2] ^= 0x40            # flips a bit inside the deflate stream: the IDAT chunk's CRC no longer matches
    trns = _raw_png(img, extra=_chunk(b"tRNS", bytes(5)))      # an RGB file's tRNS is six bytes: a malformed one is the host's
    too_short = _raw_png(img[:40])    # IHDR says 64 rows... built with 40: patch the header
    too_short = too_short[:16] + struct.pack(">II", 64, 64) + too_short[24:]
    pngs = [good, pal.getvalue(), deep.getvalue(), other_geom, gray, truncated, b"GIF89a" + bytes(80), trns, too_short, good]
    assert image.png_probe(good) == (0, 64, 64, image.PIX_RGB8)
    # (the probe reads the IHDR only: chunks behind it are the device's business, below)
    assert image.png_probe(deep.getvalue())[0] == image.NEEDS_HOST
    assert image.png_probe(b"GIF89a" + bytes(80))[0] < 0
    rec, st = image.fingerprint_pngs(pngs, 64, 64, image.PIX_RGB8, algo=image.MULTI, ctx=gpu_ctx)
    assert list(st[:5]) == [0, 1, 1, 1, 1], st
    assert st[5] < 0 and st[6] < 0 and st[7] == 1 and st[8] < 0 and st[9] == 0, st
    from ucfp_amd.blake3 import blake3_digest          # no `exact` handed in: the front end hashes the files itself
    ref, _ = oracle.image_hash_batch(img[None], 7, pixfmt=1, exact=np.frombuffer(blake3_digest(good), np.uint8)[None])
    assert np.array_equal(rec[0], ref[0]) and np.array_equal(rec[9], ref[0])
    assert not rec[1:9].any()
    for i in (1, 2, 7):
        assert oracle.png_decode(pngs[i])[0] == oracle.PNG_NEEDS_HOST
    for i in (5, 6, 8):
        assert oracle.png_decode(pngs[i])[0] == oracle.PNG_CORRUPT
    _, st2 = image.fingerprint_pngs([bytes(z)], 64, 64, image.PIX_RGB8, ctx=gpu_ctx)   # must not hang or fault
    assert st2[0] == -1
    # damage outside the image data: one byte of an ancillary chunk (its CRC is verified like every chunk's), and a file
    # whose IDAT CRC was "repaired" after the flip, so that only the stream's own checks (codes, length, Adler-32) are left
    text_chunk = _raw_png(img, extra=_chunk(b"tEXt", b"key\0value"))
    dmg = bytearray(text_chunk)
    dmg[dmg.index(b"value")] ^= 0x20
    repaired = bytearray(_raw_png(img, comp=lambda d: zlib.compress(d, 6)))
    i0 = bytes(repaired).index(b"IDAT")
    ln = struct.unpack(">I", repaired[i0 - 4:i0])[0]
    repaired[i0 + 4 + ln // 2] ^= 0x08
    repaired[i0 + 4 + ln:i0 + 8 + ln] = struct.pack(">I", zlib.crc32(bytes(repaired[i0:i0 + 4 + ln])))
    # P5: a checksum-ONLY failure (ancillary chunk CRC; Adler-32 of a stream that inflated to the right length) is
    # not the device's call -- decoders differ on it -- so the file goes to the host's decoder (NEEDS_HOST = 1); damage to
    # a critical chunk's CRC or to the stream itself stays -1
    _, st4 = image.fingerprint_pngs([text_chunk, bytes(dmg), bytes(repaired)], 64, 64, image.PIX_RGB8, ctx=gpu_ctx)
    o_rep = oracle.png_decode(bytes(repaired))[0]
    assert o_rep in (oracle.PNG_CORRUPT, oracle.PNG_NEEDS_HOST)
    assert list(st4) == [0, 1, -1 if o_rep == oracle.PNG_CORRUPT else 1], st4
    assert oracle.png_decode(bytes(dmg))[0] == oracle.PNG_NEEDS_HOST
    # a stored (level 0) stream with one payload byte changed inflates to the right length: only the Adler-32 can tell
    good_stored = _raw_png(img, comp=lambda d: zlib.compress(d, 0))
    stored = bytearray(good_stored)
    i0 = bytes(stored).index(b"IDAT")
    ln = struct.unpack(">I", stored[i0 - 4:i0])[0]
    stored[i0 + 4 + ln // 2] ^= 0x01
    stored[i0 + 4 + ln:i0 + 8 + ln] = struct.pack(">I", zlib.crc32(bytes(stored[i0:i0 + 4 + ln])))   # chunk CRC repaired
    no_trailer = bytearray(good_stored)         # the 4 Adler bytes cut off the end of the (single) IDAT chunk
    body = bytes(no_trailer[i0:i0 + 4 + ln - 4])
    no_trailer = bytes(no_trailer[:i0 - 4]) + struct.pack(">I", ln - 4) + body + struct.pack(">I", zlib.crc32(body)) + \
        _chunk(b"IEND", b"")
    _, st3 = image.fingerprint_pngs([bytes(stored), good_stored, no_trailer], 64, 64, image.PIX_RGB8, ctx=gpu_ctx)
    assert list(st3) == [1, 0, 1], st3
    assert oracle.png_decode(bytes(stored))[0] == oracle.PNG_NEEDS_HOST
    assert oracle.png_decode(no_trailer)[0] == oracle.PNG_NEEDS_HOST


def test_tiny_junk_files_between_valid_ones_at_unaligned_offsets(gpu_ctx, oracle):
    """a rejected file of 0..15 bytes has the same 16-byte-aligned gather address as the file after it; its
    wave must not touch that address (it used to zero 4 bytes there -- the neighbour's zlib header -- in the same
    launch).  0-, 1- and 15-byte junk in front of valid files, packed back to back so that offsets are unaligned."""
    from ucfp_amd import image
    files, want = [], []
    for i in range(120):
        junk = [b"", b"\x89", b"\x89PNG\r\n\x1a\n" + bytes(7), bytes(3)][i % 4]
        p, im = config1_png(i % 7, side=64, level=1 if i % 2 else 6)
        files += [junk, p]
        want += [None, im]
    fr, st = image.decode_pngs(files, 64, 64, image.PIX_RGB8, ctx=gpu_ctx)
    for i, im in enumerate(want):
        if im is None:
            assert st[i] == -1, (i, st[i])
        else:
            assert st[i] == 0 and np.array_equal(fr[i], im), i


def test_palette_and_grey_alpha_files_decode_on_the_device(gpu_ctx, oracle):
    """Round 3: 8-bit indexed colour -> RGB8 through the file's PLTE (short palettes: missing entries are
    black), 8-bit grey + alpha -> GRAY8; in the SAME batch as plain RGB / grey files of the geometry; pixels equal to
    Pillow's and the oracle's, records equal to the oracle's records of those pixels."""
    from ucfp_amd import image
    rng = np.random.default_rng(77)
    for h, w in ((64, 64), (97, 131), (1, 1), (33, 300)):
        files, want = [], []
        for i in range(24):
            if i % 3 == 0:
                arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                files.append(_png(arr, "RGB", compress_level=(1, 6, 9)[i % 3]))
                want.append(arr)
            else:
                p, rgb = palette_png(rng, h, w, (256, 5, 77, 200)[i % 4], compress_level=(1, 6)[i % 2])
                if p[24] != 8:
                    continue
                files.append(p)
                want.append(rgb)
        fr, st = image.decode_pngs(files, w, h, image.PIX_RGB8, ctx=gpu_ctx)
        assert not st.any(), st
        for i, f in enumerate(files):
            assert image.png_probe(f) == (0, w, h, image.PIX_RGB8)
            assert np.array_equal(fr[i], want[i]), (h, w, i)
            assert np.array_equal(fr[i], oracle.png_decode(f)[1])
        if h >= 32 and w >= 32:
            rec, st = image.fingerprint_pngs(files, w, h, image.PIX_RGB8, algo=image.MULTI, ctx=gpu_ctx)
            from ucfp_amd.blake3 import blake3_digest
            ex = np.stack([np.frombuffer(blake3_digest(f), np.uint8) for f in files])
            ref, _ = oracle.image_hash_batch(np.stack(want), 7, pixfmt=1, exact=ex)
            assert not st.any() and np.array_equal(rec, ref)
        files, want = [], []
        for i in range(20):
            if i % 2:
                p, g = grey_alpha_png(rng, h, w, compress_level=(1, 9)[i % 4 == 1])
            else:
                g = rng.integers(0, 256, (h, w), dtype=np.uint8)
                p = _png(g, "L")
            files.append(p)
            want.append(g)
        fr, st = image.decode_pngs(files, w, h, image.PIX_GRAY8, ctx=gpu_ctx)
        assert not st.any(), st
        for i, f in enumerate(files):
            assert np.array_equal(fr[i], want[i]), (h, w, i)
            assert np.array_equal(fr[i], np.asarray(PIL.open(io.BytesIO(f)).getchannel("L")))
    # an indexed file that lost its PLTE chunk is damaged, not a host case
    p, _ = palette_png(rng, 64, 64, 256)
    i0 = p.index(b"PLTE")
    ln = struct.unpack(">I", p[i0 - 4:i0])[0]
    no_plte = p[:i0 - 4] + p[i0 + 8 + ln:]
    _, st = image.decode_pngs([no_plte, p], 64, 64, image.PIX_RGB8, ctx=gpu_ctx)
    assert list(st) == [-1, 0] and oracle.png_decode(no_plte)[0] == oracle.PNG_CORRUPT


def test_damaged_files_never_hang_or_fault(gpu_ctx, oracle):
    """300 random corruptions of valid files (bit flips in the stream, truncations, garbage after IHDR, swapped halves): the
    decoder must come back with a status for every file -- and where it says 0, the pixels are what Pillow decodes when
    Pillow accepts the file at all."""
    from ucfp_amd import image
    rng = np.random.default_rng(2024)
    base = [config1_png(i, side=64, level=lv)[0] for i, lv in ((0, 1), (1, 6), (2, 9))]
    base.append(_png(np.tile(np.arange(64, dtype=np.uint8), (64, 1))[..., None].repeat(3, 2), "RGB", compress_level=9))
    pngs = []
    for t in range(300):
        b = bytearray(base[t % len(base)])
        kind = t % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(41, len(b) - 12))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[: int(rng.integers(34, len(b)))]
        elif kind == 2:
            cut = int(rng.integers(41, len(b) - 20))
            b[cut:cut + 16] = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
        elif kind == 3:
            h = len(b) // 2
            b = b[:41] + b[h:] + b[41:h]
        else:
            b[int(rng.integers(41, 60))] = int(rng.integers(0, 256))      # the zlib header / first block header
        pngs.append(bytes(b))
    fr, st = image.decode_pngs(pngs, 64, 64, image.PIX_RGB8, ctx=gpu_ctx)
    assert set(np.unique(st)) <= {0, 1, -1}
    assert (st != 0).sum() >= 295           # chunk CRCs and the Adler-32 leave nothing to slip through (a flip may hit a spare bit of the zlib header)
    for i in np.nonzero(st == 0)[0]:
        try:
            want = np.asarray(PIL.open(io.BytesIO(pngs[i])).convert("RGB"))
        except Exception:
            continue
        assert np.array_equal(fr[i], want), i


def test_micro_batcher_for_encoded_uploads(gpu_ctx, oracle):
    """SURVEY 8f N1 + N4: 32 request threads each submit one PNG upload at a time; every thread gets ITS record -- decode,
    BLAKE3 of the file and hashes all from the device -- bit-exact, from far fewer launch sequences than uploads.  An
    upload of another kind comes back NEEDS_HOST, a damaged one as a modality error, without disturbing its neighbours."""
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    rng = np.random.default_rng(12)
    pngs, imgs = zip(*[config1_png(i, side=64, level=1 + i % 6) for i in range(200)])
    pngs = list(pngs)
    ex = np.stack([np.frombuffer(blake3_digest(p), np.uint8) for p in pngs])
    ref, _ = oracle.image_hash_batch(np.stack(imgs), 7, pixfmt=1, exact=ex)
    gray = _png(rng.integers(0, 256, (64, 64), dtype=np.uint8), "L")
    pngs[17], pngs[90] = gray, pngs[90][: len(pngs[90]) // 2]
    b = image.PngBatcher(64, 64, image.PIX_RGB8, max_batch=64, max_delay_us=2000, ctx=gpu_ctx)
    try:
        with ThreadPoolExecutor(32) as pool:
            got = list(pool.map(b.submit, pngs))
        for i, (rec, st) in enumerate(got):
            if i == 17:
                assert st == image.NEEDS_HOST and not any(rec)
            elif i == 90:
                assert st < 0 and not any(rec)
            else:
                assert st == 0 and rec == ref[i].tobytes(), i
        batches, items = b.stats()
        assert items == len(pngs) and batches < len(pngs) // 3, (batches, items)
    finally:
        b.close()


@pytest.mark.parametrize("n", [520, 1100, 1300, 1700, 3300])
def test_every_round_shape_of_the_inflate_kernel(gpu_ctx, oracle, n):
    """The Huffman pass is launched with 4 / 2 / 1 waves per file by batch size (<= 1200 / <= 3000 / beyond), and batches of
    1600+ files decode as two halves on two streams; the other tests use small batches, these reach every shape (round 3's
    one-kernel inflate: three round shapes by batch size, the same sizes reach them under UCFP_PNG_TWO_PASS=0)."""
    from ucfp_amd import image
    rng = np.random.default_rng(n)
    base = [config1_png(i, side=64, level=(1, 6, 9)[i % 3]) for i in range(40)]
    flat = _png(np.full((64, 64, 3), 200, np.uint8), "RGB", compress_level=9)
    pngs = [base[i % 40][0] for i in range(n)]
    pngs[5] = pngs[n - 3] = flat
    fr, st = image.decode_pngs(pngs, 64, 64, image.PIX_RGB8, ctx=gpu_ctx)
    assert not st.any()
    for i in range(n):
        want = np.full((64, 64, 3), 200, np.uint8) if i in (5, n - 3) else base[i % 40][1]
        assert np.array_equal(fr[i], want), i


def test_simple_transparency_changes_no_pixel(gpu_ctx, oracle):
    """tRNS (PNG 11.3.2.1) for grey, RGB and indexed colour: the chunk adds an alpha channel and changes no colour sample, and
    luma takes no alpha (DESIGN I1) -- the device skips a well-formed one; pixels and records equal the host path's (Pillow
    decode -> L / RGB, alpha dropped) and the oracle's."""
    from ucfp_amd import image
    from ucfp_amd.image import PreprocessConfig
    rng = np.random.default_rng(77)
    h, w = 72, 90
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    files, want, fmts = [], [], []
    b = io.BytesIO()
    PIL.fromarray(grey, "L").save(b, "PNG", transparency=37)
    files.append(b.getvalue()); want.append(grey); fmts.append(image.PIX_GRAY8)
    b = io.BytesIO()
    PIL.fromarray(rgb, "RGB").save(b, "PNG", transparency=(1, 2, 3))
    files.append(b.getvalue()); want.append(rgb); fmts.append(image.PIX_RGB8)
    pimg = PIL.fromarray(rgb, "RGB").quantize(40)
    b = io.BytesIO()
    pimg.save(b, "PNG", transparency=bytes(range(0, 200, 5)))           # an alpha per palette entry
    files.append(b.getvalue()); want.append(np.asarray(pimg.convert("RGB"))); fmts.append(image.PIX_RGB8)
    b = io.BytesIO()
    pimg.save(b, "PNG", transparency=5)                                  # one fully transparent entry
    files.append(b.getvalue()); want.append(np.asarray(pimg.convert("RGB"))); fmts.append(image.PIX_RGB8)
    for f in files:
        assert b"tRNS" in f
    got, st = image.decode_uploads(files, ctx=gpu_ctx)
    assert not st.any(), st
    for i in range(len(files)):
        assert np.array_equal(got[i], want[i]), i
        rc, px = oracle.png_decode(files[i])
        assert rc == 0 and np.array_equal(px, want[i]), i
    rec, st = image.fingerprint_uploads(files, ctx=gpu_ctx)
    for i, f in enumerate(files):
        assert bytes(image.fingerprint_with(f, 0, i, PreprocessConfig()).fingerprint) == rec[i].tobytes(), i
    # the uniform entry takes them too
    fr, st = image.decode_pngs(files[1:], w, h, image.PIX_RGB8, ctx=gpu_ctx)
    assert not st.any() and all(np.array_equal(fr[i], want[i + 1]) for i in range(3))


@pytest.mark.parametrize("env", [{"UCFP_PNG_TWO_PASS": "1", "UCFP_PNG_HUFF_BITS": "128"}, {"UCFP_PNG_TWO_PASS": "2", "UCFP_PNG_HUFF_BITS": "512"},
                                 {"UCFP_PNG_TWO_PASS": "8"}, {"UCFP_PNG_TWO_PASS": "4", "UCFP_PNG_HUFF_MAX_ITER": "1"},
                                 {"UCFP_PNG_TWO_PASS": "4", "UCFP_PNG_HUFF_WARM": "0"}, {"UCFP_PNG_TWO_PASS": "0"}])
def test_inflate_variants_in_a_process_of_their_own(env):
    """Every instantiation of the two-pass inflate (subsequence length, waves per file, chain-iteration cap, warm-up) and the
    one-kernel form decode the same assorted files to Pillow's pixels.  The switches are read once per process, hence a child."""
    import os
    import subprocess
    import sys
    code = r"""
import io, sys
import numpy as np
from PIL import Image
sys.path.insert(0, 'tests')
from test_oracle_png import config1_png, _png
from ucfp_amd import image, _lib
ctx = _lib.default_context(0)
rng = np.random.default_rng(3)
pngs, want = [], []
for i in range(70):
    if i % 5 == 0:
        a = rng.integers(0, 256, (96, 96, 3), dtype=np.uint8)                      # incompressible: stored / literal-only blocks
        p = _png(a, 'RGB', compress_level=(0, 1, 9)[i % 3])
    elif i % 5 == 1:
        a = np.full((96, 96, 3), i, np.uint8)
        a[::7, ::5] = 255 - i                                                      # long runs: self-overlapping matches
        p = _png(a, 'RGB', compress_level=9)
    else:
        p, a = config1_png(i, side=96, level=(1, 6, 9)[i % 3])
    pngs.append(p)
    want.append(a)
fr, st = image.decode_pngs(pngs, 96, 96, image.PIX_RGB8, ctx=ctx)
assert not st.any(), st
for i in range(len(pngs)):
    assert np.array_equal(fr[i], want[i]), i
bad = bytearray(pngs[3]); bad[len(bad) // 2] ^= 0x40
fr, st = image.decode_pngs([pngs[0], bytes(bad), pngs[1]], 96, 96, image.PIX_RGB8, ctx=ctx)
assert st[0] == 0 and st[1] != 0 and st[2] == 0 and np.array_equal(fr[2], want[1])
print('ok')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])
