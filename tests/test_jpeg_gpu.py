"""GPU parity of the JPEG front end (ucfp_image_jpeg_*, ucfp_amd/csrc/jpeg.hip): decoded luma planes bit-equal to libjpeg's
(Pillow draft "L") and to the oracle's, records equal to the oracle's records of those planes (DESIGN J1)."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")

from test_oracle_jpeg import jpeg_of, libjpeg_luma, picture, with_quantiser   # noqa: E402


@pytest.mark.parametrize("h,w", [(256, 256), (64, 64), (33, 77), (1, 1), (17, 8), (100, 300), (250, 123)])
def test_luma_planes_equal_libjpeg_and_oracle(gpu_ctx, oracle, h, w):
    from ucfp_amd import image
    files = []
    for i in range(6):
        img = picture(h, w, seed=i)
        for q in (25, 75, 98):
            for sub in (0, 1, 2):
                for extra in ({}, {"optimize": True}, {"restart_marker_rows": 1}, {"restart_marker_blocks": 2}):
                    files.append(jpeg_of(img, quality=q, subsampling=sub, **extra))
        files.append(jpeg_of(np.asarray(PIL.fromarray(img, "RGB").convert("L")), "L", quality=80))
    fr, st = image.decode_jpegs(files, w, h, ctx=gpu_ctx)
    assert not st.any(), st
    for i, f in enumerate(files):
        assert image.jpeg_probe(f) == (0, w, h)
        assert np.array_equal(fr[i], libjpeg_luma(f)), (h, w, i)
        rc, px = oracle.jpeg_decode_luma(f)
        assert rc == 0 and np.array_equal(fr[i], px), (h, w, i)


def test_records_of_config1_images_as_jpeg(gpu_ctx, oracle):
    """BASELINE config 1's images re-encoded as JPEG: records = the oracle's records of libjpeg's luma planes, the `exact`
    field = BLAKE3 of each file (computed on the device), for every algorithm mask; the host path (fingerprint_with, Pillow
    decode) gives the same records."""
    from test_oracle_png import config1_png
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    from ucfp_amd.image import PreprocessConfig
    files, planes = [], []
    for i in range(64):
        _, img = config1_png(i)
        f = jpeg_of(img, quality=(60, 85, 95)[i % 3], subsampling=i % 3, **({"optimize": True} if i % 4 == 0 else {}))
        files.append(f)
        planes.append(libjpeg_luma(f))
    ex = np.stack([np.frombuffer(blake3_digest(f), np.uint8) for f in files])
    for algo in (image.PHASH, image.AHASH, image.DHASH, image.MULTI):
        rec, st = image.fingerprint_jpegs(files, 256, 256, algo=algo, ctx=gpu_ctx)
        ref, _ = oracle.image_hash_batch(np.stack(planes), algo, pixfmt=0, exact=ex)
        assert not st.any() and np.array_equal(rec, ref), algo
    for i in (0, 1, 2, 33):
        r = image.fingerprint_with(files[i], 1, 100 + i, PreprocessConfig())
        assert bytes(r.fingerprint) == ref[i].tobytes()           # (ref of the last mask: MULTI)


def test_files_handed_to_the_host_and_damaged_ones(gpu_ctx, oracle):
    from ucfp_amd import image
    img = picture(64, 64)
    good = jpeg_of(img, quality=80)
    progressive = jpeg_of(img, progressive=True)
    b = io.BytesIO()
    PIL.fromarray(img, "RGB").convert("CMYK").save(b, "JPEG")
    cmyk = b.getvalue()
    other_geom = jpeg_of(picture(32, 64), quality=80)
    truncated = good[: len(good) // 2]
    rst = bytearray(jpeg_of(img, quality=80, restart_marker_rows=1))
    i = rst.index(b"\xff\xd1")
    rst[i + 1] = 0xD3
    flipped = bytearray(good)
    flipped[len(flipped) * 3 // 4] ^= 0x5A                       # somewhere in the entropy-coded data
    files = [good, progressive, cmyk, other_geom, truncated, bytes(rst), b"\x89PNG\r\n\x1a\n" + bytes(80), b"", good]
    fr, st = image.decode_jpegs(files, 64, 64, ctx=gpu_ctx)
    assert list(st[:4]) == [0, 1, 1, 1] and st[4] == 1 and st[5] == 1 and st[6] < 0 and st[7] < 0 and st[8] == 0, st
    assert np.array_equal(fr[0], libjpeg_luma(good)) and np.array_equal(fr[8], fr[0])
    assert image.jpeg_probe(progressive)[0] == image.NEEDS_HOST and image.jpeg_probe(files[6])[0] < 0
    for f in (progressive, cmyk, truncated, bytes(rst)):
        assert oracle.jpeg_decode_luma(f)[0] == oracle.JPG_NEEDS_HOST
    # a flipped byte either still decodes (then to the oracle's pixels: both read the same damaged stream the same way)
    # or is handed to the host by both
    o_rc, o_px = oracle.jpeg_decode_luma(bytes(flipped))
    fr2, st2 = image.decode_jpegs([bytes(flipped)], 64, 64, ctx=gpu_ctx)
    assert st2[0] == o_rc and (o_rc != 0 or np.array_equal(fr2[0], o_px))
    rec, st3 = image.fingerprint_jpegs([good, progressive, good], 64, 64, algo=image.MULTI, ctx=gpu_ctx)
    assert list(st3) == [0, 1, 0] and not rec[1].any() and np.array_equal(rec[0], rec[2])


def test_out_of_range_coefficients_take_the_oracles_status(gpu_ctx, oracle):
    """Crafted quantisers (DESIGN J4's range guards): the device hands over exactly the files the oracle hands over, and
    decodes the others to the oracle's pixels."""
    from ucfp_amd import image
    img = picture(64, 64, seed=3)
    img[::2, ::2] = 255 - img[::2, ::2]
    files = [with_quantiser(jpeg_of(img, quality=q, subsampling=s), v) for q in (30, 60, 95) for v in (1, 16, 40, 120, 255)
             for s in (0, 2)]
    fr, st = image.decode_jpegs(files, 64, 64, ctx=gpu_ctx)
    want = [oracle.jpeg_decode_luma(f) for f in files]
    assert [int(x) for x in st] == [w[0] for w in want]
    assert 1 in st and 0 in st
    for i, (rc, px) in enumerate(want):
        if rc == 0:
            assert np.array_equal(fr[i], px), i
    rec, st2 = image.fingerprint_jpegs(files, 64, 64, ctx=gpu_ctx)
    assert np.array_equal(st2, st) and not rec[st != 0].any()


def test_stray_bytes_behind_the_last_block_are_ignored(gpu_ctx, oracle):
    """Found by tools/soak_jpeg.py: once the image's last block is decoded, what follows in the scan (stray bytes in front of
    EOI; the tail of a stream whose bit flip made blocks end early) is nobody's business -- a sequential decoder stops there,
    and so must the speculative one (its later lanes parse that junk and may well run into invalid codes)."""
    from ucfp_amd import image
    rng = np.random.default_rng(6)
    files = []
    for i in range(12):
        f = jpeg_of(picture(96, 80, seed=i), quality=(40, 85, 97)[i % 3], subsampling=i % 3)
        junk = bytes(int(x) for x in rng.integers(0, 255, int(rng.integers(30, 900))))     # no 0xFF: no markers
        files.append(f[:-2] + junk + f[-2:])
    fr, st = image.decode_jpegs(files, 80, 96, ctx=gpu_ctx)
    assert not st.any(), st
    for i, f in enumerate(files):
        rc, px = oracle.jpeg_decode_luma(f)
        assert rc == 0 and np.array_equal(fr[i], px), i
        assert np.array_equal(fr[i], libjpeg_luma(f)), i


def test_randomly_damaged_files_never_hang_and_agree_with_the_oracle(gpu_ctx, oracle):
    """300 corruptions of valid files (bit flips in the entropy-coded data, truncations, garbage runs, stray markers): a
    status for every file, equal to the oracle's; where both decode, the same pixels."""
    from ucfp_amd import image
    rng = np.random.default_rng(2025)
    img = picture(64, 64)
    base = [jpeg_of(img, quality=q, subsampling=s, **e) for q, s, e in
            ((75, 2, {}), (90, 0, {"optimize": True}), (50, 1, {"restart_marker_rows": 1}), (95, 2, {"restart_marker_blocks": 2}))]
    files = []
    for t in range(300):
        b = bytearray(base[t % len(base)])
        sos = bytes(b).index(b"\xff\xda")
        kind = t % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(sos + 14, len(b) - 2))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[: int(rng.integers(sos + 14, len(b)))]
        elif kind == 2:
            cut = int(rng.integers(sos + 14, len(b) - 20))
            b[cut:cut + 12] = rng.integers(0, 256, 12, dtype=np.uint8).tobytes()
        elif kind == 3:
            b[int(rng.integers(sos + 14, len(b) - 2))] = 0xFF
        else:
            b[int(rng.integers(2, sos))] ^= 1 << int(rng.integers(0, 8))          # the headers
        files.append(bytes(b))
    fr, st = image.decode_jpegs(files, 64, 64, ctx=gpu_ctx)
    agree = 0
    for i, f in enumerate(files):
        o_rc, o_px = oracle.jpeg_decode_luma(f)
        if o_rc == 0 and oracle.jpeg_probe(f)[1:] != (64, 64):
            o_rc = 1                                             # decodes, but to another geometry than the batch announced
        assert st[i] == o_rc, (i, st[i], o_rc)
        if o_rc == 0:
            assert np.array_equal(fr[i], o_px), i
            agree += 1
    assert agree >= 20


def test_micro_batcher_for_jpeg_uploads(gpu_ctx, oracle):
    """48 request threads submit JPEG uploads one at a time (the reference's per-request shape, handlers.rs:232-302): every
    answer equals the oracle's record of libjpeg's luma plane with the file's BLAKE3; a progressive file comes back NEEDS_HOST."""
    import threading
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    files, want = [], []
    for i in range(96):
        f = jpeg_of(picture(64, 64, seed=i), quality=(50, 80, 95)[i % 3], subsampling=i % 3)
        files.append(f)
        ex = np.frombuffer(blake3_digest(f), np.uint8)[None]
        want.append(oracle.image_hash_batch(libjpeg_luma(f)[None], 7, pixfmt=0, exact=ex)[0][0].tobytes())
    prog = jpeg_of(picture(64, 64), progressive=True)
    b = image.JpegBatcher(64, 64, algo=image.MULTI, max_batch=64, ctx=gpu_ctx)
    got, errs = [None] * len(files), []

    def worker(t):
        try:
            for i in range(t, len(files), 48):
                got[i] = b.submit(files[i])
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(48)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs
    for i, (rec, st) in enumerate(got):
        assert st == 0 and rec == want[i], i
    rec, st = b.submit(prog)
    assert st == image.NEEDS_HOST and not any(rec)
    assert b.stats()[1] == len(files) + 1
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("side,n,quality,sub", [(768, 6, 92, 2), (1024, 5, 85, 0), (1024, 300, 92, 2), (512, 40, 95, 1), (640, 700, 80, 2)])
def test_large_files_take_several_waves(gpu_ctx, side, n, quality, sub):
    """Batches whose files average more than 24 KB run the speculative decoder with 2 or 4 waves (128 / 256 subsequences) per
    file -- 8 for batches of at most 256 large files -- (jpeg.hip launch_jpeg_decode): block scans and the chain walk then cross waves.  Planes must equal libjpeg's."""
    from ucfp_amd import image
    base = picture(side, side, seed=n)
    files, want = [], []
    for i in range(n):
        img = np.roll(base, (7 * i, 13 * i), axis=(0, 1)) if i % 5 else picture(side, side, seed=n + i)
        j = jpeg_of(img, quality=quality, subsampling=sub)
        files.append(j)
        want.append(libjpeg_luma(j))
    planes, status = image.decode_jpegs(files, side, side, ctx=gpu_ctx)
    assert not status.any(), status
    for i in range(n):
        assert np.array_equal(planes[i], want[i]), i
