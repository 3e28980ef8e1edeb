"""GPU parity of the any-size, any-format upload entries (ucfp_image_probe, ucfp_image_upload_decode_batch_dev,
ucfp_image_upload_hash_batch_dev, ucfp_upload_batcher_*; upload.hip over png.hip / jpeg.hip / image.hip): PNG and JPEG files
of different geometries in ONE batch -- what the reference's route receives (src/server/handlers.rs:232-302 ->
src/modality/image.rs:54-88) -- decoded to Pillow's / libjpeg's pixels and hashed to the oracle's records."""
import io
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")

from test_oracle_jpeg import jpeg_of, libjpeg_luma, picture   # noqa: E402


def _png(arr, mode=None, **kw):
    b = io.BytesIO()
    PIL.fromarray(arr, mode).save(b, "PNG", **kw)
    return b.getvalue()


def _mixed_files(rng, n):
    """n uploads: PNG (grey, grey + alpha, RGB, palette, RGBA) and JPEG (4:4:4, 4:2:2, 4:2:0, grey, restart markers,
    optimised tables) of random sizes -> (files, expected decoded frames as the HOST path sees them)."""
    files, frames = [], []
    for i in range(n):
        h, w = int(rng.integers(32, 420)), int(rng.integers(32, 520))
        if i % 11 == 0:
            w, h = int(rng.integers(520, 1300)), int(rng.integers(32, 200))     # (a PNG's scanlines stay under the upload path's 1 MiB)
        img = picture(h, w, seed=int(rng.integers(1 << 30)))
        kind = i % 9
        if kind == 0:
            g = np.asarray(PIL.fromarray(img, "RGB").convert("L"))
            files.append(_png(g, "L", compress_level=int(rng.integers(1, 7))))
            frames.append(g)
        elif kind == 1:
            files.append(_png(img, "RGB", compress_level=int(rng.integers(1, 7))))
            frames.append(img)
        elif kind == 2:
            a = np.dstack([img, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)])
            files.append(_png(a, "RGBA"))
            frames.append(a)
        elif kind == 3:
            p = PIL.fromarray(img, "RGB").quantize(64)
            b = io.BytesIO()
            p.save(b, "PNG")
            files.append(b.getvalue())
            frames.append(np.asarray(p.convert("RGB")))
        elif kind == 4:
            g = np.asarray(PIL.fromarray(img, "RGB").convert("L"))
            la = np.dstack([g, rng.integers(0, 256, (h, w), dtype=np.uint8)])
            files.append(_png(la, "LA"))
            frames.append(g)
        else:
            extra = [{}, {"optimize": True}, {"restart_marker_rows": 1}, {}][kind - 5]
            if kind == 8:
                f = jpeg_of(np.asarray(PIL.fromarray(img, "RGB").convert("L")), "L", quality=int(rng.integers(40, 96)))
            else:
                f = jpeg_of(img, quality=int(rng.integers(40, 96)), subsampling=int(rng.integers(0, 3)), **extra)
            files.append(f)
            frames.append(libjpeg_luma(f))
    return files, frames


def _fmt_of(fr):
    return 0 if fr.ndim == 2 else (1 if fr.shape[2] == 3 else 2)


def test_probe_tells_kind_geometry_and_who_decodes(gpu_ctx):
    from ucfp_amd import image
    rng = np.random.default_rng(1)
    files, frames = _mixed_files(rng, 18)
    for f, fr in zip(files, frames):
        p = image.probe(f)
        assert p.status == 0 and (p.height, p.width) == fr.shape[:2] and p.pixfmt == _fmt_of(fr)
        assert p.format == (image.UPLOAD_PNG if f[:4] == b"\x89PNG" else image.UPLOAD_JPEG)
    img = picture(64, 64)
    b = io.BytesIO()
    PIL.fromarray(img, "RGB").save(b, "BMP")
    for other in (b.getvalue(), b"GIF89a" + bytes(40), b"RIFF\x10\0\0\0WEBPVP8 " + bytes(20), bytes(100)):
        p = image.probe(other)
        assert p.format == image.UPLOAD_OTHER and p.status == image.NEEDS_HOST
    assert image.probe(b"").status < 0
    assert image.probe(jpeg_of(img, progressive=True)).status == image.NEEDS_HOST
    assert image.probe(_png(np.zeros((8, 8), np.uint16))).status == image.NEEDS_HOST            # 16-bit PNG
    assert image.probe(b"\x89PNG\r\n\x1a\n" + bytes(40)).status < 0
    # a PNG of more than 1 MiB of scanlines is the host's (one wave inflates a file: ~33 MB/s); the same picture as JPEG is not
    big = picture(700, 700)
    assert image.probe(_png(big, "RGB")).status == image.NEEDS_HOST
    assert image.probe(jpeg_of(big, quality=80)).status == 0


def test_mixed_uploads_decode_to_the_host_decoders_pixels(gpu_ctx):
    from ucfp_amd import image
    rng = np.random.default_rng(2)
    files, frames = _mixed_files(rng, 90)
    got, st = image.decode_uploads(files, ctx=gpu_ctx)
    assert not st.any(), st
    for i, (g, fr) in enumerate(zip(got, frames)):
        assert g is not None and g.shape == fr.shape and np.array_equal(g, fr), (i, fr.shape)


@pytest.mark.parametrize("on_device", [False, True])
def test_mixed_uploads_hash_to_the_oracles_records(gpu_ctx, oracle, on_device):
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    rng = np.random.default_rng(3)
    files, frames = _mixed_files(rng, 120)
    # things the device hands back or refuses, in between
    img = picture(80, 90)
    b = io.BytesIO()
    PIL.fromarray(img, "RGB").save(b, "BMP")
    odd = {7: (jpeg_of(img, progressive=True), 1), 23: (b.getvalue(), 1), 40: (b"", -1), 55: (_png(picture(20, 300)), -1),
           71: (files[5][: len(files[5]) // 2], None), 90: (_png(np.zeros((40, 40), np.uint16)), 1)}
    for k, (f, _) in odd.items():
        files[k] = f
    rec, st = image.fingerprint_uploads(files, probe_on_device=on_device, ctx=gpu_ctx)
    for i, f in enumerate(files):
        if i in odd:
            want = odd[i][1]
            assert (st[i] != 0 if want is None else st[i] == want), (i, st[i])
            assert not rec[i].any()
            continue
        ex = np.frombuffer(blake3_digest(f), np.uint8)
        ref, rst = oracle.image_hash_batch(frames[i][None], 7, pixfmt=_fmt_of(frames[i]), exact=ex[None])
        assert st[i] == 0 and rst[0] == 0 and np.array_equal(rec[i], ref[0]), (i, frames[i].shape)
    # one algorithm, host-supplied exact digests
    ex = rng.integers(0, 256, (len(files), 32), dtype=np.uint8)
    rec1, st1 = image.fingerprint_uploads(files[:30], algo=image.PHASH, exact=ex[:30], ctx=gpu_ctx)
    for i in range(30):
        if i in odd:
            continue
        ref, _ = oracle.image_hash_batch(frames[i][None], image.PHASH, pixfmt=_fmt_of(frames[i]), exact=ex[i][None])
        assert st1[i] == 0 and np.array_equal(rec1[i], ref[0]), i


def test_the_per_request_path_gives_the_same_records(gpu_ctx, oracle):
    """One record whichever path decodes: the host adapter (Pillow decode -> fingerprint_with) and the device front end."""
    from ucfp_amd import image
    from ucfp_amd.image import PreprocessConfig
    rng = np.random.default_rng(4)
    files, _ = _mixed_files(rng, 27)
    rec, st = image.fingerprint_uploads(files, ctx=gpu_ctx)
    assert not st.any()
    for i, f in enumerate(files):
        r = image.fingerprint_with(f, 1, i, PreprocessConfig())
        assert bytes(r.fingerprint) == rec[i].tobytes(), i


def test_one_batcher_for_every_size_and_format_under_48_threads(gpu_ctx, oracle):
    """The any-upload micro-batcher: 48 request threads, PNG and JPEG files of different sizes and kinds plus things the
    device does not decode, through ONE batcher created without any geometry; every record bit-equal to the oracle's."""
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    rng = np.random.default_rng(6)
    files, frames = _mixed_files(rng, 96)
    b = io.BytesIO()
    PIL.fromarray(picture(50, 60), "RGB").save(b, "BMP")
    extra = [(b.getvalue(), 1), (jpeg_of(picture(64, 64), progressive=True), 1), (b"", -1), (_png(picture(16, 16)), -1)]
    want = []
    for f, fr in zip(files, frames):
        ex = np.frombuffer(blake3_digest(f), np.uint8)
        want.append(oracle.image_hash_batch(fr[None], 7, pixfmt=_fmt_of(fr), exact=ex[None])[0][0].tobytes())
    bt = image.UploadBatcher(max_batch=64, max_bytes=64 << 20, max_delay_us=300, ctx=gpu_ctx)
    errors = []

    def worker(tid):
        try:
            r = np.random.default_rng(tid)
            for _ in range(12):
                k = int(r.integers(0, len(files) + len(extra)))
                if k < len(files):
                    rec, st = bt.submit(files[k])
                    if st != 0 or rec != want[k]:
                        errors.append((tid, k, st))
                else:
                    f, code = extra[k - len(files)]
                    rec, st = bt.submit(f)
                    if st != code or any(rec):
                        errors.append((tid, "extra", k, st))
        except Exception as e:   # noqa: BLE001
            errors.append((tid, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(48)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    batches, items = bt.stats()
    bt.close()
    assert not errors, errors[:5]
    assert items > 0 and batches < items            # requests were coalesced
