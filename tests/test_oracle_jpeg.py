"""Pins the oracle's JPEG front end (oracle/ucfp_oracle_jpeg.c: T.81 baseline Huffman decoding + the IJG accurate integer
inverse DCT) against libjpeg itself, which IS importable here through Pillow: `draft("L")` makes libjpeg(-turbo) decode the
luma component alone (out_color_space = JCS_GRAYSCALE, JDCT_ISLOW) -- the plane DESIGN J1 hashes.  Every pixel must agree."""
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
from PIL import ImageFile  # noqa: E402

ImageFile.MAXBLOCK = 1 << 22          # Pillow's encoder needs room for optimize=True on the larger test images


def picture(h, w, seed=0):
    rng = np.random.default_rng(seed + 131 * h + w)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 3 + 7) & 255, (yy * 2 + xx) & 255, (xx + yy * 5) & 255], -1).astype(np.uint8)
    return base ^ rng.integers(0, 32, (h, w, 3), dtype=np.uint8)


def jpeg_of(img, mode="RGB", **kw):
    b = io.BytesIO()
    PIL.fromarray(img, mode).save(b, "JPEG", **kw)
    return b.getvalue()


def libjpeg_luma(jpg):
    im = PIL.open(io.BytesIO(jpg))
    im.draft("L", im.size)
    assert im.mode == "L"
    return np.asarray(im)


@pytest.mark.parametrize("h,w", [(256, 256), (64, 64), (33, 77), (1, 1), (17, 8), (100, 300), (8, 8), (250, 123)])
def test_luma_plane_equals_libjpegs(oracle, h, w):
    img = picture(h, w)
    n = 0
    for q in (20, 75, 95, 100):
        for sub in (0, 1, 2):                         # 4:4:4, 4:2:2, 4:2:0
            for extra in ({}, {"optimize": True}, {"restart_marker_rows": 1}, {"restart_marker_blocks": 3}):
                jpg = jpeg_of(img, quality=q, subsampling=sub, **extra)
                rc, px = oracle.jpeg_decode_luma(jpg)
                assert rc == 0 and oracle.jpeg_probe(jpg) == (0, w, h), (q, sub, extra)
                assert np.array_equal(px, libjpeg_luma(jpg)), (h, w, q, sub, extra)
                n += 1
    grey = np.asarray(PIL.fromarray(img, "RGB").convert("L"))
    for q in (40, 90):
        jpg = jpeg_of(grey, "L", quality=q)
        rc, px = oracle.jpeg_decode_luma(jpg)
        assert rc == 0 and np.array_equal(px, np.asarray(PIL.open(io.BytesIO(jpg))))
    assert n == 48


def test_flat_extreme_and_noisy_content(oracle):
    """Saturated blocks (range limiting), pure noise at quality 100 (long codes, every coefficient set), flat images (all EOB)."""
    rng = np.random.default_rng(5)
    cases = [np.zeros((40, 40, 3), np.uint8), np.full((40, 40, 3), 255, np.uint8),
             rng.integers(0, 256, (96, 96, 3), dtype=np.uint8),
             (rng.integers(0, 2, (64, 64, 1), dtype=np.uint8) * 255).repeat(3, 2)]
    for img in cases:
        for q, sub in ((100, 0), (100, 2), (1, 2), (50, 1)):
            jpg = jpeg_of(img, quality=q, subsampling=sub)
            rc, px = oracle.jpeg_decode_luma(jpg)
            assert rc == 0 and np.array_equal(px, libjpeg_luma(jpg)), (img.shape, q, sub)


def test_what_goes_back_to_the_host(oracle):
    img = picture(64, 64)
    assert oracle.jpeg_decode_luma(jpeg_of(img, progressive=True))[0] == oracle.JPG_NEEDS_HOST
    cmyk = PIL.fromarray(img, "RGB").convert("CMYK")
    b = io.BytesIO()
    cmyk.save(b, "JPEG")
    assert oracle.jpeg_decode_luma(b.getvalue())[0] == oracle.JPG_NEEDS_HOST          # four components
    good = jpeg_of(img, quality=80)
    assert oracle.jpeg_decode_luma(b"\x89PNG\r\n\x1a\n" + bytes(64))[0] == oracle.JPG_CORRUPT
    assert oracle.jpeg_decode_luma(good[: len(good) // 2])[0] == oracle.JPG_NEEDS_HOST   # the data ends early
    # a restart marker out of sequence
    rst = bytearray(jpeg_of(img, quality=80, restart_marker_rows=1))
    i = rst.index(b"\xff\xd1")
    rst[i + 1] = 0xD3
    assert oracle.jpeg_decode_luma(bytes(rst))[0] == oracle.JPG_NEEDS_HOST
    # RGB-coded (Adobe transform 0) files have no luma component to take
    try:
        b = io.BytesIO()
        PIL.fromarray(img, "RGB").save(b, "JPEG", keep_rgb=True)
        assert oracle.jpeg_decode_luma(b.getvalue())[0] == oracle.JPG_NEEDS_HOST
    except TypeError:
        pass


def with_quantiser(jpg, value):
    """The file with every entry of its 8-bit quantisation tables set to `value` (crafted: coefficients x quantiser leave the
    range an encoder produces)."""
    b = bytearray(jpg)
    pos = 2
    while pos + 4 <= len(b) and b[pos] == 0xFF:
        m, ln = b[pos + 1], b[pos + 2] << 8 | b[pos + 3]
        if m == 0xDB:
            o = pos + 4
            while o < pos + 2 + ln:
                assert b[o] >> 4 == 0
                b[o + 1:o + 65] = bytes([value]) * 64
                o += 65
        if m == 0xDA:
            break
        pos += 2 + ln
    return bytes(b)


def test_out_of_range_coefficients_go_to_the_host(oracle):
    """DESIGN J4's guards: dequantised coefficients beyond +-16383, column-pass results beyond +-23000 or samples beyond the
    range table -- where libjpeg's C code, its SIMD code and a 32-bit restatement part ways -- are NEEDS_HOST; files an
    encoder made never are, whatever the quality."""
    img = picture(64, 64, seed=3)
    img[::2, ::2] = 255 - img[::2, ::2]                                  # strong high frequencies: large AC coefficients
    for q in (1, 5, 30, 60, 100):
        f = jpeg_of(img, quality=q)
        rc, px = oracle.jpeg_decode_luma(f)
        assert rc == 0 and np.array_equal(px, libjpeg_luma(f)), q
    seen = set()
    for q in (30, 60, 95):
        for v in (40, 120, 255):
            rc, _ = oracle.jpeg_decode_luma(with_quantiser(jpeg_of(img, quality=q), v))
            assert rc in (0, oracle.JPG_NEEDS_HOST)
            seen.add(rc)
    assert oracle.JPG_NEEDS_HOST in seen
    assert oracle.jpeg_decode_luma(with_quantiser(jpeg_of(img, quality=95), 255))[0] == oracle.JPG_NEEDS_HOST


def test_j1_luma_route_vs_rgb_route_hash_distance(oracle):
    """DESIGN J1 hashes a JPEG's coded Y plane; the reference decodes to RGB and lets the SDK grey it.  How far
    apart are the two routes' hashes?  Natural-statistics pictures (smooth gradients + texture), qualities 50-95, the three
    common samplings: per 64-bit hash the two routes differ by a fraction of a bit on average -- the Y plane IS the luma of
    the decoded RGB up to chroma-upsampling and rounding noise, which the 8x8 / 32x32 box means average away.  The bound
    asserted here is what a near-duplicate threshold (the reference's UI uses <= 10 of 64 bits) has to absorb."""
    rng = np.random.default_rng(12)
    tot_bits, n_hash, worst = 0, 0, 0
    for i in range(24):
        h, w = int(rng.integers(120, 400)), int(rng.integers(120, 400))
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(xx / (17 + i) + c) * np.cos(yy / (23 + 2 * i) - c) for c in range(3)], -1)
        img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
        jpg = jpeg_of(img, quality=int(rng.integers(50, 96)), subsampling=i % 3)
        y = libjpeg_luma(jpg)
        rgb = np.asarray(PIL.open(io.BytesIO(jpg)).convert("RGB"))
        ra, _ = oracle.image_hash_batch(y[None], 7, pixfmt=0)
        rb, _ = oracle.image_hash_batch(rgb[None], 7, pixfmt=1)
        a = np.frombuffer(ra[0].tobytes()[32:], dtype="<u8")      # global + 16 block hashes
        b = np.frombuffer(rb[0].tobytes()[32:], dtype="<u8")
        d = [bin(int(x) ^ int(z)).count("1") for x, z in zip(a, b)]
        tot_bits += sum(d)
        n_hash += len(d)
        worst = max(worst, max(d))
    mean = tot_bits / n_hash
    print(f"J1 vs RGB route: mean {mean:.3f} bits of 64 per hash, worst {worst} over {n_hash} hashes")
    assert mean <= 1.0 and worst <= 8, (mean, worst)
