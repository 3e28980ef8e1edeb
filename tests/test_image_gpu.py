"""GPU parity: HIP image hashing (through the C ABI) vs the CPU oracle, bit-exact.

Covers the fused GRAY8 paths (256/512/1024 square), the generic normalise path (odd
geometry, RGB, RGBA), per-item status for geometry guards, single-algorithm records, the
host-pointer and device-pointer entry points, adversarial near-tie frames, and
size-independent properties at the BASELINE batch geometry (512x512)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(rng, n, h, w, c=None, kind="noise"):
    shape = (n, h, w) if c is None else (n, h, w, c)
    if kind == "noise":
        return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == "smooth":  # low-frequency content: realistic DCT spectra
        yy, xx = np.mgrid[0:h, 0:w]
        out = np.zeros(shape, np.uint8)
        for i in range(n):
            f = rng.uniform(0.5, 6.0, 4)
            ph = rng.uniform(0, 6.28, 4)
            img = 128 + 50 * np.sin(f[0] * xx / w * 6.28 + ph[0]) + 40 * np.cos(f[1] * yy / h * 6.28 + ph[1]) \
                + 25 * np.sin((f[2] * xx + f[3] * yy) / (w + h) * 6.28 + ph[2])
            img = np.clip(img + rng.normal(0, 3, (h, w)), 0, 255).astype(np.uint8)
            out[i] = img if c is None else np.repeat(img[..., None], c, axis=2)
        return out
    if kind == "flat":   # constant frames: every comparison is a tie
        out = np.zeros(shape, np.uint8)
        for i in range(n):
            out[i] = rng.integers(0, 256)
        return out
    raise ValueError(kind)


def _gpu_host(frames, algo, pixfmt=0, exact=None, pre=None):
    from ucfp_amd import image
    return image.fingerprint_frames(frames, algo=algo, pixfmt=pixfmt, exact=exact, preprocess=pre)


def _assert_same(gpu, ref, what):
    if not np.array_equal(gpu, ref):
        bad = np.argwhere(gpu != ref)
        i, off = bad[0]
        raise AssertionError(f"{what}: {len(bad)} differing bytes; first at frame {i} byte {off} "
                             f"gpu={gpu[i, off]:#x} oracle={ref[i, off]:#x}")


@pytest.mark.parametrize("side", [256, 512, 1024])
@pytest.mark.parametrize("kind", ["noise", "smooth", "flat"])
def test_multi_fused_gray_matches_oracle(gpu_ctx, oracle, side, kind):
    rng = np.random.default_rng(side * 7 + len(kind))
    n = 24 if side <= 512 else 6
    fr = _frames(rng, n, side, side, kind=kind)
    ex = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    gpu, st = _gpu_host(fr, 7, exact=ex)
    ref, rst = oracle.image_hash_batch(fr, 7, exact=ex)
    assert gpu.shape == (n, 536)
    assert np.array_equal(st, rst) and not st.any()
    _assert_same(gpu, ref, f"multi {side} {kind}")


@pytest.mark.parametrize("algo", [1, 2, 4])
def test_single_algorithm_records(gpu_ctx, oracle, algo):
    rng = np.random.default_rng(algo)
    fr = np.concatenate([_frames(rng, 8, 512, 512, kind="smooth"), _frames(rng, 8, 512, 512)])
    gpu, st = _gpu_host(fr, algo)
    ref, _ = oracle.image_hash_batch(fr, algo)
    assert gpu.shape == (16, 168)
    _assert_same(gpu, ref, f"single algo {algo}")
    # the single record equals the matching 168-B slice of the bundle
    multi, _ = _gpu_host(fr, 7)
    off = {1: 32, 2: 32 + 168, 4: 32 + 336}[algo]
    assert np.array_equal(multi[:, off:off + 168], gpu)


@pytest.mark.parametrize("geom", [(300, 200), (32, 32), (33, 47), (640, 480), (257, 256), (1000, 777),
                                  # streaming normaliser: up- and down-scaling, column parts (w > 1008), tall / wide,
                                  # widths not divisible by 4 (gather fallback)
                                  (1920, 1080), (100, 60), (60, 100), (300, 2000), (2000, 300), (1366, 768),
                                  (4096, 36), (36, 4096), (1012, 1012), (260, 252)])
def test_generic_geometry_gray(gpu_ctx, oracle, geom):
    w, h = geom
    rng = np.random.default_rng(w * 1000 + h)
    fr = np.concatenate([_frames(rng, 3, h, w), _frames(rng, 3, h, w, kind="smooth")])
    gpu, st = _gpu_host(fr, 7)
    ref, _ = oracle.image_hash_batch(fr, 7)
    assert not st.any()
    _assert_same(gpu, ref, f"generic {w}x{h}")


@pytest.mark.parametrize("pixfmt,c", [(1, 3), (2, 4)])
@pytest.mark.parametrize("geom", [(512, 512), (256, 256), (320, 240), (1920, 1080), (3840, 2160), (100, 60),
                                  (1000, 1000), (1364, 40), (8192, 32)])
def test_rgb_rgba(gpu_ctx, oracle, pixfmt, c, geom):
    w, h = geom
    rng = np.random.default_rng(pixfmt * 100 + w)
    fr = _frames(rng, 5 if w * h <= 1 << 21 else 2, h, w, c)
    gpu, st = _gpu_host(fr, 7, pixfmt=pixfmt)
    ref, _ = oracle.image_hash_batch(fr, 7, pixfmt=pixfmt)
    _assert_same(gpu, ref, f"pixfmt {pixfmt} {w}x{h}")


def test_reference_synthetic_png_pattern(gpu_ctx, oracle):
    """The reference's own test image: RGB (x%256, y%256, 128) (src/server/tests.rs:227-235,
    benches/end_to_end.rs:77-85) at 64x64 and 256x256 -> 536-byte bundle."""
    for side in (64, 256):
        yy, xx = np.mgrid[0:side, 0:side]
        img = np.stack([xx % 256, yy % 256, np.full_like(xx, 128)], axis=-1).astype(np.uint8)
        gpu, st = _gpu_host(img[None], 7, pixfmt=1)
        ref, _ = oracle.image_hash_batch(img[None], 7, pixfmt=1)
        assert gpu.shape == (1, 536) and st[0] == 0
        _assert_same(gpu, ref, f"synthetic_png {side}")


def test_geometry_guards_set_status(gpu_ctx, oracle):
    from ucfp_amd.image import PreprocessConfig
    rng = np.random.default_rng(5)
    small = _frames(rng, 3, 16, 64)   # height 16 < min_dimension 32
    gpu, st = _gpu_host(small, 7)
    ref, rst = oracle.image_hash_batch(small, 7)
    assert (st == -1).all() and np.array_equal(st, rst)
    assert not gpu.any()
    big = _frames(rng, 2, 300, 300)
    gpu, st = _gpu_host(big, 2, pre=PreprocessConfig(max_dimension=256))
    assert (st == -1).all()
    ok, st = _gpu_host(big, 2, pre=PreprocessConfig(max_dimension=300, min_dimension=300))
    assert not st.any()


def test_adversarial_near_ties(gpu_ctx, oracle):
    """Frames engineered so many comparisons sit on or next to the threshold: two-level
    images (aHash mean ties), 1-LSB gradients (dHash ties), checkerboards (DCT energy in few
    coefficients, the rest ~0 so the median lands among near-equal values)."""
    rng = np.random.default_rng(11)
    fr = []
    yy, xx = np.mgrid[0:512, 0:512]
    fr.append(((xx // 64 + yy // 64) % 2 * 255).astype(np.uint8))
    fr.append(((xx // 8 + yy // 8) % 2 * 255).astype(np.uint8))
    fr.append((xx // 2 % 256).astype(np.uint8))
    fr.append((yy // 2 % 256).astype(np.uint8))
    fr.append(np.where(xx < 256, 100, 101).astype(np.uint8))
    fr.append(np.where(yy < 256, 7, 8).astype(np.uint8))
    fr.append(((xx + yy) % 2 * 1 + 127).astype(np.uint8))
    for _ in range(9):
        base = rng.integers(0, 256)
        fr.append((base + rng.integers(0, 2, (512, 512))).clip(0, 255).astype(np.uint8))
    fr = np.stack(fr)
    gpu, _ = _gpu_host(fr, 7)
    ref, _ = oracle.image_hash_batch(fr, 7)
    _assert_same(gpu, ref, "adversarial")


def test_device_pointer_entry_and_synth(gpu_ctx, oracle, torch_cuda):
    """The *_dev entry points on torch-allocated HBM, on torch's current stream; the on-device
    synthetic generator equals the oracle's generator."""
    torch = torch_cuda
    from ucfp_amd import _lib, image
    n, side = 64, 512
    frames = torch.empty((n, side, side), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.load().ucfp_image_synth_dev(gpu_ctx.handle, frames.data_ptr(), n, side, side, 1000, stream))
    out = torch.zeros((n, 536), dtype=torch.uint8, device="cuda")
    status = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    image.fingerprint_frames_dev(frames.data_ptr(), n, side, side, algo=7, out_ptr=out.data_ptr(),
                                 status_ptr=status.data_ptr(), stream=stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    host_frames = oracle.image_synth(n, side, side, 1000)
    assert np.array_equal(frames.cpu().numpy(), host_frames)
    ref, _ = oracle.image_hash_batch(host_frames, 7)
    _assert_same(out.cpu().numpy(), ref, "dev entry")
    assert not status.cpu().numpy().any()


def test_properties_at_batch_geometry(gpu_ctx, torch_cuda):
    """Size-independent properties on a larger 512x512 batch (no oracle involved):
    determinism across launches, batch-order independence, a 256-px-periodic shift maps block
    hashes onto each other, and the single-algorithm path agrees with the bundle."""
    torch = torch_cuda
    from ucfp_amd import _lib, image
    n, side = 2048, 512
    frames = torch.empty((n, side, side), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.load().ucfp_image_synth_dev(gpu_ctx.handle, frames.data_ptr(), n, side, side, 0, stream))
    out1 = torch.zeros((n, 536), dtype=torch.uint8, device="cuda")
    out2 = torch.zeros_like(out1)
    image.fingerprint_frames_dev(frames.data_ptr(), n, side, side, out_ptr=out1.data_ptr(), stream=stream, ctx=gpu_ctx)
    image.fingerprint_frames_dev(frames.data_ptr(), n, side, side, out_ptr=out2.data_ptr(), stream=stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    perm = torch.randperm(n, device="cuda")
    shuffled = frames[perm].contiguous()
    out3 = torch.zeros_like(out1)
    image.fingerprint_frames_dev(shuffled.data_ptr(), n, side, side, out_ptr=out3.data_ptr(), stream=stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    assert torch.equal(out3, out1[perm])
    # cyclic shift by one block (128 source px = 64 normalised px) permutes the block hashes
    rolled = torch.roll(frames[:64], shifts=128, dims=2).contiguous()
    out4 = torch.zeros((64, 536), dtype=torch.uint8, device="cuda")
    image.fingerprint_frames_dev(rolled.data_ptr(), 64, side, side, out_ptr=out4.data_ptr(), stream=stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    a = out1[:64].cpu().numpy()
    b = out4.cpu().numpy()
    for slot in range(3):
        base = 32 + 168 * slot + 32 + 8
        blocks_a = a[:, base:base + 128].reshape(64, 4, 4, 8)
        blocks_b = b[:, base:base + 128].reshape(64, 4, 4, 8)
        assert np.array_equal(np.roll(blocks_a, 1, axis=2), blocks_b)
    # hashes are not degenerate on this workload
    glob = a[:, 32 + 168 + 32:32 + 168 + 40]
    assert len({bytes(g) for g in glob}) > 8


def test_strided_and_unaligned_frames_dev(gpu_ctx, oracle, torch_cuda):
    """Row/frame strides with padding, and a base pointer that is not 16-byte aligned (falls back
    to the generic normalise path): same records as the dense layout."""
    torch = torch_cuda
    from ucfp_amd import image
    rng = np.random.default_rng(21)
    n, side = 6, 512
    fr = _frames(rng, n, side, side, kind="smooth")
    ref, _ = oracle.image_hash_batch(fr, 7)
    stream = torch.cuda.current_stream().cuda_stream
    # (a) padded rows and frames, still 16-byte aligned -> fused path
    rs, fs = side + 64, (side + 64) * (side + 3)
    buf = torch.zeros(n * fs + 64, dtype=torch.uint8, device="cuda")
    view = buf[: n * fs].view(n, fs)[:, : side * rs].view(n, side, rs)
    view[:, :, :side] = torch.from_numpy(fr).cuda()
    out = torch.zeros((n, 536), dtype=torch.uint8, device="cuda")
    image.fingerprint_frames_dev(buf.data_ptr(), n, side, side, row_stride=rs, frame_stride=fs, out_ptr=out.data_ptr(),
                                 stream=stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    _assert_same(out.cpu().numpy(), ref, "padded strides")
    # (b) base pointer offset by 5 bytes, odd row stride -> generic path
    rs2, fs2 = side + 7, (side + 7) * side + 11
    buf2 = torch.zeros(n * fs2 + 64, dtype=torch.uint8, device="cuda")
    v2 = buf2[5: 5 + n * fs2].view(n, fs2)[:, : side * rs2].view(n, side, rs2)
    v2[:, :, :side] = torch.from_numpy(fr).cuda()
    out2 = torch.zeros((n, 536), dtype=torch.uint8, device="cuda")
    image.fingerprint_frames_dev(buf2.data_ptr() + 5, n, side, side, row_stride=rs2, frame_stride=fs2,
                                 out_ptr=out2.data_ptr(), stream=stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    _assert_same(out2.cpu().numpy(), ref, "unaligned base")


def test_large_generic_batch_crosses_workspace_chunks(gpu_ctx, oracle):
    """More frames than the 2048-plane normalise workspace holds (generic path chunks the batch)."""
    rng = np.random.default_rng(22)
    fr = _frames(rng, 2100, 96, 80)
    gpu, st = _gpu_host(fr, 7)
    ref, _ = oracle.image_hash_batch(fr, 7)
    assert not st.any()
    _assert_same(gpu, ref, "generic 2100 frames")


def test_encoded_image_adapters(gpu_ctx, oracle):
    """The reference's call shape: encoded bytes in, Record out (src/modality/image.rs:56-194), with
    `exact` = BLAKE3 of the upload and Error::Modality on undecodable input."""
    import io
    from PIL import Image
    from ucfp_amd import image
    from ucfp_amd.blake3 import blake3_digest
    from ucfp_amd.errors import ModalityError
    yy, xx = np.mgrid[0:256, 0:256]
    px = np.stack([xx % 256, yy % 256, np.full_like(xx, 128)], axis=-1).astype(np.uint8)   # benches/end_to_end.rs:77-85
    bio = io.BytesIO()
    Image.fromarray(px, "RGB").save(bio, format="PNG")
    png = bio.getvalue()
    rec = image.fingerprint(png, 7, 42)
    assert rec.algorithm == "imgfprint-multihash-v1" and rec.tenant_id == 7 and rec.record_id == 42
    assert rec.config_hash == 0 and rec.format_version == 1 and len(rec.fingerprint) == 536
    ex = np.frombuffer(blake3_digest(png), np.uint8)[None]
    ref, _ = oracle.image_hash_batch(px[None], 7, pixfmt=1, exact=ex)
    assert rec.fingerprint == ref[0].tobytes()
    ph = image.fingerprint_phash(png, image.PreprocessConfig(), 7, 42)
    assert ph.algorithm == "imgfprint-phash-v1" and ph.fingerprint == ref[0, 200:368].tobytes()
    with pytest.raises(ModalityError):
        image.fingerprint(b"not an image", 0, 0)
    with pytest.raises(ModalityError):
        image.fingerprint_with(png, 0, 0, image.PreprocessConfig(min_dimension=512))
    with pytest.raises(ModalityError):
        image.fingerprint_with(png, 0, 0, image.PreprocessConfig(max_input_bytes=10))


def test_micro_batcher_coalesces_concurrent_requests(gpu_ctx, oracle):
    """48 threads each submit frames one at a time (the reference's per-request shape);
    the batcher must return every thread ITS record, bit-exact, using far fewer launches than frames."""
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import image
    rng = np.random.default_rng(33)
    n = 600
    fr = np.concatenate([_frames(rng, n // 2, 256, 256), _frames(rng, n // 2, 256, 256, kind="smooth")])
    ex = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ref, _ = oracle.image_hash_batch(fr, 7, exact=ex)
    b = image.ImageBatcher(256, 256, max_batch=64, max_delay_us=2000, ctx=gpu_ctx)
    try:
        with ThreadPoolExecutor(48) as pool:
            got = list(pool.map(lambda i: b.submit(fr[i], ex[i].tobytes()), range(n)))
        for i, (rec, st) in enumerate(got):
            assert st == 0 and rec == ref[i].tobytes(), i
        batches, items = b.stats()
        assert items == n and batches < n // 4, (batches, items)
        # a lone request is flushed by the deadline, not stuck waiting for a full batch
        rec, st = b.submit(fr[0], ex[0].tobytes())
        assert rec == ref[0].tobytes()
    finally:
        b.close()


def test_random_geometries_match_oracle(gpu_ctx, oracle):
    """Thirty random (width, height, pixel format) draws through the streaming normaliser and its fallbacks:
    widths around the strip / column-part boundaries, extreme aspect ratios, both scaling directions."""
    rng = np.random.default_rng(424242)
    special_w = [32, 36, 252, 256, 260, 508, 1004, 1008, 1012, 2016, 2020, 4096]
    for trial in range(30):
        w = int(rng.choice(special_w)) if trial % 3 == 0 else int(rng.integers(8, 700)) * 4
        h = int(rng.integers(32, 1400)) if trial % 5 else int(rng.choice([32, 33, 255, 256, 257, 512]))
        if trial % 7 == 6:
            w += int(rng.integers(1, 4))          # not a multiple of 4: gather fallback
        pixfmt = int(rng.integers(0, 3))
        c = (1, 3, 4)[pixfmt]
        shape = (2, h, w) if c == 1 else (2, h, w, c)
        fr = rng.integers(0, 256, shape, dtype=np.uint8)
        fr[1] = (fr[1].astype(np.uint16) // 8 + (np.arange(w, dtype=np.uint16)[None, :, None] if c > 1
                                                 else np.arange(w, dtype=np.uint16)[None, :]) % 200).astype(np.uint8)
        gpu, st = _gpu_host(fr, 7, pixfmt=pixfmt)
        ref, _ = oracle.image_hash_batch(fr, 7, pixfmt=pixfmt)
        assert not st.any(), (w, h, pixfmt)
        _assert_same(gpu, ref, f"random geometry {w}x{h} pixfmt {pixfmt}")
