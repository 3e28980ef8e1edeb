"""GPU parity of the RAGGED image entry (ucfp_image_hash_ragged[_dev], the fused any-geometry kernels of image.hip): frames
of any mix of sizes, strides and pixel formats in ONE call, each record bit-equal to the oracle's for that frame alone.
The reference's image route takes any upload (src/server/handlers.rs:232-302 -> src/modality/image.rs:54-88)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BPP = {0: 1, 1: 3, 2: 4}


def _frame(rng, h, w, fmt, kind):
    shape = (h, w) if fmt == 0 else (h, w, BPP[fmt])
    if kind == 0:
        return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == 1:          # smooth content: realistic spectra, many near-ties after normalisation
        yy, xx = np.mgrid[0:h, 0:w]
        img = 128 + 60 * np.sin(xx / max(w, 1) * rng.uniform(2, 20)) + 50 * np.cos(yy / max(h, 1) * rng.uniform(2, 20))
        img = np.clip(img + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)
        return img if fmt == 0 else np.repeat(img[..., None], BPP[fmt], axis=2) ^ rng.integers(0, 4, shape, dtype=np.uint8)
    return np.full(shape, rng.integers(0, 256), np.uint8)      # flat: every comparison a tie


def _oracle_one(oracle, f, fmt, algo, ex=None):
    rec, st = oracle.image_hash_batch(f[None], algo, pixfmt=fmt, exact=None if ex is None else ex[None])
    return rec[0], int(st[0])


def _check(oracle, frames, fmts, rec, st, algo=7, exact=None):
    for i, (f, fmt) in enumerate(zip(frames, fmts)):
        ref, rst = _oracle_one(oracle, f, fmt, algo, None if exact is None else exact[i])
        assert int(st[i]) == rst, (i, f.shape, fmt, int(st[i]), rst)
        if not np.array_equal(rec[i], ref):
            bad = np.flatnonzero(rec[i] != ref)
            raise AssertionError(f"frame {i} {f.shape} fmt {fmt}: {bad.size} differing bytes, first at {bad[0]}")


def test_mixed_sizes_and_formats_in_one_call(gpu_ctx, oracle):
    from ucfp_amd import image
    rng = np.random.default_rng(20240)
    sizes = [(32, 32), (33, 47), (200, 300), (200, 301), (480, 640), (481, 641), (256, 256), (255, 257), (64, 500),
             (500, 64), (100, 512), (100, 513), (300, 1023), (767, 1023), (90, 1024), (64, 1025), (720, 1280), (40, 2050),
             (37, 3001), (1000, 40), (512, 512), (512, 516), (129, 253), (57, 255)]
    frames, fmts = [], []
    for i, (h, w) in enumerate(sizes):
        for fmt in (0, 1, 2):
            frames.append(_frame(rng, h, w, fmt, (i + fmt) % 3))
            fmts.append(fmt)
    ex = rng.integers(0, 256, (len(frames), 32), dtype=np.uint8)
    rec, st = image.fingerprint_frames_ragged(frames, fmts, exact=ex, ctx=gpu_ctx)
    assert rec.shape == (len(frames), 536) and not st.any()
    _check(oracle, frames, fmts, rec, st, exact=ex)
    for algo in (image.PHASH, image.AHASH, image.DHASH):
        rec1, st1 = image.fingerprint_frames_ragged(frames[:30], fmts[:30], algo=algo, exact=ex[:30], ctx=gpu_ctx)
        assert rec1.shape == (30, 168)
        _check(oracle, frames[:30], fmts[:30], rec1, st1, algo=algo, exact=ex[:30])


def test_random_geometries_strides_and_offsets_on_the_device(gpu_ctx, oracle, torch_cuda):
    """Frames packed at arbitrary byte offsets with padded rows (every alignment class of the loaders), device entry."""
    torch = torch_cuda
    from ucfp_amd import image
    rng = np.random.default_rng(777)
    frames, fmts, geoms = [], [], []
    blob = bytearray(rng.integers(0, 256, 13, dtype=np.uint8).tobytes())
    for i in range(150):
        fmt = int(rng.integers(0, 3))
        w = int(rng.integers(32, 700)) if i % 9 else int(rng.integers(700, 2600))
        h = int(rng.integers(32, 500))
        f = _frame(rng, h, w, fmt, i % 3)
        pad = int(rng.integers(0, 9)) if i % 2 else 0
        rs = w * BPP[fmt] + pad
        if i % 4 == 0:                     # an aligned frame now and then: the fast loaders
            while len(blob) % 16:
                blob.append(0)
            rs = (w * BPP[fmt] + 15) & ~15
        off = len(blob)
        rows = np.zeros((h, rs), np.uint8)
        rows[:, :w * BPP[fmt]] = f.reshape(h, -1)
        rows[:, w * BPP[fmt]:] = rng.integers(0, 256, (h, rs - w * BPP[fmt]), dtype=np.uint8)      # padding is never read as pixels
        blob += rows.tobytes()
        blob += rng.integers(0, 256, int(rng.integers(0, 7)), dtype=np.uint8).tobytes()
        frames.append(f)
        fmts.append(fmt)
        geoms.append((off, w, h, rs, fmt))
    n = len(frames)
    d_blob = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).cuda()
    d_out = torch.zeros((n, 536), dtype=torch.uint8, device="cuda")
    d_st = torch.full((n,), 99, dtype=torch.int32, device="cuda")
    image.fingerprint_frames_ragged_dev(d_blob.data_ptr(), d_blob.numel(), geoms, out_ptr=d_out.data_ptr(),
                                        status_ptr=d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, ctx=gpu_ctx)
    torch.cuda.synchronize()
    _check(oracle, frames, fmts, d_out.cpu().numpy(), d_st.cpu().numpy())


def test_guards_reject_single_frames_not_the_batch(gpu_ctx, oracle):
    from ucfp_amd import errors, image
    from ucfp_amd.image import PreprocessConfig
    rng = np.random.default_rng(5)
    frames = [_frame(rng, 100, 100, 0, 0), _frame(rng, 31, 100, 1, 0), _frame(rng, 100, 20, 0, 0), _frame(rng, 64, 64, 2, 0),
              _frame(rng, 300, 90, 1, 0)]
    fmts = [0, 1, 0, 2, 1]
    rec, st = image.fingerprint_frames_ragged(frames, fmts, ctx=gpu_ctx)
    assert list(st) == [0, -1, -1, 0, 0] and not rec[1].any() and not rec[2].any()
    _check(oracle, frames, fmts, rec, st)
    # a narrower window: the 300-row frame now falls outside max_dimension
    rec, st = image.fingerprint_frames_ragged(frames, fmts, preprocess=PreprocessConfig(max_dimension=256, min_dimension=50), ctx=gpu_ctx)
    assert list(st) == [0, -1, -1, 0, -1]
    # an item that reaches beyond the buffer is the caller's bug: the whole call is refused
    with pytest.raises(errors.InvalidArgument):
        image.fingerprint_frames_ragged_dev(1 << 20, 1000, [(0, 64, 64, 64, 0)], out_ptr=1 << 20, ctx=gpu_ctx)
    assert image.fingerprint_frames_ragged([], [], ctx=gpu_ctx)[0].shape == (0, 536)


def test_large_frames_inside_a_ragged_batch(gpu_ctx, oracle):
    """Frames beyond the fused kernel's size (more than 2^20 pixels by default) leave through the many-waves-per-frame path,
    in the same call; rows of up to 8192 pixels are cut into column parts."""
    from ucfp_amd import image
    rng = np.random.default_rng(99)
    frames = [_frame(rng, 1200, 1600, 0, 1), _frame(rng, 64, 64, 1, 0), _frame(rng, 40, 8192, 0, 0), _frame(rng, 100, 4100, 1, 1),
              _frame(rng, 2048, 1100, 2, 0), _frame(rng, 300, 300, 0, 2), _frame(rng, 128, 8190, 2, 1)]
    fmts = [0, 1, 0, 1, 2, 0, 2]
    rec, st = image.fingerprint_frames_ragged(frames, fmts, ctx=gpu_ctx)
    assert not st.any()
    _check(oracle, frames, fmts, rec, st)


def test_uniform_calls_of_odd_geometries_equal_the_ragged_ones(gpu_ctx, oracle):
    """The uniform entry routes geometries the square kernels do not take through the same fused kernel (a table-free
    launch): same records as the ragged call and the oracle."""
    from ucfp_amd import image
    rng = np.random.default_rng(31337)
    for (h, w, fmt) in [(200, 300, 1), (200, 301, 1), (481, 641, 0), (480, 640, 0), (600, 1000, 2), (65, 33, 1)]:
        fr = np.stack([_frame(rng, h, w, fmt, k % 3) for k in range(9)])
        rec, st = image.fingerprint_frames(fr, pixfmt=fmt, ctx=gpu_ctx)
        rec2, st2 = image.fingerprint_frames_ragged(list(fr), [fmt] * 9, ctx=gpu_ctx)
        assert np.array_equal(rec, rec2) and not st.any() and not st2.any()
        _check(oracle, list(fr), [fmt] * 9, rec, st)
