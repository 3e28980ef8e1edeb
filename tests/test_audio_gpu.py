"""GPU parity: HIP Wang / Haitsma (through the C ABI) vs the CPU oracle, bit-exact on the integer
outputs (hash words, anchor times, sub-fingerprint bits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _signal(kind, seconds, sr, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * sr)) / sr
    if kind == "sine440":   # the reference's own fixture: src/server/tests.rs:322-331, benches/end_to_end.rs:55-75
        return (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    if kind == "chirps":    # log-spaced chirps + noise at -30 dB
        x = np.zeros_like(t)
        for i in range(8):
            f0 = 100.0 * (1.5 ** i)
            x += 0.06 * np.sin(2 * np.pi * (f0 * t + 0.5 * (f0 / 4) * t * t / max(seconds, 1e-3)))
        x += 0.0316 * 0.5 * rng.standard_normal(t.size)
        return np.clip(x, -0.5, 0.5).astype(np.float32)
    if kind == "noise":
        return (0.2 * rng.standard_normal(t.size)).astype(np.float32)
    if kind == "silence":
        return np.zeros(t.size, np.float32)
    if kind == "clicks":
        x = np.zeros(t.size, np.float32)
        x[:: sr // 7] = 0.9
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("kind,seconds", [("sine440", 1.0), ("sine440", 4.0), ("chirps", 12.0), ("noise", 6.5),
                                          ("silence", 3.0), ("clicks", 5.0), ("chirps", 0.2), ("noise", 0.128)])
def test_wang_matches_oracle(gpu_ctx, oracle, kind, seconds):
    from ucfp_amd import audio
    x = _signal(kind, seconds, 8000, seed=int(seconds * 10))
    g = audio.wang_hashes(x, 8000, ctx=gpu_ctx)
    o = oracle.wang(x)
    assert g.shape == o.shape, (g.shape, o.shape)
    assert np.array_equal(g, o)
    if kind in ("chirps", "noise") and seconds > 2:
        assert g.shape[0] > 50
        # layout: anchor frequency in bits 31..23, dt in the low 14 bits (LandmarkScatter.svelte:31-37)
        dt = g[:, 0] & 0x3FFF
        assert dt.min() >= 1 and dt.max() <= 63
        assert (np.diff(g[:, 1].astype(np.int64)) >= 0).all()   # anchors in time order


@pytest.mark.parametrize("frames", [1, 7, 8, 12, 13, 24, 25, 47, 48, 49, 55, 56, 96, 255, 256, 257, 264, 531])
def test_wang_frame_counts_around_segment_and_round_edges(gpu_ctx, oracle, frames):
    """The streaming kernel walks segments (48 frames for inputs this short, up to 512 for hours of audio) in rounds
    of 12 with a +-7-frame halo and judges a frame one round after its window completes: every count near those
    edges must still give the oracle's landmarks."""
    from ucfp_amd import audio
    x = _signal("noise", (1024 + 128 * (frames - 1) + 5) / 8000.0, 8000, seed=frames)
    assert 1 + (x.size - 1024) // 128 == frames
    g = audio.wang_hashes(x, 8000, ctx=gpu_ctx)
    o = oracle.wang(x)
    assert g.shape == o.shape and np.array_equal(g, o)


@pytest.mark.parametrize("frames", [90, 300])
def test_wang_ties_across_rows(gpu_ctx, oracle, frames):
    """A signal with the hop as its period under a staircase envelope (runs of 20 hops, levels apart by powers of
    two) gives runs of frames with bit-identical spectra: cells tie with their time neighbours, so which of them is
    the peak is decided by the earliest-cell-wins rule."""
    from ucfp_amd import audio
    rng = np.random.default_rng(7)
    n = 1024 + 128 * (frames - 1)
    x = np.tile((0.3 * rng.standard_normal(128)).astype(np.float32), frames + 8)[:n]
    env = np.repeat(np.array([1.0, 0.5, 1.0, 1.0, 0.25, 1.0, 0.5], np.float32), 128 * 20)
    x = (x * np.tile(env, 1 + n // env.size)[:n]).astype(np.float32)
    g = audio.wang_hashes(x, 8000, ctx=gpu_ctx)
    o = oracle.wang(x)
    assert o.shape[0] > 40
    assert g.shape == o.shape and np.array_equal(g, o)


def test_wang_config_variants(gpu_ctx, oracle):
    from ucfp_amd import audio
    x = _signal("chirps", 8.0, 8000, seed=3)
    for cfg in (dict(fan_out=3, target_zone_t=20, target_zone_f=10, peaks_per_sec=12, min_anchor_mag_db=-30.0),
                dict(fan_out=64, target_zone_t=200, target_zone_f=300, peaks_per_sec=80, min_anchor_mag_db=-90.0)):
        g = audio.wang_hashes(x, 8000, audio.WangConfig(**cfg), ctx=gpu_ctx)
        ocfg = oracle.WangCfg(cfg["fan_out"], cfg["target_zone_t"], cfg["target_zone_f"], cfg["peaks_per_sec"],
                              cfg["min_anchor_mag_db"])
        o = oracle.wang(x, ocfg, cap=200000)
        assert np.array_equal(g, o)


@pytest.mark.parametrize("seconds", [75.0, 530.0, 1100.0])
def test_wang_long_stream_crosses_chunks(gpu_ctx, oracle, seconds):
    """Long streams: the stream kernel walks 256-frame segments with a +-7-frame halo and the per-second cap,
    scan and pairing stages span many workgroups.  Bit-exact against the (unsegmented) oracle."""
    from ucfp_amd import audio
    x = _signal("chirps", seconds, 8000, seed=9)
    g = audio.wang_hashes(x, 8000, ctx=gpu_ctx)
    o = oracle.wang(x)
    assert g.shape == o.shape and np.array_equal(g, o)


def test_wang_hours_long_stream_uses_the_longest_segments(gpu_ctx, oracle):
    """Beyond 524 288 frames (2.3 h at 8 kHz) a workgroup's segment is capped at 512 frames; bit-exact against the
    oracle over 2.5 h of noise + tones."""
    from ucfp_amd import audio
    rng = np.random.default_rng(3)
    n = 9000 * 8000
    t = np.arange(n, dtype=np.float32) * np.float32(1.0 / 8000.0)
    x = (0.1 * rng.standard_normal(n).astype(np.float32) + 0.2 * np.sin(2 * np.pi * 523.25 * t)).astype(np.float32)
    assert 1 + (n - 1024) // 128 > 512 * 1024
    g = audio.wang_hashes(x, 8000, ctx=gpu_ctx)
    o = oracle.wang(x)
    assert g.shape == o.shape and g.shape[0] > 100_000 and np.array_equal(g, o)


def test_haitsma_long_stream_crosses_chunks(gpu_ctx, oracle):
    """Haitsma chunks hold 131 072 frames (1678 s at 5 kHz) plus one frame of history."""
    from ucfp_amd import audio
    x = _signal("noise", 1750.0, 5000, seed=4)
    g = audio.haitsma_frames(x, 5000, ctx=gpu_ctx)
    o = oracle.haitsma(x, 5000)
    assert g.shape == o.shape and g.shape[0] > 131072 and np.array_equal(g, o)


def test_wang_rejects_other_rates(gpu_ctx):
    from ucfp_amd import audio
    from ucfp_amd.errors import ModalityError
    x = _signal("sine440", 1.0, 44100)
    with pytest.raises(ModalityError, match="8 kHz"):     # src/modality/audio.rs:424-428
        audio.wang_hashes(x, 44100, ctx=gpu_ctx)
    with pytest.raises(ModalityError):
        audio.wang_hashes(x, 0, ctx=gpu_ctx)


@pytest.mark.parametrize("kind,seconds,sr", [("chirps", 6.0, 5000), ("chirps", 6.0, 8000), ("noise", 3.0, 44100),
                                             ("sine440", 2.0, 16000), ("silence", 1.0, 5000), ("noise", 0.3, 5000)])
def test_haitsma_matches_oracle(gpu_ctx, oracle, kind, seconds, sr):
    from ucfp_amd import audio
    x = _signal(kind, seconds, sr, seed=sr % 97)
    g = audio.haitsma_frames(x, sr, ctx=gpu_ctx)
    o = oracle.haitsma(x, sr)
    assert g.shape == o.shape
    assert np.array_equal(g, o)
    if seconds >= 2:
        assert abs(g.shape[0] / seconds - 78.125) < 78.125 * 0.25   # 312 B/s (algorithms_manifest.rs:654)


def test_resample_dev_matches_oracle(gpu_ctx, oracle, torch_cuda):
    torch = torch_cuda
    from ucfp_amd import _lib
    x = _signal("chirps", 3.0, 44100, seed=5)
    for sr_out in (8000, 5000, 48000):
        m = int(_lib.load().ucfp_audio_resample_len(x.size, 44100, sr_out))
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros(m, dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().ucfp_audio_resample_linear_dev(gpu_ctx.handle, d_in.data_ptr(), x.size, 44100, sr_out,
                                                              d_out.data_ptr(), m,
                                                              torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        o = oracle.resample_linear(x, 44100, sr_out)
        assert o.size == m and np.array_equal(d_out.cpu().numpy().view(np.uint32), o.view(np.uint32))


def test_records_and_streaming_session(gpu_ctx):
    from ucfp_amd import audio
    x = _signal("chirps", 4.0, 8000, seed=1)
    rec = audio.fingerprint_wang(x, 8000, 3, 9)
    assert rec.algorithm == "audiofp-wang-v1" and rec.format_version == 1 and rec.config_hash == 0
    assert len(rec.fingerprint) % 8 == 0 and len(rec.fingerprint) > 0     # algorithmView.ts:22
    hk = audio.fingerprint_haitsma(x, 8000, 3, 9)
    assert hk.algorithm == "audiofp-haitsma-v1" and len(hk.fingerprint) % 4 == 0
    s = audio.StreamingWangSession(8000, 3, 9)
    assert s.push(x[:10000]) == [] and s.push(x[10000:]) == []
    out = s.finalize()
    assert len(out) == 1 and out[0].fingerprint == rec.fingerprint


# ---- round 2: ragged batches and the resampler fused into the stream kernel ------------------------------------
def _clip(rng, n, kind):
    t = np.arange(n, dtype=np.float64)
    if kind == 0:
        x = 0.4 * np.sin(2 * np.pi * (200.0 + 1500.0 * rng.random()) * t / 8000.0 * (1 + 0.1 * t / max(n, 1)))
    elif kind == 1:
        x = 0.2 * rng.standard_normal(n)
    elif kind == 2:
        x = np.zeros(n)
        x[rng.integers(0, max(n, 1), size=max(1, n // 900))] = rng.uniform(-1, 1, size=max(1, n // 900))
    else:
        x = 0.3 * np.sign(np.sin(2 * np.pi * 311.0 * t / 8000.0)) + 0.02 * rng.standard_normal(n)
    return x.astype(np.float32)


def test_wang_batch_of_4096_random_length_clips_matches_oracle(gpu_ctx, oracle):
    """4 096 clips of 0 .. 6 s at 8 kHz (empty, shorter than one frame, one frame exactly, several seconds) through ONE
    call of ucfp_audio_wang_batch_dev; every clip's hashes equal the oracle's for that clip alone."""
    from ucfp_amd import audio
    rng = np.random.default_rng(4096)
    lens = rng.integers(0, 48_000, size=4096)
    lens[:8] = [0, 1, 1023, 1024, 1025, 1151, 1152, 8000]
    clips = [_clip(rng, int(n), i % 4) for i, n in enumerate(lens)]
    got = audio.wang_hashes_batch(clips, 8000, ctx=gpu_ctx)
    assert len(got) == len(clips)
    total = 0
    for i, (c, g) in enumerate(zip(clips, got)):
        o = oracle.wang(c)
        assert g.shape == o.shape and np.array_equal(g, o), (i, c.size, g.shape, o.shape)
        total += o.shape[0]
    assert total > 100_000          # the batch is not trivially empty


@pytest.mark.parametrize("sr", [44100, 16000, 11025, 48000, 7999])
def test_wang_batch_fused_resample_matches_resample_then_wang(gpu_ctx, oracle, sr):
    """A clip at another rate: the stream kernel's own A1 resampling = oracle.resample_linear followed by oracle.wang."""
    from ucfp_amd import audio
    rng = np.random.default_rng(sr)
    clips = []
    for i, secs in enumerate((0.05, 0.129, 1.0, 3.7, 12.3)):
        n = int(secs * sr)
        t = np.arange(n) / sr
        x = 0.3 * np.sin(2 * np.pi * (300 + 90 * i + 40 * t) * t) + 0.05 * rng.standard_normal(n)
        clips.append(x.astype(np.float32))
    cfg = audio.WangConfig(fan_out=5, target_zone_t=40, target_zone_f=80, peaks_per_sec=20, min_anchor_mag_db=-60.0)
    got = audio.wang_hashes_batch(clips, sr, cfg, ctx=gpu_ctx)
    ocfg = oracle.WangCfg(5, 40, 80, 20, -60.0)
    for c, g in zip(clips, got):
        o = oracle.wang(oracle.resample_linear(c, sr, 8000), ocfg)
        assert g.shape == o.shape and np.array_equal(g, o), (sr, c.size, g.shape, o.shape)


def test_wang_batch_long_clip_between_short_ones(gpu_ctx, oracle):
    """A 20-minute clip (many workgroup segments) between two short ones: segment -> clip and second -> clip maps,
    and pairing never crosses a clip boundary."""
    from ucfp_amd import audio
    rng = np.random.default_rng(77)
    clips = [_clip(rng, 20_000, 0), _signal("chirps", 1200.0, 8000, seed=3), _clip(rng, 30_000, 3)]
    got = audio.wang_hashes_batch(clips, 8000, ctx=gpu_ctx)
    for c, g in zip(clips, got):
        o = oracle.wang(c)
        assert g.shape == o.shape and np.array_equal(g, o)


@pytest.mark.parametrize("sr", [8000, 44100])
def test_micro_batcher_coalesces_concurrent_clips(gpu_ctx, oracle, sr):
    """32 threads each submit one clip at a time (handlers.rs:704-918); every thread gets the
    hashes of ITS clip (t_anchor relative to the clip), bit-exact, from far fewer launch sequences than clips."""
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import audio
    rng = np.random.default_rng(sr)
    lens = rng.integers(0, 5 * sr, size=300)
    lens[:4] = [0, 1, sr // 8, 4 * sr]
    clips = [_clip(rng, int(n), i % 4) for i, n in enumerate(lens)]
    ref = [oracle.wang(c if sr == 8000 else oracle.resample_linear(c, sr, 8000)) for c in clips]
    b = audio.WangBatcher(sr, max_batch=64, max_samples=64 * 5 * sr, max_delay_us=3000, ctx=gpu_ctx)
    try:
        with ThreadPoolExecutor(32) as pool:
            got = list(pool.map(b.submit, clips))
        for i, (g, o) in enumerate(zip(got, ref)):
            assert g.shape == o.shape and np.array_equal(g, o), (i, clips[i].size, g.shape, o.shape)
        batches, items = b.stats()
        assert items == len(clips) and batches < len(clips) // 3, (batches, items)
        g = b.submit(clips[3])                          # a lone clip is flushed by the deadline
        assert np.array_equal(g, ref[3])
    finally:
        b.close()


@pytest.mark.parametrize("sr", [5000, 8000, 44100])
def test_haitsma_batch_of_random_length_clips_matches_oracle(gpu_ctx, oracle, sr):
    """700 clips of 0 .. 3 s (empty, shorter than one frame, exactly one frame, several seconds) through ONE call of
    ucfp_audio_haitsma_batch_dev; every clip's sub-fingerprints equal the oracle's for that clip alone -- in particular a
    clip's first frame has a zero history whatever clip precedes it."""
    from ucfp_amd import audio
    rng = np.random.default_rng(sr + 1)
    one = (2048 * sr + 4999) // 5000                   # source samples that give one 2048-sample frame at 5 kHz
    lens = rng.integers(0, 3 * sr, size=700)
    lens[:8] = [0, 1, one - 2, one, one + 1, one + 64 * sr // 5000 + 2, sr, 2 * sr]
    clips = [_clip(rng, int(n), i % 4) for i, n in enumerate(lens)]
    got = audio.haitsma_frames_batch(clips, sr, ctx=gpu_ctx)
    assert len(got) == len(clips)
    total = 0
    for i, (c, g) in enumerate(zip(clips, got)):
        o = oracle.haitsma(c, sr)
        assert g.shape == o.shape and np.array_equal(g, o), (i, c.size, g.shape, o.shape)
        total += o.size
    assert total > 20_000
    cfg = audio.HaitsmaConfig(fmin=400.0, fmax=1800.0)
    got = audio.haitsma_frames_batch(clips[:40], sr, cfg, ctx=gpu_ctx)
    for c, g in zip(clips[:40], got):
        assert np.array_equal(g, oracle.haitsma(c, sr, 400.0, 1800.0))
