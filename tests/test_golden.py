"""Golden vectors (tests/golden/golden_v1.npz, made by tools/gen_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must reproduce them too (GPU)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
TEXTS = [b"the quick brown fox jumps over the lazy dog",
         b"the quick brown fox jumps over the lazy dog. " * 128,
         b"Hello world, this is a test of the pipeline inspector.",
         b"It's 3.14 o'clock in the U.S.A., isn't it? 1,000,000 x:y foo_bar"]


def _png(side):
    yy, xx = np.mgrid[0:side, 0:side]
    return np.stack([xx % 256, yy % 256, np.full_like(xx, 128)], axis=-1).astype(np.uint8)


def _sine(secs):
    t = np.arange(int(secs * 8000), dtype=np.float32) / np.float32(8000)
    return (np.sin(np.float32(2.0 * np.pi) * np.float32(440.0) * t) * np.float32(0.5)).astype(np.float32)


def _rand512():
    return np.random.default_rng(int(G["image_rand512_seed"][0])).integers(0, 256, (4, 512, 512), dtype=np.uint8)


# ------------------------------------------------------------------ CPU: oracle vs golden
def test_oracle_image_golden(oracle):
    for side in (64, 256):
        rec, _ = oracle.image_hash_batch(_png(side)[None], 7, pixfmt=1)
        assert np.array_equal(rec[0], G[f"image_synthpng{side}_multi"])
    assert np.array_equal(oracle.image_hash_batch(_rand512(), 7)[0], G["image_rand512_multi"])
    assert np.array_equal(oracle.image_hash_batch(oracle.image_synth(8, 512, 512, 0), 7)[0],
                          G["image_synth512_first8_multi"])


def test_oracle_text_golden(oracle):
    assert np.array_equal(oracle.text_minhash_batch(TEXTS)[0], G["text_minhash"])
    assert np.array_equal(oracle.text_simhash_batch(TEXTS)[0], G["text_simhash"])


def test_oracle_audio_golden(oracle):
    for secs in (1, 4):
        assert np.array_equal(oracle.wang(_sine(secs)), G[f"audio_sine440_{secs}s_wang"])
        assert np.array_equal(oracle.haitsma(_sine(secs), 8000), G[f"audio_sine440_{secs}s_haitsma"])
    assert np.array_equal(oracle.wang(G["audio_chirp_pcm"]), G["audio_chirp_wang"])
    assert np.array_equal(oracle.haitsma(G["audio_chirp_pcm"], 8000), G["audio_chirp_haitsma"])


def test_oracle_index_golden(oracle):
    hi, hd, _ = oracle.hamming_topk(G["hamming_ids"], G["hamming_codes"], G["hamming_q"], 10)
    assert np.array_equal(hi, G["hamming_top_ids"]) and np.array_equal(hd, G["hamming_top_d"])
    ci, cs = oracle.cosine_knn(np.arange(512, dtype=np.uint64), G["cosine_rows"], G["cosine_q"], 10)
    assert np.array_equal(ci, G["cosine_top_ids"]) and np.array_equal(cs, G["cosine_top_scores"])


# ------------------------------------------------------------------ GPU: HIP path vs golden
@pytest.mark.gpu
def test_hip_image_golden(gpu_ctx):
    from ucfp_amd import image
    for side in (64, 256):
        rec, st = image.fingerprint_frames(_png(side)[None], algo=image.MULTI, pixfmt=image.PIX_RGB8, ctx=gpu_ctx)
        assert st[0] == 0 and np.array_equal(rec[0], G[f"image_synthpng{side}_multi"])
    rec, _ = image.fingerprint_frames(_rand512(), algo=image.MULTI, ctx=gpu_ctx)
    assert np.array_equal(rec, G["image_rand512_multi"])


@pytest.mark.gpu
def test_hip_text_golden(gpu_ctx):
    from ucfp_amd import text
    assert np.array_equal(text._run("minhash", TEXTS, 0, 5, gpu_ctx)[0], G["text_minhash"])
    assert np.array_equal(text._run("simhash", TEXTS, 0, 5, gpu_ctx)[0], G["text_simhash"])


@pytest.mark.gpu
def test_hip_audio_golden(gpu_ctx):
    from ucfp_amd import audio
    for secs in (1, 4):
        assert np.array_equal(audio.wang_hashes(_sine(secs), 8000, ctx=gpu_ctx), G[f"audio_sine440_{secs}s_wang"])
        assert np.array_equal(audio.haitsma_frames(_sine(secs), 8000, ctx=gpu_ctx),
                              G[f"audio_sine440_{secs}s_haitsma"])
    assert np.array_equal(audio.wang_hashes(G["audio_chirp_pcm"], 8000, ctx=gpu_ctx), G["audio_chirp_wang"])
    assert np.array_equal(audio.haitsma_frames(G["audio_chirp_pcm"], 8000, ctx=gpu_ctx), G["audio_chirp_haitsma"])


@pytest.mark.gpu
def test_hip_index_golden(gpu_ctx):
    from ucfp_amd import index
    ix = index.DeviceIndex(index.HAMMING64, ctx=gpu_ctx)
    ix.upsert(0, G["hamming_ids"], G["hamming_codes"])
    gi, _, gd, _ = ix.search(0, G["hamming_q"], 10)
    assert np.array_equal(gi, G["hamming_top_ids"]) and np.array_equal(gd, G["hamming_top_d"])
    cx = index.DeviceIndex(index.COSINE_F32, 48, ctx=gpu_ctx)
    cx.upsert(0, np.arange(512, dtype=np.uint64), G["cosine_rows"])
    ci, cs, _, cc = cx.search(0, G["cosine_q"][None], 10)
    assert cc[0] == 10 and np.abs(cs[0] - G["cosine_top_scores"]).max() <= 1e-5   # north_star tolerance
    assert np.array_equal(ci[0], G["cosine_top_ids"])
