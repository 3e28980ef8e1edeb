"""GPU parity: HIP MinHash-128 / SimHash-64 (through the C ABI) vs the CPU oracle, bit-exact."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORDS = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "don't", "U.S.A.", "3.14", "1,000",
         "e.g.", "x:y", "foo_bar", "a", "I", "HELLO", "World", "42", "ab12cd", "o'clock"]
SEPS = [" ", "  ", ", ", ". ", "\n", "\t", " - ", "; ", "! ", "? ", " (", ") ", "/", "\"", " ... "]


def _doc(rng, nbytes):
    out = []
    size = 0
    while size < nbytes:
        w = rng.choice(WORDS) if rng.random() < 0.7 else "".join(rng.choice("abcdefghijklmnopqrstuvwxyzABC0123456789")
                                                              for _ in range(rng.randint(1, 12)))
        s = rng.choice(SEPS)
        out.append(w + s)
        size += len(w) + len(s)
    return "".join(out)[:nbytes]


def _gpu(kind, docs, mode=0, k=5):
    from ucfp_amd import text
    return text._run(kind, docs, mode, k)


@pytest.mark.parametrize("kind", ["minhash", "simhash"])
def test_reference_inputs(gpu_ctx, oracle, kind):
    docs = [b"the quick brown fox jumps over the lazy dog",
            b"the quick brown fox jumps over the lazy dog. " * 128,       # benches/end_to_end.rs:24-38
            b"Hello world, this is a test of the pipeline inspector.",
            b"one", b"one two", b"one two three four", b"a b c d e", b"a b c d e f"]
    fo = oracle.text_minhash_batch if kind == "minhash" else oracle.text_simhash_batch
    g, gs = _gpu(kind, docs)
    o, os_ = fo(docs)
    assert np.array_equal(gs, os_) and not gs.any()
    assert np.array_equal(g, o)
    if kind == "minhash":
        assert g.shape[1] == 1032 and bytes(g[0, :8]) == b"\x01" + b"\x00" * 7   # src/server/tests.rs:1114-1118


@pytest.mark.parametrize("kind", ["minhash", "simhash"])
@pytest.mark.parametrize("nbytes", [40, 300, 4095, 4096, 4097, 9000, 40000])
def test_random_docs_match_oracle(gpu_ctx, oracle, kind, nbytes):
    rng = random.Random(nbytes)
    docs = [_doc(rng, rng.randint(max(1, nbytes - 37), nbytes)).encode() for _ in range(24)]
    fo = oracle.text_minhash_batch if kind == "minhash" else oracle.text_simhash_batch
    g, gs = _gpu(kind, docs)
    o, os_ = fo(docs)
    assert np.array_equal(gs, os_)
    bad = [i for i in range(len(docs)) if not np.array_equal(g[i], o[i])]
    assert not bad, f"{kind} {nbytes}: docs {bad[:5]} differ"


@pytest.mark.parametrize("k", [1, 2, 5, 9])
def test_shingle_width(gpu_ctx, oracle, k):
    rng = random.Random(k)
    docs = [_doc(rng, 700).encode() for _ in range(16)] + [b"only three tokens"]
    g, gs = _gpu("minhash", docs, k=k)
    o, os_ = oracle.text_minhash_batch(docs, k=k)
    assert np.array_equal(gs, os_) and np.array_equal(g, o)


def test_status_codes(gpu_ctx, oracle):
    docs = [b"", b"   ...  !!! ", b"caf\xc3\xa9 au lait", b"fine text here", b"x" * 5000 + b" tail"]
    g, gs = _gpu("minhash", docs)
    o, os_ = oracle.text_minhash_batch(docs)
    assert list(gs[:4]) == [-1, -1, 1, 0] == list(os_[:4])
    assert gs[4] == -2                       # one token longer than the tile: unsupported on the HIP path
    assert not g[0].any() and not g[1].any() and not g[2].any()
    assert np.array_equal(g[3], o[3])


def test_pretokenized_and_unicode_host_path(gpu_ctx, oracle):
    from ucfp_amd import text
    docs = ["Café au lait, naïve façade — STRASSE straße ﬁne",
            "你好 世界 hello world 123", "plain ascii goes raw"]
    recs, st = text.minhash_batch(docs)
    assert not st.any()
    for i, d in enumerate(docs):
        b, mode = text._prepare(d, text.TextOpts())
        o, os_ = oracle.text_minhash_batch([b], mode=mode)
        assert os_[0] == 0 and np.array_equal(recs[i], o[0])
    assert text._prepare(docs[2], text.TextOpts())[1] == text.RAW_ASCII
    assert text._prepare(docs[0], text.TextOpts())[1] == text.PRETOKENIZED
    # case and compatibility forms fold together on the host path
    a, _ = text.minhash_batch(["STRASSE ﬁne café one two three"])
    b, _ = text.minhash_batch(["strasse fine café one two three"])
    assert np.array_equal(a, b)


def test_record_adapters(gpu_ctx):
    from ucfp_amd import text
    from ucfp_amd.errors import ModalityError
    rec = text.fingerprint_minhash("the quick brown fox jumps over the lazy dog", 0, 1)
    assert rec.algorithm == "minhash-h128" and len(rec.fingerprint) == 1032
    assert rec.config_hash == 2_212_816_233_060_047_056        # src/server/tests.rs:1158-1161
    assert rec.text == "the quick brown fox jumps over the lazy dog" and rec.modality.name == "Text"
    lsh = text.fingerprint_lsh("the quick brown fox jumps over the lazy dog", text.TextOpts(), 0, 1)
    assert lsh.algorithm == "minhash-lsh-h128" and lsh.fingerprint == rec.fingerprint   # text.rs:437-446
    sh = text.fingerprint_simhash_tf("the quick brown fox jumps over the lazy dog", text.TextOpts(), 0, 1)
    assert sh.algorithm == "simhash-b64-tf" and len(sh.fingerprint) == 8
    with pytest.raises(ModalityError):
        text.fingerprint_minhash("   ", 0, 1)
    assert text.lsh_band_keys(rec.fingerprint).shape == (1, 16)


def test_minhash_estimates_jaccard(gpu_ctx):
    """Property at scale: slot agreement between near-duplicates tracks shingle Jaccard."""
    from ucfp_amd import text
    rng = random.Random(5)
    base = [rng.choice(WORDS[:8]) + str(rng.randint(0, 500)) for _ in range(600)]
    edited = list(base)
    for i in rng.sample(range(600), 30):
        edited[i] = "zzz" + str(i)
    recs, _ = text.minhash_batch([" ".join(base), " ".join(edited), " ".join(reversed(base))])
    slots = recs[:, 8:].copy().view(np.uint64)
    sh = lambda t: {" ".join(t[i:i + 5]) for i in range(len(t) - 4)}  # noqa: E731
    j = len(sh(base) & sh(edited)) / len(sh(base) | sh(edited))
    est = float((slots[0] == slots[1]).mean())
    assert abs(est - j) < 0.15, (est, j)
    assert (slots[0] == slots[2]).mean() < 0.1


@pytest.mark.parametrize("kind", ["minhash", "simhash"])
def test_micro_batcher_coalesces_concurrent_documents(gpu_ctx, oracle, kind):
    """48 threads each submit one document at a time (the reference's per-request shape,
    handlers.rs:304-460); every thread gets ITS record, bit-exact, from far fewer launches than documents.  ASCII and
    host-canonicalised documents are mixed: they flow through the two C batchers behind TextBatcher."""
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import text
    rng = random.Random(48)
    docs = [_doc(rng, rng.choice([30, 200, 900, 4096, 12000])) for _ in range(900)]
    for i in range(0, len(docs), 9):
        docs[i] = "Café naïve façade — " + docs[i] + " straße ﬁne 你好 世界"
    docs[5], docs[6] = "", "  ... !!"                    # no tokens: status -1 for that document only
    fo = oracle.text_minhash_batch if kind == "minhash" else oracle.text_simhash_batch
    ref, ref_st = [], []
    for d in docs:
        b, mode = text._prepare(d, text.TextOpts())
        o, s = fo([b], mode=mode)
        ref.append(o[0].tobytes())
        ref_st.append(int(s[0]))
    b = text.TextBatcher(kind, max_batch=128, max_bytes=256 << 10, max_delay_us=2000, ctx=gpu_ctx)
    try:
        with ThreadPoolExecutor(48) as pool:
            got = list(pool.map(b.submit, docs))
        for i, (rec, st) in enumerate(got):
            assert st == ref_st[i], (i, st, ref_st[i])
            if st == 0:
                assert rec == ref[i], i
        assert ref_st[5] == -1 and ref_st[6] == -1
        batches, items = b.stats()
        assert items == len(docs) and batches < len(docs) // 4, (batches, items)
        # a lone request is flushed by the deadline; one larger than the blob is refused, not truncated
        rec, st = b.submit(docs[1])
        assert st == 0 and rec == ref[1]
        from ucfp_amd.errors import UcfpError
        with pytest.raises(UcfpError):
            b.submit("word " * (60 << 10))
    finally:
        b.close()


def test_micro_batcher_byte_budget_closes_the_set(gpu_ctx, oracle):
    """Documents that together exceed max_bytes never share a flush: the set is closed early and the rest follow."""
    from concurrent.futures import ThreadPoolExecutor
    from ucfp_amd import text
    rng = random.Random(7)
    docs = [_doc(rng, 3000) for _ in range(64)]
    o, _ = oracle.text_minhash_batch([d.encode() for d in docs])
    b = text.TextBatcher("minhash", max_batch=1024, max_bytes=10_000, max_delay_us=50_000, ctx=gpu_ctx)
    try:
        with ThreadPoolExecutor(16) as pool:
            got = list(pool.map(b.submit, docs))
        assert all(st == 0 and rec == o[i].tobytes() for i, (rec, st) in enumerate(got))
        batches, items = b.stats()
        assert items == 64 and batches >= 64 // 3        # at most three 3000-byte documents fit one flush
    finally:
        b.close()
