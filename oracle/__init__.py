"""CPU oracle for the UCFP hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (ucfp_amd/) must never import it: a product path that routes
through the oracle voids every parity claim.

PARITY STATUS: see the header of each ucfp_oracle_*.c.  The image / audio / text
arithmetic of the reference lives in un-vendored crates (imgfprint 0.4.1, audiofp 0.3.0,
txtfp 0.2.0) -> "parity unpinned" beyond the sizes/layouts the reference's tests hold;
cosine kNN is restated from in-tree code and is pinned by the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libucfp_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (no-op when the .so is newer than the sources)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    srcs.append(os.path.join(_HERE, "..", "include", "ucfp_dct32.h"))
    if not force and os.path.exists(_SO):
        so_m = os.path.getmtime(_SO)
        if all(os.path.getmtime(s) <= so_m for s in srcs if os.path.exists(s)):
            return _SO
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


_SO_NATIVE = os.path.join(_HERE, "_build", "libucfp_oracle_native.so")


def _cpu_tag() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def use_native() -> str:
    """Switch this process to the TIMED build (-O3 -march=native), compiled on THIS machine (a stamp file
    records the CPU it was built for; a library built elsewhere is rebuilt).  Same sources, same float results
    (-ffp-contract=off, no fast-math).  bench.py's cpu_baseline legs call this; tests use the portable build."""
    global _lib
    stamp = _SO_NATIVE + ".cpu"
    tag = _cpu_tag()
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    fresh = (os.path.exists(_SO_NATIVE) and os.path.exists(stamp) and open(stamp).read() == tag and
             all(os.path.getmtime(s) <= os.path.getmtime(_SO_NATIVE) for s in srcs))
    if not fresh:
        subprocess.run(["make", "-C", _HERE, "-B", "native"], check=True, capture_output=True)
        open(stamp, "w").write(tag)
    _lib = C.CDLL(_SO_NATIVE)
    _declare(_lib)
    return _SO_NATIVE


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def _declare(l):
    l.ucfp_oracle_image_hash_batch.restype = C.c_int
    l.ucfp_oracle_image_hash_batch.argtypes = [
        C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t,
        C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    l.ucfp_oracle_image_normalize.restype = None
    l.ucfp_oracle_image_normalize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t,
                                              C.c_int, C.c_void_p]
    l.ucfp_oracle_image_hashes17.restype = None
    l.ucfp_oracle_image_hashes17.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    l.ucfp_oracle_image_region_gray32.restype = None
    l.ucfp_oracle_image_region_gray32.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    l.ucfp_oracle_image_phash_coefs.restype = None
    l.ucfp_oracle_image_phash_coefs.argtypes = [C.c_void_p, C.c_void_p]
    l.ucfp_oracle_image_synth.restype = None
    l.ucfp_oracle_image_synth.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t]


def cosine_knn(ids, rows, query, k, ref_fold=False):
    """Cosine kNN with the reference's exact score arithmetic. Order: (score desc, id asc);
    ref_fold=True runs the reference's insert_topk as one sequential fold instead (its tie
    order differs, see ucfp_oracle_index.c). Returns (ids, scores)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    n = ids.shape[0]
    dim = query.shape[0]
    assert rows.size == n * dim
    out_ids = np.zeros(max(k, 1), np.uint64)
    out_sc = np.zeros(max(k, 1), np.float32)
    f = lib().ucfp_oracle_cosine_knn_ref_fold if ref_fold else lib().ucfp_oracle_cosine_knn
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                  C.c_void_p, C.c_void_p]
    m = f(ids.ctypes.data, rows.ctypes.data, n, dim, query.ctypes.data, k, out_ids.ctypes.data,
          out_sc.ctypes.data)
    return out_ids[:m].copy(), out_sc[:m].copy()


def cosine_knn_batch_omp(ids, rows, queries, k):
    """Timed-baseline form: nq queries, OpenMP over (row chunk, query) tiles with the reference's dot_product /
    insert_topk, merged in (score desc, id asc).  -> (ids u64 [nq,k], scores f32 [nq,k], counts u32 [nq])."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, rows.shape[1])
    nq = queries.shape[0]
    o_ids = np.zeros((nq, max(k, 1)), np.uint64)
    o_sc = np.zeros((nq, max(k, 1)), np.float32)
    o_c = np.zeros(nq, np.uint32)
    f = lib().ucfp_oracle_cosine_knn_batch_omp
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                  C.c_void_p, C.c_void_p]
    f(ids.ctypes.data, rows.ctypes.data, rows.shape[0], rows.shape[1], queries.ctypes.data, nq, k, o_ids.ctypes.data,
      o_sc.ctypes.data, o_c.ctypes.data)
    return o_ids[:, :k], o_sc[:, :k], o_c


def hamming_simd() -> bool:
    """True when the timed Hamming scan uses its AVX-512 VPOPCNTDQ tile on this CPU."""
    f = lib().ucfp_oracle_hamming_simd
    f.restype = C.c_int
    return bool(f())


def hamming_topk_omp(ids, codes, queries, k, force_scalar=False):
    """Timed-baseline form of hamming_topk: OpenMP over (corpus slice, query run) tiles, so one query uses every core;
    AVX-512 VPOPCNTDQ inner loop where the CPU has it (force_scalar keeps the scalar tile)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    queries = np.ascontiguousarray(queries, dtype=np.uint64).reshape(-1)
    nq = queries.shape[0]
    o_ids = np.zeros((nq, max(k, 1)), np.uint64)
    o_d = np.zeros((nq, max(k, 1)), np.uint32)
    o_c = np.zeros(nq, np.uint32)
    f = lib().ucfp_oracle_hamming_topk_omp2
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_int]
    f(ids.ctypes.data, codes.ctypes.data, codes.shape[0], queries.ctypes.data, nq, k, o_ids.ctypes.data, o_d.ctypes.data,
      o_c.ctypes.data, 1 if force_scalar else 0)
    return o_ids[:, :k], o_d[:, :k], o_c


def hamming_topk(ids, codes, queries, k):
    """Returns (ids [nq,k], dist [nq,k], counts [nq]); order (d asc, id asc)."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    queries = np.ascontiguousarray(queries, dtype=np.uint64)
    nq = queries.shape[0]
    out_ids = np.zeros((nq, k), np.uint64)
    out_d = np.zeros((nq, k), np.uint32)
    out_c = np.zeros(nq, np.uint32)
    f = lib().ucfp_oracle_hamming_topk
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                  C.c_void_p, C.c_void_p, C.c_void_p]
    f(ids.ctypes.data, codes.ctypes.data, ids.shape[0], queries.ctypes.data, nq, k,
      out_ids.ctypes.data, out_d.ctypes.data, out_c.ctypes.data)
    return out_ids, out_d, out_c


def xxh3_64(data: bytes) -> int:
    f = lib().ucfp_oracle_xxh3_64
    f.restype = C.c_uint64
    f.argtypes = [C.c_char_p, C.c_size_t]
    return int(f(data, len(data)))


def text_canon(text: bytes, mode: int = 0):
    """Canonical token stream (tokens joined by one space). Returns (stream bytes, n_tokens);
    n_tokens = -1 for non-ASCII input in raw mode."""
    out = C.create_string_buffer(len(text) + 2)
    olen = C.c_size_t(0)
    f = lib().ucfp_oracle_text_canon
    f.restype = C.c_long
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    nt = f(text, len(text), mode, out, C.byref(olen))
    return out.raw[:olen.value], int(nt)


def _pack_docs(docs):
    offs = np.zeros(len(docs) + 1, np.uint64)
    for i, d in enumerate(docs):
        offs[i + 1] = offs[i] + len(d)
    blob = np.frombuffer(b"".join(docs) + b"\0", np.uint8).copy()
    return blob, offs


def text_minhash_batch(docs, mode: int = 0, k: int = 5):
    """docs: list of bytes. Returns (records [n,1032] u8, status [n] i32)."""
    blob, offs = _pack_docs(docs)
    n = len(docs)
    out = np.zeros((n, 1032), np.uint8)
    st = np.zeros(n, np.int32)
    f = lib().ucfp_oracle_text_minhash_batch
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    f(blob.ctypes.data, offs.ctypes.data, n, mode, k, out.ctypes.data, st.ctypes.data)
    return out, st


def text_simhash_batch(docs, mode: int = 0):
    blob, offs = _pack_docs(docs)
    n = len(docs)
    out = np.zeros((n, 8), np.uint8)
    st = np.zeros(n, np.int32)
    f = lib().ucfp_oracle_text_simhash_batch
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    f(blob.ctypes.data, offs.ctypes.data, n, mode, out.ctypes.data, st.ctypes.data)
    return out, st


class WangCfg(C.Structure):
    """audiofp::classical::WangConfig; defaults src/server/algorithms_manifest.rs:553-592."""
    _fields_ = [("fan_out", C.c_uint32), ("target_zone_t", C.c_uint32), ("target_zone_f", C.c_uint32),
                ("peaks_per_sec", C.c_uint32), ("min_anchor_mag_db", C.c_float)]

    @classmethod
    def default(cls):
        return cls(10, 63, 64, 30, -50.0)


def resample_linear(x, sr_in: int, sr_out: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    l = lib()
    l.ucfp_oracle_resample_len.restype = C.c_size_t
    l.ucfp_oracle_resample_len.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
    m = l.ucfp_oracle_resample_len(x.size, sr_in, sr_out)
    out = np.zeros(m, np.float32)
    l.ucfp_oracle_resample_linear.restype = None
    l.ucfp_oracle_resample_linear.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p]
    if m:
        l.ucfp_oracle_resample_linear(x.ctypes.data, x.size, sr_in, sr_out, out.ctypes.data)
    return out


def stft_power(x, n_fft: int, hop: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    T = 1 + (x.size - n_fft) // hop if x.size >= n_fft else 0
    P = np.zeros((T, n_fft // 2), np.float32)
    f = lib().ucfp_oracle_stft_power
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    if T:
        f(x.ctypes.data, x.size, n_fft, hop, P.ctypes.data)
    return P


def wang_peaks(P, peaks_per_sec: int = 30):
    P = np.ascontiguousarray(P, dtype=np.float32)
    T = P.shape[0]
    f = lib().ucfp_oracle_wang_peaks
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    n = f(P.ctypes.data, T, peaks_per_sec, None, None, None)
    t = np.zeros(n, np.uint32)
    k = np.zeros(n, np.uint32)
    p = np.zeros(n, np.float32)
    if n:
        f(P.ctypes.data, T, peaks_per_sec, t.ctypes.data, k.ctypes.data, p.ctypes.data)
    return t, k, p


def wang(x, cfg=None, cap: int = 0) -> np.ndarray:
    """8 kHz mono f32 -> WangHash array [n, 2] u32 (hash, t_anchor)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    cfg = cfg or WangCfg.default()
    cap = cap or max(64, (x.size // 8000 + 2) * cfg.peaks_per_sec * cfg.fan_out)
    out = np.zeros((cap, 2), np.uint32)
    f = lib().ucfp_oracle_wang
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(WangCfg), C.c_void_p, C.c_size_t]
    n = f(x.ctypes.data, x.size, C.byref(cfg), out.ctypes.data, cap)
    assert n <= cap
    return out[:n].copy()


def haitsma(x, sample_rate: int, fmin: float = 300.0, fmax: float = 2000.0) -> np.ndarray:
    """mono f32 at any rate (linear resample to 5 kHz like src/modality/audio.rs:194-200) -> u32 frames."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if sample_rate != 5000:
        x = resample_linear(x, sample_rate, 5000)
    T = 1 + (x.size - 2048) // 64 if x.size >= 2048 else 0
    out = np.zeros(T, np.uint32)
    f = lib().ucfp_oracle_haitsma_5k
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]
    if T:
        f(x.ctypes.data, x.size, fmin, fmax, out.ctypes.data)
    return out


def num_threads() -> int:
    return int(lib().ucfp_oracle_num_threads())


def set_threads(n: int) -> None:
    lib().ucfp_oracle_set_threads(int(n))


ALGO = {"ahash": 1, "phash": 2, "dhash": 4, "multi": 7}
_BPP = {0: 1, 1: 3, 2: 4}


def image_hash_batch(frames: np.ndarray, algo: int, pixfmt: int = 0, exact=None,
                     min_dim: int = 32, max_dim: int = 8192):
    """frames: uint8 [n, h, w] (GRAY8) or [n, h, w, 3|4]. Returns (records [n, rec] u8, status [n] i32)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, h, w = frames.shape[:3]
    bpp = _BPP[pixfmt]
    assert frames.size == n * h * w * bpp
    rec = 536 if algo == 7 else 168
    out = np.zeros((n, rec), np.uint8)
    status = np.zeros(n, np.int32)
    ex = None
    if exact is not None:
        ex = np.ascontiguousarray(exact, dtype=np.uint8)
        assert ex.shape == (n, 32)
    lib().ucfp_oracle_image_hash_batch(
        algo, frames.ctypes.data, n, w, h, w * bpp, h * w * bpp, pixfmt, min_dim, max_dim,
        ex.ctypes.data if ex is not None else None, out.ctypes.data, status.ctypes.data)
    return out, status


PNG_OK, PNG_NEEDS_HOST, PNG_CORRUPT = 0, 1, -1


def inflate(z: bytes, cap: int):
    """zlib stream -> (status, bytes) by the restated RFC 1950/1951 decoder (ucfp_oracle_png.c)."""
    f = lib().ucfp_oracle_inflate
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    out = np.zeros(max(cap, 1), np.uint8)
    n = C.c_size_t(0)
    rc = f(z, len(z), out.ctypes.data, cap, C.byref(n))
    return rc, out[: n.value].tobytes()


def png_probe(png: bytes):
    """-> (status, width, height, pixfmt)."""
    f = lib().ucfp_oracle_png_probe
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    w, h, fmt = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
    rc = f(png, len(png), C.byref(w), C.byref(h), C.byref(fmt))
    return rc, w.value, h.value, fmt.value


def png_decode(png: bytes):
    """-> (status, pixels uint8 [h, w] or [h, w, c] (None unless status == PNG_OK))."""
    rc, w, h, fmt = png_probe(png)
    if rc != PNG_OK:
        return rc, None
    c = _BPP[fmt]
    px = np.zeros((h, w) if c == 1 else (h, w, c), np.uint8)
    f = lib().ucfp_oracle_png_decode
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rc = f(png, len(png), px.ctypes.data, px.size)
    return rc, (px if rc == PNG_OK else None)


JPG_OK, JPG_NEEDS_HOST, JPG_CORRUPT = 0, 1, -1


def jpeg_probe(jpg: bytes):
    """-> (status, width, height)."""
    f = lib().ucfp_oracle_jpeg_probe
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    w, h = C.c_uint32(0), C.c_uint32(0)
    rc = f(jpg, len(jpg), C.byref(w), C.byref(h))
    return rc, w.value, h.value


def jpeg_decode_luma(jpg: bytes):
    """-> (status, luma uint8 [h, w] (None unless status == JPG_OK)): the file's Y component, accurate integer IDCT."""
    rc, w, h = jpeg_probe(jpg)
    if rc != JPG_OK:
        return rc, None
    px = np.zeros((h, w), np.uint8)
    f = lib().ucfp_oracle_jpeg_decode_luma
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rc = f(jpg, len(jpg), px.ctypes.data, px.size)
    return rc, (px if rc == JPG_OK else None)


def image_normalize(frame: np.ndarray, pixfmt: int = 0) -> np.ndarray:
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape[:2]
    norm = np.zeros((256, 256), np.uint8)
    lib().ucfp_oracle_image_normalize(frame.ctypes.data, w, h, w * _BPP[pixfmt], pixfmt,
                                      norm.ctypes.data)
    return norm


def image_hashes17(norm: np.ndarray, which: int) -> np.ndarray:
    norm = np.ascontiguousarray(norm, dtype=np.uint8)
    hs = np.zeros(17, np.uint64)
    lib().ucfp_oracle_image_hashes17(norm.ctypes.data, which, hs.ctypes.data)
    return hs


def image_region_gray32(norm: np.ndarray, r: int) -> np.ndarray:
    norm = np.ascontiguousarray(norm, dtype=np.uint8)
    g = np.zeros((32, 32), np.uint8)
    lib().ucfp_oracle_image_region_gray32(norm.ctypes.data, r, g.ctypes.data)
    return g


def image_phash_coefs(g32: np.ndarray) -> np.ndarray:
    g32 = np.ascontiguousarray(g32, dtype=np.uint8)
    co = np.zeros(64, np.float32)
    lib().ucfp_oracle_image_phash_coefs(g32.ctypes.data, co.ctypes.data)
    return co.reshape(8, 8)


def image_synth(n: int, w: int, h: int, first: int = 0) -> np.ndarray:
    out = np.zeros((n, h, w), np.uint8)
    lib().ucfp_oracle_image_synth(out.ctypes.data, n, w, h, first)
    return out


# ------------------------------------------------------------------------------------------
# banded MinHash LSH (DESIGN.md "LSH"; the reference has no band index)
# ------------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


def _records_slots(records) -> np.ndarray:
    r = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, 1032)
    return np.ascontiguousarray(r[:, 8:]).view("<u8").reshape(-1, 128)


def lsh_band_keys(records, bands: int = 16, rows: int = 8) -> np.ndarray:
    """uint64 [n, bands]: h = FNV offset; h = (h ^ slot) * FNV prime per slot; splitmix64 finaliser."""
    slots = _records_slots(records)
    n = slots.shape[0]
    with np.errstate(over="ignore"):
        out = np.empty((n, bands), np.uint64)
        for b in range(bands):
            h = np.full(n, 0xCBF29CE484222325, np.uint64)
            for r in range(rows):
                h = (h ^ slots[:, b * rows + r]) * np.uint64(0x100000001B3)
            h ^= h >> np.uint64(30)
            h *= np.uint64(0xBF58476D1CE4E5B9)
            h ^= h >> np.uint64(27)
            h *= np.uint64(0x94D049BB133111EB)
            h ^= h >> np.uint64(31)
            out[:, b] = h
    return out


def lsh_query(ids, records, queries, k: int, bands: int = 16, rows: int = 8, cand_per_band: int = 64,
              max_cand: int = 1024):
    """Candidate list = for band 0..bands-1 the first `cand_per_band` corpus rows (ascending row) whose
    band key equals the query's, the concatenation cut at `max_cand`; duplicates dropped; score =
    equal slots / 128; best k by (score desc, id asc).  -> (ids [nq,k], scores [nq,k], counts [nq])."""
    ids = np.asarray(ids, np.uint64)
    cs, qs = _records_slots(records), _records_slots(queries)
    ck, qk = lsh_band_keys(records, bands, rows), lsh_band_keys(queries, bands, rows)
    nq = qs.shape[0]
    o_ids = np.full((nq, k), _M64, np.uint64)
    o_sc = np.full((nq, k), -1.0, np.float32)
    o_ct = np.zeros(nq, np.uint32)
    order = [np.argsort(ck[:, b], kind="stable") for b in range(bands)] if cs.shape[0] else []
    skeys = [ck[order[b], b] for b in range(bands)] if cs.shape[0] else []
    for q in range(nq):
        cand = []
        for b in range(bands if cs.shape[0] else 0):
            lo = np.searchsorted(skeys[b], qk[q, b], "left")
            hi = np.searchsorted(skeys[b], qk[q, b], "right")
            cand.extend(order[b][lo:min(hi, lo + cand_per_band)].tolist())
        cand = cand[:max_cand]
        seen, uniq = set(), []
        for r in cand:
            if r not in seen:
                seen.add(r)
                uniq.append(r)
        scored = sorted(((-int((cs[r] == qs[q]).sum()), int(ids[r])) for r in uniq))[:k]
        o_ct[q] = len(scored)
        for j, (na, i) in enumerate(scored):
            o_ids[q, j] = i
            o_sc[q, j] = np.float32(-na) * np.float32(1.0 / 128.0)
    return o_ids, o_sc, o_ct
