"""Text fingerprinting -- host-side mirror of src/modality/text.rs.

    fingerprint_minhash(text, tenant_id, record_id)                 text.rs:172-174
    fingerprint_minhash_with(text, opts, tenant_id, record_id)      text.rs:182-236  (H = 128)
    fingerprint_simhash_tf / fingerprint_simhash_idf                text.rs:328-362
    fingerprint_lsh                                                 text.rs:437-446

plus the batched form (`minhash_batch` / `simhash_batch`).  Hashing runs in the HIP library.
ASCII documents go to the GPU raw (it lower-cases and segments them); a document with non-ASCII
characters is canonicalised (NFKC + case fold + Bidi/Cf stripping, text.rs:112-114) and segmented
(UAX#29 via the `regex` module) here on the host, then submitted pre-tokenised -- Unicode tables are
host business (SURVEY "hard parts").
"""
import ctypes as C
import unicodedata
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .core import Modality, Record
from .errors import ModalityError, UnsupportedError

DEFAULT_K = 5        # text.rs:39
DEFAULT_H = 128      # text.rs:41
ALGORITHM_MINHASH_128 = "minhash-h128"
ALGORITHM_SIMHASH_TF = "simhash-b64-tf"
ALGORITHM_SIMHASH_IDF = "simhash-b64-idf"
ALGORITHM_LSH = "minhash-lsh-h128"
FORMAT_VERSION = 1   # txtfp::FORMAT_VERSION as stored by text.rs:227
# MinHash / LSH records: the slot derivation (DESIGN T5) is KNOWN not to be txtfp 0.2.0's (the reference's
# golden slot 0, src/server/tests.rs:1153-1157, is not reproduced), so these records carry their own
# format_version: a corpus mixing them with upstream-produced `minhash-h128` records is then rejected as
# incompatible instead of yielding meaningless Jaccard estimates.  Layout, schema word and tag are upstream's.
FORMAT_VERSION_MINHASH_HIP = 0x48500001

RAW_ASCII, PRETOKENIZED = 0, 1
NEEDS_HOST = 1
MINHASH_BYTES, SIMHASH_BYTES = 1032, 8

# txtfp::config_hash is not available offline and could not be reconstructed (DESIGN section 2 lists what was
# tried).  The ONE value the reference's tests show (src/server/tests.rs:1158-1161: default canonicalizer,
# "shingle-k=5/word-uax29", "minhash-h128") is CARRIED here as a constant -- it is not computed, so it pins nothing.
_CARRIED_CONFIG_HASH = {("nfkc", True, True, True, "shingle-k=5/word-uax29", ALGORITHM_MINHASH_128):
                        2_212_816_233_060_047_056}


@dataclass
class Canonicalizer:
    """txtfp::Canonicalizer knobs as exposed by handlers.rs:547-586."""
    normalization: str = "nfkc"     # nfc | nfkc | none
    case_fold: bool = True
    strip_bidi: bool = True
    strip_format: bool = True

    def apply(self, s: str) -> str:
        if self.normalization == "nfkc":
            s = unicodedata.normalize("NFKC", s)
        elif self.normalization == "nfc":
            s = unicodedata.normalize("NFC", s)
        if self.case_fold:
            s = s.casefold()
            if self.normalization in ("nfkc", "nfc"):
                s = unicodedata.normalize(self.normalization.upper(), s)
        if self.strip_bidi or self.strip_format:
            s = "".join(ch for ch in s if unicodedata.category(ch) != "Cf")
        return s

    def is_default(self) -> bool:
        return self == Canonicalizer()


@dataclass
class TextOpts:
    """text.rs:116-147."""
    canonicalizer: Canonicalizer = field(default_factory=Canonicalizer)
    tokenizer: str = "word"          # word | grapheme | cjk-jp | cjk-ko (text.rs:71-82)
    k: int = DEFAULT_K
    h: int = DEFAULT_H
    preprocess: Optional[str] = None  # html | markdown | pdf: not on the hot path

    def tokenizer_tag(self) -> str:   # text.rs:152-159
        return {"word": f"shingle-k={self.k}/word-uax29", "grapheme": f"shingle-k={self.k}/grapheme-uax29",
                "cjk-jp": f"shingle-k={self.k}/cjk-jp", "cjk-ko": f"shingle-k={self.k}/cjk-ko"}[self.tokenizer]


def config_hash(canon: Canonicalizer, tokenizer_tag: str, algorithm: str) -> int:
    """txtfp::config_hash (text.rs:221): only the default MinHash configuration's value is known (carried as a
    constant from the reference's test); for any other configuration the reference's value cannot be produced,
    so this raises instead of inventing one."""
    key = (canon.normalization, canon.case_fold, canon.strip_bidi, canon.strip_format, tokenizer_tag, algorithm)
    if key in _CARRIED_CONFIG_HASH:
        return _CARRIED_CONFIG_HASH[key]
    raise UnsupportedError(f"txtfp::config_hash of {key} is unknown: only the default MinHash configuration's value "
                           "is carried (src/server/tests.rs:1158-1161)")


def _host_tokens(s: str) -> List[str]:
    import regex  # UAX#29 default word boundaries
    return [t for t in regex.split(r"(?w)\b", s, flags=regex.V1) if any(ch.isalnum() for ch in t)]


def _prepare(text: str, opts: TextOpts) -> Tuple[bytes, int]:
    """-> (bytes for the GPU, mode)."""
    if opts.tokenizer != "word":
        raise UnsupportedError(f"tokenizer `{opts.tokenizer}` is not built into the HIP path")
    c = opts.canonicalizer
    if text.isascii() and c.case_fold:
        return text.encode("ascii"), RAW_ASCII
    toks = _host_tokens(c.apply(text))
    return " ".join(toks).encode("utf-8"), PRETOKENIZED


def _pack(docs: Sequence[bytes]):
    offs = np.zeros(len(docs) + 1, np.uint64)
    np.cumsum([len(d) for d in docs], out=offs[1:])
    blob = np.frombuffer(b"".join(docs) + b"\0" * 16, np.uint8)
    return blob, offs


def _run(kind: str, docs: Sequence[bytes], mode: int, k: int, ctx=None):
    ctx = ctx or _lib.current_context()
    lib = _lib.load()
    blob, offs = _pack(docs)
    n = len(docs)
    rec = SIMHASH_BYTES if kind == "simhash" else MINHASH_BYTES
    out = np.zeros((n, rec), np.uint8)
    status = np.zeros(n, np.int32)
    if kind == "simhash":
        _lib.check(lib.ucfp_text_simhash_batch(ctx.handle, blob.ctypes.data, offs.ctypes.data, n, mode,
                                               out.ctypes.data, status.ctypes.data))
    else:
        _lib.check(lib.ucfp_text_minhash_batch(ctx.handle, blob.ctypes.data, offs.ctypes.data, n, mode, k,
                                               out.ctypes.data, status.ctypes.data))
    return out, status


def _batch(kind: str, texts: Sequence[str], opts: TextOpts, ctx=None):
    """Split into the raw-ASCII and the host-pretokenised group, one launch each."""
    prepared = [_prepare(t, opts) for t in texts]
    rec = SIMHASH_BYTES if kind == "simhash" else MINHASH_BYTES
    out = np.zeros((len(texts), rec), np.uint8)
    status = np.zeros(len(texts), np.int32)
    for mode in (RAW_ASCII, PRETOKENIZED):
        idx = [i for i, (_, m) in enumerate(prepared) if m == mode]
        if not idx:
            continue
        o, s = _run(kind, [prepared[i][0] for i in idx], mode, opts.k, ctx)
        out[idx] = o
        status[idx] = s
    return out, status


def minhash_batch(texts: Sequence[str], opts: Optional[TextOpts] = None, ctx=None):
    """-> (records uint8 [n, 1032], status int32 [n])."""
    return _batch("minhash", texts, opts or TextOpts(), ctx)


def simhash_batch(texts: Sequence[str], opts: Optional[TextOpts] = None, ctx=None):
    return _batch("simhash", texts, opts or TextOpts(), ctx)


ALGO_MINHASH, ALGO_SIMHASH = 1, 2


class TextBatcher:
    """Host micro-batcher (handlers.rs:304-460 fingerprints one document per request): many request
    threads call `submit` concurrently, the library packs them into one GPU launch.  Two C batchers sit behind this
    object -- one for raw ASCII documents, one for the documents the host had to canonicalise and tokenise."""

    def __init__(self, kind: str = "minhash", opts: Optional[TextOpts] = None, *, max_batch: int = 4096,
                 max_bytes: int = 8 << 20, max_delay_us: int = 200, ctx=None):
        if kind not in ("minhash", "simhash"):
            raise UnsupportedError(f"text batcher kind `{kind}`")
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.opts = opts or TextOpts()
        self.kind = kind
        self.rec = MINHASH_BYTES if kind == "minhash" else SIMHASH_BYTES
        algo = ALGO_MINHASH if kind == "minhash" else ALGO_SIMHASH
        self._handles = {}
        for mode in (RAW_ASCII, PRETOKENIZED):
            h = C.c_void_p()
            _lib.check(self._lib.ucfp_text_batcher_create(self.ctx.handle, algo, mode, self.opts.k, max_batch, max_bytes,
                                                          max_delay_us, C.byref(h)))
            self._handles[mode] = h

    def submit(self, text: str):
        """-> (record bytes, status).  Blocks until this document's record is ready."""
        doc, mode = _prepare(text, self.opts)
        out = (C.c_uint8 * self.rec)()
        st = C.c_int32(0)
        _lib.check(self._lib.ucfp_text_batcher_submit(self._handles[mode], doc, len(doc), out, C.byref(st)))
        return bytes(out), int(st.value)

    def stats(self):
        """-> (launches, documents) summed over both modes."""
        tb = ti = 0
        for h in self._handles.values():
            b, i = C.c_uint64(0), C.c_uint64(0)
            _lib.check(self._lib.ucfp_text_batcher_stats(h, C.byref(b), C.byref(i)))
            tb, ti = tb + int(b.value), ti + int(i.value)
        return tb, ti

    def close(self):
        for h in getattr(self, "_handles", {}).values():
            self._lib.ucfp_text_batcher_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _raise_for(status: int):
    if status == -1:
        raise ModalityError("text has no tokens after canonicalisation")
    if status == -2:
        raise UnsupportedError("a token (or a run of fewer than k tokens) exceeds the ~1.4 KiB LDS batch")
    if status != 0:
        raise ModalityError(f"text fingerprint failed with status {status}")


CONFIG_HASH_UNKNOWN = 0


def _record_config_hash(opts: TextOpts, tokenizer_tag: str, algorithm: str, supplied: Optional[int]) -> int:
    """`config_hash` of a record.  In the drop-in the Rust host keeps calling txtfp::config_hash itself (a host-side
    function of the configuration, text.rs:221,406 -- not on the hot path) and passes the value in (`supplied`).
    Without it only the carried default-MinHash constant is known; every other configuration gets
    CONFIG_HASH_UNKNOWN (0), never an invented value."""
    if supplied is not None:
        return int(supplied)
    try:
        return config_hash(opts.canonicalizer, tokenizer_tag, algorithm)
    except UnsupportedError:
        return CONFIG_HASH_UNKNOWN


def fingerprint_minhash(text: str, tenant_id: int, record_id: int) -> Record:
    return fingerprint_minhash_with(text, TextOpts(), tenant_id, record_id)


def fingerprint_minhash_with(text: str, opts: TextOpts, tenant_id: int, record_id: int,
                             config_hash_value: Optional[int] = None) -> Record:
    if opts.h != DEFAULT_H:
        raise UnsupportedError("only H = 128 is built (the reference's public entry point, text.rs:172-174)")
    recs, status = minhash_batch([text], opts)
    _raise_for(int(status[0]))
    return Record(tenant_id=tenant_id, record_id=record_id, modality=Modality.Text,
                  format_version=FORMAT_VERSION_MINHASH_HIP, algorithm=ALGORITHM_MINHASH_128,
                  config_hash=_record_config_hash(opts, opts.tokenizer_tag(), ALGORITHM_MINHASH_128, config_hash_value),
                  fingerprint=recs[0].tobytes(), embedding=None, model_id=None, metadata=b"", text=text)


def _simhash(text: str, opts: TextOpts, tag: str, tenant_id: int, record_id: int,
             config_hash_value: Optional[int] = None) -> Record:
    recs, status = simhash_batch([text], opts)
    _raise_for(int(status[0]))
    tok_tag = {"word": "word-uax29", "grapheme": "grapheme-uax29", "cjk-jp": "cjk-jp", "cjk-ko": "cjk-ko"}[opts.tokenizer]
    return Record(tenant_id=tenant_id, record_id=record_id, modality=Modality.Text, format_version=FORMAT_VERSION,
                  algorithm=tag, config_hash=_record_config_hash(opts, tok_tag, tag, config_hash_value),
                  fingerprint=recs[0].tobytes(), embedding=None, model_id=None, metadata=b"", text=text)


def fingerprint_simhash_tf(text: str, opts: TextOpts, tenant_id: int, record_id: int,
                           config_hash_value: Optional[int] = None) -> Record:
    return _simhash(text, opts, ALGORITHM_SIMHASH_TF, tenant_id, record_id, config_hash_value)


def fingerprint_simhash_idf(text: str, opts: TextOpts, idf, tenant_id: int, record_id: int,
                            config_hash_value: Optional[int] = None) -> Record:
    """The server always passes IdfTable::default() (handlers.rs:410): an empty table weights every
    token 1.0, i.e. TF weighting; a non-empty table is not supported on the HIP path."""
    if idf:
        raise UnsupportedError("non-empty IdfTable is not built into the HIP path")
    return _simhash(text, opts, ALGORITHM_SIMHASH_IDF, tenant_id, record_id, config_hash_value)


def fingerprint_lsh(text: str, opts: TextOpts, tenant_id: int, record_id: int,
                    config_hash_value: Optional[int] = None) -> Record:
    rec = fingerprint_minhash_with(text, opts, tenant_id, record_id, config_hash_value)
    rec.algorithm = ALGORITHM_LSH
    return rec


def _dev(a: np.ndarray):
    """Host array -> device tensor (torch is memory plumbing only)."""
    import torch
    if not torch.cuda.is_available():
        raise UnsupportedError("no HIP device: the LSH index only exists on the GPU")
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a if a.flags.writeable else a.copy()).cuda()


def _records(records) -> np.ndarray:
    if isinstance(records, (bytes, bytearray)):
        records = np.frombuffer(bytes(records), np.uint8)
    r = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, MINHASH_BYTES)
    return r


def lsh_band_keys(records, bands: int = 16, rows: int = 8, ctx=None) -> np.ndarray:
    """Band keys of MinHash-128 records (the reference has no band index).
    -> uint64 [n, bands].  Key spec: DESIGN.md "LSH" (slot-wise FNV fold + splitmix64 finaliser);
    computed by ucfp_text_lsh_band_keys_dev."""
    import torch
    ctx = ctx or _lib.current_context()
    r = _records(records)
    n = r.shape[0]
    d_r = _dev(r)
    d_k = torch.empty((bands, max(n, 1)), dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().ucfp_text_lsh_band_keys_dev(ctx.handle, d_r.data_ptr(), n, bands, rows, d_k.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream or None))
    return d_k[:, :n].t().contiguous().cpu().numpy().view(np.uint64)


class LshIndex:
    """Banded MinHash LSH shard on the GPU: `build` sorts (band key, row) per band, `query` returns
    the best k candidates by slot agreement (the MinHash Jaccard estimate)."""

    def __init__(self, bands: int = 16, rows: int = 8, cand_per_band: int = 64, ctx=None):
        self._lib = _lib.load()
        self.ctx = ctx or _lib.current_context()
        self.bands, self.rows, self.cand_per_band = bands, rows, cand_per_band
        h = C.c_void_p()
        _lib.check(self._lib.ucfp_lsh_create(self.ctx.handle, bands, rows, cand_per_band, C.byref(h)))
        self.handle = h
        self.n = 0

    def build_dev(self, ids_ptr: int, records_ptr: int, n: int, stream: int = 0) -> None:
        _lib.check(self._lib.ucfp_lsh_build_dev(self.handle, ids_ptr or None, records_ptr or None, n, stream or None))
        self.n = n

    def build(self, ids, records) -> None:
        import torch
        ids = np.ascontiguousarray(ids, dtype=np.uint64).reshape(-1)
        r = _records(records)
        if r.shape[0] != ids.shape[0]:
            raise ModalityError("ids and records disagree on the number of rows")
        if ids.shape[0] == 0:
            self.build_dev(0, 0, 0)
            return
        d_ids, d_r = _dev(ids.view(np.int64)), _dev(r)
        self.build_dev(d_ids.data_ptr(), d_r.data_ptr(), ids.shape[0], torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().synchronize()   # the build copies what it needs; inputs may go now

    def query_dev(self, records_ptr: int, nq: int, k: int, out_ids_ptr: int, out_scores_ptr: int,
                  out_counts_ptr: int, stream: int = 0) -> None:
        _lib.check(self._lib.ucfp_lsh_query_dev(self.handle, records_ptr, nq, k, out_ids_ptr, out_scores_ptr,
                                                out_counts_ptr, stream or None))

    def query(self, records, k: int = 10):
        """-> (ids uint64 [nq, k] (INVALID = 2^64-1), scores float32 [nq, k], counts uint32 [nq])."""
        import torch
        r = _records(records)
        nq = r.shape[0]
        d_r = _dev(r)
        o_ids = torch.empty((max(nq, 1), k), dtype=torch.int64, device="cuda")
        o_sc = torch.empty((max(nq, 1), k), dtype=torch.float32, device="cuda")
        o_ct = torch.empty(max(nq, 1), dtype=torch.int32, device="cuda")
        self.query_dev(d_r.data_ptr(), nq, k, o_ids.data_ptr(), o_sc.data_ptr(), o_ct.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
        return (o_ids[:nq].cpu().numpy().view(np.uint64), o_sc[:nq].cpu().numpy(),
                o_ct[:nq].cpu().numpy().view(np.uint32))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ucfp_lsh_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
