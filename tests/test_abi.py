"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly the
symbols include/ucfp_hip.h declares; host-only entry points work; device entry points fail
loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from ucfp_amd import _lib, errors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "ucfp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ucfp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    syms = _header_symbols()
    assert len(syms) >= 8
    lib = _lib.load()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in ucfp_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in ucfp_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in ucfp_hip.h"


def test_abi_version_and_record_sizes():
    lib = _lib.load()
    assert lib.ucfp_abi_version() == 2
    # pinned by the reference: 536-B bundle (src/server/tests.rs:1206), 168-B single
    # (web/src/lib/components/charts/algorithmView.ts:11-17)
    assert lib.ucfp_image_record_bytes(7) == 536
    for a in (1, 2, 4):
        assert lib.ucfp_image_record_bytes(a) == 168
    assert lib.ucfp_image_record_bytes(3) == 0


def test_blake3_known_answers():
    from ucfp_amd.blake3 import blake3_digest
    # official BLAKE3 test vectors (input byte i = i % 251)
    assert blake3_digest(b"").hex() == \
        "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert blake3_digest(b"abc").hex() == \
        "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
    assert blake3_digest(bytes(i % 251 for i in range(1025))).hex() == \
        "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444"
    assert blake3_digest(bytes(i % 251 for i in range(31744))).hex().startswith(
        "62b6960e1a44bcc1eb1a611a8d6235b6")


def test_error_mapping_matches_reference_http_codes():
    # src/server/error.rs:24-34
    assert errors.ModalityError.http_status == 400
    assert errors.UnsupportedError.http_status == 501
    assert errors.IndexError_.http_status == 500
    assert errors.RecordNotFound.http_status == 404
    assert isinstance(errors.from_status(-1, "x"), errors.ModalityError)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present; the loud-failure path is exercised on CPU-only hosts")
    with pytest.raises(errors.UcfpError):
        _lib.Context(0)
    msg = _lib.load().ucfp_last_error().decode()
    assert "no CPU path" in msg or "HIP" in msg


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure; nothing under ucfp_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ucfp_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), (dp, fn)
                assert not re.search(r'#include\s*"[^"]*oracle', txt), (dp, fn)
                assert "libucfp_oracle" not in txt, (dp, fn)


def test_host_only_entry_points_need_no_gpu():
    """Sizing helpers and argument validation run on the host: usable (and testable) without a device."""
    lib = _lib.load()
    cfg = _lib.WangConfig(10, 63, 64, 30, -50.0)
    # 10 s at 8 kHz: 618 frames -> 10 one-second buckets -> 10 * 30 peaks * 10 targets
    assert lib.ucfp_audio_wang_max_hashes(80000, C.byref(cfg)) == 10 * 30 * 10
    assert lib.ucfp_audio_wang_max_hashes(1000, C.byref(cfg)) == 0          # shorter than one frame
    assert lib.ucfp_audio_resample_len(44100, 44100, 8000) == 8000
    assert lib.ucfp_audio_resample_len(44101, 44100, 5000) == 5000
    assert lib.ucfp_audio_haitsma_frames(5000 * 10, 5000) == 1 + (50000 - 2048) // 64
    assert lib.ucfp_audio_haitsma_frames(8000 * 10, 8000) == 1 + (50000 - 2048) // 64   # resampled to 5 kHz
    assert lib.ucfp_audio_haitsma_frames(100, 5000) == 0
    # NULL context / bad enums are rejected before any device call
    out = (C.c_uint8 * 536)()
    assert lib.ucfp_image_hash_batch(None, 7, out, 1, 64, 64, 64, 4096, 0, None, None, out, None) == -4
    assert b"ctx" in lib.ucfp_last_error()
    assert lib.ucfp_text_minhash_batch(None, None, None, 0, 0, 5, None, None) == -4
    n = C.c_size_t(0)
    assert lib.ucfp_audio_wang(None, None, 0, 8000, None, None, 0, C.byref(n)) == -4
    assert lib.ucfp_index_create(None, 1, 0, 0, C.byref(C.c_void_p())) == -4
    assert lib.ucfp_blake3(None, 5, out) == -4


def test_query_request_wire_format():
    """POST /v1/query body (src/server/dto.rs:74-87, handlers.rs:153): the reference's body parses unchanged;
    `hash` is the one additive field."""
    from ucfp_amd.core import Hit, HitSource, Modality, QueryRequest, hit_to_json
    from ucfp_amd.errors import InvalidArgument
    r = QueryRequest.from_json({"tenant_id": 7, "modality": "Image", "vector": [0.6, 0.6, 0]})
    assert (r.tenant_id, r.modality, r.k, r.vector, r.hash) == (7, Modality.Image, 10, [0.6, 0.6, 0.0], None)
    assert QueryRequest.from_json({"tenant_id": 1, "modality": "Text", "k": 0, "vector": [1]}).k == 1
    h = QueryRequest.from_json({"tenant_id": 1, "modality": "Image", "hash": list((0x1122334455667788).to_bytes(8, "little")),
                                "algorithm": "imgfprint-phash-v1"})
    assert h.hash == 0x1122334455667788 and h.vector is None and h.algorithm == "imgfprint-phash-v1"
    for bad in ({"tenant_id": 1, "modality": "Image"}, {"modality": "Image", "vector": [1]},
                {"tenant_id": 1, "modality": "image", "vector": [1]}, {"tenant_id": 1, "modality": "Image", "hash": [1, 2]}):
        with pytest.raises(InvalidArgument):
            QueryRequest.from_json(bad)
    out = hit_to_json(Hit(tenant_id=1, record_id=9, score=0.5, source=HitSource.Vector, vector_score=0.5, vector_rank=1))
    assert list(out) == ["tenant_id", "record_id", "score", "source", "vector_score", "bm25_score", "vector_rank",
                         "bm25_rank", "term_hits"]                      # HitOut field order, dto.rs:94-116
    assert "distance" in hit_to_json(Hit(tenant_id=1, record_id=9, score=0.9, source=HitSource.Hamming, distance=6))
