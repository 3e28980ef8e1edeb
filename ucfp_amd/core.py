"""Schema types mirroring src/core/mod.rs of the reference (Record :33-72, Hit :107-131,
Query :152-189, Modality :19-26). Field names and meaning are kept 1:1 so the parity tests read
like the reference's own."""
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional


class Modality(IntEnum):
    # catalog discriminants: src/index/embedded/mod.rs:106,411-414
    Audio = 0
    Image = 1
    Text = 2


class HitSource:
    Vector = "vector"
    Bm25 = "bm25"
    Fused = "fused"
    Hamming = "hamming"   # new capability behind /v1/query; not in the reference


@dataclass
class Record:
    tenant_id: int
    record_id: int
    modality: Modality
    format_version: int
    algorithm: str
    config_hash: int
    fingerprint: bytes
    embedding: Optional[List[float]] = None
    model_id: Optional[str] = None
    metadata: bytes = b""
    text: Optional[str] = None


@dataclass
class Hit:
    tenant_id: int
    record_id: int
    score: float                      # higher is better (src/core/mod.rs:113-115)
    source: str = HitSource.Vector
    vector_score: Optional[float] = None
    bm25_score: Optional[float] = None
    vector_rank: Optional[int] = None
    bm25_rank: Optional[int] = None
    term_hits: list = field(default_factory=list)
    distance: Optional[int] = None    # Hamming distance when source == "hamming"


FORMAT_VERSION = 1  # src/lib.rs:62


# ---- /v1/query wire types (src/server/dto.rs:74-116, handlers.rs:143-187) -----------------------------
# The reference's request needs `vector`; the Hamming search adds ONE additive, backward-compatible field
# (SURVEY 8b / 8f N3): `hash` (u64, or 8 little-endian bytes) with `algorithm` naming the hash space.  A body
# the reference accepts parses to the same query here.

DEFAULT_K = 10   # dto.rs:85-87


@dataclass
class QueryRequest:
    tenant_id: int
    modality: Modality
    k: int = DEFAULT_K
    vector: Optional[List[float]] = None
    hash: Optional[int] = None
    algorithm: Optional[str] = None

    @classmethod
    def from_json(cls, body: dict) -> "QueryRequest":
        from .errors import InvalidArgument
        try:
            tenant_id = int(body["tenant_id"])
            modality = Modality[body["modality"]]          # "Image" | "Audio" | "Text" (tests.rs:60,199)
        except (KeyError, TypeError, ValueError) as e:
            raise InvalidArgument(f"bad query body: {e}") from None
        k = int(body.get("k", DEFAULT_K))
        vector, h = body.get("vector"), body.get("hash")
        if vector is None and h is None:
            raise InvalidArgument("query needs `vector` (dto.rs:80-82) or `hash`")
        if isinstance(h, (list, bytes, bytearray)):
            if len(h) != 8:
                raise InvalidArgument("`hash` bytes must be 8 little-endian bytes")
            h = int.from_bytes(bytes(h), "little")
        if h is not None and not 0 <= int(h) < 1 << 64:
            raise InvalidArgument("`hash` must be a u64")
        return cls(tenant_id=tenant_id, modality=modality, k=max(k, 1),       # handlers.rs:153: k.max(1)
                   vector=[float(x) for x in vector] if vector is not None else None,
                   hash=int(h) if h is not None else None, algorithm=body.get("algorithm"))


def hit_to_json(h: Hit) -> dict:
    """HitOut (dto.rs:94-116); `distance` only appears on Hamming hits, so vector hits stay byte-stable."""
    out = {"tenant_id": h.tenant_id, "record_id": h.record_id, "score": h.score, "source": h.source,
           "vector_score": h.vector_score, "bm25_score": h.bm25_score, "vector_rank": h.vector_rank,
           "bm25_rank": h.bm25_rank, "term_hits": list(h.term_hits)}
    if h.distance is not None:
        out["distance"] = h.distance
    return out
