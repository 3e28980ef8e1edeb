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
