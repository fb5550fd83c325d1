"""Chunk records consumed by index_chunks, plus a small sentence-window chunker.

OUT OF THE HOT PATH: only the record types and the chunk-id format
matter to search, because `chunk_id` is the id every search returns
(reference chunking.py:13-64 types, :361-364 id = "{doc_id}::{idx}::{sha256(text)[:8]}").
The chunker is a compact host counterpart, not a restatement of the
reference's adaptive heuristics (:158-201).
"""
from __future__ import annotations

import hashlib
import re
from dataclasses import dataclass, field
from datetime import datetime
from typing import Any, Dict, List, Optional


@dataclass
class ChunkMetadata:
    chunk_id: str
    doc_id: str
    chunk_index: int
    char_start: int
    char_end: int
    token_count: int
    entropy: float
    redundancy: float
    domain_density: float
    coherence_score: float
    source: str
    timestamp: str
    version: str
    extra: Dict[str, Any] = field(default_factory=dict)

    def to_dict(self) -> Dict[str, Any]:
        base = {k: getattr(self, k) for k in ("chunk_id", "doc_id", "chunk_index", "char_start", "char_end",
                                              "token_count", "entropy", "redundancy", "domain_density",
                                              "coherence_score", "source", "timestamp", "version")}
        base.update(self.extra)
        return base


@dataclass
class Chunk:
    text: str
    metadata: ChunkMetadata

    def __hash__(self):
        return hash(self.metadata.chunk_id)


_WORD = re.compile(r"\b\w+\b")
_SENT = re.compile(r"(?<=[.!?])\s+")


def tokenize(text: str) -> List[str]:
    return _WORD.findall(text.lower())


def chunk_id_for(doc_id: Any, index: int, content: str) -> str:
    return f"{doc_id}::{index}::{hashlib.sha256(content.encode()).hexdigest()[:8]}"


class AdaptiveChunker:
    def __init__(self, base_chunk_size: int = 512, max_chunk_size: int = 1024, min_chunk_size: int = 128,
                 overlap_ratio: float = 0.15, semantic_boundary_detection: bool = True):
        self.base_chunk_size = base_chunk_size
        self.max_chunk_size = max_chunk_size
        self.min_chunk_size = min_chunk_size
        self.overlap_ratio = overlap_ratio
        self.semantic_boundary_detection = semantic_boundary_detection

    def _target_size(self, diagnostics) -> int:
        size = self.base_chunk_size
        ent = getattr(diagnostics, "information_entropy", 0.5)
        size = int(size * (1.3 if ent > 0.8 else 0.8 if ent < 0.4 else 1.0))
        if getattr(diagnostics, "redundancy_score", 0.0) > 0.6:
            size = int(size * 0.7)
        if getattr(diagnostics, "domain_density", 0.0) > 0.3:
            size = int(size * 0.85)
        if getattr(diagnostics, "semantic_coherence", 1.0) < 0.3:
            size = int(size * 0.75)
        return max(self.min_chunk_size, min(size, self.max_chunk_size))

    def chunk_document(self, text: str, diagnostics, metadata: Optional[Dict[str, Any]] = None) -> List[Chunk]:
        metadata = metadata or {}
        target = self._target_size(diagnostics)
        overlap = int(target * self.overlap_ratio)
        sentences = [s.strip() for s in _SENT.split(text) if s.strip()]
        windows, cur, cur_n, cur_tok, pos, start = [], [], [], 0, 0, 0   # cur_n: token counts of the sentences in cur
        for s in sentences:
            n = len(tokenize(s))
            if cur and cur_tok + n > target:
                windows.append((" ".join(cur), start, pos))
                keep, keep_n, kept = [], [], 0
                for prev, pn in zip(reversed(cur), reversed(cur_n)):
                    if kept + pn > overlap:
                        break
                    keep.insert(0, prev)
                    keep_n.insert(0, pn)
                    kept += pn
                cur, cur_n, cur_tok = keep, keep_n, kept
                start = pos - sum(len(x) + 1 for x in keep)
            cur.append(s)
            cur_n.append(n)
            cur_tok += n
            pos += len(s) + 1
        if cur:
            windows.append((" ".join(cur), start, len(text)))
        doc_id = metadata.get("doc_id", hashlib.sha256(text.encode()).hexdigest()[:16])
        chunks = []
        for i, (body, a, b) in enumerate(windows):
            toks = tokenize(body)
            uniq = len(set(toks)) / len(toks) if toks else 1.0
            chunks.append(Chunk(text=body, metadata=ChunkMetadata(
                chunk_id=chunk_id_for(doc_id, i, body), doc_id=doc_id, chunk_index=i, char_start=a, char_end=b,
                token_count=len(toks), entropy=float(getattr(diagnostics, "information_entropy", 0.0)),
                redundancy=float(1.0 - uniq), domain_density=float(getattr(diagnostics, "domain_density", 0.0)),
                coherence_score=float(getattr(diagnostics, "semantic_coherence", 0.0)),
                source=metadata.get("source", "unknown"),
                timestamp=metadata.get("timestamp", datetime.now().isoformat()),
                version=metadata.get("version", "1.0"),
                extra={k: v for k, v in metadata.items() if k not in ("doc_id", "source", "timestamp", "version")})))
        return chunks
