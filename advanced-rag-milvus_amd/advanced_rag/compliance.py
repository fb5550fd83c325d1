"""In-memory audit trail.  OUT OF THE HOT PATH (bookkeeping; reference
compliance.py:28-60 record, :157-190 log_retrieval); kept because
RetrievalResult.audit_trail carries an AuditLog."""
from __future__ import annotations

import random
import uuid
from dataclasses import dataclass, field
from datetime import datetime
from enum import Enum
from typing import Any, Dict, List, Optional


class AuditEventType(Enum):
    INGESTION = "ingestion"
    RETRIEVAL = "retrieval"
    DELETION = "deletion"


@dataclass
class AuditLog:
    event_id: str
    event_type: AuditEventType
    timestamp: str
    user_id: Optional[str]
    session_id: Optional[str]
    event_data: Dict[str, Any]
    parent_event_id: Optional[str] = None
    related_event_ids: List[str] = field(default_factory=list)
    compliance_flags: List[str] = field(default_factory=list)
    retention_policy: str = "standard"

    def to_dict(self) -> Dict[str, Any]:
        d = dict(self.__dict__)
        d["event_type"] = self.event_type.value
        return d


_ID_RNG = random.Random(uuid.uuid4().int)   # seeded once from the OS


class ComplianceManager:
    def __init__(self, enable_audit: bool = True, enable_versioning: bool = True):
        self.enable_audit = enable_audit
        self.enable_versioning = enable_versioning
        self.audit_logs: List[AuditLog] = []

    def _log(self, kind: AuditEventType, data: Dict[str, Any]) -> Optional[AuditLog]:
        if not self.enable_audit:
            return None
        # a random 128-bit id with the version / variant bits of a UUID4, from the process PRNG: uuid.uuid4() is one urandom
        # system call per entry, and a retrieve() writes an entry per returned chunk (reference pipeline.py:279-293)
        event_id = "%032x" % ((_ID_RNG.getrandbits(128) & ~(0xF << 76) & ~(0x3 << 62)) | (4 << 76) | (0x2 << 62))
        entry = AuditLog(event_id=event_id, event_type=kind, timestamp=datetime.now().isoformat(),
                         user_id=None, session_id=None, event_data=data)
        self.audit_logs.append(entry)
        return entry

    async def log_ingestion(self, document_count: int, chunk_count: int, report: Dict[str, Any]):
        return self._log(AuditEventType.INGESTION, {"document_count": document_count, "chunk_count": chunk_count,
                                                    "total_time_ms": report.get("total_time_ms")})

    async def log_retrieval(self, query: str, chunk_id: str, score: float, latency_ms: float):
        return self._log(AuditEventType.RETRIEVAL, {"query": query, "chunk_id": chunk_id, "score": score,
                                                    "latency_ms": latency_ms})

    async def close(self):
        return None
