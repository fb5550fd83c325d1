"""Embedding cache with the reference's API and statistics.

Semantics follow reference src/advanced_rag/embedding_cache.py:38-245:
sha256 keys (`text` or `model:text`), TTL checked on read, capacity eviction of
the FIRST-inserted entry, hit/miss/eviction counters, and get/put/
get_or_compute that work both awaited and called synchronously; the three
process-wide singletons of :248-285.

North-star change: values may be device-resident.  `DeviceEmbeddingTable`
keeps the vectors in ONE preallocated HBM tensor (slot per key) so a cache hit
hands the search kernel a device pointer instead of re-uploading the query.
"""
from __future__ import annotations

import hashlib
import threading
import time
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple


@dataclass
class CacheStats:
    hits: int = 0
    misses: int = 0
    evictions: int = 0
    current_size: int = 0

    @property
    def hit_rate(self) -> float:
        seen = self.hits + self.misses
        return self.hits / seen if seen else 0.0


class _Ready:
    """Awaitable that also carries its value for synchronous callers."""

    __slots__ = ("value",)

    def __init__(self, value: Any):
        self.value = value

    def __await__(self):
        async def _v():
            return self.value
        return _v().__await__()


class EmbeddingCache:
    def __init__(self, max_size: Optional[int] = None, ttl_seconds: int = 3600, enabled: bool = True,
                 maxsize: Optional[int] = None):
        cap = max_size if max_size is not None else (maxsize if maxsize is not None else 10000)
        self.max_size = self.maxsize = cap
        self.ttl_seconds = ttl_seconds
        self.enabled = enabled
        self._cache: Dict[str, Tuple[float, Any]] = {}
        self._lock = threading.RLock()
        self._stats = CacheStats()

    @staticmethod
    def _materialize_key(*parts: Any) -> str:
        raw = str(parts[0]) if len(parts) == 1 else (f"{parts[1]}:{parts[0]}" if len(parts) >= 2 else "")
        return hashlib.sha256(raw.encode("utf-8")).hexdigest()

    # -- synchronous core ------------------------------------------------------
    def _sync_get(self, *key_parts: Any):
        if not self.enabled:
            return None
        key = self._materialize_key(*key_parts)
        with self._lock:
            hit = self._cache.get(key)
            if hit is None:
                self._stats.misses += 1
                return None
            ts, value = hit
            if self.ttl_seconds > 0 and time.time() - ts > self.ttl_seconds:
                del self._cache[key]
                self._stats.current_size = len(self._cache)
                self._stats.misses += 1
                return None
            self._stats.hits += 1
            return value

    def _sync_put(self, *args: Any) -> None:
        if not self.enabled:
            return
        if len(args) == 2:
            key, value = self._materialize_key(args[0]), args[1]
        elif len(args) >= 3:
            key, value = self._materialize_key(args[0], args[1]), args[2]
        else:
            return
        with self._lock:
            if key not in self._cache and len(self._cache) >= self.max_size and self._cache:
                del self._cache[next(iter(self._cache))]
                self._stats.evictions += 1
            self._cache[key] = (time.time(), value)
            self._stats.current_size = len(self._cache)

    # -- public, awaitable-or-direct --------------------------------------------
    def get(self, *key_parts: Any) -> _Ready:
        return _Ready(self._sync_get(*key_parts))

    def put(self, *args: Any) -> _Ready:
        self._sync_put(*args)
        return _Ready(None)

    def get_or_compute(self, *args: Any):
        if len(args) == 2:
            key, fn = args
            key_parts: Tuple[Any, ...] = (key,)
        elif len(args) >= 3:
            key, fn, key_parts = args[0], args[2], (args[0], args[1])
        else:
            raise TypeError("get_or_compute requires at least key and compute_fn")
        return _Compute(self, key, fn, key_parts)

    def clear(self) -> None:
        with self._lock:
            self._cache.clear()
            self._stats = CacheStats()

    def get_stats(self) -> dict:
        with self._lock:
            return {"size": len(self._cache), "hits": self._stats.hits, "misses": self._stats.misses,
                    "evictions": self._stats.evictions, "hit_rate": self._stats.hit_rate,
                    "max_size": self.max_size, "ttl_seconds": self.ttl_seconds, "enabled": self.enabled}

    def reset_stats(self) -> None:
        with self._lock:
            self._stats = CacheStats(current_size=len(self._cache))


class _Compute:
    """The awaitable get_or_compute returns (one class for all calls: building a class per request showed in the profile)."""
    __slots__ = ("cache", "key", "fn", "key_parts")

    def __init__(self, cache, key, fn, key_parts):
        self.cache, self.key, self.fn, self.key_parts = cache, key, fn, key_parts

    async def _run(self):
        got = self.cache._sync_get(*self.key_parts)
        if got is not None:
            return got
        fn = self.fn
        takes_key = getattr(getattr(fn, "__code__", None), "co_argcount", 0) != 0
        value = await (fn(self.key) if takes_key else fn())
        self.cache._sync_put(self.key, value)  # stored under the plain key, as the reference does
        return value

    def __await__(self):
        return self._run().__await__()


class DeviceEmbeddingTable:
    """Device-resident value store for an EmbeddingCache: one [capacity, dim]
    fp32 tensor in HBM; `slot_for(key)` hands out rows FIFO like the host cache."""

    def __init__(self, capacity: int, dim: int, device: str = "cuda:0"):
        import torch
        self.capacity, self.dim = capacity, dim
        self.table = torch.zeros((capacity, dim), dtype=torch.float32, device=device)
        self._slots: Dict[str, int] = {}
        self._next = 0
        self._lock = threading.Lock()

    def lookup(self, key: str):
        with self._lock:
            s = self._slots.get(key)
        return None if s is None else self.table[s]

    def store(self, key: str, vec) -> Any:
        """Rows are handed out as VIEWS (a hit is a device pointer the search kernel reads in place), so a row may only be
        overwritten once every search already enqueued with the evicted key's view has finished: an eviction waits for the
        device before it copies (a miss has just paid for an encoder forward; first fills of a slot need no wait).  A view
        is valid for searches enqueued before the next store() of a new key; holders re-`lookup` after that."""
        import torch
        with self._lock:
            s = self._slots.get(key)
            evicting = False
            if s is None:
                if len(self._slots) >= self.capacity:
                    victim = next(iter(self._slots))
                    s = self._slots.pop(victim)
                    evicting = True
                else:
                    s = self._next
                    self._next += 1
                self._slots[key] = s
            if evicting:
                torch.cuda.synchronize(self.table.device)
            self.table[s].copy_(torch.as_tensor(vec, dtype=torch.float32), non_blocking=True)
            return self.table[s]


_semantic_cache: Optional[EmbeddingCache] = None
_sparse_cache: Optional[EmbeddingCache] = None
_domain_cache: Optional[EmbeddingCache] = None


def initialize_caches(maxsize: int = 10000, ttl_seconds: int = 3600, enabled: bool = True) -> None:
    global _semantic_cache, _sparse_cache, _domain_cache
    _semantic_cache = EmbeddingCache(maxsize, ttl_seconds, enabled)
    _sparse_cache = EmbeddingCache(maxsize, ttl_seconds, enabled)
    _domain_cache = EmbeddingCache(maxsize // 2, ttl_seconds, enabled)


def get_semantic_cache() -> EmbeddingCache:
    if _semantic_cache is None:
        initialize_caches()
    return _semantic_cache


def get_sparse_cache() -> EmbeddingCache:
    if _sparse_cache is None:
        initialize_caches()
    return _sparse_cache


def get_domain_cache() -> EmbeddingCache:
    if _domain_cache is None:
        initialize_caches()
    return _domain_cache
