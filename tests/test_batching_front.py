"""Host logic of the batching front (advanced_rag/batching.py) that needs no GPU: what happens to a round when a caller
goes away (its Future is cancelled by asyncio.wrap_future on a timeout) and when the worker thread dies."""
import queue
import threading
import time
from concurrent.futures import CancelledError, Future

import pytest

from advanced_rag.batching import SearchCoalescer, _deliver, _fail, _Request


class _Mgr:
    device = 0
    _main = None
    collections = {}


class _EchoFront(SearchCoalescer):
    """The worker loop without a device: a round answers every request with (key, payload)."""

    def __init__(self, fail_rounds=0):
        super().__init__(_Mgr())
        self.fail_rounds = fail_rounds
        self.gate = threading.Event()
        self.gate.set()

    def _run(self):
        while True:
            reqs = self._collect()
            if reqs is None:
                return
            self.gate.wait(5)
            if self.fail_rounds > 0:
                self.fail_rounds -= 1
                raise RuntimeError("HIP error: stream synchronisation failed (injected)")
            for r in reqs:
                _deliver(r.future, (r.key, r.payload))
            self._inflight = []


def test_deliver_and_fail_tolerate_cancelled_futures():
    f = Future()
    assert f.cancel()
    _deliver(f, 1)          # set_result on a cancelled future raises InvalidStateError: must not escape
    _fail(f, ValueError())
    with pytest.raises(CancelledError):
        f.result(0)
    g = Future()
    _deliver(g, 7)
    _deliver(g, 8)          # already answered: the first answer stands
    _fail(g, ValueError())
    assert g.result(0) == 7


def test_collect_drops_cancelled_requests_and_claims_the_rest():
    front = _EchoFront()
    reqs = [_Request("dense", ("k",), i) for i in range(5)]
    for r in reqs:
        front._q.put(r)
    assert reqs[1].future.cancel() and reqs[3].future.cancel()
    live = front._collect()
    assert [r.payload for r in live] == [0, 2, 4]
    assert front.stats["cancelled_before_launch"] == 2
    # a claimed request cannot be cancelled under the worker any more (what asyncio.wrap_future tries on a timeout)
    assert not live[0].future.cancel()
    _deliver(live[0].future, "hit")
    assert live[0].future.result(0) == "hit"


def test_worker_death_fails_the_waiting_requests_and_the_next_submit_restarts_it():
    front = _EchoFront(fail_rounds=1)
    front.gate.clear()
    futs = [front.submit("dense", ("k",), i) for i in range(4)]
    time.sleep(0.05)                    # the worker has collected (some of) them and waits at the gate
    late = front.submit("dense", ("k",), 99)
    front.gate.set()
    for f in futs + [late]:
        with pytest.raises(RuntimeError, match="search front worker failed"):
            f.result(5)
    t0 = front._thread
    if t0 is not None:
        t0.join(5)
    again = front.submit("dense", ("k2",), 5)      # a fresh worker serves it
    assert again.result(5) == (("k2",), 5)
    assert front.stats["worker_failures"] == 1
    front.close()


def test_one_cancelled_caller_among_co_batched_requests():
    front = _EchoFront()
    front.gate.clear()
    first = front.submit("dense", ("k",), 0)       # claimed by the worker, which then waits at the gate
    time.sleep(0.05)
    more = [front.submit("dense", ("k",), i) for i in range(1, 6)]
    assert more[2].cancel()                          # still queued: goes away before the next round
    first.cancel()                                   # claimed: cancel() is refused, the result is simply unread
    front.gate.set()
    assert first.result(5) == (("k",), 0)
    got = [f.result(5) for i, f in enumerate(more) if i != 2]
    assert got == [(("k",), i) for i in (1, 2, 4, 5)]
    front.close()
