"""The N > 1 form of bench.py, exactly as the driver launches it (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), rehearsed on the one GPU of the
test box: BENCH_REHEARSAL=1 puts both ranks on cuda:0 and exchanges over gloo (RCCL refuses two ranks on one device) —
everything else (rank env, row sharding, the per-batch exchange, the post kernel's merge, the barrier + max-over-ranks
timing, the single JSON line of rank 0) is the code an 8-GPU node runs.  The launcher is a CHILD process: nothing in
this test process replaces itself, and the launcher itself never touches the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_rehearsal_prints_one_valid_line(gpu):
    env = dict(os.environ, BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--no-api-concurrent", "--no-config4-full", "--latency-queries", "8"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ranks_agree"] is True and out["all_lists_proven_exact"] is True
    assert out["value"] > 0 and out["unit"] and out["higher_is_better"] is True and out["scaling"] in ("strong", "weak")
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    assert out["config"]["workload"]


def test_bench_two_ranks_rehearsal_with_the_cross_encoder_hook(gpu):
    """The same launch with `--rerank cross-encoder`: every rank scores its share of the queries' candidates in the
    engine's post hook — which is deferred by one batch (engine.post_hook_exclusive) and carries the all-gather of the kept
    ids — so the collectives inside deferred hooks must line up across the ranks, and the last one runs in synchronize()."""
    env = dict(os.environ, BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline", "--no-api-concurrent", "--no-config4-full", "--no-latency",
           "--rerank", "cross-encoder", "--ce-seq-len", "32", "--batch", "16"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_agree"] is True and out["all_lists_proven_exact"] is True
    assert out["cross_encoder"]["forwards_timed"] == 4 and out["cross_encoder"]["pairs_per_step"] == 8 * 20
