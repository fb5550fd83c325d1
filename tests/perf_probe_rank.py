"""Ad-hoc perf probe (not a test): the pipelined hybrid step on a rank-sized shard under several stream / finishing
arrangements, ONE shard build, every arrangement timed in the same process (so the A/B shares box and data):

    python tests/perf_probe_rank.py [rows=1250000] [steps=300] [simulate_ranks=8] > gpurun_out/rank_probe.jsonl

One JSON line per arrangement: ms/step, per-phase event times (hr_set_profiling(2)), finish / post kernels alone."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (corpus generators)
from advanced_rag import _native as nat  # noqa: E402
from advanced_rag.engine import EngineConfig, PipelinedSearchEngine, pack_sparse_queries  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
SIM = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ONLY = sys.argv[4].split(",") if len(sys.argv) > 4 else None
D, B, BLK = 768, 128, 250_000
dev = torch.device("cuda:0")

h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, bench.SPARSE_DIM, 0)
h.reserve(N)
t0 = time.time()
for b in range(-(-N // BLK)):
    n = min(BLK, N - b * BLK)
    h.add_dense(bench.dense_block(b, n, D))
    h.add_sparse(*bench.sparse_block(b, n))
h.finalize()
print(f"shard {N} rows built in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)

Q, SQ = bench.make_queries(8, B, D)
cfg = EngineConfig(top_k=20)
dQ = [torch.from_numpy(Q[i]).to(dev) for i in range(8)]

ARRANGEMENTS = [
    # name, finish mode, prep stream, light CUs, depth
    ("r2_like_chain_noprep", 1, False, 0, 3),
    ("chain_prep", 1, True, 0, 3),
    ("fused_noprep", 2, False, 0, 3),
    ("fused_prep", 2, True, 0, 3),
    ("fused_prep_depth2", 2, True, 0, 2),
    ("fused_prep_depth4", 2, True, 0, 4),
    ("fused_prep_cu16", 2, True, 16, 3),
    ("fused_prep_cu24", 2, True, 24, 3),
    ("fused_prep_cu32", 2, True, 32, 3),
    ("fused_prep_cu48", 2, True, 48, 3),
    ("fused_prep_cu64", 2, True, 64, 3),
    ("chain_prep_cu32", 1, True, 32, 3),
]

for name, mode, prep, cus, depth in ARRANGEMENTS:
    if ONLY and name not in ONLY:
        continue
    nat.debug_option(nat.HR_DEBUG_FINISH_MODE, mode)
    try:
        eng = PipelinedSearchEngine(h, cfg, device=str(dev), depth=depth, simulate_ranks=SIM, light_cus=cus, prep_stream=prep)
    except Exception as e:  # e.g. CU masks refused on this box
        print(json.dumps({"arrangement": name, "error": str(e)}), flush=True)
        continue
    dS = [eng.upload_sparse(pack_sparse_queries(SQ[i], 0.2)) for i in range(8)]
    for i in range(20):
        out = eng.submit(dQ[i % 8], dS[i % 8])
    torch.cuda.synchronize()
    h.kernel_ms()
    res = {}
    for prof in (0, 2):  # untimed-phase run first (the step time that counts), then with every phase bracketed
        h.set_profiling(prof)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(STEPS):
            out = eng.submit(dQ[i % 8], dS[i % 8])
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[f"ms_per_step_prof{prof}"] = dt / STEPS * 1e3
        res[f"host_enqueue_ms_prof{prof}"] = host / STEPS * 1e3
    h.set_profiling(0)
    phases = {k: round(v[0], 4) for k, v in h.kernel_ms().items() if v[1]}
    exact = eng.all_flags_exact()
    # finish + post alone on an idle chip
    st = torch.cuda.current_stream(dev)
    slot = (eng._n - 1) % eng.depth
    q_, (ip_, ix_, iv_, mx_) = dQ[(STEPS - 1) % 8], dS[(STEPS - 1) % 8]

    def timed(fn, reps=30):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    fin_us = timed(lambda: h.hybrid_finish_dev(q_.data_ptr(), ip_.data_ptr(), ix_.data_ptr(), iv_.data_ptr(), B, int(mx_), 40, slot,
                                               out["ids"].data_ptr(), out["scores"].data_ptr(), out["flags"].data_ptr(), st.cuda_stream))
    pa = eng._post_args(out, B, eng.n_lists, out.get("gathered"))
    post_us = timed(lambda: nat.post_lists_dev(pa, B, st.cuda_stream))
    pieces = {}
    if eng.n_lists > 1:   # the parts of the post kernel as the separate entry points
        g, lay = out["gathered"], out["layout"]
        def merges():
            for m in range(2):
                sc_off, id_off, sc_stride, id_stride = lay.merge_args(m)
                nat.merge_topk_dev(g.data_ptr() + sc_off, g.data_ptr() + id_off, eng.n_lists, B, 40, 40, out["m_ids"][m].data_ptr(),
                                   out["m_scores"][m].data_ptr(), st.cuda_stream, score_stride=sc_stride, id_stride=id_stride)
        pieces["merge_2_modalities_us"] = timed(merges)
        pieces["fuse_us"] = timed(lambda: nat.fuse_rrf_dev(out["m_ids"][0].data_ptr(), 40, out["m_ids"][1].data_ptr(), 40, 0, 0, B, 0.7, 0.3, 0.2, 60,
                                                           20, out["fused_ids"].data_ptr(), out["fused_scores"].data_ptr(),
                                                           out["fused_methods"].data_ptr(), out["fused_n"].data_ptr(), st.cuda_stream))
        pieces["rerank_us"] = timed(lambda: nat.rerank_linear_dev(out["fused_ids"].data_ptr(), out["fused_scores"].data_ptr(),
                                                                  out["fused_methods"].data_ptr(), out["fused_n"].data_ptr(), B, 20, 1.0, 0.1,
                                                                  0.0, 5, out["rr_ids"].data_ptr(), out["rr_scores"].data_ptr(),
                                                                  out["rr_orig"].data_ptr(), st.cuda_stream))
        pieces["empty_launch_pair_us"] = timed(lambda: (torch.cuda._sleep(1), torch.cuda._sleep(1)))
    print(json.dumps({"arrangement": name, "rows": N, "simulate_ranks": SIM, "depth": depth, "light_cus": cus, "prep_stream": prep,
                      "finish": {1: "chain", 2: "fused"}[mode], **res, "phases_ms": phases, "all_exact": exact,
                      "finish_alone_us": fin_us, "post_alone_us": post_us, "post_pieces": pieces,
                      "dense_scan_frac_hbm": h.dense_scan_bytes / (phases.get("dense_scan", 1e9) * 1e-3) / 8e12}), flush=True)
    eng.close()
nat.debug_option(nat.HR_DEBUG_FINISH_MODE, 0)
h.close()
