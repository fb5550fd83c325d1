"""Opt-in kernel variants that the library selects by environment variables read once per process
(HBMRAG_GEMM128: 65..128 queries through dense_scan_gemm_kernel<8>; HBMRAG_GEMM: 129..256 queries through
dense_scan_gemm_kernel<16> at D = 768 too; HBMRAG_REFINE_DPW: docs per wave of the sparse refine): each is run in
a child process against the oracle — ids and score bits must match as on the default paths."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/advanced-rag-milvus_amd")
    import numpy as np
    import oracle
    from advanced_rag import _native as nat
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    rng = np.random.default_rng(7)
    for n, d, B in ((70001, 768, 128), (70001, 768, 256), (9000, 256, 100), (3333, 1024, 300), (63, 768, 130)):
        X = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
        X[n // 2] = X[7]
        Q = rng.standard_normal((B, d)).astype(np.float32)
        h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE)
        h.add_dense(X[: n // 3]); h.add_dense(X[n // 3:]); h.finalize()
        pick = sorted(set([0, 1, 15, 16, 64, 65, 127, 128, 129, 255, 256, B - 1]) & set(range(B)))
        k = min(40, n)
        for m in (None, np.packbits(rng.random(n) < 0.5, bitorder="little")):
            ids, sc = h.search_dense(Q, k, m)
            oids, osc = oracle.dense_search(X, Q[pick], k, nat.HR_METRIC_COSINE, m)
            assert np.array_equal(ids[pick], oids), (n, d, B)
            assert np.array_equal(bits(sc[pick]), bits(osc)), (n, d, B)
        h.close()
    # sparse refine with short wave chains
    V, nd = 3000, 20000
    idx = [np.sort(rng.choice(V, size=rng.integers(1, 60), replace=False)).astype(np.int32) for _ in range(nd)]
    val = [np.abs(rng.standard_normal(len(i))).astype(np.float32) + 0.01 for i in idx]
    indptr = np.concatenate([[0], np.cumsum([len(i) for i in idx])]).astype(np.int64)
    h = nat.ShardHandle(64, nat.HR_F16, nat.HR_METRIC_COSINE, sparse_dim=V)
    h.add_dense(rng.standard_normal((nd, 64)).astype(np.float32))
    h.add_sparse(indptr, np.concatenate(idx), np.concatenate(val)); h.finalize()
    qs = [(np.sort(rng.choice(V, size=12, replace=False)).astype(np.int32), np.abs(rng.standard_normal(12)).astype(np.float32))
          for _ in range(70)]
    ids, sc = h.search_sparse(qs, 40)
    oids, osc = oracle.sparse_search(indptr, np.concatenate(idx), np.concatenate(val), qs, 40)
    assert np.array_equal(ids, oids) and np.array_equal(bits(sc), bits(osc))
    print("OK")
""")


@pytest.mark.parametrize("env", [{"HBMRAG_GEMM128": "1", "HBMRAG_GEMM": "1"},
                                 {"HBMRAG_REFINE_DPW": "8", "HBMRAG_NO_QREG": "1", "HBMRAG_SPLIT_FINISH": "1"}])
def test_opt_in_kernel_variants_match_the_oracle(gpu, env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
