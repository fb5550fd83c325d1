"""Every dense scan kernel at every shape it can serve: hr_debug_option(HR_DEBUG_DENSE_KERNELS) takes kernels out of the
selection so that the others step in (no register-resident 256-query pass -> the tiled contraction at D = 768 too; no
tiled contraction -> two 128-query passes; ...), HR_DEBUG_SPARSE_RPB / HR_DEBUG_GROUP_ROWS pin the sparse scan's ranges
per block and the candidate-group size.  Ids and score bits must match the oracle on every path, as on the defaults.
(The library reads no environment variables; these hooks are process-wide and reset after each test.)"""
import numpy as np
import pytest

import oracle
from advanced_rag import _native as nat

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture
def options():
    def set_(key, value):
        nat.debug_option(key, value)
    yield set_
    for key in (nat.HR_DEBUG_DENSE_KERNELS, nat.HR_DEBUG_SPARSE_RPB, nat.HR_DEBUG_GROUP_ROWS, nat.HR_DEBUG_FINISH_MODE):
        nat.debug_option(key, 0)


@pytest.mark.parametrize("mask", [8,        # tiled contraction preferred for 129..256 queries at D = 768
                                  1,        # no register-resident pass: tiled contraction
                                  1 | 2,    # neither 256-query pass: 128-query passes
                                  4,        # no large-batch pass at all: 64-query passes
                                  16])      # the 4 x 64-query register form of the 256-query pass (dense_scan_q64_kernel)
def test_dense_kernel_variants_match_the_oracle(gpu, options, mask):
    options(nat.HR_DEBUG_DENSE_KERNELS, mask)
    rng = np.random.default_rng(7)
    for n, d, B in ((70001, 768, 128), (70001, 768, 256), (9000, 256, 100), (3333, 1024, 300), (63, 768, 130)):
        X = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
        X[n // 2] = X[7]
        Q = rng.standard_normal((B, d)).astype(np.float32)
        h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE)
        h.add_dense(X[: n // 3])
        h.add_dense(X[n // 3:])
        h.finalize()
        pick = sorted(set([0, 1, 15, 16, 64, 65, 127, 128, 129, 255, 256, B - 1]) & set(range(B)))
        k = min(40, n)
        for m in (None, np.packbits(rng.random(n) < 0.5, bitorder="little")):
            ids, sc = h.search_dense(Q, k, m)
            oids, osc = oracle.dense_search(X, Q[pick], k, nat.HR_METRIC_COSINE, m)
            assert np.array_equal(ids[pick], oids), (n, d, B)
            assert np.array_equal(bits(sc[pick]), bits(osc)), (n, d, B)
        h.close()


@pytest.mark.parametrize("rpb,group_rows", [(1, 64), (3, 16), (16, 0)])
def test_sparse_scan_geometry_variants_match_the_oracle(gpu, options, rpb, group_rows):
    options(nat.HR_DEBUG_SPARSE_RPB, rpb)
    options(nat.HR_DEBUG_GROUP_ROWS, group_rows)
    rng = np.random.default_rng(11)
    V, nd = 3000, 60000
    idx = [np.sort(rng.choice(V, size=rng.integers(1, 60), replace=False)).astype(np.int32) for _ in range(nd)]
    val = [np.abs(rng.standard_normal(len(i))).astype(np.float32) + 0.01 for i in idx]
    indptr = np.concatenate([[0], np.cumsum([len(i) for i in idx])]).astype(np.int64)
    h = nat.ShardHandle(0, sparse_dim=V)
    h.add_sparse(indptr, np.concatenate(idx), np.concatenate(val))
    h.finalize()
    qs = [(np.sort(rng.choice(V, size=12, replace=False)).astype(np.int32), np.abs(rng.standard_normal(12)).astype(np.float32))
          for _ in range(70)]
    ids, sc = h.search_sparse(qs, 40)
    oids, osc = oracle.sparse_search(indptr, np.concatenate(idx), np.concatenate(val), qs, 40)
    assert np.array_equal(ids, oids) and np.array_equal(bits(sc), bits(osc))
    h.close()


@pytest.mark.parametrize("group_rows", [16, 64])
def test_q64_pass_proves_its_lists_like_the_qreg_pass(gpu, options, group_rows):
    """The 4 x 64-query form of the 256-query pass (HR_DEBUG_DENSE_KERNELS bit 16) through the DEVICE form: the same lists
    and, above all, the same exactness flags as the default 8 x 32-query form — a wrong group maximum (the first draft
    scaled the first row block of every block with scales that had not landed) hides behind the host form's escalation,
    but not behind the flags.  Several super-groups per block, a ragged last one, both group sizes."""
    torch = pytest.importorskip("torch")
    options(nat.HR_DEBUG_GROUP_ROWS, group_rows)
    rng = np.random.default_rng(group_rows)
    n, d, B, k = 300 * 64 * 5 + 37, 768, 256, 40
    X = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE)
    h.add_dense(X)
    h.finalize()
    q = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32)).cuda()
    out = {}
    for mask in (0, 16):
        options(nat.HR_DEBUG_DENSE_KERNELS, mask)
        ids = torch.empty((B, k), dtype=torch.int64, device="cuda")
        sc = torch.empty((B, k), dtype=torch.float32, device="cuda")
        fl = torch.zeros((B,), dtype=torch.int32, device="cuda")
        h.search_dense_dev(q.data_ptr(), B, k, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out[mask] = (ids.cpu().numpy(), sc.cpu().numpy(), fl.cpu().numpy())
    assert out[0][2].min() == 1 and out[16][2].min() == 1          # every list proven exact by both forms
    assert np.array_equal(out[0][0], out[16][0]) and np.array_equal(bits(out[0][1]), bits(out[16][1]))
    oids, osc = oracle.dense_search(X, q.cpu().numpy()[[0, 63, 64, 200, 255]], k, nat.HR_METRIC_COSINE)
    assert np.array_equal(out[16][0][[0, 63, 64, 200, 255]], oids)
    h.close()
