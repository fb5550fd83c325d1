import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag import _native as nat
from advanced_rag.encoder_kernels import fr_rows, from_fragment_order
for n_seq, T, heads, hd in ((7, 128, 12, 32), (5, 40, 4, 32), (5, 40, 4, 64), (2, 200, 3, 32)):
    H = heads * hd
    qkv = torch.randn((n_seq, T, 3, heads, hd), device="cuda").half()
    out = torch.zeros((n_seq, T, H), dtype=torch.float16, device="cuda")
    ofr = torch.zeros((fr_rows(n_seq * T), H), dtype=torch.float16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), n_seq, T, heads, hd, hd ** -0.5, s)
    nat.attention_fr_f16_dev(qkv.data_ptr(), 0, ofr.data_ptr(), n_seq, T, heads, hd, hd ** -0.5, s)
    torch.cuda.synchronize()
    got = from_fragment_order(ofr, n_seq * T)
    want = out.reshape(n_seq * T, H)
    bad = (got != want)
    print(n_seq, T, heads, hd, "mismatch", int(bad.sum()), "of", bad.numel(), "max abs diff", float((got.float() - want.float()).abs().max()))
    if bad.any():
        idx = bad.nonzero()[:8].tolist()
        print("  first:", [(r, c, float(got[r, c]), float(want[r, c])) for r, c in idx])
        rows = bad.any(dim=1).nonzero().flatten()
        cols = bad.any(dim=0).nonzero().flatten()
        print("  rows", rows[:10].tolist(), "...", int(rows.numel()), " cols", cols[:16].tolist(), "...", int(cols.numel()))
