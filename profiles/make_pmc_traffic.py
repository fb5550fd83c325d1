"""Turn the two rocprofv3 PMC passes into profiles/rN_pmc_traffic.json (HBM bytes per launch, per kernel).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir_f> -o f -- python3 bench.py <args>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir_w> -o w -- python3 bench.py <args>
    python profiles/make_pmc_traffic.py <f_counter_collection.csv> <w_counter_collection.csv> rows dim batch [uniform|zipf] > profiles/rN_pmc_traffic.json

Counters are KiB per dispatch.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B,
/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as is.  Only dispatches of the full-size
workload are averaged (the parity gate's and the warm-up's small launches are filtered by grid size)."""
import csv
import json
import sys
from collections import defaultdict

KERNELS = {"dense_scan": ("dense_scan_kernel", "dense_scan_bigq_kernel", "dense_scan_qreg_kernel", "dense_scan_gemm_kernel"),
           "sparse_scan": ("sparse_scan_kernel",), "refine_dense": ("refine_dense_kernel",),
           "refine_sparse": ("refine_sparse_kernel",), "select_groups": ("select_groups_kernel",),
           "bucket_max": ("bucket_max_kernel",), "select_topk": ("select_topk_kernel",),
           "finish_fused": ("finish_kernel",), "post_lists": ("post_lists_kernel",), "hybrid_prep": ("hybrid_prep_kernel",)}


def per_kernel(path, counter):
    out = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for key, names in KERNELS.items():
                if any(n in r["Kernel_Name"] for n in names):
                    out[key].append((int(r["Grid_Size"]) if "Grid_Size" in r else 0, float(r["Counter_Value"]) * 1024.0))
    return out


def main():
    f_csv, w_csv, rows, dim, batch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    sparse_dist = sys.argv[6] if len(sys.argv) > 6 else "uniform"
    fetch, write = per_kernel(f_csv, "FETCH_SIZE"), per_kernel(w_csv, "WRITE_SIZE")
    kernels = {}
    for key in KERNELS:
        fv, wv = fetch.get(key, []), write.get(key, [])
        if not fv:
            continue
        big = max(v for _, v in fv)
        fsel = [v for _, v in fv if v > 0.5 * big]            # the full-size launches
        wbig = max((v for _, v in wv), default=0.0)
        wsel = [v for _, v in wv if v > 0.5 * wbig] or [0.0]
        fb, wb = 2.0 * sum(fsel) / len(fsel), sum(wsel) / len(wsel)
        kernels[key] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                        "launches_averaged": len(fsel)}
    dpad = -(-dim // 128) * 128
    if "dense_scan" in kernels:
        kernels["dense_scan"]["algorithmic_bytes_per_launch"] = rows * dpad * 2 + 4 * rows
    json.dump({"_how": __doc__.strip().splitlines()[0] + " — see profiles/make_pmc_traffic.py",
               "workload": {"rows": rows, "dim": dim, "batch": batch, "top_k": 20, "sparse_dist": sparse_dist}, "kernels": kernels},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
