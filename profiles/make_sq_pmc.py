"""Turn rocprofv3 SQ-counter passes into profiles/rN_sparse_scan_pmc*.json (mean per launch of one kernel + derived shares).

    rocprofv3 --kernel-trace --pmc <up to 8 SQ counters> --output-format csv -d <dir> -o p -- python3 tests/perf_probe_sparse.py 10000000 128 [zipf]
    python profiles/make_sq_pmc.py sparse_scan_kernel "<what was run>" <counter_collection.csv> [<second pass csv> ...] > profiles/r3_sparse_scan_pmc.json

SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over the chip (MI355X_MICROARCH.md, PMC slots)."""
import csv
import json
import sys
from collections import defaultdict


def main():
    kernel, what, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
    vals, dur = defaultdict(list), []
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel not in r["Kernel_Name"]:
                    continue
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r and r.get("End_Timestamp"):
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    c = {k: sum(v) / len(v) for k, v in vals.items()}
    d = {}
    g = c.get
    if g("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        for name, key in (("waves_parked_share (SQ_WAIT_ANY / SQ_WAVE_CYCLES)", "SQ_WAIT_ANY"),
                          ("issue_stall_share (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)", "SQ_WAIT_INST_ANY"),
                          ("issuing_share (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)", "SQ_ACTIVE_INST_ANY")):
            if g(key) is not None:
                d[name] = c[key] / wc
    if g("SQ_WAIT_INST_LDS") is not None and g("SQ_WAIT_INST_ANY"):
        d["lds_share_of_issue_stalls (SQ_WAIT_INST_LDS / SQ_WAIT_INST_ANY)"] = c["SQ_WAIT_INST_LDS"] / c["SQ_WAIT_INST_ANY"]
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if g("SQ_LDS_IDX_ACTIVE") is not None and g("SQ_BUSY_CYCLES"):
        d["lds_busy_share_of_cu_cycles (SQ_LDS_IDX_ACTIVE / (256 CUs x SQ_BUSY_CYCLES / 32 shader engines))"] = \
            c["SQ_LDS_IDX_ACTIVE"] / (256.0 * c["SQ_BUSY_CYCLES"] / 32.0)
    if g("SQ_INSTS_VALU") and g("SQ_INSTS_LDS"):
        d["valu_instructions_per_lds_instruction"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_LDS"]
    json.dump({"_what": what, "kernel": kernel, "launches_averaged": max((len(v) for v in vals.values()), default=0),
               "mean_launch_ms_under_the_profiler": (sum(dur) / len(dur) if dur else None), "counters": c, "derived": d},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
