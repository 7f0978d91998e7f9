#!/usr/bin/env python3
"""Developer tool: fold the rocprofv3 passes of ONE bench command into the per-kernel table bench.py reads
(profiles/r0N_pmc.json) and a kernel-stats csv.

    rocprofv3 --kernel-trace --stats --output-format csv -d D/stats -o s -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D/fetch -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D/write -o w -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE \
              SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d D/sq -o q -- ...
    python tools/pmc_summary.py D RN50 1024 f32 > profiles/r03_pmc.json

Separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing domains
beside --pmc).  Units and gfx950 correction from the same guide: the TCC counters are KiB, FETCH_SIZE reports half of
a wide coalesced read stream and is doubled; both are fabric-side (Infinity-Cache hits included).  Durations come from
the un-instrumented --stats pass, never from a counter pass (counter collection serialises kernels)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dbmm_amd  # noqa: E402,F401
from dbmm_amd import _lib  # noqa: E402


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:80]


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def counters(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    if not path:
        return tot, n
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    return tot, n


def main():
    d, arch, bl, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    dur, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(find(os.path.join(d, "stats"), "*kernel_trace.csv"))):
        k = short(r["Kernel_Name"])
        dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
        calls[k] += 1
    fetch, nf = counters(find(os.path.join(d, "fetch"), "*counter_collection.csv"))
    write, nw = counters(find(os.path.join(d, "write"), "*counter_collection.csv"))
    sq, _ = counters(find(os.path.join(d, "sq"), "*counter_collection.csv"))
    total = sum(dur.values())
    out = {"arch": arch, "batch_per_gpu": bl, "dtype": dtype,
           "note": "rocprofv3 passes of `python3 bench.py` at this configuration (tools/pmc_summary.py): durations from the "
                   "--kernel-trace --stats pass; hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 FETCH_SIZE "
                   "counts half of wide coalesced reads, MI355X_MICROARCH.md) per launch, fabric-side (Infinity-Cache hits "
                   "included); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); waves_per_simd = SQ_WAVE_CYCLES / "
                   "SQ_BUSY_CU_CYCLES; lds_bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
           "kernels": {}}
    for k in sorted(dur, key=lambda k: -dur[k]):
        if dur[k] < 0.002 * total:
            continue
        e = {"launches": calls[k], "avg_launch_ms": dur[k] / calls[k], "share_of_gpu_time": dur[k] / total}
        if k in fetch and nf[k]["FETCH_SIZE"]:
            f = 2.0 * 1024.0 * fetch[k]["FETCH_SIZE"] / nf[k]["FETCH_SIZE"]
            w = 1024.0 * write[k]["WRITE_SIZE"] / max(nw[k]["WRITE_SIZE"], 1)
            e.update(fetch_bytes_per_launch=f, write_bytes_per_launch=w, hbm_bytes_per_launch=f + w,
                     hbm_tbps=(f + w) / (e["avg_launch_ms"] * 1e-3) / 1e12)
        c = sq.get(k)
        if c and c.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
            e["waves_per_simd"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CU_CYCLES"]
            if c.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
                e["lds_active_frac_of_cu_cycles"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"]
            if c.get("SQ_WAVE_CYCLES"):
                e["wave_time_waiting_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
                e["wave_time_issuing_frac"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        e["source_hash"] = _lib.kernel_source_hash(k)      # bench.py drops these counters once the kernel's sources change
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
