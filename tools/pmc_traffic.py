#!/usr/bin/env python3
"""Developer tool: turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the same bench
command into the per-kernel HBM-side traffic table bench.py reads (profiles/r01_traffic.json).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv

Units and the gfx950 correction follow MI355X_MICROARCH.md: the counters are KiB, and
FETCH_SIZE under-reports wide coalesced reads by half (so it is doubled)."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(igemm_\w+<[^>]*>|\w+_kernel)\(", name)
    return m.group(1) if m else name[:80]


def collect(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k] += 1
    return tot, n


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, _ = collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline at B=512; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of "
                   "wide coalesced reads); counters are KiB; fabric-side requests, i.e. Infinity-Cache hits are included",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
        if "igemm" not in k and "avgpool" not in k and "stem" not in k:
            continue
        f, w = 2.0 * fetch[k] / nf[k], write.get(k, 0.0) / nf[k]
        out["kernels"][k] = {"launches_profiled": nf[k], "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                             "hbm_bytes_per_launch": f + w}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
