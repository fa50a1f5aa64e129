#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name.

    python tools/pmc_by_kernel.py <..._counter_collection.csv> FETCH_SIZE > profiles/rN/bench_vK_pmc_fetch_size_by_kernel.csv

Output columns: kernel, dispatches, total_<COUNTER>_KB_raw, per_dispatch_KB_raw (raw counter units: KB for
FETCH_SIZE / WRITE_SIZE; the gfx950 x2 correction of FETCH_SIZE is applied by the reader, bench.py)."""
import csv, sys, collections

path, counter = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float)
disp = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter:
        continue
    k = r["Kernel_Name"]
    tot[k] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
w = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL)
w.writerow(["kernel", "dispatches", f"total_{counter}_KB_raw", "per_dispatch_KB_raw"])
for k in sorted(tot, key=lambda k: -tot[k]):
    n = len(disp[k])
    w.writerow([k, n, f"{tot[k]:.6e}", f"{tot[k] / max(n, 1):.6e}"])
