#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

usage: tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<command>" [<workload key, e.g. cox2_x64>] [<round tag>] [<passes = steps + warmup of the command>]
The output file holds one entry per workload key ("<workload>_x<replicas>", what bench.py looks up);
an existing file is updated in place.
Per kernel and launch: raw counters (KB) and HBM bytes corrected as MI355X_MICROARCH.md (HBM
section) prescribes for gfx950: FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced
reads -> x2; WRITE_SIZE is exact for 16-B-per-lane stores.  Other access widths are uncalibrated,
so the corrected read figure is an upper bound for kernels that mix widths (noted per kernel)."""
import collections
import csv
import json
import sys


def load(path, cname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("desco::", "").replace("gf16::", "").replace("small::", "").strip()
        if k.startswith("shmp_layer16_kernel<"):           # <NW, KB, ST, LD64, POOL, F16[, SELFDEG]>
            a = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            k = f"shmp_layer16_kernel<{a[1]},{a[2]}{',f16x3' if len(a) > 5 and a[5] == 'true' else ''}{',selfdeg' if len(a) > 6 and a[6] == 'true' else ''}>"
        elif k.startswith("shmp_layer_f32_kernel<"):
            # template <KB, ST, X6, LD64> -> the profiler key of desco_amd/ops.py
            a = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            k = f"shmp_layer_f32_kernel<{a[0]},{a[1]},{'x6' if a[2] == 'true' else 'f32'}>"
        else:
            k = k.split("<")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    key = sys.argv[5] if len(sys.argv) > 5 else "cox2_x64"
    passes = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    entry = {"command": sys.argv[4] if len(sys.argv) > 4 else "", "round": sys.argv[6] if len(sys.argv) > 6 else "",
             "passes": passes, "kernels": {}}
    for k in sorted(f, key=lambda k: -f[k][1]):
        if "_kernel" not in k:
            continue
        fk = f[k][1] / f[k][0]
        wk = w[k][1] / w[k][0] if k in w and w[k][0] else 0.0
        # per STEP figures: bench.py compares launches_per_step with the live run before using the bytes
        entry["kernels"][k] = {"launches": f[k][0], "launches_per_step": f[k][0] / passes,
                               "hbm_bytes_per_step": (2 * fk + wk) * 1024 * f[k][0] / passes,
                               "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                               "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
                               "hbm_bytes_per_launch_uncorrected": (fk + wk) * 1024}
    try:
        out = json.load(open(sys.argv[3]))
    except (OSError, ValueError):
        out = {}
    out.setdefault("unit_note", "FETCH_SIZE/WRITE_SIZE in KB per launch; hbm_bytes = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024 (gfx950 correction for 16 B/lane reads)")
    out.setdefault("workloads", {})[key] = entry
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(entry["kernels"], indent=1)[:1500])


if __name__ == "__main__":
    main()
