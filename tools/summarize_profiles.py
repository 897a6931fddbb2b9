#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into
gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_hbm_traffic.json (bytes per launch, with
the gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request, so
it is doubled; WRITE_SIZE is taken as is; both are reported in KB by rocprofv3)."""
import collections
import csv
import glob
import json
import shutil
import sys

CLASS = {"k_extract<1, false>": "extract_hist", "k_extract<2, false>": "extract_hist",
         "k_extract_staged": "extract_scatter", "k_extract<1, true>": "extract_scatter",
         "k_extract<2, true>": "extract_scatter", "k_bucket_sort_rle": "bucket_sort_rle",
         "k_setop": "setop", "k_range_bounds": "range_bounds", "k_union_tagged": "union_tagged",
         "k_union_hash": "union_tagged", "k_grid_bucket": "bucket_sort_rle", "k_grid_oversize": "grid_oversize",
         "k_skm_scatter": "skm_scatter", "k_skm_regroup": "skm_regroup", "k_skm_union": "skm_union",
         "k_skm2_scatter": "skm_scatter", "k_skm2_regroup": "skm_regroup", "k_skm2_union": "skm_union",
         "k_skm_pack": "skm_pack", "k_skm_phased": "skm_phased", "k_skm_big": "skm_big"}


def cls(name):
    for key, c in CLASS.items():
        if key in name:
            return c
    return None



if __name__ == "__main__":
    out, tag = sys.argv[1], sys.argv[2]
    stats = glob.glob(f"{out}/stats/*/*kernel_stats.csv")
    if stats:
        shutil.copy(stats[0], f"gpurun_out/{tag}_kernel_stats.csv")
    traffic = collections.defaultdict(lambda: {"fetch_kb": 0.0, "write_kb": 0.0, "launches": 0})
    for kind in ("fetch", "write"):
        files = glob.glob(f"{out}/{kind}/*/*counter_collection.csv")
        if not files:
            continue
        seen = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            c = cls(r["Kernel_Name"])
            if not c:
                continue
            traffic[c][f"{kind}_kb"] += float(r["Counter_Value"])
            seen[c] += 1
        for c, n in seen.items():
            traffic[c]["launches"] = n
    res = {}
    for c, t in traffic.items():
        n = max(1, t["launches"])
        res[c] = {"launches_profiled": n,
                  "fetch_bytes_per_launch": int(2 * t["fetch_kb"] * 1024 / n),
                  "write_bytes_per_launch": int(t["write_kb"] * 1024 / n)}
        res[c]["bytes_per_launch"] = res[c]["fetch_bytes_per_launch"] + res[c]["write_bytes_per_launch"]
    # the sources these counters were collected on (bench.py refuses the file for any other sources)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash  # noqa: E402
    res["_source_sha"] = source_hash()
    steps_profiled = 3          # collect_profiles.sh: --steps 2 --warmup 1 in the PMC passes
    res["_bytes_per_step"] = int(sum(v["bytes_per_launch"] * v["launches_profiled"] for k, v in res.items()
                                     if isinstance(v, dict)) / steps_profiled)
    json.dump(res, open(f"gpurun_out/{tag}_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))
