#!/usr/bin/env python3
"""Condense the two SQ counter passes of tools/pmc_sq.sh into gpurun_out/<tag>_sq_counters.json: per kernel
class and launch, wave-level instruction counts (SQ_INSTS_*) and wave / wait cycles (quad-cycles), tied to the
sources they were measured on (bench.py reads the file only for exactly those sources)."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402
from tools.summarize_profiles import CLASS  # noqa: E402


def cls(name):
    for key, c in CLASS.items():
        if key in name:
            return c
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.Counter())
for sub in ("a", "b"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = cls(r["Kernel_Name"])
            if not c:
                continue
            acc[c][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[c][r["Counter_Name"]] += 1
res = {}
for c, v in acc.items():
    res[c] = {name: int(x / max(1, launches[c][name])) for name, x in v.items()}
    res[c]["launches_profiled"] = max(launches[c].values())
res["_source_sha"] = source_hash()
res["_units"] = "per launch; SQ_INSTS_* count wave instructions, SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves"
json.dump(res, open(f"gpurun_out/{tag}_sq_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
