#!/usr/bin/env python3
"""BASELINE configs[2] shape WITH the across-group step (steps 1-8, 10 x 10 x 5 Mbp: more than 64 genomes):
time of kh_exp1_run per k.    python tools/bench_cfg3_full.py [k ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from khoice_amd import engine as E, synth
ks = [int(x) for x in sys.argv[1:]] or [31, 41]
items = synth.species_set(10, 10, 5_000_000)
group_of = [s - 1 for s, _, _ in items]
dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for _, _, t in items]
seqs = [(d.data_ptr(), d.numel()) for d in dev]
eng = E.Engine(0)
rows = []
for k in ks:
    eng.exp1_run(seqs, group_of, k); eng.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = eng.exp1_run(seqs, group_of, k); eng.sync(); ts.append(time.perf_counter() - t0)
    eng.stats_reset(); eng.profile(True)
    eng.exp1_run(seqs, group_of, k); eng.sync()
    kern = {n: round(v["ms"], 3) for n, v in eng.stats()["kernels"].items() if v["ms"]}
    eng.profile(False)
    rows.append({"k": k, "ms": round(1e3 * sorted(ts)[1], 3), "across_distinct": int(r["across_hist"][1:].sum()), "kernel_ms": kern})
print(json.dumps({"workload": "10 x 10 x 5 Mbp, steps 1-8", "per_k": rows}))
