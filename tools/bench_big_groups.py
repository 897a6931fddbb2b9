#!/usr/bin/env python3
"""The configs[4] shape (groups of 10 / 70 / 200 / 70 / 10 genomes x 1 Mbp, steps 1-8) with the time of every kernel
class: python tools/bench_big_groups.py [k ...]   (GPU; default k = 31)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from khoice_amd import engine as E, synth

ks = [int(a) for a in sys.argv[1:]] or [31]
eng = E.Engine(0)
sizes = [10, 70, 200, 70, 10]
seqs, group_of = [], []
for g, n in enumerate(sizes):
    anc = synth.ancestor(g + 1, 1_000_000)
    for j in range(n):
        seqs.append(synth.clean_text(synth.genome_records(g + 1, j, 1_000_000, anc)))
        group_of.append(g)
dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for t in seqs]
ptrs = [(x.data_ptr(), x.numel()) for x in dev]
for k in ks:
    eng.exp1_run(ptrs, group_of, k, cs=5000, hist_len=5001); eng.sync()
    t0 = time.perf_counter(); eng.exp1_run(ptrs, group_of, k, cs=5000, hist_len=5001); eng.sync(); t1 = time.perf_counter()
    eng.profile(True); eng.stats_reset()
    eng.exp1_run(ptrs, group_of, k, cs=5000, hist_len=5001); eng.sync()
    st = eng.stats(); eng.profile(False)
    print("k", k, "ms", round(1e3 * (t1 - t0), 2), "replans", st["retries"], "big slots", st["big_slots"])
    for n, v in sorted(st["kernels"].items(), key=lambda kv: -kv[1].get("ms", 0)):
        if v["launches"]: print("   ", n, v["launches"], round(v.get("ms", 0), 3))
