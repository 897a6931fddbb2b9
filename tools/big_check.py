#!/usr/bin/env python3
"""One-off cross-check above the sizes the test-suite runs: 12 species x 5 genomes x 8 Mbp (480 Mbp, 60 genomes,
more than 256 x 512 slots: the super-k-mer form with 512 coarse buckets) at k = 31 and k = 41 — the super-k-mer
form and the key-array form of kh_exp1_run must agree on every histogram and distinct count; prints both timings.

    python tools/big_check.py          (on the GPU box)
"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from khoice_amd import engine as E, synth
items = synth.species_set(12, 5, 8_000_000)
group_of = [s - 1 for s, _, _ in items]
dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for _, _, t in items]
seqs = [(d.data_ptr(), d.numel()) for d in dev]
eng = E.Engine(0)
eng.profile(True)
os.environ["KHOICE_SKM_DEBUG"] = "1"
for k in (31, 41):
    a = eng.exp1_run(seqs, group_of, k)
    t0 = time.perf_counter(); a = eng.exp1_run(seqs, group_of, k); eng.sync(); t1 = time.perf_counter()
    n_skm = eng.stats()["kernels"]["skm_union"]["launches"]
    os.environ["KHOICE_NO_SKM"] = "1"
    b = eng.exp1_run(seqs, group_of, k)
    t2 = time.perf_counter(); b = eng.exp1_run(seqs, group_of, k); eng.sync(); t3 = time.perf_counter()
    del os.environ["KHOICE_NO_SKM"]
    ok = (a["distinct_per_seq"] == b["distinct_per_seq"]).all() and (a["within_hist"] == b["within_hist"]).all() and (a["across_hist"] == b["across_hist"]).all()
    print("k", k, "skm launches so far", n_skm, "equal", bool(ok), "skm ms %.2f keyarray ms %.2f" % (1e3 * (t1 - t0), 1e3 * (t3 - t2)), "distinct", int(a["distinct_per_seq"].sum()), flush=True)
