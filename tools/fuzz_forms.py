#!/usr/bin/env python3
"""Randomised cross-check of the two forms of the fused path: for random genome sets (lengths 0 .. 300 kbp, related
and unrelated genomes, N runs, repeats, low-complexity stretches, 1 .. 90 genomes in 1 .. 12 groups, now and
then two groups of 65 .. 140 genomes) and random
k in 17 .. 63, kh_exp1_run must give the same histograms and distinct counts in the super-k-mer form and with
KHOICE_NO_SKM=1 (key arrays).  GPU only, no oracle: python tools/fuzz_forms.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from khoice_amd import engine as E

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = E.Engine(0)


def dna(n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(s, rate):
    t = list(s)
    for _ in range(int(len(t) * rate)):
        t[rng.randrange(len(t))] = rng.choice("ACGTN" if rng.random() < 0.02 else "ACGT")
    return "".join(t)


bad = 0
skm_runs = 0
for it in range(cases):
    ngroups = rng.randint(1, 12)
    seqs, group_of = [], []
    shared = dna(rng.randint(0, 3000))
    big = rng.random() < 0.15          # a case with groups wider than the 64-genome mask (sub-batches / phases)
    for g in range(ngroups):
        L = rng.choice([0, 10, 200, 5000, 40_000, 120_000, 300_000])
        if big: L = min(L, 40_000)
        anc = dna(L) if L else ""
        for j in range(rng.randint(65, 140) if big and g < 2 and L else rng.randint(1, 9)):
            s = mutate(anc, rng.choice([0.0, 0.001, 0.01, 0.1])) if anc else ""
            r = rng.random()
            if r < 0.15: s += "\n" + shared
            elif r < 0.25 and len(s) > 4000: s += "\n" + s[1000:3000]
            elif r < 0.30: s += "A" * rng.randint(1, 20000)
            elif r < 0.35: s += ("AC" * rng.randint(1, 5000))
            elif r < 0.40: s = s[: len(s) // 2] + "N" * rng.randint(1, 200) + s[len(s) // 2:]
            seqs.append(s.encode())
            group_of.append(g)
    order = list(range(len(seqs)))
    rng.shuffle(order)
    seqs = [seqs[i] for i in order]
    group_of = [group_of[i] for i in order]
    k = rng.randint(17, 63)
    cs = rng.choice([1, 2, 7, 5000])
    hl = rng.choice([2, 9, 300, 5001])
    across = rng.random() < 0.8
    eng.profile(True)
    before = eng.stats()["kernels"]["skm_union"]["launches"]
    a = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hl, across=across)
    skm_runs += eng.stats()["kernels"]["skm_union"]["launches"] > before
    eng.profile(False)
    os.environ["KHOICE_NO_SKM"] = "1"
    b = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hl, across=across)
    del os.environ["KHOICE_NO_SKM"]
    ok = (a["distinct_per_seq"] == b["distinct_per_seq"]).all() and (a["within_hist"] == b["within_hist"]).all()
    if across:
        ok = ok and (a["across_hist"] == b["across_hist"]).all()
    if not ok:
        bad += 1
        print("MISMATCH case", it, "k", k, "genomes", len(seqs), "groups", ngroups, "cs", cs, "hist_len", hl, flush=True)
print(f"{cases} cases, {skm_runs} took the super-k-mer form, {bad} mismatches")
sys.exit(1 if bad else 0)
