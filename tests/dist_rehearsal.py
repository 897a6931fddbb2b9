"""Rehearsal of the N>1 path on ONE GPU (gloo collectives on host copies, all ranks on device 0), run by
tests/test_gpu_dist_rehearsal.py through torch.distributed.run with 2 and 3 ranks.  The global step_8 histogram and
the local step_4 histograms are checked against the single-process engine AND against the C restatement
(oracle/kh_oracle.c), for the exchange of minimizer records (k = 31, 24; with the owner forced into rounds of key
subsets) and for the exchange of key sets (k = 41)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from khoice_amd import dist as kdist  # noqa: E402
from khoice_amd import engine as E  # noqa: E402
from khoice_amd import synth  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

torch.cuda.init()
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per_rank, L = 3, 120_000
eng = E.Engine(0)
items = synth.species_set(per_rank, 2 + rank, L, first_species=1 + rank * per_rank)     # (ranks differ in genomes per group)
seqs = [t for _, _, t in items]
group_of = [s - 1 - rank * per_rank for s, _, _ in items]
allitems = []
for r in range(world):
    allitems += synth.species_set(per_rank, 2 + r, L, first_species=1 + r * per_rank)
allseqs = [t for _, _, t in allitems]
allgroups = [s - 1 for s, _, _ in allitems]
total = 0
for k, env, form in ((31, {}, "records"), (24, {}, "records"), (31, {"KHOICE_SKM_EXCHANGE_MEAN": "3500"}, "records"),
                     (41, {}, "sets"), (31, {"KHOICE_DIST_SET_EXCHANGE": "1"}, "sets")):
    os.environ.update(env)
    eng.profile(True)
    eng.stats_reset()
    got = kdist.exp1_step(eng, seqs, group_of, k, cs=5000, hist_len=64)
    st = eng.stats()
    eng.profile(False)
    for name in env:
        del os.environ[name]
    assert got.get("exchange", "sets") == form, (k, got.get("exchange"))
    if form == "records":       # every rank ran the super-k-mer kernels, nobody built a key set
        kern = st["kernels"]
        assert kern["skm_union"]["launches"] >= 1 and kern["skm_pack"]["launches"] >= 1 and kern["skm_phased"]["launches"] >= 1, kern
        assert kern["union_tagged"]["launches"] == 0, kern
    if rank == 0:
        want = eng.exp1_run(allseqs, allgroups, k, cs=5000, hist_len=64)
        ref = CO.exp1(allseqs, allgroups, k, cs=5000, hist_len=64, nthreads=4)
        assert (want["across_hist"] == ref["across_hist"]).all() and (want["within_hist"] == ref["within_hist"]).all()
        assert (got["across_hist"] == ref["across_hist"]).all(), (k, env, got["across_hist"][:6], ref["across_hist"][:6])
        assert (got["within_hist"] == ref["within_hist"][:per_rank]).all()
        assert (got["distinct_per_seq"] == ref["distinct_per_seq"][:len(seqs)]).all()
        assert int(ref["across_hist"][2:].sum()) > 0
        total += int(got["across_hist"].sum())
    dist.barrier()
if rank == 0:
    print("REHEARSAL_OK", total)
eng.close()
dist.destroy_process_group()
