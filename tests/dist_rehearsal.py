"""Two-rank rehearsal of the N>1 path on ONE GPU (gloo collectives on host copies, both ranks on
device 0), run by tests/test_gpu_dist_rehearsal.py through torch.distributed.run.  Checks the
global step_8 histogram against the single-process engine result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from khoice_amd import dist as kdist  # noqa: E402
from khoice_amd import engine as E  # noqa: E402
from khoice_amd import synth  # noqa: E402

torch.cuda.init()
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
k, per_rank, L = 31, 3, 120_000
eng = E.Engine(0)
items = synth.species_set(per_rank, 2, L, first_species=1 + rank * per_rank)
seqs = [t for _, _, t in items]
group_of = [s - 1 - rank * per_rank for s, _, _ in items]
got = kdist.exp1_step(eng, seqs, group_of, k, cs=5000, hist_len=64)
if rank == 0:
    allitems = synth.species_set(per_rank * world, 2, L)
    want = eng.exp1_run([t for _, _, t in allitems], [s - 1 for s, _, _ in allitems], k, cs=5000, hist_len=64)
    assert (got["across_hist"] == want["across_hist"]).all(), (got["across_hist"][:6], want["across_hist"][:6])
    assert (got["within_hist"] == want["within_hist"][:per_rank]).all()
    assert int(want["across_hist"][2:].sum()) > 0
    print("REHEARSAL_OK", int(got["across_hist"].sum()))
eng.close()
dist.destroy_process_group()
