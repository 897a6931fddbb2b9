"""TEST INFRASTRUCTURE: oracle-backed stand-ins for `kmc` / `kmc_tools`, used by the CPU tests
of the workflow runner only (the checker plays the engine so that the DAG plumbing can be
exercised without a GPU).  Databases are JSON files named <prefix>.kmc_pre (+ empty .kmc_suf)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kmer_oracle as O  # noqa: E402


def save(prefix, k, db, cmax):
    with open(prefix + ".kmc_pre", "w") as fh:
        json.dump({"k": k, "cmax": cmax, "db": {str(c): n for c, n in db.items()}}, fh)
    open(prefix + ".kmc_suf", "w").close()


def load(prefix):
    with open(prefix + ".kmc_pre") as fh:
        d = json.load(fh)
    return d["k"], {int(c): n for c, n in d["db"].items()}, d["cmax"]


def lines_for(cmax):
    return 255 if cmax <= 255 else 65535
