"""Shared helpers for the parity tests (oracle dict <-> engine arrays)."""
import numpy as np

MASK64 = (1 << 64) - 1


def words(k):
    return 1 if k <= 32 else 2


def db_to_arrays(db, k):
    """oracle dict{int code: count} -> (keys[n, W] uint64 sorted by k-mer, counts[n])."""
    w = words(k)
    codes = sorted(db)
    keys = np.zeros((len(codes), w), dtype=np.uint64)
    for i, c in enumerate(codes):
        keys[i, 0] = c & MASK64
        if w == 2:
            keys[i, 1] = c >> 64
    counts = np.array([db[c] for c in codes], dtype=np.uint32)
    return keys, counts


def set_to_db(kset):
    keys, counts = kset.download()
    if keys.shape[1] == 1:
        codes = [int(x) for x in keys[:, 0]]
    else:
        codes = [int(lo) | (int(hi) << 64) for lo, hi in zip(keys[:, 0], keys[:, 1])]
    db = dict(zip(codes, (int(c) for c in counts)))
    assert len(db) == len(codes), "engine returned duplicate keys"
    return db


def random_dna(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))
