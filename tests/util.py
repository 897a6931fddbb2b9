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


def multiset_text(keys, counts, k):
    """(keys[n, W] uint64 sorted by k-mer, counts[n]) -> the sorted "KMER<TAB>count<LF>" text of a dump -s."""
    alpha = "ACGT"
    out = []
    for i in range(keys.shape[0]):
        code = int(keys[i, 0]) | ((int(keys[i, 1]) << 64) if keys.shape[1] == 2 else 0)
        out.append("".join(alpha[(code >> (2 * (k - 1 - j))) & 3] for j in range(k)) + "\t" + str(int(counts[i])) + "\n")
    return "".join(out)


def check_multiset_case(case, keys, counts, cs):
    """One case of tests/golden/kmer_multiset.json (the reference's own get_canonical_kmer over
    process_read_into_kmers, Counter'ed) against an implementation's sorted (keys, counts); `cs` is the counter
    ceiling the implementation ran with (it must not be reached unless the case says so)."""
    import hashlib
    k = case["k"]
    assert keys.shape[0] == case["distinct"], (case["kind"], k, keys.shape[0], case["distinct"])
    if case["max_count"] <= cs:
        assert int(counts.sum()) == case["total"]
        text = multiset_text(keys, counts, k)
        assert hashlib.sha256(text.encode()).hexdigest() == case["sha256"], (case["kind"], k, len(case["seq"]))
        if "multiset" in case:
            assert text == "".join(f"{km}\t{c}\n" for km, c in case["multiset"])
    else:   # saturating counters (KMC's documented -cs; not a reference-pinned rule): the k-mer set still has to agree
        assert "multiset" not in case or [km for km, _ in case["multiset"]] == [ln.split("\t")[0] for ln in multiset_text(keys, counts, k).splitlines()]
