"""TEST INFRASTRUCTURE ONLY -- brute-force CPU oracle for the khoice k-mer hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product (``khoice_amd``) never does.

What it restates
----------------
khoice has no k-mer arithmetic of its own: every operation is an external call to
KMC 3.2.1 (pinned at ``workflow/envs/khoice_exps.yaml:97``; upstream refresh-bio/KMC,
NOT present in /root/reference or in this container).  This file restates the
*documented* KMC behaviour at the seven call forms khoice uses,
anchored on the reference's own statements of the boundary:

* k-mer windowing ``read[i:i+k]`` for ``i in range(len(read)-k+1)``
  -> ``src/merge_lists.py:53-58`` (``process_read_into_kmers``)
* canonical k-mer = lexicographic min(kmer, reverse complement) over ACGT
  -> ``src/merge_lists.py:60-73`` (``get_canonical_kmer``)
* ``kmc -fm -m64 -k{k} -ci1 IN OUT tmp/`` -> ``workflow/rules/exp_type_1.smk:163``
* ``kmc_tools transform X set_counts 1 Y`` -> ``exp_type_1.smk:173``
* ``kmc_tools complex OPS``; ops grammar written at ``exp_type_1.smk:52-61``
* ``kmc_tools transform X histogram H``; consumer ``exp_type_1.smk:210-212``
* ``kmc_tools simple A B intersect OUT -ocsum`` -> ``exp_type_2.smk:363-365``
* ``kmc_tools simple A B kmers_subtract OUT`` -> ``exp_type_2.smk:377-379``
* ``kmc_tools transform X dump -s OUT``; consumer ``src/merge_lists.py:19-22``

PARITY STATUS: the canonical-form and windowing functions are pinned against golden
vectors captured from the reference's own ``src/merge_lists.py`` (tests/golden/
canonical_kmers.json).  The counting / set-operation semantics are **parity unpinned**:
the reference holds no test, fixture or golden output for them and the KMC binary is
unavailable offline, so they follow KMC's published CLI semantics only.

Representation: a k-mer database is a ``dict[int, int]`` mapping the k-mer's integer
code (A=0,C=1,G=2,T=3, first base most significant, so integer order == lexicographic
order) to its counter.
"""
from __future__ import annotations

import gzip
import io
from typing import Dict, Iterable, List, Sequence

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}
_ALPHA = "ACGT"

KMC_DEFAULT_CS = 255          # kmc -cs default
KMC_DEFAULT_CX = 1_000_000_000


# ----------------------------------------------------------------------------- strings
def reverse_complement(kmer: str) -> str:
    """Reverse complement over ACGT; KeyError on anything else (merge_lists.py:63-67)."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    return "".join(comp[ch] for ch in reversed(kmer))


def canonical_str(kmer: str) -> str:
    """merge_lists.py:60-73: the smaller of the k-mer and its reverse complement."""
    rc = reverse_complement(kmer)
    return kmer if kmer < rc else rc


def windows(read: str, k: int) -> List[str]:
    """merge_lists.py:53-58."""
    return [read[i:i + k] for i in range(0, len(read) - k + 1)]


def encode(kmer: str) -> int:
    v = 0
    for ch in kmer:
        v = (v << 2) | _CODE[ch]
    return v


def decode(code: int, k: int) -> str:
    return "".join(_ALPHA[(code >> (2 * (k - 1 - i))) & 3] for i in range(k))


# ------------------------------------------------------------------------------- FASTA
def read_fasta_bytes(path: str) -> bytes:
    """Return the decompressed bytes of a (possibly gzip'ed) FASTA file."""
    with open(path, "rb") as fh:
        head = fh.read(2)
    if head == b"\x1f\x8b":
        with gzip.open(path, "rb") as fh:
            return fh.read()
    with open(path, "rb") as fh:
        return fh.read()


def fasta_records(data: bytes) -> List[str]:
    """Multi-FASTA (-fm): '>' starts a header line; sequence lines are concatenated;
    k-mers never span records."""
    recs: List[str] = []
    cur: List[str] | None = None
    for line in io.BytesIO(data):
        line = line.rstrip(b"\r\n")
        if line.startswith(b">"):
            if cur is not None:
                recs.append("".join(cur))
            cur = []
        else:
            if cur is None:          # sequence before any header: treat as a record
                cur = []
            cur.append(line.decode("latin-1"))
    if cur is not None:
        recs.append("".join(cur))
    return recs


# ------------------------------------------------------------------------- K1: build
def count_records(records: Iterable[str], k: int, ci: int = 1,
                  cx: int = KMC_DEFAULT_CX, cs: int = KMC_DEFAULT_CS) -> Dict[int, int]:
    """`kmc -k{k} -ci{ci}`: canonical counting; any symbol outside ACGTacgt ends the
    current run; keep ci <= count <= cx; counter saturates at cs."""
    raw: Dict[int, int] = {}
    for rec in records:
        run: List[str] = []
        for ch in rec + "$":                     # '$' flushes the last run
            if ch in _CODE:
                run.append(ch.upper())
                continue
            seq = "".join(run)
            run = []
            for w in windows(seq, k):
                c = encode(canonical_str(w))
                raw[c] = raw.get(c, 0) + 1
    return {c: min(n, cs) for c, n in raw.items() if ci <= n <= cx}


def build(fasta: bytes, k: int, ci: int = 1, cx: int = KMC_DEFAULT_CX,
          cs: int = KMC_DEFAULT_CS) -> Dict[int, int]:
    return count_records(fasta_records(fasta), k, ci, cx, cs)


# ----------------------------------------------------------------- K2..K7: kmc_tools
def set_counts(db: Dict[int, int], value: int) -> Dict[int, int]:
    return {c: value for c in db}


def union_sum(dbs: Sequence[Dict[int, int]], cs: int) -> Dict[int, int]:
    """`complex`: out = (set1 + set2 + ...), OUTPUT_PARAMS -cs{cs}: counters add,
    saturate at cs."""
    out: Dict[int, int] = {}
    for db in dbs:
        for c, n in db.items():
            out[c] = out.get(c, 0) + n
    return {c: min(n, cs) for c, n in out.items()}


_OC = {
    "min": min, "max": max, "sum": lambda a, b: a + b,
    "diff": lambda a, b: a - b, "left": lambda a, b: a, "right": lambda a, b: b,
}


def intersect(a: Dict[int, int], b: Dict[int, int], oc: str = "min",
              cs: int = KMC_DEFAULT_CS) -> Dict[int, int]:
    """`simple A B intersect OUT [-oc<mode>]`; khoice passes -ocsum."""
    f = _OC[oc]
    out = {}
    for c, n in a.items():
        if c in b:
            v = min(f(n, b[c]), cs)
            if v > 0:
                out[c] = v
    return out


def kmers_subtract(a: Dict[int, int], b: Dict[int, int]) -> Dict[int, int]:
    return {c: n for c, n in a.items() if c not in b}


def counters_subtract(a: Dict[int, int], b: Dict[int, int]) -> Dict[int, int]:
    out = {}
    for c, n in a.items():
        v = n - b.get(c, 0)
        if v > 0:
            out[c] = v
    return out


def union2(a: Dict[int, int], b: Dict[int, int], oc: str = "sum",
           cs: int = KMC_DEFAULT_CS) -> Dict[int, int]:
    f = _OC[oc]
    out = dict(a)
    for c, n in b.items():
        out[c] = f(a[c], n) if c in a else n
    return {c: min(n, cs) for c, n in out.items() if n > 0}


def histogram(db: Dict[int, int], cmax: int) -> List[int]:
    """hist[c] for c in 0..cmax (index 0 unused); the text form is lines c<TAB>n
    for c = 1..cmax."""
    h = [0] * (cmax + 1)
    for n in db.values():
        h[min(n, cmax)] += 1
    return h


def histogram_text(db: Dict[int, int], cmax: int) -> str:
    h = histogram(db, cmax)
    return "".join(f"{c}\t{h[c]}\n" for c in range(1, cmax + 1))


def dump_sorted_text(db: Dict[int, int], k: int) -> str:
    return "".join(f"{decode(c, k)}\t{db[c]}\n" for c in sorted(db))


# ------------------------------------------------------------------ complex ops file
def parse_complex_ops(text: str):
    """Grammar written by exp_type_1.smk:52-61.  Returns (inputs: dict name->prefix,
    out_prefix, [names], cs)."""
    section = None
    inputs: Dict[str, str] = {}
    out_prefix = None
    names: List[str] = []
    cs = KMC_DEFAULT_CS
    for raw in text.splitlines():
        line = raw.strip()
        if not line:
            continue
        if line in ("INPUT:", "OUTPUT:", "OUTPUT_PARAMS:"):
            section = line[:-1]
            continue
        if section == "INPUT":
            name, path = [p.strip() for p in line.split("=", 1)]
            inputs[name] = path
        elif section == "OUTPUT":
            out_prefix, expr = [p.strip() for p in line.split("=", 1)]
            expr = expr.strip()
            assert expr.startswith("(") and expr.endswith(")"), expr
            names = [t.strip() for t in expr[1:-1].split("+")]
            names = [t for t in names if t]
        elif section == "OUTPUT_PARAMS":
            for tok in line.split():
                if tok.startswith("-cs"):
                    cs = int(tok[3:])
    return inputs, out_prefix, names, cs
