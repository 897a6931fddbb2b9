"""Deterministic synthetic "RefSeq-like" genome sets.

No reference code or data is involved: the reference's inputs come from NCBI
(src/download_genomes.py) and cannot be fetched offline, so benchmarks and tests run on
this generator instead.

PRNG: counter-based splitmix64, seed = 0x6B686F696365 ("khoice") ^ (species << 32) ^ genome.
Model
  * every species has an independent uniform ACGT ancestor of length L that carries one
    copy of a global 50 kb "rRNA-like" block diverged by 3 % (so across-group
    histograms have mass above bin 1);
  * genome g of a species = ancestor with 1 % substitutions, 20 indels of 1-50 bp, one
    5 kb segment duplicated, runs of N (1-100 bp) over ~0.01 % of the bases, cut into
    1-3 contigs; upper case; FASTA with 80-column lines.
"""
from __future__ import annotations

import gzip
import os
from typing import List, Tuple

import numpy as np

SEED = 0x6B686F696365
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


class SplitMix:
    """splitmix64 stream; `take(n)` returns the next n outputs as uint64."""

    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def take(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * _GAMMA
            self.state = self.state + np.uint64(n) * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def one(self) -> int:
        return int(self.take(1)[0])

    def below(self, bound: int) -> int:
        return self.one() % bound

    def bases(self, n: int) -> np.ndarray:
        """n uniform base codes 0..3 (32 per draw)."""
        words = self.take((n + 31) // 32)
        shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
        codes = (words[:, None] >> shifts[None, :]) & np.uint64(3)
        return codes.reshape(-1)[:n].astype(np.uint8)


def _substitute(codes: np.ndarray, rate: float, rng: SplitMix) -> np.ndarray:
    r = rng.take(codes.size)
    hit = (r >> np.uint64(40)) < np.uint64(int(rate * (1 << 24)))
    delta = ((r & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.uint8) + np.uint8(1)
    out = codes.copy()
    out[hit] = (out[hit] + delta[hit]) & np.uint8(3)
    return out


def rrna_block(length: int = 50_000) -> np.ndarray:
    return SplitMix(SEED ^ 0x72524E41).bases(length)


def ancestor(species: int, length: int) -> np.ndarray:
    rng = SplitMix(SEED ^ (species << 32))
    codes = rng.bases(length)
    block = rrna_block(min(50_000, max(0, length // 4)))
    if block.size:
        block = _substitute(block, 0.03, rng)
        at = rng.below(length - block.size + 1)
        codes[at:at + block.size] = block
    return codes


def genome_codes(species: int, genome: int, length: int, anc: np.ndarray | None = None) -> np.ndarray:
    """Base codes 0..3, 4 = N."""
    if anc is None:
        anc = ancestor(species, length)
    rng = SplitMix(SEED ^ (species << 32) ^ (genome + 1))
    g = _substitute(anc, 0.01, rng)
    # 20 indels of 1-50 bp
    pieces: List[np.ndarray] = []
    cuts = sorted(rng.below(max(1, g.size)) for _ in range(20))
    prev = 0
    for c in cuts:
        ln = 1 + rng.below(50)
        pieces.append(g[prev:c])
        if rng.below(2):
            pieces.append(rng.bases(ln))          # insertion
            prev = c
        else:
            prev = min(g.size, c + ln)            # deletion
    pieces.append(g[prev:])
    g = np.concatenate(pieces)
    # one 5 kb duplication
    dl = min(5000, g.size // 4)
    if dl:
        src = rng.below(g.size - dl + 1)
        dst = rng.below(g.size + 1)
        g = np.concatenate([g[:dst], g[src:src + dl], g[dst:]])
    # runs of N over ~0.01 % of the bases
    n_runs = max(1, round(1e-4 * g.size / 50))
    for _ in range(n_runs):
        ln = 1 + rng.below(100)
        at = rng.below(max(1, g.size - ln))
        g[at:at + ln] = 4
    return g


_LETTERS = np.frombuffer(b"ACGTN", dtype=np.uint8)


def genome_records(species: int, genome: int, length: int,
                   anc: np.ndarray | None = None) -> List[Tuple[str, bytes]]:
    g = genome_codes(species, genome, length, anc)
    rng = SplitMix(SEED ^ (species << 32) ^ (genome + 1) ^ 0xC0)
    n_contigs = 1 + rng.below(3)
    cuts = sorted({rng.below(max(1, g.size)) for _ in range(n_contigs - 1)})
    bounds = [0] + cuts + [g.size]
    recs = []
    for i in range(len(bounds) - 1):
        seq = _LETTERS[g[bounds[i]:bounds[i + 1]]].tobytes()
        recs.append((f"sp{species}_g{genome}_contig{i + 1} synthetic", seq))
    return recs


def fasta_bytes(records: List[Tuple[str, bytes]], width: int = 80) -> bytes:
    out = bytearray()
    for name, seq in records:
        out += b">" + name.encode() + b"\n"
        for i in range(0, len(seq), width):
            out += seq[i:i + width] + b"\n"
    return bytes(out)


def clean_text(records: List[Tuple[str, bytes]]) -> bytes:
    """What kh_read_fasta yields for fasta_bytes(records): sequences joined by '\\n'."""
    return b"\n".join(seq for _, seq in records)


def write_dataset_tree(root: str, n_species: int, n_genomes: int, length: int) -> None:
    """data/dataset_{s}/{name}.fna.gz as exp_type_1.smk:44-47,158 expects."""
    for s in range(1, n_species + 1):
        d = os.path.join(root, "data", f"dataset_{s}")
        os.makedirs(d, exist_ok=True)
        anc = ancestor(s, length)
        for g in range(n_genomes):
            path = os.path.join(d, f"sp{s}_g{g}.fna.gz")
            with gzip.open(path, "wb", compresslevel=1) as fh:
                fh.write(fasta_bytes(genome_records(s, g, length, anc)))


def write_type4_tree(root: str, n_species: int, n_genomes: int, length: int) -> None:
    """input_type4/{rest_of_set/dataset_{s}/*.fna.gz, pivot/pivot_{s}.fna.gz} as exp_type_4.smk:31-51
    stages them (out-pivot: the pivot genome is NOT among its dataset's rest-of-set genomes)."""
    os.makedirs(os.path.join(root, "input_type4", "pivot"), exist_ok=True)
    for s in range(1, n_species + 1):
        d = os.path.join(root, "input_type4", "rest_of_set", f"dataset_{s}")
        os.makedirs(d, exist_ok=True)
        anc = ancestor(s, length)
        for g in range(n_genomes + 1):
            path = (os.path.join(root, "input_type4", "pivot", f"pivot_{s}.fna.gz") if g == n_genomes
                    else os.path.join(d, f"sp{s}_g{g}.fna.gz"))
            with gzip.open(path, "wb", compresslevel=1) as fh:
                fh.write(fasta_bytes(genome_records(s, g, length, anc)))


def write_type2_tree(root: str, n_species: int, n_genomes: int, length: int) -> None:
    """input_type_2/{rest_of_set/dataset_{s}/*.fna.gz, pivot/dataset_{s}/pivot_{s}.fna.gz}
    (exp_type_2.smk:31-48)."""
    for s in range(1, n_species + 1):
        d = os.path.join(root, "input_type_2", "rest_of_set", f"dataset_{s}")
        pd = os.path.join(root, "input_type_2", "pivot", f"dataset_{s}")
        os.makedirs(d, exist_ok=True)
        os.makedirs(pd, exist_ok=True)
        anc = ancestor(s, length)
        for g in range(n_genomes + 1):
            path = os.path.join(pd, f"pivot_{s}.fna.gz") if g == n_genomes else os.path.join(d, f"sp{s}_g{g}.fna.gz")
            with gzip.open(path, "wb", compresslevel=1) as fh:
                fh.write(fasta_bytes(genome_records(s, g, length, anc)))


def species_set(n_species: int, n_genomes: int, length: int, first_species: int = 1):
    """[(species, genome, cleaned sequence text)] for the device-resident benchmarks."""
    out = []
    for s in range(first_species, first_species + n_species):
        anc = ancestor(s, length)
        for g in range(n_genomes):
            out.append((s, g, clean_text(genome_records(s, g, length, anc))))
    return out


# ------------------------------------------------------------------------------------------------
# "Hard" genomes: what real bacterial chromosomes have and i.i.d. sequence does not — skewed base
# composition, insertion-sequence copies, rRNA operons, tandem repeats, homopolymer runs.  They put many
# instances of the same minimizer into one slot of the super-k-mer form and are what its region / slot
# capacities are tested against (tests/test_gpu_scale.py, tools/bench_hard.py).
def _skewed_bases(rng: SplitMix, n: int, gc: float) -> np.ndarray:
    """n base codes with P(G or C) = gc."""
    r = rng.take(n)
    is_gc = (r >> np.uint64(40)) < np.uint64(int(gc * (1 << 24)))
    low = (r & np.uint64(1)).astype(np.uint8)
    return np.where(is_gc, np.uint8(1) + low, np.uint8(3) * low).astype(np.uint8)   # C/G or A/T


def hard_ancestor(species: int, length: int, gc: float = 0.70) -> np.ndarray:
    rng = SplitMix(SEED ^ 0x48415244 ^ (species << 32))
    codes = _skewed_bases(rng, length, gc)

    def paste(piece, copies, divergence):
        for _ in range(copies):
            if piece.size >= codes.size:
                return
            at = rng.below(codes.size - piece.size)
            codes[at:at + piece.size] = _substitute(piece, divergence, rng) if divergence else piece

    scale = max(1, length // 5_000_000)
    paste(_skewed_bases(rng, min(1500, length // 8), gc), 50 * scale, 0.0)        # an insertion sequence, 50 exact copies
    paste(_skewed_bases(rng, min(5000, length // 8), 0.5), 7 * scale, 0.002)      # rRNA-like operons, nearly identical
    for _ in range(40 * scale):                                                   # tandem repeats: a 2-60 bp unit, 10-200 times
        unit = _skewed_bases(rng, 2 + rng.below(59), gc)
        paste(np.tile(unit, 10 + rng.below(191))[:max(1, length // 16)], 1, 0.0)
    for _ in range(60 * scale):                                                   # homopolymer runs
        paste(np.full(12 + rng.below(60), rng.below(4), dtype=np.uint8), 1, 0.0)
    block = rrna_block(min(50_000, max(0, length // 4)))                          # the block all species share
    if block.size:
        at = rng.below(length - block.size + 1)
        codes[at:at + block.size] = _substitute(block, 0.03, rng)
    return codes


def hard_species_set(n_species: int, n_genomes: int, length: int, gc: float = 0.70):
    """[(species, genome, cleaned sequence text)] like species_set, on hard ancestors."""
    out = []
    for s in range(1, n_species + 1):
        anc = hard_ancestor(s, length, gc)
        for g in range(n_genomes):
            out.append((s, g, clean_text(genome_records(s, g, length, anc))))
    return out
