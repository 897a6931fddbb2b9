"""Experiment type 4's confusion matrix (feature level) on the MI355X engine.

Host-side mirror of the reference's src/merge_lists.py — same command line, same three output
files, byte for byte — with the text-dump dictionaries (build_dictionary / update_dictionary,
src/merge_lists.py:14-33) replaced by one membership search on the device
(`kh_confusion_row`, include/khoice_hip.h).  Two entries:

  * `main(argv)`                 drop-in for `python3 src/merge_lists.py -p .. -i .. -o .. -n .. -k ..`
                                 (exp_type_4.smk:280-288): reads the `dump -s` text files.
  * `run_from_databases(...)`    skips the dumps and the D x D `simple intersect` databases
                                 (exp_type_4.smk:216-271): pivot databases against the D
                                 rest-of-set unions directly.

Read-level analysis (`-r`, src/merge_lists.py:147-182) breaks ties with `random.choice`, so it
has no reproducible answer; it is refused.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Sequence

import numpy as np


# ------------------------------------------------------------------ matrices (host arithmetic)
def assemble_matrices(rows: Sequence[Sequence[float]], uniques: Sequence[int], num_datasets: int):
    """Rows as `kh_confusion_row` returns them -> the two matrices of src/merge_lists.py:101-145.
    Cells keep the reference's int/float typing (it decides how they print): the matrices start
    as int 0 (:104,108), a cell turns float at its first `+=`; the last column is reset to int 0
    (:139,145); the "regular" matrix then gets 1/num_datasets * unique_pivot_count in every
    column (:140-141), the `_with_unidentified` one does not."""
    cm, cm_ucol = [], []
    for row, unique in zip(rows, uniques):
        touched = [float(x) if x > 0 else 0 for x in row]          # additions are > 0
        reg = []
        for x in touched:
            x += 1 / num_datasets * unique
            reg.append(x)
        cm.append(reg + [0])
        cm_ucol.append(touched + [0])
    return cm, cm_ucol


def calculate_accuracy_values(confusion_matrix, num_datasets: int, k) -> List[list]:
    """[k, pivot, TP, TN, FP, FN] per pivot, cells added in row-major order starting from int 0
    (src/merge_lists.py:35-51)."""
    out = []
    for pivot in range(num_datasets):
        tp = confusion_matrix[pivot][pivot]
        fp = fn = tn = 0
        for r in range(num_datasets):
            for c in range(num_datasets + 1):
                cell = confusion_matrix[r][c]
                if r != pivot and c == pivot:
                    fp += cell
                elif r == pivot and c != pivot:
                    fn += cell
                elif r != pivot:
                    tn += cell
        out.append([k, pivot, tp, tn, fp, fn])
    return out


def format_outputs(cm, cm_ucol, num_datasets: int, k) -> dict:
    """{relative path: text} of the three files src/merge_lists.py:187-210 writes."""
    def matrix(m):
        return "".join(",".join(str(x) for x in row) + "\n" for row in m)
    a, b = calculate_accuracy_values(cm, num_datasets, k), calculate_accuracy_values(cm_ucol, num_datasets, k)
    acc = "".join(",".join(str(x) for x in r1) + "," + ",".join(str(x) for x in r2[2:]) + "\n"
                  for r1, r2 in zip(a, b))
    return {f"confusion_matrix/k_{k}_confusion_matrix.txt": matrix(cm),
            f"confusion_matrix/k_{k}_confusion_matrix_with_unidentified.txt": matrix(cm_ucol),
            f"values/k_{k}_accuracy_values.csv": acc}


def write_outputs(output_path: str, files: dict):
    for rel, text in files.items():
        with open(output_path + rel, "w+") as fh:      # output_path ends in "/" (exp_type_4.smk:285)
            fh.write(text)


# ------------------------------------------------------------------ device side
def confusion_from_sets(eng, pivots: Sequence, sets_per_pivot: Sequence[Sequence], num_datasets: int, k):
    """pivots[p] = KmerSet with counts; sets_per_pivot[p][d] = KmerSet whose members mark
    "pivot k-mer occurs in dataset d"."""
    rows, uniques = [], []
    for pv, sets in zip(pivots, sets_per_pivot):
        if len(sets) != num_datasets:
            raise ValueError("every pivot needs one set per dataset")
        row, unique = eng.confusion_row(pv, list(sets))
        rows.append(row.tolist())
        uniques.append(unique)
    cm, cm_ucol = assemble_matrices(rows, uniques, num_datasets)
    return format_outputs(cm, cm_ucol, num_datasets, k)


def run_from_databases(eng, pivot_prefixes: Sequence[str], union_prefixes: Sequence[str], k,
                       output_path: str) -> dict:
    """Pivot databases (`kmc -ci1` output, exp_type_4.smk:141-153) against the rest-of-set group
    sets (exp_type_4.smk:200-214) without the D x D intersections and without any text dump."""
    unions = [eng.load(p) for p in union_prefixes]
    pivots = [eng.load(p) for p in pivot_prefixes]
    files = confusion_from_sets(eng, pivots, [unions] * len(pivots), len(unions), k)
    write_outputs(output_path, files)
    return files


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _CODE[_ch] = _i


def read_dump(path: str, k: int):
    """`KMER<TAB>count` lines (exp_type_4.smk:255-257; consumer src/merge_lists.py:19-22) ->
    (keys[n, W] uint64 canonical, counts[n] uint32)."""
    w = 1 if k <= 32 else 2
    data = np.fromfile(path, dtype=np.uint8)
    if data.size == 0:
        return np.zeros((0, w), dtype=np.uint64), np.zeros(0, dtype=np.uint32)
    if data[-1] != 10:
        data = np.concatenate([data, np.array([10], dtype=np.uint8)])
    ends = np.flatnonzero(data == 10)
    starts = np.concatenate([[0], ends[:-1] + 1])
    n = starts.size
    keys = np.zeros((n, w), dtype=np.uint64)
    for i in range(k):
        code = _CODE[data[starts + i]]
        if (code > 3).any():
            raise ValueError(f"{path}: symbol outside ACGT in a k-mer of length {k}")
        bit = 2 * (k - 1 - i)
        keys[:, bit // 64] |= code.astype(np.uint64) << np.uint64(bit % 64)
    counts = np.zeros(n, dtype=np.uint64)
    pos = starts + k + 1                     # first digit after the separator
    if (_CODE[data[starts + k]] <= 3).any():
        raise ValueError(f"{path}: k-mers longer than k = {k}")
    live = pos < ends
    while live.any():
        d = data[np.where(live, pos, 0)].astype(np.int64) - 48
        ok = live & (d >= 0) & (d <= 9)
        counts = np.where(ok, counts * np.uint64(10) + d.astype(np.uint64), counts)
        pos = pos + 1
        live = ok & (pos < ends)
    return keys, counts.astype(np.uint32)


def parse_arguments(argv=None):
    """Same options as src/merge_lists.py:212-223."""
    ap = argparse.ArgumentParser(description="experiment type 4: merge k-mer lists into a confusion matrix "
                                             "(MI355X engine)")
    ap.add_argument("-n", "--num", dest="num_datasets", required=True, type=int)
    ap.add_argument("-p", "--pivot_list", dest="pivot_filelist", required=True)
    ap.add_argument("-i", "--intersect_list", dest="intersect_list", required=True)
    ap.add_argument("-o", "--output_path", dest="output_path", required=True)
    ap.add_argument("-k", "--k_value", dest="k", required=True)
    ap.add_argument("-r", "--read-level", dest="read_level", nargs=1)
    ap.add_argument("--device", type=int, default=int(os.environ.get("KHOICE_GPU_DEVICE", "0")))
    return ap.parse_args(argv)


def main(argv=None) -> int:
    args = parse_arguments(argv)
    for f in (args.pivot_filelist, args.intersect_list):
        if not os.path.isfile(f):
            print("Error: One of the provided files is not valid: " + f)
            return 1
    if args.num_datasets <= 0:
        print("Error: The number of datasets needs to be positive integer.")
        return 1
    if args.read_level is not None:
        print("Error: read-level analysis draws random tie-breaks (src/merge_lists.py:180) and is not "
              "reproducible; only the feature level is implemented.", file=sys.stderr)
        return 1
    with open(args.pivot_filelist) as fh:
        pivot_files = [x.strip() for x in fh.readlines()]
    with open(args.intersect_list) as fh:
        intersect_files = [x.strip() for x in fh.readlines()]
    for path in pivot_files + intersect_files:
        if not os.path.isfile(path):
            print(f"Error: At least one of the file paths in the file lists is not valid ({path})")
            return 1
    n, k = args.num_datasets, int(args.k)
    if len(intersect_files) < n * len(pivot_files):
        print("Error: the intersection list needs num_datasets entries per pivot", file=sys.stderr)
        return 1
    from khoice_amd import engine as E
    with E.Engine(args.device) as eng:
        pivots, per_pivot = [], []
        for p, path in enumerate(pivot_files):
            keys, counts = read_dump(path, k)
            pivots.append(eng.upload(k, keys, counts))
            # the reference maps list position -> column with `intersect_num % num_datasets` (:32)
            cols = [None] * n
            for j in range(n):
                pos = p * n + j
                ikeys, _ = read_dump(intersect_files[pos], k)
                cols[pos % n] = eng.upload(k, ikeys, None)
            per_pivot.append(cols)
        files = confusion_from_sets(eng, pivots, per_pivot, n, args.k)
    write_outputs(args.output_path, files)
    return 0


if __name__ == "__main__":
    sys.exit(main())
