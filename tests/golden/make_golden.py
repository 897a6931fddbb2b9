#!/usr/bin/env python3
"""Generate tests/golden/*.json|csv by RUNNING the reference's own pure-Python code.

Runs only in the build container (needs /root/reference); the GPU box and the test
suite read the committed fixtures, never the reference.  Nothing from the reference
is copied: the fixtures hold inputs and the outputs the reference computed for them.

Sources executed:
  * workflow/rules/exp_type_1.smk:26-84    parse-time writer of the `kmc_tools complex` files
  * workflow/rules/exp_type_1.smk:115-150  summarize_histogram_type1
  * workflow/rules/exp_type_1.smk:199-231  body of rule within_group_union_analysis
  * workflow/rules/exp_type_1.smk:268-297  body of rule across_group_union_analysis
  * workflow/rules/exp_type_2.smk:171-216  summarize_histogram_type2
  * workflow/rules/exp_type_2.smk:404-438  body of rule within_group_analysis_exp_type2
  * workflow/rules/exp_type_2.smk:521-554  body of rule across_group_analysis_exp_type2
  * src/merge_lists.py                     get_canonical_kmer, process_read_into_kmers
  * src/merge_lists.py:53-73 composed      kmer_multiset.json: Counter of get_canonical_kmer(w) over
                                           process_read_into_kmers(s, k) — the multiset of canonical k-mers
                                           of an ACGT string by the reference's own two functions
  * src/merge_lists.py main()              feature-level confusion matrix + accuracy values
                                           (merge_lists.json; its text-dump inputs are written
                                           by OUR oracle, the three output files by the reference)
"""
import importlib.util
import json
import os
import random
import sys
import tempfile
import textwrap

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _lines(path, lo, hi):
    with open(path) as fh:
        src = fh.readlines()
    return textwrap.dedent("".join(src[lo - 1:hi]))


def load_reference():
    ns1, ns2 = {}, {}
    exec(_lines(f"{REF}/workflow/rules/exp_type_1.smk", 115, 150), ns1)
    exec(_lines(f"{REF}/workflow/rules/exp_type_2.smk", 171, 216), ns2)
    spec = importlib.util.spec_from_file_location("ref_merge_lists", f"{REF}/src/merge_lists.py")
    ml = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ml)
    return ns1["summarize_histogram_type1"], ns2["summarize_histogram_type2"], ml


def rand_hist(rng, length, n_members, total_scale):
    """A histogram shaped like a union-of-sets histogram: mass in bins 1..n_members."""
    h = [0] * length
    for i in range(min(n_members, length)):
        h[i] = rng.randrange(0, total_scale) if rng.random() < 0.85 else 0
    if rng.random() < 0.3:
        h[0] = rng.randrange(total_scale, 10 * total_scale)
    if sum(h) == 0:
        h[0] = 1
    return h


def gen_type1(s1, rng):
    cases = []
    for length in (255, 5000, 65535):
        for n_members in (1, 2, 4, 5, 10, 100):
            for across in (False, True):
                for _ in range(3):
                    k = rng.choice([7, 15, 21, 27, 30, 31, 34, 41, 49, 63])
                    h = rand_hist(rng, length, n_members if not across else 30,
                                  rng.choice([10, 1000, 5_000_000]))
                    try:
                        m = s1(list(h), n_members, across, k)
                    except AssertionError:
                        continue
                    nz = [(i, v) for i, v in enumerate(h) if v]
                    cases.append({"len": length, "nonzero": nz, "n_members": n_members,
                                  "across": across, "k": k, "metrics": m})
    # edge: everything in bin 1; saturated last bin
    for length in (255, 65535):
        h = [0] * length
        h[0] = 123456
        cases.append({"len": length, "nonzero": [(0, 123456)], "n_members": 5,
                      "across": False, "k": 31, "metrics": s1(h, 5, False, 31)})
        h = [0] * length
        h[0] = 10
        h[length - 1] = 7
        cases.append({"len": length, "nonzero": [(0, 10), (length - 1, 7)], "n_members": 5,
                      "across": False, "k": 31, "metrics": s1(h, 5, False, 31)})
    return cases


def gen_type2(s2, rng):
    cases = []
    for length in (255, 65535):
        for n in (2, 4, 5, 10, 50):
            for across in (False, True):
                for _ in range(3):
                    k = rng.choice([7, 21, 31, 41])
                    sub = [0] * length
                    sub[0] = rng.randrange(1, 5_000_000)
                    inter = [0] * length
                    for i in range(1, min(n + 1, length)):
                        inter[i] = rng.randrange(0, 1_000_000)
                    try:
                        m = s2(list(sub), list(inter), n, across, k)
                    except AssertionError:
                        continue
                    cases.append({"len": length, "sub0": sub[0],
                                  "inter_nonzero": [(i, v) for i, v in enumerate(inter) if v],
                                  "n": n, "across": across, "k": k, "metrics": m})
    return cases


def gen_canonical(ml, rng):
    out = {"canonical": [], "windows": []}
    for k in (1, 2, 7, 15, 21, 27, 30, 31, 32, 33, 41, 63, 64):
        for _ in range(12):
            s = "".join(rng.choice("ACGT") for _ in range(k))
            out["canonical"].append([s, ml.get_canonical_kmer(s)])
        if k % 2 == 0:   # even-k reverse-complement palindromes
            half = "".join(rng.choice("ACGT") for _ in range(k // 2))
            comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
            pal = half + "".join(comp[c] for c in reversed(half))
            out["canonical"].append([pal, ml.get_canonical_kmer(pal)])
    for s in ("A" * 31, "T" * 31, "ACGT" * 8, "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTG"):
        out["canonical"].append([s, ml.get_canonical_kmer(s)])
    for L, k in ((0, 3), (2, 3), (3, 3), (10, 4), (40, 31), (70, 63)):
        s = "".join(rng.choice("ACGT") for _ in range(L))
        out["windows"].append([s, k, ml.process_read_into_kmers(s, k)])
    return out


def gen_kmer_multiset(ml, rng):
    """The multiset of canonical k-mers of ACGT strings by the reference's own functions only:
    collections.Counter(ml.get_canonical_kmer(w) for w in ml.process_read_into_kmers(s, k)).
    Small cases carry the whole multiset, large ones its size, its total and the sha256 of its sorted
    "KMER<TAB>count<LF>" text (the text `kmc_tools transform dump -s` would print for an unsaturated database)."""
    import collections
    import hashlib
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def dna(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def make(kind, k, n):
        if kind == "random":
            return dna(n)
        if kind == "repeat":              # a unit repeated: counters above 1 (above 255 for the long ones)
            unit = dna(rng.choice([1, 2, 3, 5, 8, 13]))
            return (unit * (n // len(unit) + 1))[:n]
        if kind == "palindromes":         # even k: reverse-complement palindromes between random stretches
            out = ""
            while len(out) < n:
                half = dna(k // 2)
                out += dna(rng.randrange(0, 7)) + half + "".join(comp[c] for c in reversed(half))
            return out[:max(n, k)]
        if kind == "both_strands":        # a stretch followed by its reverse complement: every k-mer meets its partner
            half = dna(n // 2)
            return half + "".join(comp[c] for c in reversed(half))
        if kind == "duplicated":          # a segment copied inside the string
            s = dna(n)
            a = rng.randrange(0, max(1, n // 2))
            return s + s[a:a + n // 3]
        raise ValueError(kind)

    cases = []
    for k in (7, 15, 21, 31, 32, 33, 41, 63):
        shapes = [("random", 0), ("random", k - 1), ("random", k), ("random", k + 1), ("random", 100), ("random", 1000),
                  ("random", 5000), ("repeat", 400), ("repeat", 3000), ("both_strands", 600), ("duplicated", 900)]
        if k % 2 == 0:
            shapes += [("palindromes", 300)]
        for kind, n in shapes:
            s = make(kind, k, n)
            cnt = collections.Counter(ml.get_canonical_kmer(w) for w in ml.process_read_into_kmers(s, k))
            items = sorted(cnt.items())
            text = "".join(f"{km}\t{c}\n" for km, c in items)
            case = {"k": k, "kind": kind, "seq": s, "distinct": len(items), "total": sum(cnt.values()),
                    "max_count": max(cnt.values()) if cnt else 0, "sha256": hashlib.sha256(text.encode()).hexdigest()}
            if len(text) <= 4000:
                case["multiset"] = items
            cases.append(case)
    return {"cases": cases}


def gen_csv(s1, rng):
    """Run the bodies of the two `run:` blocks on synthetic histogram files."""
    k_values = ["7", "21", "31"]
    num_datasets = 3
    members = {1: 1, 2: 5, 3: 10}
    smk = f"{REF}/workflow/rules/exp_type_1.smk"
    within_body = _lines(smk, 199, 231)
    across_body = _lines(smk, 268, 297)
    hists = {}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            within_inputs, across_inputs = [], []
            for k in k_values:
                for num in range(1, num_datasets + 1):
                    p = f"step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt"
                    os.makedirs(os.path.dirname(p), exist_ok=True)
                    h = rand_hist(rng, 5000, members[num], 3_000_000)
                    hists[p] = [(i, v) for i, v in enumerate(h) if v]
                    with open(p, "w") as fh:
                        fh.write("".join(f"{i + 1}\t{v}\n" for i, v in enumerate(h)))
                    within_inputs.append(p)
            # the expand() order of exp_type_1.smk:195 is k-major? expand iterates the
            # product with the LAST keyword varying fastest: k_len outer, num inner.
            for k in k_values:
                p = f"step_8/k_{k}/all_datasets_k{k}_hist.txt"
                os.makedirs(os.path.dirname(p), exist_ok=True)
                h = rand_hist(rng, 5000, num_datasets, 9_000_000)
                hists[p] = [(i, v) for i, v in enumerate(h) if v]
                with open(p, "w") as fh:
                    fh.write("".join(f"{i + 1}\t{v}\n" for i, v in enumerate(h)))
                across_inputs.append(p)
            os.makedirs("step_5")
            os.makedirs("step_9")
            ns = {"summarize_histogram_type1": s1, "num_datasets": num_datasets,
                  "get_num_of_dataset_members": lambda d: members[int(d)],
                  "input": within_inputs, "output": ["step_5/within_datasets_analysis.csv"]}
            exec(within_body, ns)
            ns.update(input=across_inputs, output=["step_9/across_datasets_analysis.csv"])
            exec(across_body, ns)
            within_csv = open("step_5/within_datasets_analysis.csv").read()
            across_csv = open("step_9/across_datasets_analysis.csv").read()
        finally:
            os.chdir(cwd)
    return {"k_values": k_values, "num_datasets": num_datasets,
            "members": {str(a): b for a, b in members.items()}, "hist_len": 5000,
            "hists": hists, "within_inputs": within_inputs, "across_inputs": across_inputs,
            "within_csv": within_csv, "across_csv": across_csv}


def gen_csv2(s2, rng):
    """Run the bodies of exp_type_2's two analysis `run:` blocks on synthetic histogram files."""
    k_values = ["9", "21", "31"]
    num_datasets = 3
    members = {1: 2, 2: 5, 3: 12}
    smk = f"{REF}/workflow/rules/exp_type_2.smk"
    within_body = _lines(smk, 404, 438)
    across_body = _lines(smk, 521, 554)
    files = {}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            inputs = {"within": [], "across": []}
            for scope, top in (("within", "within_dataset_results_type_2"), ("across", "across_dataset_results_type_2")):
                for num in range(1, num_datasets + 1):          # exp_type_2.smk:153-169 order
                    for k in k_values:
                        n_bins = members[num] if scope == "within" else num_datasets - 1
                        total = rng.randrange(1000, 4_000_000)
                        sub = [rng.randrange(0, total)] + [0] * 254
                        inter = [0] * 255
                        for c in range(1, n_bins + 1):          # -ocsum counters are 1 + occurrences
                            inter[c] = rng.randrange(0, total // n_bins + 1)
                        for op, h in (("subtract", sub), ("intersect", inter)):
                            p = f"{top}/k_{k}/dataset_{num}/{op}/dataset_{num}_pivot_{op}_group.hist.txt"
                            os.makedirs(os.path.dirname(p), exist_ok=True)
                            text = "".join(f"{i + 1}\t{v}\n" for i, v in enumerate(h))
                            open(p, "w").write(text)
                            files[p] = [(i, v) for i, v in enumerate(h) if v]
                            inputs[scope].append(p)
            os.makedirs("within_dataset_analysis_type_2")
            os.makedirs("across_dataset_analysis_type_2")
            ns = {"summarize_histogram_type2": s2, "num_datasets": num_datasets,
                  "get_num_of_dataset_members_exp2": lambda d: members[int(d)],
                  "input": inputs["within"], "output": ["within_dataset_analysis_type_2/within_dataset_analysis.csv"]}
            exec(within_body, ns)
            ns.update(input=inputs["across"], output=["across_dataset_analysis_type_2/across_dataset_analysis.csv"])
            exec(across_body, ns)
            within_csv = open("within_dataset_analysis_type_2/within_dataset_analysis.csv").read()
            across_csv = open("across_dataset_analysis_type_2/across_dataset_analysis.csv").read()
        finally:
            os.chdir(cwd)
    return {"k_values": k_values, "num_datasets": num_datasets, "members": {str(a): b for a, b in members.items()},
            "hist_len": 255, "hists": files, "within_inputs": inputs["within"], "across_inputs": inputs["across"],
            "within_csv": within_csv, "across_csv": across_csv}


def gen_complex_ops():
    """Run the parse-time section exp_type_1.smk:26-84 in a scratch WORK_ROOT."""
    body = _lines(f"{REF}/workflow/rules/exp_type_1.smk", 26, 84)
    k_values, num_datasets = ["7", "31"], 3
    listing = {"1": ["only.fna.gz"], "2": ["b.fna.gz", "a.fna.gz", "notes.txt"],
               "3": ["g3.fna.gz", "g1.fna.gz", "g2.fna.gz", "g0.fna.gz"]}
    files = {}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            for num, names in listing.items():
                os.makedirs(f"data/dataset_{num}")
                for n in names:
                    open(f"data/dataset_{num}/{n}", "w").close()
            real = os.listdir
            fake_os = type("FakeOs", (), {})()
            for attr in dir(os):
                setattr(fake_os, attr, getattr(os, attr))
            fake_os.listdir = lambda p: list(listing[os.path.basename(os.path.normpath(p)).split("_")[1]]) \
                if os.path.basename(os.path.normpath(p)).startswith("dataset_") else real(p)
            exec(body, {"exp_type": 1, "k_values": k_values, "num_datasets": num_datasets, "os": fake_os})
            for d, _, names in os.walk("complex_ops"):
                for n in names:
                    files[os.path.join(d, n)] = open(os.path.join(d, n)).read()
        finally:
            os.chdir(cwd)
    return {"k_values": k_values, "num_datasets": num_datasets, "listing": listing, "files": files}


def _mutate(rng, seq, rate):
    t = list(seq)
    for i in range(len(t)):
        if rng.random() < rate:
            t[i] = rng.choice("ACGT")
    return "".join(t)


def gen_merge_lists(ml, rng):
    """Run the reference's merge_lists.main (feature level) on exp_type_4-shaped text dumps."""
    import argparse
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import kmer_oracle as O
    cases = []
    for num_datasets, k, length in [(2, 5, 500), (3, 11, 700), (4, 21, 800), (3, 33, 700), (5, 7, 400)]:
        shared = "".join(rng.choice("ACGT") for _ in range(length // 3))
        pivots, rests = [], []
        for d in range(num_datasets):
            anc = shared + "".join(rng.choice("ACGT") for _ in range(length - len(shared)))
            rests.append([_mutate(rng, anc, 0.03) + "N" + _mutate(rng, anc[:60], 0.0) for _ in range(2)])
            pivots.append(_mutate(rng, anc, 0.05))
        pivot_dbs = [O.count_records([p], k) for p in pivots]                     # kmc -ci1 (counts kept)
        unions = [O.set_counts(O.union_sum([O.set_counts(O.count_records([g], k), 1) for g in gs], 5000), 1)
                  for gs in rests]                                                # exp_type_4.smk:160-214
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(f"{tmp}/out/confusion_matrix")
            os.makedirs(f"{tmp}/out/values")
            pivot_paths, inter_paths, pivot_txt, inter_txt = [], [], [], []
            for p in range(num_datasets):
                path = f"{tmp}/pivot_{p + 1}.txt"
                txt = O.dump_sorted_text(pivot_dbs[p], k)
                open(path, "w").write(txt)
                pivot_paths.append(path)
                pivot_txt.append(txt)
                for d in range(num_datasets):                                      # exp_type_4.smk:216-230
                    ipath = f"{tmp}/pivot_{p + 1}_intersect_dataset_{d + 1}.txt"
                    itxt = O.dump_sorted_text(O.intersect(unions[d], pivot_dbs[p], "sum", 255), k)
                    open(ipath, "w").write(itxt)
                    inter_paths.append(ipath)
                    inter_txt.append(itxt)
            open(f"{tmp}/pivots.txt", "w").write("".join(x + "\n" for x in pivot_paths))
            open(f"{tmp}/inters.txt", "w").write("".join(x + "\n" for x in inter_paths))
            ns = argparse.Namespace(num_datasets=num_datasets, pivot_filelist=f"{tmp}/pivots.txt",
                                    intersect_list=f"{tmp}/inters.txt", output_path=f"{tmp}/out/",
                                    k=str(k), read_level=None)
            ml.args = ns            # calculate_accuracy_values reads the module-level `args`
            ml.main(ns)
            outs = {name: open(f"{tmp}/out/{name}").read() for name in (
                f"confusion_matrix/k_{k}_confusion_matrix.txt",
                f"confusion_matrix/k_{k}_confusion_matrix_with_unidentified.txt",
                f"values/k_{k}_accuracy_values.csv")}
        cases.append({"k": k, "num_datasets": num_datasets, "pivots": pivots, "rest_of_set": rests,
                      "pivot_dumps": pivot_txt, "intersection_dumps": inter_txt, "outputs": outs})
    return {"cases": cases}


def main():
    s1, s2, ml = load_reference()
    rng = random.Random(0x6B686F696365)
    out = {
        "summarize_type1.json": gen_type1(s1, rng),
        "summarize_type2.json": gen_type2(s2, rng),
        "canonical_kmers.json": gen_canonical(ml, rng),
        "exp1_csv.json": gen_csv(s1, rng),
        "complex_ops.json": gen_complex_ops(),
        "merge_lists.json": gen_merge_lists(ml, random.Random(0x6D65726765)),
        "exp2_csv.json": gen_csv2(s2, random.Random(0x74797065)),
        "kmer_multiset.json": gen_kmer_multiset(ml, random.Random(0x6D756C7469)),
    }
    for name, obj in out.items():
        with open(os.path.join(HERE, name), "w") as fh:
            json.dump(obj, fh, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")
    print("python", sys.version.split()[0])


if __name__ == "__main__":
    main()
