"""ctypes binding of libkhoice_hip.so (C ABI: include/khoice_hip.h).

The Python layer is orchestration only: every k-mer operation runs in the HIP library.
If the library is missing or no MI355X is visible the calls raise; there is no CPU path.

Method names follow the KMC command they stand in for at khoice's call sites
(workflow/rules/exp_type_1.smk:156-259, exp_type_2.smk:354-380).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkhoice_hip.so")

UNION, INTERSECT, KMERS_SUBTRACT, COUNTERS_SUBTRACT = 0, 1, 2, 3
MODE = {"min": 0, "max": 1, "sum": 2, "diff": 3, "left": 4, "right": 5}
NO_MAX = 0xFFFFFFFF
KMC_DEFAULT_CS = 255


class KhoiceError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"khoice_hip error {code}: {msg}")
        self.code = code


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the engine.  Raises if it was not built (run `python -m khoice_amd.build`)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("KHOICE_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `python -m khoice_amd.build` (needs hipcc). "
            "khoice_amd has no CPU fallback.")
    lib = C.CDLL(p)
    vp, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    sig = {
        "kh_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "kh_ctx_destroy": (None, [vp]),
        "kh_last_error": (C.c_char_p, []),
        "kh_device_count": (C.c_int, []),
        "kh_stats": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
        "kh_profile_enable": (C.c_int, [vp, C.c_int]),
        "kh_stats_reset": (C.c_int, [vp]),
        "kh_sync": (C.c_int, [vp]),
        "kh_trim": (C.c_int, [vp]),
        "kh_build_batch": (C.c_int, [vp, C.c_int, C.POINTER(vp), u64p, C.c_int, C.c_int, C.c_uint32,
                                     C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
        "kh_build_fasta": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(vp)]),
        "kh_read_fasta": (C.c_int, [C.c_char_p, C.POINTER(vp), u64p]),
        "kh_free_host": (None, [vp]),
        "kh_ingest_fasta": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]),
        "kh_seqs_count": (C.c_int, [vp]),
        "kh_seqs_get": (C.c_int, [vp, C.c_int, C.POINTER(vp), u64p]),
        "kh_seqs_free": (None, [vp]),
        "kh_write_histogram_text": (C.c_int, [C.c_char_p, u64p, C.c_uint32, C.c_uint32]),
        "kh_set_counts": (C.c_int, [vp, vp, C.c_uint32, C.POINTER(vp)]),
        "kh_union_sum": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_uint32, C.POINTER(vp), u64p, C.c_uint32]),
        "kh_union_histogram": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_uint32, u64p, C.c_uint32]),
        "kh_simple": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_uint32, C.POINTER(vp)]),
        "kh_histogram": (C.c_int, [vp, vp, u64p, C.c_uint32]),
        "kh_histogram_file": (C.c_int, [vp, vp, C.c_uint32, C.c_char_p]),
        "kh_membership": (C.c_int, [vp, vp, C.POINTER(vp), C.c_int, vp, vp, vp]),
        "kh_confusion_row": (C.c_int, [vp, vp, C.POINTER(vp), C.c_int, C.POINTER(C.c_double), u64p]),
        "kh_table_add_set": (C.c_int, [vp, vp, vp, C.c_uint32]),
        "kh_table_histogram": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, u64p,
                                         C.c_uint32]),
        "kh_dump_sorted": (C.c_int, [vp, vp, C.c_char_p]),
        "kh_set_free": (None, [vp]),
        "kh_set_info": (C.c_int, [vp, u64p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), u32p]),
        "kh_set_counter_max": (C.c_int, [vp, u32p]),
        "kh_set_download": (C.c_int, [vp, vp, vp, vp]),
        "kh_set_upload": (C.c_int, [vp, C.c_int, C.c_uint64, vp, vp, C.POINTER(vp)]),
        "kh_set_device_ptrs": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "kh_set_from_device": (C.c_int, [vp, C.c_int, C.c_uint64, vp, vp, C.POINTER(vp)]),
        "kh_set_export_device": (C.c_int, [vp, vp, vp, vp]),
        "kh_set_export_range": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp, vp]),
        "kh_set_wrap_device": (C.c_int, [vp, C.c_int, C.c_uint64, vp, vp, C.c_uint32, C.POINTER(vp)]),
        "kh_set_partition_bounds": (C.c_int, [vp, vp, C.c_uint32, u64p]),
        "kh_sets_partition_bounds": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_uint32, u64p]),
        "kh_save": (C.c_int, [vp, vp, C.c_char_p]),
        "kh_load": (C.c_int, [vp, C.c_char_p, C.POINTER(vp)]),
        "kh_exp1_run": (C.c_int, [vp, C.c_int, C.POINTER(vp), u64p, C.c_int, C.POINTER(C.c_int), C.c_int,
                                  C.c_int, C.c_uint32, u64p, u64p, C.c_uint32, u64p, C.POINTER(vp),
                                  C.POINTER(vp)]),
        "kh_comm_unique_id": (C.c_int, [C.c_char_p]),
        "kh_comm_init": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]),
        "kh_comm_destroy": (None, [vp]),
        "kh_across_exchange_histogram": (C.c_int, [vp, vp, vp, C.c_uint32, u64p, C.c_uint32]),
        "kh_mix_host": (None, [C.c_int, u64p, u64p]),
        "kh_unmix_host": (None, [C.c_int, u64p, u64p]),
        "kh_skm_exchange_plan": (C.c_int, [vp, C.c_int, C.c_uint64, C.c_uint32, C.c_int, u32p, u32p, u64p]),
        "kh_skm_pack": (C.c_int, [vp, C.c_int, C.POINTER(vp), u64p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_uint32,
                                  C.c_int, C.c_uint64, vp, vp, vp, vp, u64p]),
        "kh_skm_phased_histogram": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                              C.POINTER(vp), C.c_uint32, C.c_uint32, u64p, C.c_uint32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


ABI_SYMBOLS = [
    "kh_ctx_create", "kh_ctx_destroy", "kh_last_error", "kh_device_count", "kh_stats",
    "kh_profile_enable", "kh_stats_reset", "kh_sync", "kh_trim", "kh_build_batch", "kh_build_fasta",
    "kh_read_fasta", "kh_free_host", "kh_ingest_fasta", "kh_seqs_count", "kh_seqs_get", "kh_seqs_free",
    "kh_write_histogram_text", "kh_set_counts", "kh_union_sum", "kh_union_histogram", "kh_simple", "kh_histogram",
    "kh_histogram_file", "kh_membership", "kh_confusion_row",
    "kh_table_add_set", "kh_table_histogram", "kh_dump_sorted", "kh_set_free", "kh_set_info", "kh_set_counter_max",
    "kh_set_download",
    "kh_set_upload", "kh_set_device_ptrs", "kh_set_from_device", "kh_set_export_device",
    "kh_set_export_range", "kh_set_wrap_device", "kh_set_partition_bounds",
    "kh_sets_partition_bounds",
    "kh_save", "kh_load", "kh_exp1_run", "kh_comm_unique_id", "kh_comm_init", "kh_comm_destroy",
    "kh_skm_exchange_plan", "kh_skm_pack", "kh_skm_phased_histogram",
    "kh_across_exchange_histogram", "kh_mix_host", "kh_unmix_host",
]


def _check(rc: int):
    if rc != 0:
        raise KhoiceError(rc, load_library().kh_last_error().decode("utf-8", "replace"))


def words_per_key(k: int) -> int:
    return 1 if k <= 32 else 2


def _u64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class KmerSet:
    """A k-mer database resident in HBM (handle owned by the library)."""

    def __init__(self, engine: "Engine", handle: int):
        self._e = engine
        self._h = C.c_void_p(handle)

    # -- lifetime
    def free(self):
        if self._h:
            self._e._lib.kh_set_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # -- introspection
    def info(self):
        n, k, w, hc, uni = C.c_uint64(), C.c_int(), C.c_int(), C.c_int(), C.c_uint32()
        _check(self._e._lib.kh_set_info(self._h, C.byref(n), C.byref(k), C.byref(w), C.byref(hc), C.byref(uni)))
        return {"n": n.value, "k": k.value, "words": w.value, "has_counts": bool(hc.value),
                "uniform": uni.value}

    def counter_max(self) -> int:
        v = C.c_uint32()
        _check(self._e._lib.kh_set_counter_max(self._h, C.byref(v)))
        return v.value

    def __len__(self):
        return self.info()["n"]

    @property
    def k(self):
        return self.info()["k"]

    def download(self):
        """(keys[n, W] uint64 little-endian words, counts[n] uint32) in storage order."""
        i = self.info()
        keys = np.empty((i["n"], i["words"]), dtype=np.uint64)
        counts = np.empty(i["n"], dtype=np.uint32)
        _check(self._e._lib.kh_set_download(self._e._ctx, self._h, keys.ctypes.data, counts.ctypes.data))
        return keys, counts

    def download_sorted(self):
        """Same, ordered by k-mer (lexicographic A<C<G<T)."""
        keys, counts = self.download()
        if keys.shape[1] == 1:
            order = np.argsort(keys[:, 0], kind="stable")
        else:
            order = np.lexsort((keys[:, 0], keys[:, 1]))
        return keys[order], counts[order]

    def export_device(self, keys_ptr: int, counts_ptr: int):
        """Copy mixed keys / counters into caller-owned device buffers (exchange send side)."""
        _check(self._e._lib.kh_set_export_device(self._e._ctx, self._h, keys_ptr, counts_ptr))

    def export_range(self, lo: int, hi: int, keys_ptr: int, counts_ptr: Optional[int]):
        """Stream-ordered copy of elements [lo, hi); call Engine.sync() before other streams read."""
        _check(self._e._lib.kh_set_export_range(self._e._ctx, self._h, lo, hi, keys_ptr, counts_ptr))

    def device_ptrs(self):
        kp, cp = C.c_void_p(), C.c_void_p()
        _check(self._e._lib.kh_set_device_ptrs(self._h, C.byref(kp), C.byref(cp)))
        return kp.value, cp.value

    # -- kmc_tools transform
    def set_counts(self, value: int) -> "KmerSet":
        out = C.c_void_p()
        _check(self._e._lib.kh_set_counts(self._e._ctx, self._h, value, C.byref(out)))
        return KmerSet(self._e, out.value)

    def histogram(self, hist_len: int) -> np.ndarray:
        h = np.zeros(hist_len, dtype=np.uint64)
        _check(self._e._lib.kh_histogram(self._e._ctx, self._h, _u64p(h), hist_len))
        return h

    def histogram_file(self, cmax: int, path: str):
        _check(self._e._lib.kh_histogram_file(self._e._ctx, self._h, cmax, path.encode()))

    def dump_sorted(self, path: str):
        _check(self._e._lib.kh_dump_sorted(self._e._ctx, self._h, path.encode()))

    def save(self, prefix: str):
        _check(self._e._lib.kh_save(self._e._ctx, self._h, prefix.encode()))

    def partition_bounds(self, nparts: int) -> np.ndarray:
        b = np.zeros(nparts + 1, dtype=np.uint64)
        _check(self._e._lib.kh_set_partition_bounds(self._e._ctx, self._h, nparts, _u64p(b)))
        return b


class DeviceTexts:
    """Cleaned sequence texts resident in HBM (Engine.ingest_fasta); `seqs` is what build_batch /
    exp1_run take: (device pointer, length) pairs.  Owned by the library until free()."""

    def __init__(self, engine: "Engine", handle: int):
        self._e, self._h = engine, C.c_void_p(handle)
        n = engine._lib.kh_seqs_count(self._h)
        self.seqs = []
        for i in range(n):
            p, ln = C.c_void_p(), C.c_uint64()
            _check(engine._lib.kh_seqs_get(self._h, i, C.byref(p), C.byref(ln)))
            self.seqs.append((p.value or 0, ln.value))

    def total_bases(self) -> int:
        return sum(n for _, n in self.seqs)

    def download(self, i: int) -> bytes:
        import ctypes
        ptr, n = self.seqs[i]
        buf = (ctypes.c_ubyte * max(n, 1))()
        hip = ctypes.CDLL("libamdhip64.so")
        if n:
            rc = hip.hipMemcpy(buf, C.c_void_p(ptr), C.c_size_t(n), 2)      # hipMemcpyDeviceToHost
            if rc != 0:
                raise RuntimeError(f"hipMemcpy failed: {rc}")
        return bytes(buf[:n])

    def free(self):
        if self._h:
            self._e._lib.kh_seqs_free(self._h)
            self._h = None
            self.seqs = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One HIP stream on one MI355X."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        ctx = C.c_void_p()
        _check(self._lib.kh_ctx_create(device, C.byref(ctx)))
        self._ctx = ctx
        self.device = device

    def close(self):
        if self._ctx:
            self._lib.kh_ctx_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- misc
    def stats(self) -> dict:
        buf = C.create_string_buffer(8192)
        _check(self._lib.kh_stats(self._ctx, buf, len(buf)))
        return json.loads(buf.value.decode())

    def profile(self, on: bool = True):
        _check(self._lib.kh_profile_enable(self._ctx, 1 if on else 0))

    def stats_reset(self):
        _check(self._lib.kh_stats_reset(self._ctx))

    def sync(self):
        _check(self._lib.kh_sync(self._ctx))

    def trim(self):
        _check(self._lib.kh_trim(self._ctx))

    # -- kmc
    def _seq_args(self, seqs):
        """seqs: list of bytes / uint8 ndarrays (host) or (device_ptr, length) tuples."""
        n = len(seqs)
        ptrs = (C.c_void_p * n)()
        lens = np.zeros(n, dtype=np.uint64)
        keep = []
        on_device = None
        for i, s in enumerate(seqs):
            if isinstance(s, tuple):
                dev = True
                ptrs[i] = C.c_void_p(int(s[0]))
                lens[i] = int(s[1])
            else:
                dev = False
                a = np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray, memoryview)) else \
                    np.ascontiguousarray(s, dtype=np.uint8)
                keep.append(a)
                ptrs[i] = C.c_void_p(a.ctypes.data if a.size else 0)
                lens[i] = a.size
            if on_device is None:
                on_device = dev
            elif on_device != dev:
                raise ValueError("mix of host and device sequences in one batch")
        return ptrs, lens, 1 if on_device else 0, keep

    def build_batch(self, seqs: Sequence, k: int, ci: int = 1, cx: int = NO_MAX,
                    cs: int = KMC_DEFAULT_CS, with_counts: bool = True) -> List[KmerSet]:
        """`kmc -k{k} -ci{ci}` on each cleaned sequence text (exp_type_1.smk:163)."""
        ptrs, lens, on_dev, keep = self._seq_args(seqs)
        n = len(seqs)
        out = (C.c_void_p * n)()
        _check(self._lib.kh_build_batch(self._ctx, n, ptrs, _u64p(lens), on_dev, k, ci, cx, cs,
                                        1 if with_counts else 0, out))
        return [KmerSet(self, out[i]) for i in range(n)]

    def build(self, seq, k: int, **kw) -> KmerSet:
        return self.build_batch([seq], k, **kw)[0]

    def build_fasta(self, path: str, k: int, ci: int = 1, cx: int = NO_MAX,
                    cs: int = KMC_DEFAULT_CS) -> KmerSet:
        out = C.c_void_p()
        _check(self._lib.kh_build_fasta(self._ctx, path.encode(), k, ci, cx, cs, C.byref(out)))
        return KmerSet(self, out.value)

    def ingest_fasta(self, paths: Sequence[str], threads: int = 0) -> DeviceTexts:
        """(gz) FASTA files -> cleaned texts in HBM: parallel inflate into (plain, not page-locked) host buffers on
        `threads` host threads, FASTA cleaning on the device, file by file as they complete."""
        n = len(paths)
        arr = (C.c_char_p * n)(*[p.encode() for p in paths])
        out = C.c_void_p()
        _check(self._lib.kh_ingest_fasta(self._ctx, n, arr, threads, C.byref(out)))
        return DeviceTexts(self, out.value)

    def write_histogram_text(self, path: str, hist: np.ndarray, cmax: int):
        """`kmc_tools transform histogram` text (lines c<TAB>n, c = 1..cmax) from a histogram array."""
        h = np.ascontiguousarray(hist, dtype=np.uint64)
        _check(self._lib.kh_write_histogram_text(path.encode(), _u64p(h), h.size, cmax))

    def read_fasta(self, path: str) -> bytes:
        p, n = C.c_void_p(), C.c_uint64()
        _check(self._lib.kh_read_fasta(path.encode(), C.byref(p), C.byref(n)))
        try:
            return C.string_at(p.value, n.value)
        finally:
            self._lib.kh_free_host(p)

    # -- kmc_tools complex / simple
    def union_sum(self, sets: Sequence[KmerSet], cs: int, hist_len: int = 0):
        """`kmc_tools complex`: (set1 + set2 + ...), -cs{cs} (exp_type_1.smk:52-61,182).
        Returns the set, or (set, histogram) when hist_len > 0."""
        n = len(sets)
        arr = (C.c_void_p * n)(*[s._h for s in sets])
        out = C.c_void_p()
        hist = np.zeros(max(hist_len, 1), dtype=np.uint64)
        _check(self._lib.kh_union_sum(self._ctx, arr, n, cs, C.byref(out),
                                      _u64p(hist) if hist_len else None, hist_len))
        res = KmerSet(self, out.value)
        return (res, hist) if hist_len else res

    def union_histogram(self, sets: Sequence[KmerSet], cs: int, hist_len: int) -> np.ndarray:
        """hist of `kmc_tools complex (set1 + ... ) -csN` + `transform histogram` without keeping
        the union (its slots then run as independent chains)."""
        hist = np.zeros(hist_len, dtype=np.uint64)
        arr = (C.c_void_p * len(sets))(*[s._h for s in sets])
        _check(self._lib.kh_union_histogram(self._ctx, arr, len(sets), cs, _u64p(hist), hist_len))
        return hist

    def simple(self, a: KmerSet, b: KmerSet, op: int, mode: str = "min", cs: Optional[int] = None) -> KmerSet:
        """cs None = no -cs on the command line: the larger counter range of the two operands (what
        bin/kmc_tools does; parity unpinned, see kh_cli.cpp do_simple)."""
        if cs is None:
            cs = max(a.counter_max(), b.counter_max())
        out = C.c_void_p()
        _check(self._lib.kh_simple(self._ctx, a._h, b._h, op, MODE[mode], cs, C.byref(out)))
        return KmerSet(self, out.value)

    def intersect(self, a, b, mode="min", cs=None):
        """`kmc_tools simple A B intersect OUT [-oc<mode>]` (exp_type_2.smk:363-365)."""
        return self.simple(a, b, INTERSECT, mode, cs)

    def kmers_subtract(self, a, b, cs=None):
        """`kmc_tools simple A B kmers_subtract OUT` (exp_type_2.smk:377-379)."""
        return self.simple(a, b, KMERS_SUBTRACT, "left", cs)

    # -- transfer / files
    def upload(self, k: int, keys: np.ndarray, counts: Optional[np.ndarray] = None) -> KmerSet:
        w = words_per_key(k)
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, w)
        cp = None
        if counts is not None:
            counts = np.ascontiguousarray(counts, dtype=np.uint32)
            cp = counts.ctypes.data
        out = C.c_void_p()
        _check(self._lib.kh_set_upload(self._ctx, k, keys.shape[0], keys.ctypes.data, cp, C.byref(out)))
        return KmerSet(self, out.value)

    def from_device(self, k: int, n: int, keys_ptr: int, counts_ptr: Optional[int] = None) -> KmerSet:
        out = C.c_void_p()
        _check(self._lib.kh_set_from_device(self._ctx, k, n, keys_ptr, counts_ptr, C.byref(out)))
        return KmerSet(self, out.value)

    def partition_bounds(self, sets: Sequence[KmerSet], nparts: int) -> np.ndarray:
        """bounds[len(sets), nparts + 1] of several sets with one launch and one synchronisation."""
        n = len(sets)
        out = np.zeros((n, nparts + 1), dtype=np.uint64)
        if n:
            arr = (C.c_void_p * n)(*[s._h for s in sets])
            _check(self._lib.kh_sets_partition_bounds(self._ctx, arr, n, nparts, _u64p(out)))
        return out

    def wrap_device(self, k: int, n: int, keys_ptr: int, counts_ptr: Optional[int] = None,
                    uniform: int = 1) -> KmerSet:
        """Zero-copy set over caller-owned device arrays (kept alive by the caller)."""
        out = C.c_void_p()
        _check(self._lib.kh_set_wrap_device(self._ctx, k, n, keys_ptr, counts_ptr, uniform, C.byref(out)))
        return KmerSet(self, out.value)

    # -- experiment type 4: membership of pivot k-mers in group sets
    def membership(self, pivot: KmerSet, sets: Sequence[KmerSet]):
        """(keys[n, W] canonical ascending, counts[n], masks[n, nwords]) — src/merge_lists.py:14-33
        without the text dumps: bit d of a mask = sets[d] holds the k-mer."""
        n, w = len(pivot), words_per_key(pivot.k)
        nw = max(1, (len(sets) + 63) // 64)
        keys = np.zeros((n, w), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint32)
        masks = np.zeros((n, nw), dtype=np.uint64)
        arr = (C.c_void_p * max(1, len(sets)))(*[s._h for s in sets])
        _check(self._lib.kh_membership(self._ctx, pivot._h, arr, len(sets), keys.ctypes.data,
                                       counts.ctypes.data, masks.ctypes.data))
        return keys, counts, masks

    def confusion_row(self, pivot: KmerSet, sets: Sequence[KmerSet]):
        """(row[len(sets)] float64, unique_pivot_count) as src/merge_lists.py:122-141 adds them up."""
        row = np.zeros(max(1, len(sets)), dtype=np.float64)
        uniq = C.c_uint64(0)
        arr = (C.c_void_p * max(1, len(sets)))(*[s._h for s in sets])
        _check(self._lib.kh_confusion_row(self._ctx, pivot._h, arr, len(sets),
                                          row.ctypes.data_as(C.POINTER(C.c_double)), C.byref(uniq)))
        return row[:len(sets)], int(uniq.value)

    # -- direct-addressed occurrence table (k <= 16)
    def table_add_set(self, s: KmerSet, table_ptr: int, cell_bytes: int):
        """table[v] += 1 (saturating) for every canonical k-mer v of `s`; `table_ptr` = device
        memory of 4^k cells of `cell_bytes` (1 or 4).  Queued on the engine's stream."""
        _check(self._lib.kh_table_add_set(self._ctx, s._h, table_ptr, cell_bytes))

    def table_histogram(self, table_ptr: int, cell_bytes: int, lo: int, hi: int, cs: int,
                        hist_len: int) -> np.ndarray:
        hist = np.zeros(hist_len, dtype=np.uint64)
        _check(self._lib.kh_table_histogram(self._ctx, table_ptr, cell_bytes, lo, hi, cs, _u64p(hist),
                                            hist_len))
        return hist

    # -- C-level RCCL exchange (what a caller without torch.distributed uses)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        _check(self._lib.kh_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, nranks: int, uid: bytes):
        out = C.c_void_p()
        _check(self._lib.kh_comm_init(self._ctx, rank, nranks, uid, C.byref(out)))
        return out

    def comm_destroy(self, comm):
        self._lib.kh_comm_destroy(comm)

    def across_exchange_histogram(self, comm, local_across_set: KmerSet, cs: int, hist_len: int) -> np.ndarray:
        hist = np.zeros(hist_len, dtype=np.uint64)
        _check(self._lib.kh_across_exchange_histogram(self._ctx, comm, local_across_set._h, cs, _u64p(hist), hist_len))
        return hist

    def load(self, prefix: str) -> KmerSet:
        out = C.c_void_p()
        _check(self._lib.kh_load(self._ctx, prefix.encode(), C.byref(out)))
        return KmerSet(self, out.value)

    # -- fused experiment type 1
    def exp1_run(self, seqs: Sequence, group_of: Sequence[int], k: int, cs: int = 5000,
                 hist_len: int = 5001, want_sets: bool = False, across: bool = True,
                 want_across_set: bool = False):
        """Device side of exp_type_1.smk:156-259 for one k.  Returns a dict with
        within_hist[ngroups, hist_len], across_hist[hist_len], distinct_per_seq[nseq]
        (and the group / across sets when want_sets; only the across-group set, counter = groups
        holding the k-mer, when want_across_set).  Without group sets the library takes its fused
        form: one batched build + one tagged union, no per-group database in between."""
        ptrs, lens, on_dev, keep = self._seq_args(seqs)
        n = len(seqs)
        ng = max(group_of) + 1
        gof = (C.c_int * n)(*[int(g) for g in group_of])
        within = np.zeros((ng, hist_len), dtype=np.uint64)
        distinct = np.zeros(n, dtype=np.uint64)
        gsets = (C.c_void_p * ng)()
        aset = C.c_void_p()
        do_across = across
        across = np.zeros(hist_len, dtype=np.uint64)
        _check(self._lib.kh_exp1_run(self._ctx, n, ptrs, _u64p(lens), on_dev, gof, ng, k, cs,
                                     _u64p(within), _u64p(across) if do_across else None, hist_len,
                                     _u64p(distinct), gsets if want_sets else None,
                                     C.byref(aset) if ((want_sets and do_across) or want_across_set) else None))
        res = {"within_hist": within, "across_hist": across if do_across else None,
               "distinct_per_seq": distinct}
        if want_sets:
            res["group_sets"] = [KmerSet(self, gsets[i]) for i in range(ng)]
        if (want_sets and do_across) or want_across_set:
            res["across_set"] = KmerSet(self, aset.value)
        return res

    # -- steps 7-8 across ranks by exchange of minimizer records (khoice_amd/dist.py; 17 <= k <= 32)
    SKM_EXCHANGE_K = (17, 32)

    def skm_exchange_plan(self, k: int, positions_max: int, fan_max: int, nparts: int):
        """(nslots, slots_per_part, part_cap): the slot geometry every rank must use, from numbers the ranks agreed on."""
        ns, spp, cap = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(self._lib.kh_skm_exchange_plan(self._ctx, k, positions_max, fan_max, nparts, C.byref(ns), C.byref(spp),
                                              C.byref(cap)))
        return ns.value, spp.value, cap.value

    def skm_pack(self, seqs: Sequence, tag_of: Sequence[int], k: int, nslots: int, nparts: int, part_cap: int,
                 rec_ptr: int, mask_ptr: int, count_ptr: int, off_ptr: int) -> np.ndarray:
        """This rank's genomes -> records tagged with tag_of[i] (local group, 0..31), identical ones merged, packed by
        owner of their slot into the caller's device buffers.  Returns the number of records per part."""
        ptrs, lens, on_dev, keep = self._seq_args(seqs)
        n = len(seqs)
        tags = (C.c_int * n)(*[int(t) for t in tag_of])
        part_n = np.zeros(nparts, dtype=np.uint64)
        _check(self._lib.kh_skm_pack(self._ctx, n, ptrs, _u64p(lens), on_dev, tags, k, nslots, nparts, part_cap,
                                     rec_ptr, mask_ptr, count_ptr, off_ptr, _u64p(part_n)))
        return part_n

    def skm_phased_histogram(self, k: int, pieces: Sequence, nslots: int, cs: int, hist_len: int) -> np.ndarray:
        """pieces: (rec_ptr, mask_ptr, count_ptr, off_ptr) per source rank, device pointers.  hist[c] = distinct k-mers of
        this rank's slots that occur in c (source, tag) pairs."""
        n = len(pieces)
        arrs = [(C.c_void_p * n)(*[int(pc[j]) for pc in pieces]) for j in range(4)]
        hist = np.zeros(hist_len, dtype=np.uint64)
        _check(self._lib.kh_skm_phased_histogram(self._ctx, k, n, arrs[0], arrs[1], arrs[2], arrs[3], nslots, cs,
                                                 _u64p(hist), hist_len))
        return hist


def mix_host(k: int, key_words: np.ndarray) -> np.ndarray:
    lib = load_library()
    a = np.ascontiguousarray(key_words, dtype=np.uint64)
    o = np.zeros_like(a)
    lib.kh_mix_host(k, _u64p(a), _u64p(o))
    return o


def unmix_host(k: int, key_words: np.ndarray) -> np.ndarray:
    lib = load_library()
    a = np.ascontiguousarray(key_words, dtype=np.uint64)
    o = np.zeros_like(a)
    lib.kh_unmix_host(k, _u64p(a), _u64p(o))
    return o
