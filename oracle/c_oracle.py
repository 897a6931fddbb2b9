"""ctypes wrapper of oracle/_build/libkh_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Databases are returned as (keys[n, W] uint64, counts[n] uint32) sorted by k-mer."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkh_oracle.so")


class _Db(C.Structure):
    _fields_ = [("n", C.c_uint64), ("kmers", C.c_uint64), ("k", C.c_int), ("w", C.c_int),
                ("keys", C.POINTER(C.c_uint64)), ("counts", C.POINTER(C.c_uint32))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_SO)
        _lib.kho_count.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_Db)]
        _lib.kho_union_sum.argtypes = [C.POINTER(C.POINTER(_Db)), C.c_int, C.c_uint32, C.POINTER(_Db)]
        _lib.kho_simple.argtypes = [C.POINTER(_Db), C.POINTER(_Db), C.c_int, C.c_int, C.c_uint32, C.POINTER(_Db)]
        _lib.kho_histogram.argtypes = [C.POINTER(_Db), C.POINTER(C.c_uint64), C.c_uint32]
        _lib.kho_set_counts.argtypes = [C.POINTER(_Db), C.c_uint32]
        _lib.kho_free.argtypes = [C.POINTER(_Db)]
        _lib.kho_exp1.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int),
                                  C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.c_uint32, C.POINTER(C.c_uint64), C.c_int]
    return _lib


class Db:
    def __init__(self):
        self.d = _Db()

    def __del__(self):
        try:
            lib().kho_free(C.byref(self.d))
        except Exception:
            pass

    def __len__(self):
        return self.d.n

    @property
    def kmers(self):
        return self.d.kmers

    def arrays(self):
        n, w = self.d.n, self.d.w
        if n == 0:
            return np.zeros((0, max(w, 1)), dtype=np.uint64), np.zeros(0, dtype=np.uint32)
        keys = np.ctypeslib.as_array(self.d.keys, shape=(n, w)).copy()
        counts = np.ctypeslib.as_array(self.d.counts, shape=(n,)).copy()
        return keys, counts

    def set_counts(self, v):
        lib().kho_set_counts(C.byref(self.d), v)
        return self

    def histogram(self, length):
        h = np.zeros(length, dtype=np.uint64)
        lib().kho_histogram(C.byref(self.d), h.ctypes.data_as(C.POINTER(C.c_uint64)), length)
        return h


def count(seq: bytes, k: int, ci=1, cx=0xFFFFFFFF, cs=255) -> Db:
    out = Db()
    buf = np.frombuffer(seq, dtype=np.uint8)
    rc = lib().kho_count(buf.ctypes.data if buf.size else None, buf.size, k, ci, cx, cs, C.byref(out.d))
    if rc:
        raise RuntimeError(f"kho_count failed: {rc}")
    return out


def union_sum(dbs, cs) -> Db:
    out = Db()
    arr = (C.POINTER(_Db) * len(dbs))(*[C.pointer(d.d) for d in dbs])
    rc = lib().kho_union_sum(arr, len(dbs), cs, C.byref(out.d))
    if rc:
        raise RuntimeError(f"kho_union_sum failed: {rc}")
    return out


def simple(a: Db, b: Db, op: int, mode: int, cs=255) -> Db:
    out = Db()
    rc = lib().kho_simple(C.byref(a.d), C.byref(b.d), op, mode, cs, C.byref(out.d))
    if rc:
        raise RuntimeError(f"kho_simple failed: {rc}")
    return out


def exp1(seqs, group_of, k, cs=5000, hist_len=5001, nthreads=0):
    n = len(seqs)
    ng = max(group_of) + 1
    bufs = [np.frombuffer(s, dtype=np.uint8) for s in seqs]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_uint64 * n)(*[b.size for b in bufs])
    gof = (C.c_int * n)(*group_of)
    within = np.zeros((ng, hist_len), dtype=np.uint64)
    across = np.zeros(hist_len, dtype=np.uint64)
    distinct = np.zeros(n, dtype=np.uint64)
    p64 = C.POINTER(C.c_uint64)
    used = lib().kho_exp1(n, ptrs, lens, gof, ng, k, cs, within.ctypes.data_as(p64), across.ctypes.data_as(p64),
                          hist_len, distinct.ctypes.data_as(p64), nthreads)
    if used < 0:
        raise RuntimeError("kho_exp1 failed")
    return {"within_hist": within, "across_hist": across, "distinct_per_seq": distinct, "threads": used}
