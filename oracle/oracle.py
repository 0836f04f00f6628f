"""ctypes binding of the CPU oracle (oracle/wfa_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under wfa_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwfa_oracle.so")

OK, ERR_EMPTY, ERR_TOO_LONG, ERR_INTERNAL = 0, 1, 2, 9
PH_INIT, PH_NEXT, PH_EXTEND, PH_REDUCE = 0, 1, 2, 3
TAGS = {1: "InsOpen", 2: "InsExt", 3: "DelOpen", 4: "DelExt", 5: "Mismatch", 6: "Match"}


class Params(C.Structure):
    _fields_ = [
        ("mismatch", C.c_uint32), ("gap_open", C.c_uint32), ("gap_ext", C.c_uint32),
        ("global_alignment", C.c_int32), ("adaptive", C.c_int32),
        ("min_wf_len", C.c_uint32), ("max_dist_diff", C.c_uint32), ("cutoff_step", C.c_uint32),
    ]


class _Result(C.Structure):
    _fields_ = [
        ("ops", C.POINTER(C.c_uint64)), ("n_ops", C.c_size_t), ("cap_ops", C.c_size_t),
        ("score", C.c_uint32),
        ("tbegin", C.c_int32), ("tend", C.c_int32), ("qbegin", C.c_int32), ("qend", C.c_int32),
        ("align_len", C.c_uint32), ("matches", C.c_uint32), ("gaps", C.c_uint32),
        ("gap_regions", C.c_uint32),
        ("cells", C.c_uint64 * 3), ("lcp_bases", C.c_uint64), ("n_scores", C.c_uint32),
    ]


_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint32)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "wfa_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "wfa_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libwfa_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.wfao_new.restype = C.c_void_p
        L.wfao_new.argtypes = [C.POINTER(Params)]
        L.wfao_free.argtypes = [C.c_void_p]
        L.wfao_set_hook.argtypes = [C.c_void_p, _HOOK, C.c_void_p]
        L.wfao_align.restype = C.c_int
        L.wfao_align.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                 C.POINTER(_Result)]
        L.wfao_result_release.argtypes = [C.POINTER(_Result)]
        L.wfao_cigar.restype = C.c_size_t
        L.wfao_cigar.argtypes = [C.POINTER(_Result), C.c_int, C.c_char_p, C.c_size_t]
        L.wfao_get_wavefront.restype = C.c_int
        L.wfao_get_wavefront.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_size_t]
        L.wfao_num_scores.restype = C.c_uint32
        L.wfao_num_scores.argtypes = [C.c_void_p, C.c_int]
        L.wfao_align_batch.restype = C.c_int
        L.wfao_align_batch.argtypes = [
            C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_uint64, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.POINTER(C.POINTER(C.c_uint64)), C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def make_params(mismatch=4, gap_open=6, gap_ext=2, global_alignment=True,
                adaptive: Optional[Tuple[int, int, int]] = None) -> Params:
    p = Params(mismatch, gap_open, gap_ext, 1 if global_alignment else 0, 0, 0, 0, 0)
    if adaptive is not None:
        p.adaptive = 1
        p.min_wf_len, p.max_dist_diff, p.cutoff_step = adaptive
    return p


def ops_to_cigar(ops: Sequence[int]) -> str:
    return "".join(f"{int(o) & 0xFFFFFFFF}{chr(int(o) >> 32)}" for o in ops)


@dataclass
class Result:
    status: int
    score: int = 0
    ops: List[int] = field(default_factory=list)
    tbegin: int = 0
    tend: int = 0
    qbegin: int = 0
    qend: int = 0
    align_len: int = 0
    matches: int = 0
    gaps: int = 0
    gap_regions: int = 0
    cells: Tuple[int, int, int] = (0, 0, 0)
    lcp_bases: int = 0
    n_scores: int = 0

    @property
    def cigar(self) -> str:
        return ops_to_cigar(self.ops)

    def cigar_trimmed(self) -> str:
        idx = [i for i, o in enumerate(self.ops) if (o >> 32) == ord("M")]
        if not idx:
            return ""
        return ops_to_cigar(self.ops[idx[0]:idx[-1] + 1])

    def key(self):
        """The tuple parity tests compare."""
        return (self.status, self.score, self.cigar, self.qbegin, self.qend, self.tbegin, self.tend,
                self.align_len, self.matches, self.gaps, self.gap_regions)


class Aligner:
    """Oracle aligner; mirrors wfa.New / AdaptiveReduction / Align (wfa.go:120-140,196)."""

    def __init__(self, params: Optional[Params] = None, **kw):
        self.params = params if params is not None else make_params(**kw)
        self._h = lib().wfao_new(C.byref(self.params))
        self._res = _Result()
        self._hook_ref = None

    def close(self):
        if self._h:
            lib().wfao_result_release(C.byref(self._res))
            lib().wfao_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hook(self, fn: Optional[Callable[[int, int], None]]):
        if fn is None:
            self._hook_ref = _HOOK(0)
        else:
            self._hook_ref = _HOOK(lambda ud, phase, s: fn(phase, s))
        lib().wfao_set_hook(self._h, self._hook_ref, None)

    def align(self, q: bytes, t: bytes) -> Result:
        st = lib().wfao_align(self._h, q, len(q), t, len(t), C.byref(self._res))
        if st != OK:
            return Result(status=st)
        r = self._res
        return Result(status=st, score=r.score, ops=[int(r.ops[i]) for i in range(r.n_ops)],
                      tbegin=r.tbegin, tend=r.tend, qbegin=r.qbegin, qend=r.qend,
                      align_len=r.align_len, matches=r.matches, gaps=r.gaps,
                      gap_regions=r.gap_regions, cells=tuple(r.cells), lcp_bases=r.lcp_bases,
                      n_scores=r.n_scores)

    def wavefront(self, comp: int, s: int):
        """(lo, hi, [raw words for k=lo..hi]) or None if (comp, s) has no wavefront."""
        lo, hi = C.c_int32(), C.c_int32()
        if not lib().wfao_get_wavefront(self._h, comp, s, C.byref(lo), C.byref(hi), None, 0):
            return None
        w = max(0, hi.value - lo.value + 1)
        buf = (C.c_uint32 * max(w, 1))()
        lib().wfao_get_wavefront(self._h, comp, s, C.byref(lo), C.byref(hi), buf, w)
        return lo.value, hi.value, [int(buf[i]) for i in range(w)]

    def num_scores(self, comp: int = 0) -> int:
        return lib().wfao_num_scores(self._h, comp)

    def dump(self):
        """All wavefronts as {comp: {s: (lo, hi, raw[])}} (only non-empty entries kept in raw)."""
        out = {}
        for comp, name in enumerate("MID"):
            d = {}
            for s in range(self.num_scores(comp)):
                w = self.wavefront(comp, s)
                if w is not None:
                    d[s] = w
            out[name] = d
        return out


@dataclass
class BatchResult:
    status: np.ndarray
    score: np.ndarray
    tbegin: np.ndarray
    tend: np.ndarray
    qbegin: np.ndarray
    qend: np.ndarray
    align_len: np.ndarray
    matches: np.ndarray
    gaps: np.ndarray
    gap_regions: np.ndarray
    cells: np.ndarray
    ops: np.ndarray
    ops_off: np.ndarray
    ops_len: np.ndarray

    def pair_ops(self, i: int) -> np.ndarray:
        return self.ops[int(self.ops_off[i]):int(self.ops_off[i]) + int(self.ops_len[i])]

    def cigar(self, i: int) -> str:
        return ops_to_cigar(self.pair_ops(i))


def align_batch(params: Params, blob: np.ndarray, q_off, q_len, t_off, t_len, n_threads: int = 1,
                want_ops: bool = True) -> BatchResult:
    """Oracle over a C-ABI-shaped batch (used by parity tests and the cpu_baseline leg)."""
    n = int(len(q_len))
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
    t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
    q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
    t_len = np.ascontiguousarray(t_len, dtype=np.uint32)
    i32 = lambda: np.zeros(n, dtype=np.int32)
    u32 = lambda: np.zeros(n, dtype=np.uint32)
    r = BatchResult(i32(), u32(), i32(), i32(), i32(), i32(), u32(), u32(), u32(), u32(),
                    np.zeros((n, 3), dtype=np.uint64), np.zeros(0, dtype=np.uint64),
                    np.zeros(n, dtype=np.uint64), u32())
    ops_ptr = C.POINTER(C.c_uint64)()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().wfao_align_batch(C.byref(params), p(blob), p(q_off), p(q_len), p(t_off), p(t_len), n,
                           n_threads, p(r.status), p(r.score), p(r.tbegin), p(r.tend), p(r.qbegin),
                           p(r.qend), p(r.align_len), p(r.matches), p(r.gaps), p(r.gap_regions),
                           p(r.cells), C.byref(ops_ptr) if want_ops else None, p(r.ops_off),
                           p(r.ops_len))
    if want_ops:
        total = int(r.ops_len.astype(np.uint64).sum())
        if total:
            r.ops = np.ctypeslib.as_array(ops_ptr, shape=(total,)).copy()
        _free = C.CDLL(None).free
        _free.argtypes = [C.c_void_p]
        _free(C.cast(ops_ptr, C.c_void_p))
    return r
