"""Host-side mirror of the reference's Go API (package wfa) over the libwfahip.so C-ABI.

Same names, argument meaning and error behaviour as the reference (file:line into shenwei356/wfa):

    Penalties / DefaultPenalties                     wfa.go:32-43
    AdaptiveReductionOption / DefaultAdaptiveOption  wfa.go:46-60
    Options / DefaultOptions                         wfa.go:64-71
    New, RecycleAligner                              wfa.go:102-131
    Aligner.AdaptiveReduction                        wfa.go:134-140
    Aligner.Align / AlignPointers                    wfa.go:196-268
    ErrEmptySeq, ErrSeqTooLong, MaxSeqLen            wfa.go:186-193
    AlignmentResult, Op, OpM.., CIGAR, AlignmentText wfa_cigar.go:30-66,236-333
    RecycleAlignmentResult / RecycleAlignmentText    wfa_cigar.go:92,347

New (not in the reference): Aligner.AlignBatch -- the batch entry a GPU needs.  All alignment work
happens in the HIP kernels; this module only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L

MaxSeqLen = L.MAX_SEQ_LEN
MaskLower32 = 4294967295
OpM, OpD, OpI, OpX, OpH = ord("M"), ord("D"), ord("I"), ord("X"), ord("H")


class WfaError(Exception):
    pass


ErrEmptySeq = WfaError("wfa: invalid empty sequence")                                 # wfa.go:187
ErrSeqTooLong = WfaError(f"wfa: sequences longer than {MaxSeqLen} are not supported")  # wfa.go:193


@dataclass
class Penalties:
    Mismatch: int = 4
    GapOpen: int = 6
    GapExt: int = 2


@dataclass
class AdaptiveReductionOption:
    MinWFLen: int = 10
    MaxDistDiff: int = 50
    CutoffStep: int = 1  # not used by the reference either (wfa.go:49)


@dataclass
class Options:
    GlobalAlignment: bool = True


DefaultPenalties = Penalties()
DefaultAdaptiveOption = AdaptiveReductionOption()
DefaultOptions = Options()


def Op(op: int) -> Tuple[str, int]:
    """wfa_cigar.go:57-59: split an op word into (letter, count)."""
    return chr(int(op) >> 32), int(op) & MaskLower32


def trimOps(ops: Sequence[int]) -> List[int]:
    """wfa_cigar.go:217-233 (returns [] where the reference would panic: no M op)."""
    idx = [i for i, o in enumerate(ops) if (int(o) >> 32) == OpM]
    if not idx:
        return []
    return list(ops[idx[0]:idx[-1] + 1])


def _i32(v: int) -> int:  # a record word that holds a signed position
    return v - (1 << 32) if v >= (1 << 31) else v


@dataclass
class AlignmentResult:
    """wfa_cigar.go:30-48.  Ops are op<<32|n, already reversed/merged (process(), :136-214)."""
    Ops: List[int] = field(default_factory=list)
    Score: int = 0
    TBegin: int = 0
    TEnd: int = 0
    QBegin: int = 0
    QEnd: int = 0
    AlignLen: int = 0
    Matches: int = 0
    Gaps: int = 0
    GapRegions: int = 0

    def CIGAR(self, onlyAlignedRegion: bool = False) -> str:  # wfa_cigar.go:236-255
        ops = trimOps(self.Ops) if onlyAlignedRegion else self.Ops
        return "".join(f"{int(o) & MaskLower32}{chr(int(o) >> 32)}" for o in ops)

    def AlignmentText(self, q0: bytes, t0: bytes, onlyAlignedRegion: bool = False) -> Tuple[bytes, bytes, bytes]:
        """wfa_cigar.go:259-333: the three display lines (query, bars, target)."""
        if not onlyAlignedRegion:
            q, t, ops = q0, t0, self.Ops
        else:
            q, t = q0[self.QBegin - 1:self.QEnd], t0[self.TBegin - 1:self.TEnd]
            ops = trimOps(self.Ops)
        Q, A, T = bytearray(), bytearray(), bytearray()
        v = h = 0
        for op in ops:
            letter, n = int(op) >> 32, int(op) & MaskLower32
            if letter == OpM or letter == OpX:
                Q += q[v:v + n]
                A += (b"|" if letter == OpM else b" ") * n
                T += t[h:h + n]
                v += n
                h += n
            elif letter == OpI:
                Q += b"-" * n
                A += b" " * n
                T += t[h:h + n]
                h += n
            elif letter in (OpD, OpH):
                Q += q[v:v + n]
                A += b" " * n
                T += b"-" * n
                v += n
        return bytes(Q), bytes(A), bytes(T)

    def key(self):
        return (0, self.Score, self.CIGAR(False), self.QBegin, self.QEnd, self.TBegin, self.TEnd,
                self.AlignLen, self.Matches, self.Gaps, self.GapRegions)


def RecycleAlignmentResult(r: Optional[AlignmentResult]) -> None:  # wfa_cigar.go:92 (pool return: no-op here)
    return None


def RecycleAlignmentText(Q, A, T) -> None:  # wfa_cigar.go:347
    return None


@dataclass
class Timing:
    kernel_ms: float
    total_ms: float
    n_launches: int
    n_retried_pairs: int
    cells_stored: int
    ops_written: int
    arena_bytes: int
    main_kernel_ms: float
    n_main_launches: int = 0
    n_packed_pairs: int = 0
    main_kernel_kind: int = 0
    ladder_start_level: int = 0  # arena level the long-pair ladder started on (learned hint)


def make_blob(qs: Sequence[bytes], ts: Sequence[bytes]):
    """Flat byte blob + offset/length arrays, the C-ABI input layout (cgo rule: no Go pointers inside)."""
    n = len(qs)
    q_len = np.fromiter((len(x) for x in qs), dtype=np.uint32, count=n)
    t_len = np.fromiter((len(x) for x in ts), dtype=np.uint32, count=n)
    # 16-byte aligned starts: lets the staging loop use aligned dword loads
    q_cap = (q_len.astype(np.uint64) + 15) & ~np.uint64(15)
    t_cap = (t_len.astype(np.uint64) + 15) & ~np.uint64(15)
    tot = q_cap + t_cap
    starts = np.zeros(n, dtype=np.uint64)
    if n > 1:
        starts[1:] = np.cumsum(tot[:-1])
    q_off = starts
    t_off = starts + q_cap
    total = int(tot.sum()) if n else 0
    blob = np.zeros(max(total, 1), dtype=np.uint8)
    for i in range(n):
        blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])] = np.frombuffer(qs[i], dtype=np.uint8)
        blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])] = np.frombuffer(ts[i], dtype=np.uint8)
    return blob, q_off, q_len, t_off, t_len


class Aligner:
    """The aligner object (wfa.go:79-87).  Not safe for concurrent use; one per thread (wfa.go:73-78)."""

    def __init__(self, p: Penalties, opt: Options, device: int = -1):
        self.p = p
        self.opt = opt
        self.ad: Optional[AdaptiveReductionOption] = None
        self._ctx = C.c_void_p()
        self._one = None  # Align's reusable record / ops buffers
        L.check(L.lib().wfahip_create(device, C.byref(self._ctx)), "wfahip_create")

    # -- reference API ---------------------------------------------------------------------------
    def AdaptiveReduction(self, ad: AdaptiveReductionOption) -> Optional[Exception]:
        """wfa.go:134-140: returns an error (not raises) iff MinWFLen == 0, like the Go method."""
        if ad.MinWFLen == 0:
            return WfaError("cutoff step should not be 0")
        self.ad = ad
        return None

    def Align(self, q: bytes, t: bytes) -> AlignmentResult:
        """wfa.go:196: raises ErrEmptySeq / ErrSeqTooLong where the Go method returns them."""
        if len(q) == 0 or len(t) == 0:
            raise ErrEmptySeq
        if len(q) > MaxSeqLen or len(t) > MaxSeqLen:
            raise ErrSeqTooLong
        # wfahip_align_pair: record + ops into two reusable buffers (no arrays to build, none to take apart); what does not
        # change between calls -- the parameter block, the entry point, the by-reference wrappers -- is built once
        # (a third of the Python side of a 150 us call)
        lq, lt = len(q), len(t)
        one = self._one
        if one is None or len(one[1]) < 2 * (lq + lt) + 64:
            rec, ops, n_ops, prm = (C.c_uint32 * L.REC_WORDS)(), (C.c_uint64 * (2 * (lq + lt) + 64))(), C.c_uint64(), self._params()
            one = self._one = (rec, ops, n_ops, C.byref(prm), C.byref(n_ops), L.lib().wfahip_align_pair, prm)
        rec, ops, n_ops, prm_ref, n_ref, entry = one[0], one[1], one[2], one[3], one[4], one[5]
        # The Go aligner reads algn.p / algn.opt / algn.ad on every Align (wfa.go:79-87,134-140): penalties or options
        # re-assigned, or an AdaptiveReductionOption mutated in place, between two calls must show in the second one -- the
        # cached block is REFILLED every call (a few attribute stores; the by-reference wrapper stays)
        prm, p_, ad = one[6], self.p, self.ad
        prm.mismatch, prm.gap_open, prm.gap_ext = p_.Mismatch, p_.GapOpen, p_.GapExt
        prm.global_alignment = 1 if self.opt.GlobalAlignment else 0
        if ad is None:
            prm.adaptive = 0
        else:
            prm.adaptive, prm.min_wf_len, prm.max_dist_diff, prm.cutoff_step = 1, ad.MinWFLen, ad.MaxDistDiff, ad.CutoffStep
        rc = entry(self._ctx, prm_ref, q, lq, t, lt, rec, ops, len(ops), n_ref)
        if rc != 0:
            L.check(rc, "wfahip_align_pair")
        if rec[L.REC_STATUS] != L.PAIR_OK:
            raise ErrSeqTooLong if rec[L.REC_STATUS] == L.PAIR_TOO_LONG else (
                ErrEmptySeq if rec[L.REC_STATUS] == L.PAIR_EMPTY else WfaError("pair could not be aligned (out of device memory)"))
        r = rec[0:10]  # (one slice instead of ten index calls: REC_STATUS .. REC_GAP_REGIONS)
        i32 = _i32
        return AlignmentResult(ops[:n_ops.value], r[L.REC_SCORE], i32(r[L.REC_TBEGIN]), i32(r[L.REC_TEND]), i32(r[L.REC_QBEGIN]),
                               i32(r[L.REC_QEND]), r[L.REC_ALIGN_LEN], r[L.REC_MATCHES], r[L.REC_GAPS], r[L.REC_GAP_REGIONS])

    AlignPointers = Align  # wfa.go:201 (pointer arguments have no Python analogue)

    # -- new: batch entry --------------------------------------------------------------------------
    def _params(self) -> L.Params:
        p = L.Params(self.p.Mismatch, self.p.GapOpen, self.p.GapExt, 1 if self.opt.GlobalAlignment else 0, 0)
        if self.ad is not None:
            p.adaptive = 1
            p.min_wf_len, p.max_dist_diff, p.cutoff_step = self.ad.MinWFLen, self.ad.MaxDistDiff, self.ad.CutoffStep
        return p

    def align_arrays(self, blob, q_off, q_len, t_off, t_len) -> "BatchResult":
        """Batch alignment over the C-ABI layout; returns struct-of-arrays results (numpy copies)."""
        n = int(len(q_len))
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
        t_len = np.ascontiguousarray(t_len, dtype=np.uint32)
        res = L.Results()
        prm = self._params()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.check(L.lib().wfahip_align_batch(self._ctx, C.byref(prm), vp(blob), blob.size, vp(q_off), vp(q_len),
                                           vp(t_off), vp(t_len), n, C.byref(res)), "wfahip_align_batch")
        return _take_results(res, n)

    def align_arrays_packed(self, packed, q_woff, q_len, t_woff, t_len) -> "BatchResult":
        """Batch alignment of pre-packed 2-bit input (include/wfa_hip.h: wfahip_align_batch_packed); pack with
        pack_pairs().  A quarter of the bytes cross PCIe; results are those of align_arrays on the unpacked bytes."""
        n = int(len(q_len))
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        q_woff = np.ascontiguousarray(q_woff, dtype=np.uint64)
        t_woff = np.ascontiguousarray(t_woff, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
        t_len = np.ascontiguousarray(t_len, dtype=np.uint32)
        res = L.Results()
        prm = self._params()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.check(L.lib().wfahip_align_batch_packed(self._ctx, C.byref(prm), vp(packed), packed.size, vp(q_woff), vp(q_len),
                                                  vp(t_woff), vp(t_len), n, C.byref(res)), "wfahip_align_batch_packed")
        return _take_results(res, n)

    # -- new: one pair at a time behind the batch ----------------------------------------------------
    def Submit(self, q: bytes, t: bytes) -> int:
        """Hand in one pair (copied) and return its ticket; Collect() aligns everything submitted as ONE batch.  This is
        how a per-pair loop like the reference's CLI (wfa-go/wfa-go.go:166-178) is served at batch throughput."""
        ticket = C.c_uint64()
        L.check(L.lib().wfahip_submit(self._ctx, q, len(q), t, len(t), C.byref(ticket)), "wfahip_submit")
        return int(ticket.value)

    def Pending(self) -> int:
        return int(L.lib().wfahip_pending(self._ctx))

    def Collect(self):
        """([]*AlignmentResult, []error) of the pairs submitted since the last Collect, indexed by ticket."""
        n = self.Pending()
        res = L.Results()
        prm = self._params()
        L.check(L.lib().wfahip_collect(self._ctx, C.byref(prm), C.byref(res)), "wfahip_collect")
        return _results_and_errors(_take_results(res, n))

    def AlignBatch(self, qs: Sequence[bytes], ts: Sequence[bytes]):
        """([]*AlignmentResult, []error): per-pair results and per-pair errors (None = ok)."""
        if len(qs) != len(ts):
            raise ValueError("qs and ts differ in length")
        if not qs:
            return [], []
        return _results_and_errors(self.align_arrays(*make_blob(qs, ts)))

    # -- diagnostics ----------------------------------------------------------------------------
    def last_timing(self) -> Timing:
        t = L.Timing()
        L.check(L.lib().wfahip_last_timing(self._ctx, C.byref(t)))
        return Timing(t.kernel_ms, t.total_ms, t.n_launches, t.n_retried_pairs, t.cells_stored, t.ops_written,
                      t.arena_bytes, t.main_kernel_ms, t.n_main_launches, t.n_packed_pairs, t.main_kernel_kind, t.ladder_start_level)

    def set_option(self, key: str, value: int) -> None:
        L.check(L.lib().wfahip_set_option(self._ctx, key.encode(), int(value)), f"set_option({key})")

    def debug_team_compact(self, q: bytes, t: bytes):
        """The rows wfa_teamc_kernel leaves in its arena for one pair (include/wfa_hip.h: wfahip_debug_team_compact; option
        team_wgs must be set): ({score: {diagonal: backtrace word}}, AlignmentResult)."""
        rows, words = C.POINTER(L.Row)(), C.POINTER(C.c_uint32)()
        n_rows, n_words = C.c_uint64(), C.c_uint64()
        res = L.Results()
        prm = self._params()
        L.check(L.lib().wfahip_debug_team_compact(self._ctx, C.byref(prm), q, len(q), t, len(t), C.byref(rows),
                                                  C.byref(n_rows), C.byref(words), C.byref(n_words), C.byref(res)),
                "wfahip_debug_team_compact")
        try:
            out = {}
            for i in range(n_rows.value):
                r = rows[i]
                out[int(r.score)] = {r.lo + j: int(words[r.word_off + j]) for j in range(r.width)}
            br = BatchResult.from_c(res, 1)
        finally:
            L.lib().wfahip_free(rows)
            L.lib().wfahip_free(words)
            L.lib().wfahip_results_free(C.byref(res))
        return out, br.result(0)

    def debug_wavefronts(self, q: bytes, t: bytes):
        """All stored M/I/D rows of one alignment: ({'M': {s: {k: raw}}, 'I': .., 'D': ..}, AlignmentResult)."""
        rows, words = C.POINTER(L.Row)(), C.POINTER(C.c_uint32)()
        n_rows, n_words = C.c_uint64(), C.c_uint64()
        res = L.Results()
        prm = self._params()
        L.check(L.lib().wfahip_debug_wavefronts(self._ctx, C.byref(prm), q, len(q), t, len(t), C.byref(rows),
                                                C.byref(n_rows), C.byref(words), C.byref(n_words), C.byref(res)),
                "wfahip_debug_wavefronts")
        try:
            out = {"M": {}, "I": {}, "D": {}}
            for i in range(n_rows.value):
                r = rows[i]
                for ci, name in enumerate("MID"):
                    d = {}
                    for j in range(r.width):
                        w = words[r.word_off + ci * r.width + j]
                        if w:
                            d[r.lo + j] = int(w)
                    if d:
                        out[name][int(r.score)] = d
            br = BatchResult.from_c(res, 1)
        finally:
            L.lib().wfahip_free(rows)
            L.lib().wfahip_free(words)
            L.lib().wfahip_results_free(C.byref(res))
        return out, br.result(0)

    def debug_compact_arena(self, pair: int):
        """(words, fmt, meta) of pair `pair` of the most recent batch: the compact backtrace arena as the first-pass
        forward kernel left it (include/wfa_hip.h: wfahip_debug_compact_arena)."""
        words = C.POINTER(C.c_uint32)()
        n_words, fmt, meta = C.c_uint64(), C.c_uint32(), (C.c_uint32 * 4)()
        L.check(L.lib().wfahip_debug_compact_arena(self._ctx, pair, C.byref(words), C.byref(n_words), C.byref(fmt),
                                                   C.byref(meta)), "wfahip_debug_compact_arena")
        try:
            arr = np.ctypeslib.as_array(words, shape=(int(n_words.value),)).copy()
        finally:
            L.lib().wfahip_free(words)
        return arr, int(fmt.value), tuple(int(x) for x in meta)

    def Plot(self, q: bytes, t: bytes, component: str = "M", notChangeToMatch: bool = False, maxScore: int = -1) -> str:
        """(*Aligner).Plot (wfa_component_plot.go:41): the component table, from the DEVICE wavefronts of one pair."""
        wf, _ = self.debug_wavefronts(q, t)
        return plot_component(q, t, wf, component, self.p, notChangeToMatch, maxScore)

    def close(self):
        if getattr(self, "_ctx", None):
            L.lib().wfahip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _take_results(res: "L.Results", n: int) -> "BatchResult":
    """numpy copies of a wfahip_results, which is then handed back to the library."""
    try:
        return BatchResult.from_c(res, n)
    finally:
        L.lib().wfahip_results_free(C.byref(res))


def _results_and_errors(br: "BatchResult"):
    results: List[Optional[AlignmentResult]] = []
    errors: List[Optional[Exception]] = []
    for i in range(len(br.status)):
        st = int(br.status[i])
        if st == L.PAIR_OK:
            results.append(br.result(i))
            errors.append(None)
        else:
            results.append(None)
            errors.append(ErrEmptySeq if st == L.PAIR_EMPTY else ErrSeqTooLong if st == L.PAIR_TOO_LONG
                          else WfaError("wfa: out of device memory for this pair"))
    return results, errors


def pack_pairs(blob, q_off, q_len, t_off, t_len, n_threads: int = 8):
    """Host-side 2-bit packer (include/wfa_hip.h: wfahip_pack_pairs) -> (packed, q_woff, t_woff).  Raises WfaHipError
    (ERR_UNSUPPORTED) when a byte outside ACGT is found: such input must use the byte entry."""
    n = int(len(q_len))
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
    t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
    q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
    t_len = np.ascontiguousarray(t_len, dtype=np.uint32)
    words = int(((q_len.astype(np.uint64) + 15) // 16 + 1).sum() + ((t_len.astype(np.uint64) + 15) // 16 + 1).sum())
    packed = np.zeros(max(words, 1), dtype=np.uint32)
    q_woff = np.zeros(n, dtype=np.uint64)
    t_woff = np.zeros(n, dtype=np.uint64)
    n_words = C.c_uint64()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(L.lib().wfahip_pack_pairs(vp(blob), vp(q_off), vp(q_len), vp(t_off), vp(t_len), n, n_threads, vp(packed),
                                      vp(q_woff), vp(t_woff), C.byref(n_words)), "wfahip_pack_pairs")
    assert int(n_words.value) == words
    return packed, q_woff, t_woff


class MultiAligner:
    """One aligner over several GPUs (include/wfa_hip.h: wfahip_create_multi): AlignBatch cuts the batch into
    contiguous shards, one per GPU, each aligned from its own host thread; results come back in pair order.  The
    reference's model is one Aligner per goroutine (wfa.go:73-78); this is that, behind one call."""

    def __init__(self, p: Penalties = None, opt: Options = None, devices: Optional[Sequence[int]] = None):
        self.p = p or DefaultPenalties
        self.opt = opt or DefaultOptions
        self.ad: Optional[AdaptiveReductionOption] = None
        self._m = C.c_void_p()
        if devices is None:
            L.check(L.lib().wfahip_create_multi(None, 0, C.byref(self._m)), "wfahip_create_multi")
        else:
            ids = (C.c_int * len(devices))(*devices)
            L.check(L.lib().wfahip_create_multi(ids, len(devices), C.byref(self._m)), "wfahip_create_multi")

    AdaptiveReduction = Aligner.AdaptiveReduction
    _params = Aligner._params

    def size(self) -> int:
        return int(L.lib().wfahip_multi_size(self._m))

    def set_option(self, key: str, value: int) -> None:
        for i in range(self.size()):
            L.check(L.lib().wfahip_set_option(L.lib().wfahip_multi_ctx(self._m, i), key.encode(), int(value)))

    def align_arrays(self, blob, q_off, q_len, t_off, t_len) -> "BatchResult":
        n = int(len(q_len))
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
        t_len = np.ascontiguousarray(t_len, dtype=np.uint32)
        res = L.Results()
        prm = self._params()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.check(L.lib().wfahip_align_batch_multi(self._m, C.byref(prm), vp(blob), blob.size, vp(q_off), vp(q_len),
                                                 vp(t_off), vp(t_len), n, C.byref(res)), "wfahip_align_batch_multi")
        return _take_results(res, n)

    def AlignBatch(self, qs: Sequence[bytes], ts: Sequence[bytes]):
        if len(qs) != len(ts):
            raise ValueError("qs and ts differ in length")
        if not qs:
            return [], []
        return _results_and_errors(self.align_arrays(*make_blob(qs, ts)))

    def close(self):
        if getattr(self, "_m", None):
            L.lib().wfahip_destroy_multi(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


WFA_ARROWS = "⊕⟼🠦↧🠧⬂⬊"  # wfa_backtrace_types.go:39: no type, InsOpen, InsExt, DelOpen, DelExt, Mismatch, Match


def plot_component(q: bytes, t: bytes, wavefronts: dict, component: str = "M", penalties: Penalties = None,
                   notChangeToMatch: bool = False, maxScore: int = -1) -> str:
    """The text table of (*Aligner).Plot (wfa_component_plot.go:41-209) for one component, built from the stored
    wavefront words {'M': {score: {k: raw}}, 'I': .., 'D': ..} (raw = offset<<3 | type) -- the form
    Aligner.debug_wavefronts returns from the device and the oracle's dump has.

    Every cell shows the first (lowest) score that reached it and the arrow of its type; with the M component,
    cells reached by extension become matches and only the cell a run started from keeps its type
    (wfa_component_plot.go:97-99,133-177)."""
    p = penalties or DefaultPenalties
    lenQ, lenT = len(q), len(t)
    M, I, D = wavefronts["M"], wavefronts["I"], wavefronts["D"]
    comp = wavefronts[component]
    isM = component == "M"
    m = [[-1] * lenT for _ in range(lenQ)]

    def after_diff(c, s, diff, k):  # Component.GetAfterDiff (wfa_component.go:158-167): 0 when missing
        if diff > s:
            return 0
        return c.get(s - diff, {}).get(k, 0) >> 3

    vp = hp = 0  # the reference declares them outside the loops (wfa_component_plot.go:66)
    for s in sorted(comp):
        if maxScore >= 0 and s > maxScore:
            break
        row = comp[s]
        for k in sorted(row):
            raw = row[k]
            offset, typ = raw >> 3, raw & 7
            h = offset - 1
            v = h - k
            if v < 0 or h < 0 or v >= lenQ or h >= lenT:
                continue
            if m[v][h] >= 0:  # recorded with a lower score
                continue
            m[v][h] = (s << 3) | typ
            if not isM or q[v] != t[h]:
                continue
            if typ == 2:  # InsExt
                offset0 = max(after_diff(M, s, p.GapOpen + p.GapExt, k - 1), after_diff(I, s, p.GapExt, k - 1)) + 1
            elif typ == 4:  # DelExt
                offset0 = max(after_diff(M, s, p.GapOpen + p.GapExt, k + 1), after_diff(D, s, p.GapExt, k + 1))
            else:
                isk = max(after_diff(M, s, p.GapOpen + p.GapExt, k - 1), after_diff(I, s, p.GapExt, k - 1)) + 1
                dsk = max(after_diff(M, s, p.GapOpen + p.GapExt, k + 1), after_diff(D, s, p.GapExt, k + 1))
                offset0 = max(isk, dsk, after_diff(M, s, p.Mismatch, k) + 1)
            h00 = offset0 - 1
            if h == h00:  # not extended at all
                continue
            v0, h0 = v, h
            if not notChangeToMatch:
                m[v0][h0] = (s << 3) | 6
            n = 0
            while True:
                h -= 1
                v -= 1
                if v < 0 or h < 0:
                    break
                n += 1
                if m[v][h] >= 0:
                    continue
                m[v][h] = (s << 3) | (typ if notChangeToMatch else 6)
                vp, hp = v, h
                if q[v] != t[h] or h == h00:
                    break
            if n == 0:
                vp, hp = v0, h0
            if not notChangeToMatch:
                m[vp][hp] = (s << 3) | typ  # the run's first cell keeps the original type
    out = ["   \t " + "".join("\t%3d" % (h + 1) for h in range(lenT)),
           "   \t " + "".join("\t%3s" % chr(b) for b in t)]
    for v in range(lenQ):
        cells = "".join("\t  ." if c < 0 else "\t%s%2d" % (WFA_ARROWS[c & 7], c >> 3) for c in m[v])
        out.append("%3d\t%s%s" % (v + 1, chr(q[v]), cells))
    return "\n".join(out) + "\n"


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
    ops_off: np.ndarray
    ops_len: np.ndarray
    ops: np.ndarray

    @staticmethod
    def from_c(res: "L.Results", n: int) -> "BatchResult":
        def arr(ptr, dt, cnt):
            if cnt == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dt, copy=True)
        return BatchResult(arr(res.status, np.int32, n), arr(res.score, np.uint32, n), arr(res.tbegin, np.int32, n),
                           arr(res.tend, np.int32, n), arr(res.qbegin, np.int32, n), arr(res.qend, np.int32, n),
                           arr(res.align_len, np.uint32, n), arr(res.matches, np.uint32, n),
                           arr(res.gaps, np.uint32, n), arr(res.gap_regions, np.uint32, n),
                           arr(res.ops_off, np.uint64, n), arr(res.ops_len, np.uint32, n),
                           arr(res.ops, np.uint64, int(res.n_ops)))

    def pair_ops(self, i: int) -> np.ndarray:
        return self.ops[int(self.ops_off[i]):int(self.ops_off[i]) + int(self.ops_len[i])]

    def result(self, i: int) -> AlignmentResult:
        return AlignmentResult(Ops=[int(o) for o in self.pair_ops(i)], Score=int(self.score[i]),
                               TBegin=int(self.tbegin[i]), TEnd=int(self.tend[i]), QBegin=int(self.qbegin[i]),
                               QEnd=int(self.qend[i]), AlignLen=int(self.align_len[i]),
                               Matches=int(self.matches[i]), Gaps=int(self.gaps[i]),
                               GapRegions=int(self.gap_regions[i]))


def New(p: Penalties = DefaultPenalties, opt: Options = DefaultOptions, device: int = -1) -> Aligner:
    """wfa.go:120."""
    return Aligner(p, opt, device)


def RecycleAligner(algn: Optional[Aligner]) -> None:
    """wfa.go:102: returns the aligner to the pool; here it releases the device context."""
    if algn is not None:
        algn.close()


def generate_pairs(seed: int, n_pairs: int, length: int, error_rate: float, first_index: int = 0,
                   n_threads: int = 8):
    """Seeded synthetic dataset (include/wfa_hip.h: wfahip_generate_pairs).  Host only, no GPU needed."""
    stride = int(L.lib().wfahip_gen_stride(length, error_rate))
    blob = np.zeros(max(n_pairs * stride, 1) + 16, dtype=np.uint8)
    q_off = np.zeros(n_pairs, dtype=np.uint64)
    t_off = np.zeros(n_pairs, dtype=np.uint64)
    q_len = np.zeros(n_pairs, dtype=np.uint32)
    t_len = np.zeros(n_pairs, dtype=np.uint32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if n_pairs:
        L.check(L.lib().wfahip_generate_pairs(seed, first_index, n_pairs, length, float(error_rate), n_threads,
                                              vp(blob), vp(q_off), vp(q_len), vp(t_off), vp(t_len)),
                "wfahip_generate_pairs")
    return blob, q_off, q_len, t_off, t_len


def generate_pairs_device(ctx_owner: "Aligner", seed: int, n_pairs: int, length: int, error_rate: float, first_index: int = 0):
    """The same dataset generated on the aligner's GPU (include/wfa_hip.h: wfahip_generate_pairs_device): returns torch
    tensors (blob u8, q_off i64, q_len i32, t_off i64, t_len i32) resident in HBM; nothing crosses PCIe.  (The buffers
    are torch's: torch.cuda must have been initialised before the first aligner of the process was created -- torch
    ships its own HIP runtime, and it does not find the GPUs once the library's runtime has come up first.)"""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    stride = int(L.lib().wfahip_gen_stride(length, error_rate))
    blob = torch.zeros(max(n_pairs * stride, 1) + 16, dtype=torch.uint8, device=dev)
    q_off = torch.zeros(n_pairs, dtype=torch.int64, device=dev)
    t_off = torch.zeros(n_pairs, dtype=torch.int64, device=dev)
    q_len = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    t_len = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)  # (the fills run on torch's stream, the generator on the library's: not ordered otherwise)
    L.check(L.lib().wfahip_generate_pairs_device(ctx_owner._ctx, seed, first_index, n_pairs, length, float(error_rate),
                                                 blob.data_ptr(), q_off.data_ptr(), q_len.data_ptr(), t_off.data_ptr(),
                                                 t_len.data_ptr(), None), "wfahip_generate_pairs_device")
    return blob, q_off, q_len, t_off, t_len
