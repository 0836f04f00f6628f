"""A second, independent CPU restatement of the reference's alignment path, in plain Python.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): tests/test_oracle_crosscheck.py diffs it against the C oracle
(oracle/wfa_oracle.c) on thousands of small random pairs, so that the oracle the HIP path is held to does not rest on one
reading of the reference alone.  It was written from the Go sources, not from the C restatement, and is built
differently on purpose: wavefronts are dictionaries keyed by diagonal (the reference uses zig-zag indexed slices), a
component is a dictionary keyed by score, scores are Python ints masked to 32 bits where the reference's uint32
arithmetic can wrap.  Pure Python loops: small cases only.

Citations are file:line into shenwei356/wfa v0.4.0.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

INS_OPEN, INS_EXT, DEL_OPEN, DEL_EXT, MISMATCH, MATCH = 1, 2, 3, 4, 5, 6  # wfa_backtrace_types.go:27-35
OPS = ".IIDDXMH"                                                           # wfa_backtrace_types.go:37
U32 = 0xFFFFFFFF
MAX_SEQ_LEN = (1 << 29) - 1                                                # wfa.go:190


class WaveFront:
    """wfa_wavefront.go:45-183.  Lo/Hi are maintained exactly as Set / Increase / Delete do it."""

    def __init__(self):
        self.lo, self.hi = 1 << 62, -(1 << 62)     # math.MaxInt / math.MinInt (:52-53)
        self.raw: Dict[int, int] = {}

    def _touch(self, k):
        if k < self.lo:
            self.lo = k
        if k > self.hi:
            self.hi = k

    def set(self, k, offset, typ):                 # :85-104
        self.raw[k] = (offset << 3) | typ
        self._touch(k)

    def increase(self, k, delta):                  # :131-150
        self.raw[k] = self.raw.get(k, 0) + (delta << 3)
        self._touch(k)

    def get(self, k) -> Tuple[int, int, bool]:     # :153-159
        if k < self.lo or k > self.hi:
            return 0, 0, False
        r = self.raw.get(k, 0)
        return r >> 3, r & 7, r > 0

    def get_raw(self, k) -> Tuple[int, bool]:      # :162-168
        if k < self.lo or k > self.hi:
            return 0, False
        r = self.raw.get(k, 0)
        return r, r > 0

    def delete(self, k):                           # :171-183
        if k < self.lo or k > self.hi:
            return
        self.raw[k] = 0
        if k == self.hi:
            self.hi -= 1
        elif k == self.lo:
            self.lo += 1


class Component:
    """wfa_component.go:37-187: a score-indexed table that starts with 2048 slots and grows by 2048 when a Set lands
    beyond it (:104-106); reads beyond the table find nothing (:82,95,141,151,163)."""

    def __init__(self):
        self.n = 2048
        self.wf: Dict[int, WaveFront] = {}

    def has_score(self, s):                        # :81-86
        return s < self.n and s in self.wf

    def krange(self, s, diff):                     # :91-101
        if diff > s:
            return 0, 0
        s -= diff
        if s >= self.n or s not in self.wf:
            return 0, 0
        w = self.wf[s]
        return w.lo, w.hi

    def set(self, s, k, offset, typ):              # :104-115
        if s >= self.n:
            self.n += 2048
        if s >= self.n:
            raise IndexError("the reference panics here: score beyond the grown table")
        self.wf.setdefault(s, WaveFront()).set(k, offset, typ)

    def get(self, s, k):                           # :139-144
        if s >= self.n or s not in self.wf:
            return 0, 0, False
        return self.wf[s].get(k)

    def get_raw(self, s, k):                       # :149-154
        if s >= self.n or s not in self.wf:
            return 0, False
        return self.wf[s].get_raw(k)

    def get_after_diff(self, s, diff, k):          # :158-167
        if diff > s:
            return 0, 0, False
        return self.get(s - diff, k)

    def delete(self, s, k):                        # :182-187
        if s >= self.n or s not in self.wf:
            return
        self.wf[s].delete(k)


@dataclass
class Result:
    status: int = 0            # 0 ok, 1 ErrEmptySeq, 2 ErrSeqTooLong
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

    @property
    def cigar(self) -> str:    # wfa_cigar.go:236-255 (all ops)
        return "".join(f"{o & U32}{chr(o >> 32)}" for o in self.ops)

    def key(self):
        return (self.status, self.score, self.cigar, self.qbegin, self.qend, self.tbegin, self.tend, self.align_len,
                self.matches, self.gaps, self.gap_regions)


class Aligner:
    def __init__(self, mismatch=4, gap_open=6, gap_ext=2, global_alignment=True,
                 adaptive: Optional[Tuple[int, int, int]] = None):
        self.x, self.o, self.e = mismatch, gap_open, gap_ext
        self.glob = global_alignment
        self.ad = adaptive                         # (MinWFLen, MaxDistDiff, CutoffStep) or None (wfa.go:134-140)
        self.M = self.I = self.D = None

    # ---- wfa.go:143-184
    def _init(self, q, t):
        self.M, self.I, self.D = Component(), Component(), Component()
        m, n = len(t), len(q)
        if q[0] == t[0]:
            self.M.set(0, 0, 1, MATCH)
        else:
            self.M.set(self.x, 0, 1, MISMATCH)
        if not self.glob:
            for k in range(1, m):                  # first row
                if q[0] == t[k]:
                    self.M.set(0, k, k + 1, MATCH)
                else:
                    self.M.set(self.x, k, k + 1, MISMATCH)
            for k in range(1, n):                  # first column
                if q[k] == t[0]:
                    self.M.set(0, -k, 1, MATCH)
                else:
                    self.M.set(self.x, -k, 1, MISMATCH)

    # ---- wfa.go:196-268
    def align(self, q: bytes, t: bytes) -> Result:
        m, n = len(t), len(q)
        if n == 0 or m == 0:
            return Result(status=1)
        if n > MAX_SEQ_LEN or m > MAX_SEQ_LEN:
            return Result(status=2)
        self._init(q, t)
        Ak, Aoffset = m - n, m
        s = 0
        reduce_on = self.ad is not None
        while True:
            if self.M.has_score(s):
                lo, hi = self._extend(q, t, s)
                offset, _, _ = self.M.get_after_diff(s, 0, Ak)
                if offset >= Aoffset:
                    break
                if reduce_on and hi - lo + 1 >= self.ad[0]:
                    self._reduce(q, t, s)
            s += 1
            self._next(q, t, s)
        min_s, last_k = s, Ak
        if not self.glob:
            min_s, last_k = self._start(q, t, s)
        return self._backtrace(q, t, min_s, last_k)

    # ---- wfa.go:381-458: the block loop + byte loop add up to the longest common prefix of q[v:], t[h:]
    def _extend(self, q, t, s):
        wf = self.M.wf[s]
        lo, hi = wf.lo, wf.hi
        n, m = len(q), len(t)
        for k in range(hi, lo - 1, -1):
            offset, _, ok = wf.get(k)
            if not ok:
                continue
            h = offset
            v = h - k
            if v <= 0 or v >= n or h >= m:
                continue
            N = 0
            while v < n and h < m and q[v] == t[h]:
                v += 1
                h += 1
                N += 1
            if N:
                wf.increase(k, N)
        return lo, hi

    # ---- wfa.go:461-540
    def _reduce(self, q, t, s):
        wf = self.M.wf[s]
        lo, hi = wf.lo, wf.hi
        n, m = len(q), len(t)
        ds = []
        min_dist = 1 << 62
        for k in range(lo, hi + 1):
            offset, _, ok = wf.get(k)
            if not ok:
                ds.append(-1)
                continue
            h = offset
            v = h - k
            if v < 0 or v >= n or h >= m:
                ds.append(-1)
                continue
            d = max(m - h, n - v)
            ds.append(d)
            if d < min_dist:
                min_dist = d
        _lo, _hi = lo, hi
        update_lo, found = True, False
        for i, d in enumerate(ds):
            if d < 0:
                continue
            if d - min_dist > self.ad[1]:
                found = True
                if update_lo:
                    _lo = lo + i + 1
                ds[i] = -1
            else:
                update_lo = False
        if found:
            for i in range(len(ds) - 1, -1, -1):
                if ds[i] >= 0:
                    _hi = lo + i
                    break
        for k in range(lo, _lo):
            wf.delete(k)
            self.I.delete(s, k)
            self.D.delete(s, k)
        for k in range(_hi + 1, hi + 1):
            wf.delete(k)
            self.I.delete(s, k)
            self.D.delete(s, k)
        wf.lo, wf.hi = _lo, _hi

    # ---- wfa.go:549-700
    def _next(self, q, t, s):
        M, I, D = self.M, self.I, self.D
        x, oe, e = self.x, self.o + self.e, self.e
        n, m = len(q), len(t)
        lo_x, hi_x = M.krange(s, x)
        lo_o, hi_o = M.krange(s, oe)
        lo_i, hi_i = I.krange(s, e)
        lo_d, hi_d = D.krange(s, e)
        hi = min(m - 1, max(hi_x, hi_o, hi_i, hi_d) + 1)
        lo = max(-(n - 1), min(lo_x, lo_o, lo_i, lo_d) - 1)
        for k in range(lo, hi + 1):
            upd_i = upd_d = False
            typ_i = typ_d = typ_m = 0
            # insertion
            v1, _, from_m = M.get_after_diff(s, oe, k - 1)
            v2, _, from_i = I.get_after_diff(s, e, k - 1)
            if from_m and v1 > m:
                from_m, v1 = False, 0
            if from_i and v2 > m:
                from_i, v2 = False, 0
            isk = max(v1, v2) + 1
            if from_m or from_i:
                if from_m and from_i:
                    typ_i = INS_OPEN if v1 >= v2 else INS_EXT
                elif from_m:
                    typ_i = INS_OPEN
                else:
                    typ_i = INS_EXT
                upd_i = True
                I.set(s, k, isk, typ_i)
            else:
                isk = 0
            # deletion
            v1, _, from_m = M.get_after_diff(s, oe, k + 1)
            v2, _, from_d = D.get_after_diff(s, e, k + 1)
            if from_m and v1 - k > n:
                from_m, v1 = False, 0
            if from_d and v2 - k > n:
                from_d, v2 = False, 0
            dsk = max(v1, v2)
            if from_m or from_d:
                if from_m and from_d:
                    typ_d = DEL_OPEN if v1 >= v2 else DEL_EXT
                elif from_m:
                    typ_d = DEL_OPEN
                else:
                    typ_d = DEL_EXT
                upd_d = True
                D.set(s, k, dsk, typ_d)
            else:
                dsk = 0
            # mismatch
            v1, _, from_m = M.get_after_diff(s, x, k)
            if from_m and (v1 > m or v1 - k > n):
                from_m, v1 = False, 0
            msk = max(isk, dsk, v1 + 1)
            if upd_i or upd_d or from_m:
                if upd_i and upd_d and from_m:
                    typ_m = MISMATCH if msk == v1 + 1 else (typ_i if msk == isk else typ_d)
                elif upd_i:
                    if upd_d:
                        typ_m = typ_i if msk == isk else typ_d
                    elif from_m:
                        typ_m = MISMATCH if msk == v1 + 1 else typ_i
                    else:
                        typ_m = typ_i
                elif upd_d:
                    if from_m:
                        typ_m = MISMATCH if msk == v1 + 1 else typ_d
                    else:
                        typ_m = typ_d
                else:
                    typ_m = MISMATCH
                M.set(s, k, msk, typ_m)

    # ---- wfa.go:270-375
    def _start(self, q, t, s):
        M = self.M
        m, n = len(t), len(q)
        min_s, Ak = s, m - n
        last_k = Ak
        _s = s
        while True:
            if M.has_score(_s):
                lo, hi = M.krange(_s, 0)
                for first, step in ((Ak, -1), (Ak + 1, +1)):
                    k = first
                    hit = False
                    while True:
                        if (step < 0 and k < lo) or (step > 0 and k > hi):
                            break
                        offset, _, ok = M.get_after_diff(_s, 0, k)
                        if not ok:
                            k += step
                            continue
                        h = offset
                        v = h - k
                        if v <= 0 or v > n or h > m:
                            break
                        if (v == n and h >= n) or (h == m and v >= m):
                            hit = True
                            break
                        k += step
                    if hit and _s <= min_s:
                        last_k, min_s = k, _s
            if _s == 0:
                break
            _s -= 1
        return min_s, last_k

    # ---- wfa.go:703-983 + AlignmentResult.process (wfa_cigar.go:136-214)
    def _backtrace(self, q, t, s, Ak):
        M, I, D = self.M, self.I, self.D
        x, oe, e = self.x, self.o + self.e, self.e
        n, m = len(q), len(t)
        semi = not self.glob
        res = Result(score=s)
        ops: List[int] = []
        add = lambda letter, cnt: ops.append((ord(letter) << 32) | cnt)
        k = Ak
        first_match = True
        raw, _ = M.get_raw(s, k)
        prev_from_m = True
        typ = raw & 7
        h = raw >> 3
        v = h - k
        q_begin = t_begin = 0
        if h < m:
            add("I", m - h)
        elif v < n:
            add("H", n - v)
        from_itself = False
        offset0 = 0
        while v > 0 and h > 0:
            s_x, s_o, s_e = (s - x) & U32, (s - oe) & U32, (s - e) & U32
            from_mi = from_md = False
            if typ == INS_EXT:
                v1, _, f_m = M.get(s_o, k - 1)
                v2, _, f_i = I.get(s_e, k - 1)
                if f_m or f_i:
                    from_mi = True
                    offset0 = max(v1, v2) + 1
                else:
                    offset0 = 0
                M0 = I
            elif typ == DEL_EXT:
                v1, _, f_m = M.get(s_o, k + 1)
                v2, _, f_d = D.get(s_e, k + 1)
                if f_m or f_d:
                    from_md = True
                    offset0 = max(v1, v2)
                else:
                    offset0 = 0
                M0 = D
            else:
                v1, _, f_m = M.get(s_o, k - 1)
                v2, _, f_i = I.get(s_e, k - 1)
                isk = max(v1, v2) + 1 if (f_m or f_i) else 0
                from_mi = f_m or f_i
                v1, _, f_m = M.get(s_o, k + 1)
                v2, _, f_d = D.get(s_e, k + 1)
                dsk = max(v1, v2) if (f_m or f_d) else 0
                from_md = f_m or f_d
                v1, _, f_m = M.get(s_x, k)
                if from_mi or from_md or f_m:
                    offset0 = max(isk, dsk, v1 + 1)
                    from_itself = False
                else:
                    from_itself = True
                M0 = M
            if from_itself or offset0 == 0:
                break
            h0 = offset0
            if prev_from_m:
                n_matches = h - h0
                if n_matches > 0:
                    if first_match:
                        first_match = False
                        res.tend, res.qend = h, v
                    add("M", n_matches)
                h = offset0
                v = h - k
                if typ == MATCH:
                    t_begin, q_begin = h, v
                elif n_matches > 0:
                    t_begin, q_begin = h + 1, v + 1
                if h <= 0 or v <= 0:
                    break
            add(OPS[typ], 1)
            if semi and (h == 1 or v == 1):
                break
            prev_from_m = True
            if typ == MISMATCH:
                s, h = s_x, h - 1
            elif typ == INS_OPEN:
                s, k, h = s_o, k - 1, h - 1
            elif typ == INS_EXT:
                s, k, h = s_e, k - 1, h - 1
                prev_from_m = False
            elif typ == DEL_OPEN:
                s, k = s_o, k + 1
            elif typ == DEL_EXT:
                s, k = s_e, k + 1
                prev_from_m = False
            else:
                break
            v = h - k
            raw, ok = M0.get_raw(s, k)
            if not ok:
                break
            typ = raw & 7
        if h > 0 and v > 0:
            n_matches = min(h, v) - 1
            if n_matches > 0:
                if first_match:
                    first_match = False
                    res.tend, res.qend = h, v
                add("M", n_matches)
                h -= n_matches
                v -= n_matches
                if typ == MATCH:
                    t_begin, q_begin = h, v
                else:
                    t_begin, q_begin = h + 1, v + 1
            elif typ == MATCH:
                t_begin, q_begin = h, v
                if first_match:
                    first_match = False
                    res.tend, res.qend = h, v
            add(OPS[typ], 1)
        if v > 1:
            add("H", v - 1)
        if h > 1:
            add("I", h - 1)
        res.tbegin, res.qbegin = t_begin, q_begin
        # process(): reverse, merge equal neighbours, statistics between the first and the last M run
        ops.reverse()
        merged: List[int] = []
        for o in ops:
            if merged and (merged[-1] >> 32) == (o >> 32):
                merged[-1] += o & U32
            else:
                merged.append(o)
        begin = next((i for i, o in enumerate(merged) if (o >> 32) == ord("M")), 0)
        end = next((i for i in range(len(merged) - 1, -1, -1) if (merged[i] >> 32) == ord("M")), 0)
        for o in merged[begin:end + 1]:
            c, letter = o & U32, chr(o >> 32)
            res.align_len += c
            if letter == "M":
                res.matches += c
            elif letter in "ID":
                res.gaps += c
                res.gap_regions += 1
        res.ops = merged
        return res
