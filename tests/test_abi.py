"""CPU: the C-ABI library loads, exports every symbol include/wfa_hip.h declares, and its host-only
entry points (generator, error strings, argument validation) behave.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(built):
    from wfa_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "wfa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only, not the comments
    declared = set(re.findall(r"\b(wfahip_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.wfahip_version() == 400


def test_struct_layouts(built):
    from wfa_amd import _lib
    assert C.sizeof(_lib.Params) == 28
    assert C.sizeof(_lib.Results) == 15 * 8
    assert C.sizeof(_lib.Timing) == 72
    assert C.sizeof(_lib.Row) == 24


def test_strerror_and_no_device(built):
    import torch
    from wfa_amd import _lib
    L = _lib.lib()
    assert L.wfahip_strerror(0) == b"ok"
    assert b"device" in L.wfahip_strerror(_lib.ERR_NO_DEVICE)
    if not torch.cuda.is_available():
        ctx = C.c_void_p()
        assert L.wfahip_create(0, C.byref(ctx)) == _lib.ERR_NO_DEVICE
        import wfa_amd
        with pytest.raises(_lib.WfaHipError):  # the product path fails loudly: there is no CPU fallback
            wfa_amd.New()


def test_generator_is_deterministic_and_sane(built):
    import wfa_amd
    a = wfa_amd.generate_pairs(seed=3, n_pairs=64, length=1000, error_rate=0.05, n_threads=1)
    b = wfa_amd.generate_pairs(seed=3, n_pairs=64, length=1000, error_rate=0.05, n_threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    c = wfa_amd.generate_pairs(seed=3, n_pairs=16, length=1000, error_rate=0.05, first_index=48)
    blob, q_off, q_len, t_off, t_len = a
    for i in range(16):  # sharding by first_index reproduces the same pairs (multi-GPU shards rely on this)
        qa = bytes(blob[int(q_off[48 + i]):int(q_off[48 + i]) + int(q_len[48 + i])])
        qc = bytes(c[0][int(c[1][i]):int(c[1][i]) + int(c[2][i])])
        ta = bytes(blob[int(t_off[48 + i]):int(t_off[48 + i]) + int(t_len[48 + i])])
        tc = bytes(c[0][int(c[3][i]):int(c[3][i]) + int(c[4][i])])
        assert qa == qc and ta == tc
    assert (q_len == 1000).all()
    assert (np.abs(t_len.astype(int) - 1000) <= 50).all()
    seqs = set(bytes(blob[int(q_off[i]):int(q_off[i]) + 1000]) for i in range(64))
    assert len(seqs) == 64
    assert set(bytes(blob[int(q_off[0]):int(q_off[0]) + 1000])) <= set(b"ACGT")
    # edit distance sanity through the oracle: ~50 edits -> score well below 50 * (o+e) and above 0
    from oracle import oracle as O
    r = O.align_batch(O.make_params(), blob, q_off, q_len, t_off, t_len, want_ops=False)
    assert (r.status == 0).all() and 100 < r.score.mean() < 400


def test_reference_api_surface(built):
    """Names and defaults of the reference's exported API (wfa.go:32-71,186-193; wfa_cigar.go:57-66)."""
    import wfa_amd as w
    assert (w.DefaultPenalties.Mismatch, w.DefaultPenalties.GapOpen, w.DefaultPenalties.GapExt) == (4, 6, 2)
    assert (w.DefaultAdaptiveOption.MinWFLen, w.DefaultAdaptiveOption.MaxDistDiff,
            w.DefaultAdaptiveOption.CutoffStep) == (10, 50, 1)
    assert w.DefaultOptions.GlobalAlignment is True
    assert w.MaxSeqLen == 536870911
    assert w.Op((ord("M") << 32) | 7) == ("M", 7)
    assert (w.OpM, w.OpD, w.OpI, w.OpX, w.OpH) == tuple(map(ord, "MDIXH"))
    r = w.AlignmentResult(Ops=[(ord("I") << 32) | 2, (ord("M") << 32) | 3, (ord("X") << 32) | 1,
                               (ord("M") << 32) | 2, (ord("H") << 32) | 1])
    assert r.CIGAR(False) == "2I3M1X2M1H"
    assert r.CIGAR(True) == "3M1X2M"
    assert str(w.ErrEmptySeq) == "wfa: invalid empty sequence"


def test_host_packer(built):
    """wfahip_pack_pairs (host only): 16 bases per word, code (ascii >> 1) & 3, every sequence at a word boundary with
    one pad word; anything outside uppercase ACGT is refused (the byte entry must take it: wfa.go:408-454 compares
    raw bytes)."""
    import wfa_amd
    from wfa_amd import _lib
    data = wfa_amd.generate_pairs(seed=4, n_pairs=300, length=77, error_rate=0.1, n_threads=2)
    blob, q_off, q_len, t_off, t_len = data
    for thr in (1, 3):
        packed, q_woff, t_woff = wfa_amd.pack_pairs(*data, n_threads=thr)
        code = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}
        pos = 0
        for i in range(300):
            for off, ln, woff in ((q_off, q_len, q_woff), (t_off, t_len, t_woff)):
                assert int(woff[i]) == pos
                seq = blob[int(off[i]):int(off[i]) + int(ln[i])]
                nw = (int(ln[i]) + 15) // 16
                for w in range(nw):
                    want = 0
                    for k, c in enumerate(seq[16 * w:16 * w + 16]):
                        want |= code[int(c)] << (2 * k)
                    assert int(packed[pos + w]) == want
                assert int(packed[pos + nw]) == 0
                pos += nw + 1
        assert pos == packed.size
    bad = wfa_amd.make_blob([b"ACGTN", b"acgt"], [b"ACGT", b"ACGT"])
    with pytest.raises(_lib.WfaHipError) as ei:
        wfa_amd.pack_pairs(*bad)
    assert ei.value.code == _lib.ERR_UNSUPPORTED
