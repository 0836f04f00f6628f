"""GPU: the register-ring forward kernels off the 2 : 4 rails (round 5).  wfa.go:32-36 takes any penalties; the sub-wave
kernels are instantiated per penalty SHAPE x/g : (o+e)/g with e/g == 1 (wfa_amd/csrc/wfa_fwd.hpp): 2 : 4 (the default 4/6/2),
1 : 3 (2/4/2), 1 : 2 (1/1/1, 2/2/2), 2 : 3 (4/4/2), 2 : 2 (4/2/2: x == o+e), 3 : 3 (6/4/2: x == o+e).  Every instance against
the oracle -- results, the kernel that produced them (main_kernel_kind), and every stored backtrace word."""
import numpy as np
import pytest

from test_parity_gpu import _aligner, _arena_word_check, _oracle_params, assert_batch_equal

pytestmark = pytest.mark.gpu

# one penalty set per new shape (+ a second member of two of them: the shape is what counts, not the numbers)
SHAPE_PENS = [(2, 4, 2), (1, 1, 1), (4, 4, 2), (4, 2, 2), (6, 4, 2), (2, 2, 2), (6, 12, 6)]


@pytest.mark.parametrize("pen", SHAPE_PENS)
@pytest.mark.parametrize("form", ["blk", "duo", "lane", "narrow"])
def test_shape_arena_word_for_word(built, pen, form):
    """Every compact backtrace word the per-shape instances leave in HBM, against the oracle's wavefronts
    (_arena_word_check: wf-adaptive, ragged lengths, cells at sequence ends, the census of stored words)."""
    if form == "blk":
        _arena_word_check(600, 0.05, pen, (10, 50, 1), 3, 1, 0, 0)
    elif form == "duo":
        _arena_word_check(1000, 0.04, pen, (10, 50, 1), 3, 0, 1, 0)
    elif form == "lane":
        _arena_word_check(150, 0.03, pen, (10, 50, 1) if pen[0] != 1 else None, 8, 1, 0, 1)
    else:
        # (32-diagonal windows: with cheap gaps -- x >= o+e -- most 120-base pairs at 5 % outgrow them and are handed on)
        _arena_word_check(120, 0.05, pen, None, 5, 0, 0, 0, min_pairs=24)


def _run(pen, ad, data, opts, kind, what):
    from oracle import oracle as O
    al = _aligner(True, ad, pen)
    for k, v in opts.items():
        al.set_option(k, v)
    for rep in range(2):  # (twice: what a context learns about a class of batches must not change a result)
        got = al.align_arrays(*data)
        if rep == 0 and kind is not None:
            assert al.last_timing().main_kernel_kind == kind, (what, al.last_timing().main_kernel_kind)
        if rep == 0:
            want = O.align_batch(_oracle_params(True, ad, pen), *data, n_threads=8)
        assert_batch_equal(got, want, f"{what} pen={pen} ad={ad} opts={opts} rep={rep}")
    n_retried = al.last_timing().n_retried_pairs
    al.close()
    return n_retried


@pytest.mark.parametrize("pen", SHAPE_PENS[:5])
def test_shape_first_pass_kernels(built, pen):
    """The first-pass kernel of every batch class, chosen by the default routing: a lane per pair (short reads), the
    variable-lanes kernel (1 kbp in GPU-filling numbers), four pairs per wave (small batches), and the poisoned arena."""
    import wfa_amd as w
    short = w.generate_pairs(seed=11 + pen[0], n_pairs=40000, length=150, error_rate=0.03)
    _run(pen, None, short, {"arena_poison": 1}, 10, "lane")
    _run(pen, (10, 50, 1), short, {}, 10, "lane, wf-adaptive")
    mid = w.generate_pairs(seed=12 + pen[1], n_pairs=12000, length=1000, error_rate=0.05)
    _run(pen, (10, 50, 1), mid, {"duo": 2, "arena_poison": 1}, 8, "duo")
    _run(pen, (10, 50, 1), mid, {"duo": 0}, 3, "blk")
    few = w.generate_pairs(seed=13, n_pairs=3000, length=150, error_rate=0.04)
    _run(pen, (10, 50, 1), few, {}, 6, "narrow")


@pytest.mark.parametrize("pen", SHAPE_PENS[:5])
def test_shape_wide_rungs_and_long_reads(built, pen):
    """Band failures climb the 128- and 256-diagonal instances of the same shape; long reads take the sliding-window
    instances (four pairs per wave, a wave per pair with one / two diagonals per lane)."""
    import wfa_amd as w
    wide = w.generate_pairs(seed=21 + pen[2], n_pairs=1500, length=900, error_rate=0.18)
    _run(pen, (10, 50, 1), wide, {}, None, "wide rungs")
    _run(pen, None, w.generate_pairs(seed=22, n_pairs=600, length=500, error_rate=0.10), {}, None, "wf-adaptive off")
    longr = w.generate_pairs(seed=23 + pen[0], n_pairs=48, length=6000, error_rate=0.04)
    for first in (11, 12, 13, 14, 15):
        _run(pen, (10, 50, 1), longr, {"long_first": first, "long_window_words": 64}, first, f"long reads, first instance {first}")
    _run(pen, (10, 50, 1), longr, {}, 15, "long reads, default routing")


@pytest.mark.parametrize("pen", SHAPE_PENS)
def test_shape_single_align(built, pen):
    """Aligner.Align (wfahip_align_pair: the lone-pair instance of the shape, rows in LDS or in global memory)."""
    import wfa_amd as w
    from oracle import oracle as O
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=31, n_pairs=24, length=700, error_rate=0.06)
    for lds in (1, 0):
        al = _aligner(True, (10, 50, 1), pen)
        al.set_option("pair_lds", lds)
        oa = O.Aligner(_oracle_params(True, (10, 50, 1), pen))
        lone = 0
        for i in range(24):
            q = bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])])
            t = bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])])
            r, want = al.Align(q, t), oa.align(q, t)
            assert (r.Score, r.CIGAR(False), r.QBegin, r.QEnd, r.TBegin, r.TEnd, r.AlignLen, r.Matches, r.Gaps, r.GapRegions) == (
                want.score, want.cigar, want.qbegin, want.qend, want.tbegin, want.tend, want.align_len, want.matches, want.gaps,
                want.gap_regions), (pen, lds, i)
            lone += al.last_timing().main_kernel_kind == 16 and al.last_timing().n_launches == 1
        assert lone >= 20, (pen, lds, lone)  # (a pair whose band leaves the instance's 64 diagonals is finished by the batch entry)
        al.close()
