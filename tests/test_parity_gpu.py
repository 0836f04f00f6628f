"""GPU: the HIP path (through the C-ABI) against the oracle and the golden fixtures.  Bit-exact: score,
CIGAR ops, match region, statistics."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len")


def _aligner(global_alignment=True, adaptive=(10, 50, 1), penalties=(4, 6, 2)):
    import wfa_amd as w
    al = w.New(w.Penalties(*penalties), w.Options(GlobalAlignment=global_alignment), device=0)
    if adaptive is not None:
        assert al.AdaptiveReduction(w.AdaptiveReductionOption(*adaptive)) is None
    # WFA_TEST_OPTS="key=value,..." runs the whole suite with library options forced (e.g.
    # bt_stream_min=1,bt_stream_single=1: the streamed backtrace on every batch the blocked kernel takes)
    for kv in filter(None, os.environ.get("WFA_TEST_OPTS", "").split(",")):
        k, v = kv.split("=")
        al.set_option(k, int(v))
    return al


def _oracle_params(global_alignment=True, adaptive=(10, 50, 1), penalties=(4, 6, 2)):
    from oracle import oracle as O
    return O.make_params(*penalties, global_alignment=global_alignment, adaptive=adaptive)


def assert_batch_equal(got, want, what=""):
    assert np.array_equal(got.status, want.status), what
    for f in FIELDS:
        a, b = getattr(got, f), getattr(want, f)
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{what}: field {f} differs at {len(bad)} pairs, first {bad[:5]}: "
                                 f"{a[bad[:5]]} vs {b[bad[:5]]}")
    # ops: both are packed in pair order
    if not np.array_equal(got.ops, want.ops):
        for i in range(len(got.status)):
            if not np.array_equal(got.pair_ops(i), want.pair_ops(i)):
                raise AssertionError(f"{what}: CIGAR differs at pair {i}")


def test_known_answers(built, known_answers):
    for ka in known_answers["vectors"]:
        al = _aligner(ka["mode"] == "global", tuple(ka["adaptive"]))
        r = al.Align(ka["q"].encode(), ka["t"].encode())
        if ka.get("cigar_exact", True):
            assert r.CIGAR(False) == ka["cigar"], ka["id"]
        for f, g in (("score", "Score"), ("qbegin", "QBegin"), ("qend", "QEnd"), ("tbegin", "TBegin"),
                     ("tend", "TEnd"), ("align_len", "AlignLen"), ("matches", "Matches"), ("gaps", "Gaps"),
                     ("gap_regions", "GapRegions")):
            if f in ka:
                assert getattr(r, g) == ka[f], (ka["id"], f)
        if "text" in ka:
            Q, A, T = r.AlignmentText(ka["q"].encode(), ka["t"].encode(), False)
            assert (Q.decode(), A.decode().rstrip(), T.decode()) == (ka["text"][0], ka["text"][1].rstrip(),
                                                                     ka["text"][2])
        al.close()


def test_reference_test_pairs_all_option_sets(built, ref_pairs, oracle_vectors):
    """Every input pair of wfa_test.go / seqs.txt under 4 option sets vs the committed oracle vectors."""
    opts = {"global+adaptive": (True, (10, 50, 1)), "global": (True, None),
            "semiglobal+adaptive": (False, (10, 50, 1)), "semiglobal": (False, None)}
    qs = [p["q"].upper().encode() for p in ref_pairs]
    ts = [p["t"].upper().encode() for p in ref_pairs]
    for name, (g, ad) in opts.items():
        al = _aligner(g, ad)
        results, errors = al.AlignBatch(qs, ts)
        exp = [e for e in oracle_vectors["results"] if e["options"] == name]
        assert len(exp) == len(qs)
        for e in exp:
            r = results[e["pair"]]
            assert errors[e["pair"]] is None
            got = (r.Score, r.CIGAR(False), r.QBegin, r.QEnd, r.TBegin, r.TEnd, r.AlignLen, r.Matches, r.Gaps,
                   r.GapRegions)
            want = (e["score"], e["cigar"], e["qbegin"], e["qend"], e["tbegin"], e["tend"], e["align_len"],
                    e["matches"], e["gaps"], e["gap_regions"])
            assert got == want, (name, ref_pairs[e["pair"]]["source"])
        al.close()


def test_wavefront_dumps_match_oracle(built, known_answers, oracle_vectors):
    """Internal state: every stored M/I/D word of KA1 and KA2 equals the oracle's final wavefronts."""
    for d in oracle_vectors["dumps"]:
        ka = [k for k in known_answers["vectors"] if k["id"] == d["id"]][0]
        al = _aligner(ka["mode"] == "global", tuple(ka["adaptive"]))
        wf, res = al.debug_wavefronts(ka["q"].encode(), ka["t"].encode())
        # the device stops at the same final score; rows above the backtrace start score exist in both
        for comp in "MID":
            want = {int(s): {int(k): v for k, v in row.items()} for s, row in d["final"][comp].items()}
            want = {s: r for s, r in want.items() if r}
            assert wf[comp] == want, (d["id"], comp)
        al.close()


def test_plot_from_device_wavefronts(built, known_answers):
    """(*Aligner).Plot over the DEVICE wavefronts prints the README's KA1 M-component table
    cell for cell, and for KA2 exactly what the oracle's wavefronts print (the README's KA2 block is stale in seven
    cells, tests/test_oracle_golden.py)."""
    import json
    import os
    from oracle import oracle as O
    from wfa_amd.aligner import plot_component
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "plot_tables.json"), encoding="utf-8"))
    cells = lambda text: [[c.strip() for c in line.split("\t")] for line in text.rstrip("\n").split("\n")]
    for ka, want in ((known_answers["vectors"][0], gold["ka1_global"]), (known_answers["vectors"][1], None)):
        al = _aligner(ka["mode"] == "global", tuple(ka["adaptive"]))
        text = al.Plot(ka["q"].encode(), ka["t"].encode())
        oa = O.Aligner(global_alignment=ka["mode"] == "global", adaptive=tuple(ka["adaptive"]))
        oa.align(ka["q"].encode(), ka["t"].encode())
        owf = {c: {s: {lo + i: v for i, v in enumerate(raw) if v} for s, (lo, hi, raw) in d.items()}
               for c, d in oa.dump().items()}
        assert text == plot_component(ka["q"].encode(), ka["t"].encode(), owf), ka["id"]
        if want is not None:
            assert cells(text) == want
        for comp in "ID":  # the I and D components plot too (no extension logic there)
            assert al.Plot(ka["q"].encode(), ka["t"].encode(), comp) == plot_component(
                ka["q"].encode(), ka["t"].encode(), owf, comp)
        al.close()


@pytest.mark.parametrize("length,err,n", [(150, 0.02, 4000), (1000, 0.05, 3000), (1000, 0.2, 300), (37, 0.1, 2000)])
@pytest.mark.parametrize("glob,adaptive", [(True, (10, 50, 1)), (True, None), (False, (10, 50, 1)), (False, None)])
def test_synthetic_batches(built, length, err, n, glob, adaptive):
    import wfa_amd as w
    from oracle import oracle as O
    if not glob and length >= 1000:
        n = max(50, n // 10)  # semi-global seeds make the oracle slow
    data = w.generate_pairs(seed=length * 7 + int(err * 100), n_pairs=n, length=length, error_rate=err)
    al = _aligner(glob, adaptive)
    got = al.align_arrays(*data)
    want = O.align_batch(_oracle_params(glob, adaptive), *data, n_threads=8)
    assert_batch_equal(got, want, f"L={length} e={err} glob={glob} ad={adaptive}")
    al.close()


# (6,4,2), (3,1,2): x == o+e; (2,4,2), (2,2,2): x == e; (4,2,2): o+e == 2e -- the shapes in which two of next()'s
# sources come from the same earlier score
@pytest.mark.parametrize("pen", [(4, 6, 2), (1, 1, 1), (2, 3, 1), (5, 0, 3), (3, 7, 2), (6, 5, 4), (2, 12, 1),
                                 (6, 4, 2), (2, 4, 2), (4, 2, 2), (3, 1, 2), (2, 2, 2),
                                 # GapExt == 0: I[s-e] is the row being written, an insertion chains along it (wfa.go:580)
                                 (4, 6, 0), (2, 3, 0), (3, 3, 0)])
def test_other_penalties(built, pen):
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=sum(pen), n_pairs=600, length=200, error_rate=0.08)
    for glob in (True, False):
        for ad in (None, (10, 50, 1), (4, 5, 1)):
            al = _aligner(glob, ad, pen)
            got = al.align_arrays(*data)
            want = O.align_batch(_oracle_params(glob, ad, pen), *data, n_threads=8)
            assert_batch_equal(got, want, f"pen={pen} glob={glob} ad={ad}")
            al.close()


def test_aggressive_adaptive_short_pairs(built):
    """Short pairs under aggressive wf-adaptive settings, where the reference's backtrace leaves the path its score was reached
    on (tests/test_oracle_crosscheck.py: CIGARs that do not cost the score, or consume a base more than a sequence has): the HIP
    path returns exactly what the oracle returns -- the two pairs named there, and 3 000 random ones of up to 30 bases."""
    import random
    import wfa_amd as w
    from oracle import oracle as O
    rng = random.Random(77)
    qs = [b"GGATGTTGTAGCCGTGCTC", b"CCTTCAGGTGCCGAGTGTTA"]
    ts = [b"TCGTGATGTTGTAGCCGTGCTCTATACGG", b"CCTTCAGGTGCAGTGT"]
    for _ in range(3000):
        q = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 30)))
        t = bytearray(q)
        for _ in range(rng.randint(0, 7)):
            kind, pos = rng.randint(0, 2), rng.randint(0, max(0, len(t) - 1))
            if kind == 0 and t:
                t[pos] = rng.choice(b"ACGT")
            elif kind == 1:
                t.insert(pos, rng.choice(b"ACGT"))
            elif len(t) > 1:
                del t[pos]
        qs.append(q), ts.append(bytes(t) or b"A")
    data = w.make_blob(qs, ts)
    for glob in (True, False):
        for ad in ((4, 5, 1), (1, 1, 1), (2, 3, 1)):
            al = _aligner(glob, ad)
            got = al.align_arrays(*data)
            want = O.align_batch(_oracle_params(glob, ad), *data, n_threads=8)
            assert_batch_equal(got, want, f"glob={glob} ad={ad}")
            if glob:
                r = al.Align(qs[0], ts[0])
                # (Align() is wfahip_align_pair, not the batch entry: its score AND its CIGAR against the oracle's for this pair)
                assert (r.Score, r.CIGAR(False)) == (int(want.score[0]), want.cigar(0)), ad
                if ad == (4, 5, 1):
                    assert (r.Score, r.CIGAR(False)) == (58, "3X2M1X3M1X1M3I8M7I")
            al.close()


def test_ragged_and_edge_inputs(built):
    """Empty / 1-base / very unequal lengths / non-ACGT bytes / lowercase, mixed in one batch."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    qs, ts = [], []
    alpha = b"ACGT"
    for i in range(400):
        n, m = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        q = bytes(alpha[j] for j in rng.integers(0, 4, n))
        if i % 3 == 0:  # related sequences
            t = bytearray(q)
            for _ in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, len(t)))
                t[p] = alpha[int(rng.integers(0, 4))]
            t = bytes(t[:m]) if len(t) > m else bytes(t)
        else:
            t = bytes(alpha[j] for j in rng.integers(0, 4, m))
        qs.append(q), ts.append(t)
    qs += [b"", b"ACGT", b"", b"A", b"A", b"acgtacgt", b"ACGTNNACGT", b"Bioinformatics helps Biology", b"C", b"CG",
           b"A" * 300, b"ACGT" * 50]
    ts += [b"ACGT", b"", b"", b"A", b"CA", b"ACGTACGT", b"ACGTACGT", b"We learn bioinformatics to help biologists",
           b"C", b"C", b"A", b"TGCA" * 50]
    for glob in (True, False):
        for ad in (None, (10, 50, 1)):
            al = _aligner(glob, ad)
            blob, q_off, q_len, t_off, t_len = w.make_blob(qs, ts)
            got = al.align_arrays(blob, q_off, q_len, t_off, t_len)
            want = O.align_batch(_oracle_params(glob, ad), blob, q_off, q_len, t_off, t_len, n_threads=4)
            assert_batch_equal(got, want, f"ragged glob={glob} ad={ad}")
            results, errors = al.AlignBatch(qs[-12:-9], ts[-12:-9])
            assert errors[0] is w.ErrEmptySeq and errors[1] is w.ErrEmptySeq and errors[2] is w.ErrEmptySeq
            with pytest.raises(w.WfaError):
                al.Align(b"", b"A")
            al.close()


def test_unaligned_blob_offsets(built):
    """Sequences at arbitrary byte offsets (the staging loop uses aligned dword loads + funnel shifts)."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    n = 300
    parts, q_off, q_len, t_off, t_len = [], [], [], [], []
    pos = 0
    for i in range(n):
        pad = int(rng.integers(0, 7))
        parts.append(b"#" * pad)
        pos += pad
        L = int(rng.integers(1, 200))
        q = bytes(b"ACGT"[j] for j in rng.integers(0, 4, L))
        t = bytearray(q)
        for _ in range(int(rng.integers(0, 8))):
            t[int(rng.integers(0, len(t)))] = b"ACGT"[int(rng.integers(0, 4))]
        if rng.integers(0, 2):
            t = t[int(rng.integers(0, 3)):]
        t = bytes(t) or b"A"
        q_off.append(pos), q_len.append(len(q)), parts.append(q)
        pos += len(q)
        t_off.append(pos), t_len.append(len(t)), parts.append(t)
        pos += len(t)
    blob = np.frombuffer(b"".join(parts), dtype=np.uint8)
    args = (blob, np.array(q_off, np.uint64), np.array(q_len, np.uint32), np.array(t_off, np.uint64),
            np.array(t_len, np.uint32))
    al = _aligner(True, (10, 50, 1))
    want = O.align_batch(_oracle_params(), *args)
    assert_batch_equal(al.align_arrays(*args), want, "unaligned")
    al.close()
    # the lane-per-pair kernel packs the bytes itself: its word-by-word path (a wave that meets an unaligned or very short
    # sequence), with the packing kernel in front of it, and -- the same pairs moved to 16-byte boundaries, lengths
    # 16 .. 199 -- its one-round path with partial last chunks of every length
    for opts in ({"lane": 2}, {"lane": 2, "lane_pack": 0}):
        al = _aligner(True, (10, 50, 1))
        for k, v in opts.items():
            al.set_option(k, v)
        assert_batch_equal(al.align_arrays(*args), want, f"unaligned {opts}")
        assert al.last_timing().main_kernel_kind == 10
        al.close()
    keep = [i for i in range(n) if q_len[i] >= 16 and t_len[i] >= 16]
    raw = b"".join(parts)
    qs = [raw[q_off[i]:q_off[i] + q_len[i]] for i in keep]
    ts = [raw[t_off[i]:t_off[i] + t_len[i]] for i in keep]
    data = w.make_blob(qs, ts)
    al = _aligner(True, (10, 50, 1))
    al.set_option("lane", 2)
    assert_batch_equal(al.align_arrays(*data), O.align_batch(_oracle_params(), *data), "aligned, every tail length")
    al.close()


@pytest.mark.parametrize("opts,kind", [({"blk": 16}, 3), ({"blk": 16, "bt_stream_min": 1, "bt_stream_single": 1}, 3),
                                       ({"blk": 16, "bt_stream_min": 1, "bt_stream_single": 1, "bt_stream": 4}, 3), ({"blk": 8}, 4),
                                       ({"blk": 0}, 2), ({"blk": 0, "reg": 0}, 1), ({"packed": 0}, 0), ({"blk": 16, "duo": 2}, 8),
                                       ({"blk": 16, "duo": 0}, 3)])
def test_forward_kernel_variants(built, opts, kind):
    """Every forward kernel (blocked register-window with 16 / 8 lanes per pair, strided register-window, LDS-ring
    packed, generic; the blocked kernel also with the streamed backtrace forced on for this small batch) on one
    mixed batch: unequal lengths (the 64-diagonal window has to follow the band up and
    down), 15 % error (bands that outgrow the window are handed down the ladder), short reads, wf-adaptive on
    and off, and a per-pair arena too small for some pairs."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    qs, ts = [], []
    for i in range(1500):
        L = int(rng.integers(20, 900))
        q = rng.integers(0, 4, L)
        t = list(q)
        ne = int(L * (0.15 if i % 5 == 0 else 0.04))
        for _ in range(ne):
            kind_e, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(1, len(t))))
            if kind_e == 0 and t:
                t[pos] = int(rng.integers(0, 4))
            elif kind_e == 1:
                t.insert(pos, int(rng.integers(0, 4)))
            elif t:
                del t[pos]
        if i % 7 == 0:  # long overhang: |m - n| up to 200
            extra = list(rng.integers(0, 4, int(rng.integers(1, 200))))
            t = (extra + t) if i % 2 else (t + extra)
        if i % 11 == 0:
            q = q[int(rng.integers(0, min(150, L - 1))):]
        qs.append(bytes(b"ACGT"[c] for c in q) or b"A")
        ts.append(bytes(b"ACGT"[c] for c in t) or b"C")
    data = w.make_blob(qs, ts)
    for ad in ((10, 50, 1), None, (4, 10, 1)):
        want = O.align_batch(_oracle_params(True, ad), *data, n_threads=8)
        ref = _aligner(True, ad)  # the census of stored wavefront cells must not depend on the kernel either
        ref.set_option("packed", 0)
        ref.set_option("census", 1)
        ref.align_arrays(*data)
        want_cells = ref.last_timing().cells_stored
        ref.close()
        for small_arena in (False, True):
            al = _aligner(True, ad)
            al.set_option("census", 1)
            for k, v in opts.items():
                al.set_option(k, v)
            if small_arena:
                al.set_option("packed_arena_bytes", 6 * 1024)
            got = al.align_arrays(*data)
            t = al.last_timing()
            assert t.main_kernel_kind == kind, (opts, t.main_kernel_kind)
            if small_arena and kind != 0:
                assert t.n_retried_pairs > 0
            assert_batch_equal(got, want, f"opts={opts} ad={ad} small_arena={small_arena}")
            assert t.cells_stored == want_cells, (opts, ad, small_arena)
            al.close()


@pytest.mark.parametrize("n_pairs,max_l", [(1, 240), (3, 240), (9, 240), (64, 240), (1237, 240), (9, 190), (1237, 190), (20000, 120)])
def test_short_read_batches(built, n_pairs, max_l):
    """Short pairs (<= 240 bases) take the blocked kernel's batch mode: a group stages 8 queue entries at a time.
    Ragged lengths, empty / 1-base / non-ACGT / lowercase entries inside batches, batch counts that do not
    divide the number of pairs; the unbatched kernel must give the same records."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + n_pairs)
    qs, ts = [], []
    for i in range(n_pairs):
        L = int(rng.integers(1, max_l))
        q = bytes(b"ACGT"[c] for c in rng.integers(0, 4, L))
        t = bytearray(q)
        for _ in range(int(rng.integers(0, 1 + L // 12))):
            pos = int(rng.integers(0, len(t)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                t[pos] = b"ACGT"[int(rng.integers(0, 4))]
            elif kind == 1:
                t.insert(pos, b"ACGT"[int(rng.integers(0, 4))])
            elif len(t) > 1:
                del t[pos]
        t = bytes(t[:max_l])
        if i % 17 == 5:
            q = b""
        elif i % 17 == 9:
            t = t.lower()
        elif i % 17 == 13:
            q = q[:len(q) // 2] + b"N" + q[len(q) // 2:]
            q = q[:max_l]
        elif i % 23 == 7:
            q, t = b"A", b"CA"
        qs.append(q), ts.append(t)
    data = w.make_blob(qs, ts)
    max_len = max(max(len(q), len(t)) for q, t in zip(qs, ts))
    for ad in ((10, 50, 1), None):
        want = O.align_batch(_oracle_params(True, ad), *data, n_threads=4)
        for batch, narrow in ((1, 1), (8, 1), (3, 1), (1, 0), (8, 0), (0, 1)):
            al = _aligner(True, ad)
            al.set_option("blk_batch", batch)
            al.set_option("blk_narrow", narrow)
            got = al.align_arrays(*data)
            # reads under 200 bases start with eight pairs per wave (32-diagonal windows); pairs that outgrow them
            # (here: up to 240 bases with up to 8 % edits) move on to the 16-lane instance
            assert al.last_timing().main_kernel_kind == (6 if (batch and narrow and max_len < 200) else 3)
            assert_batch_equal(got, want, f"short reads n={n_pairs} ad={ad} blk_batch={batch} blk_narrow={narrow}")
            al.close()


def test_hand_over_does_not_depend_on_wave_mates(built):
    """Which pairs a forward kernel hands on (band touching the window edge) must be a property of the pair, not of
    the pairs that happen to share its wave: the queue order varies from run to run, the count must not."""
    import wfa_amd as w
    data = w.generate_pairs(seed=108, n_pairs=300000, length=100, error_rate=0.06, n_threads=16)
    al = _aligner(True, (10, 50, 1))
    al.set_option("census", 1)
    seen = set()
    for _ in range(4):
        al.align_arrays(*data)
        t = al.last_timing()
        seen.add((t.n_retried_pairs, t.cells_stored))
    assert len(seen) == 1 and next(iter(seen))[0] > 0, seen
    al.close()


@pytest.mark.parametrize("length,err,n", [(4000, 0.03, 120), (9000, 0.02, 40), (2500, 0.08, 120)])
def test_mid_length_global(built, length, err, n):
    """Global pairs of a few kbp: the blocked kernel near its LDS limit (offsets of several thousand, hundreds of
    rows per pair); without wf-adaptive the bands outgrow the 64-diagonal window and the ladder takes over."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=length, n_pairs=n, length=length, error_rate=err, n_threads=8)
    for ad in ((10, 50, 1), None):
        al = _aligner(True, ad)
        want = O.align_batch(_oracle_params(True, ad), *data, n_threads=8)
        # (round 4: reads beyond 4 000 bases start on the sliding-window instances -- a batch this small with a wave per pair;
        # option long = 0 keeps them on the whole-sequence instance this test was written for)
        for long_opt in (1, 0):
            al.set_option("long", long_opt)
            got = al.align_arrays(*data)
            assert al.last_timing().main_kernel_kind == (15 if long_opt and length >= 4000 else 3)
            assert_batch_equal(got, want, f"L={length} ad={ad} long={long_opt}")
        al.close()


@pytest.mark.parametrize("wait_us,chunk_pairs", [(20000, 0), (0, 0), (20000, 1000)])
def test_streamed_backtrace_mid_length(built, wait_us, chunk_pairs):
    """Streamed backtrace forced on for a small batch of 4 kbp pairs (hundreds of score steps per pair, streaming
    waves that wait for their entries).  wait_us = 0: every streaming wave gives up at its first wait, so the sweep
    kernel after the launch has to find and walk what they left behind.  chunk_pairs = 1000: three chunks, each with
    its own queue."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=44, n_pairs=3000, length=4000, error_rate=0.04, n_threads=8)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), *data, n_threads=8)
    al = _aligner(True, (10, 50, 1))
    al.set_option("long", 0)  # (the whole-sequence instance: the streamed backtrace is its)
    al.set_option("bt_stream_min", 1)
    al.set_option("bt_stream_single", 1)
    al.set_option("bt_stream_wait_us", wait_us)
    al.set_option("chunk_pairs", chunk_pairs)
    for rep in range(2):
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == 3
        assert_batch_equal(got, want, f"wait_us={wait_us} chunk_pairs={chunk_pairs} repeat {rep}")
    al.close()


def test_wide_band_retry_kernel(built):
    """Pairs whose band leaves the 64-diagonal window retry on the wave-per-pair blocked kernel (256 diagonals);
    what outgrows that too goes on to the generic kernel.  Same records with the rung switched off."""
    import wfa_amd as w
    from oracle import oracle as O
    # without wf-adaptive a wavefront is about as wide as its score / gap_ext
    parts = [w.generate_pairs(seed=5, n_pairs=300, length=400, error_rate=0.06, n_threads=8),    # ~100 diagonals
             w.generate_pairs(seed=6, n_pairs=60, length=2500, error_rate=0.15, n_threads=8)]    # > 256 diagonals
    qs, ts = [], []
    for blob, q_off, q_len, t_off, t_len in parts:
        for i in range(len(q_len)):
            qs.append(bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]))
            ts.append(bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]))
    data = w.make_blob(qs, ts)
    want = O.align_batch(_oracle_params(True, None), *data, n_threads=8)
    packed = {}
    for wide in (1, 0):
        al = _aligner(True, None)
        al.set_option("blk_wide", wide)
        got = al.align_arrays(*data)
        t = al.last_timing()
        assert t.main_kernel_kind == 3
        packed[wide] = t.n_packed_pairs
        assert_batch_equal(got, want, f"blk_wide={wide}")
        al.close()
    assert packed[1] >= packed[0] + 100, packed   # the 256-diagonal rung finished most of the 400 bp pairs ...
    assert packed[1] < len(qs)                    # ... and handed the widest ones on


@pytest.mark.parametrize("length,err,route", [(400, 0.12, "generic"), (400, 0.01, "blk"), (300, 0.08, "wide")])
def test_pilot_chunk(built, length, err, route):
    """wf-adaptive off on a large batch of 200+ base reads: the first 4 096 pairs are a pilot.  When most of them
    outgrow the 64-diagonal window the rest goes straight to the wave-per-pair kernel (256 diagonals) if that one
    takes the pilot's leftovers, else to the generic kernel; otherwise the blocked kernel takes the rest.  Either
    way every record equals the oracle's."""
    import os
    import wfa_amd as w
    from oracle import oracle as O
    n = 70000
    data = w.generate_pairs(seed=int(err * 1000), n_pairs=n, length=length, error_rate=err, n_threads=16)
    al = _aligner(True, None)
    got = al.align_arrays(*data)
    t = al.last_timing()
    if route == "generic":
        assert t.main_kernel_kind == 3 and t.n_packed_pairs < 4096, t
    elif route == "wide":
        assert t.main_kernel_kind == 5 and t.n_packed_pairs > n - n // 10, t
    else:
        assert t.main_kernel_kind == 3 and t.n_packed_pairs > n - n // 10, t
    want = O.align_batch(_oracle_params(True, None), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    assert_batch_equal(got, want, f"pilot L={length} err={err}")
    al.set_option("pilot", 0)
    assert_batch_equal(al.align_arrays(*data), want, f"no pilot L={length} err={err}")
    al.close()


def test_mixed_lengths_keep_the_fast_path(built):
    """A batch of short reads with a few long ones: the sub-wave pipeline is sized for the pairs that fit its LDS
    budget, the kernels hand the long pairs on (ST_REDO_LDS) and the generic / team kernels finish them."""
    import wfa_amd as w
    from oracle import oracle as O
    short = w.generate_pairs(seed=41, n_pairs=3000, length=300, error_rate=0.05)
    longp = w.generate_pairs(seed=42, n_pairs=6, length=15000, error_rate=0.05)
    blob = np.concatenate([short[0], longp[0]])
    off = np.uint64(len(short[0]))
    data = (blob, np.concatenate([short[1], longp[1] + off]), np.concatenate([short[2], longp[2]]),
            np.concatenate([short[3], longp[3] + off]), np.concatenate([short[4], longp[4]]))
    for ad in ((10, 50, 1), None):
        al = _aligner(True, ad)
        got = al.align_arrays(*data)
        t = al.last_timing()
        assert t.main_kernel_kind == 3, t
        if ad is not None:  # (without wf-adaptive many 300-base pairs outgrow the 64-diagonal window on their own)
            assert t.n_packed_pairs >= 2900, t
        assert_batch_equal(got, O.align_batch(_oracle_params(True, ad), *data, n_threads=8), f"mixed lengths ad={ad}")
        al.close()


@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_fuzz_configs(built, seed):
    """Seeded random configurations: penalties (the 2:4:1 shapes the blocked kernel takes and others), global /
    semi-global, wf-adaptive parameters, ragged lengths from 1 to 1 500 bases, error rates up to 30 %."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    pens = [(4, 6, 2), (2, 3, 1), (8, 12, 4), (6, 9, 3), (1, 1, 1), (3, 5, 2), (2, 6, 1), (5, 3, 3)]
    for _ in range(6):
        pen = pens[int(rng.integers(0, len(pens)))]
        glob = bool(rng.integers(0, 4))  # mostly global
        ad = None if rng.integers(0, 3) == 0 else (int(rng.integers(1, 24)), int(rng.integers(1, 120)), 1)
        n = int(rng.integers(50, 400))
        qs, ts = [], []
        for i in range(n):
            L = int(rng.integers(1, 1500 if glob else 400))
            q = rng.integers(0, 4, L)
            t = list(q)
            for _e in range(int(L * rng.uniform(0, 0.3))):
                kind, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(1, len(t))))
                if kind == 0 and t:
                    t[pos] = int(rng.integers(0, 4))
                elif kind == 1:
                    t.insert(pos, int(rng.integers(0, 4)))
                elif len(t) > 1:
                    del t[pos]
            qs.append(bytes(b"ACGT"[c] for c in q)), ts.append(bytes(b"ACGT"[c] for c in t) or b"G")
        data = w.make_blob(qs, ts)
        al = _aligner(glob, ad, pen)
        got = al.align_arrays(*data)
        want = O.align_batch(_oracle_params(glob, ad, pen), *data, n_threads=8)
        assert_batch_equal(got, want, f"fuzz seed={seed} pen={pen} glob={glob} ad={ad} n={n}")
        al.close()


def test_small_arena_forces_retry_ladder(built):
    """A deliberately tiny wavefront arena: pairs overflow, are re-run with 8x slots, results unchanged."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=21, n_pairs=500, length=400, error_rate=0.1)
    al = _aligner(True, None)
    al.set_option("arena_bytes_per_slot", 16 * 1024)
    got = al.align_arrays(*data)
    t = al.last_timing()
    assert t.n_launches >= 2 and t.n_retried_pairs > 0
    assert_batch_equal(got, O.align_batch(_oracle_params(True, None), *data, n_threads=8), "retry")
    al.close()


def test_multiwave_configuration(built):
    """256- and 1024-thread workgroups per pair (the long-read configuration) on short inputs."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=33, n_pairs=300, length=700, error_rate=0.1)
    for tpp in (256, 1024):
        for glob in (True, False):
            al = _aligner(glob, (10, 50, 1))
            al.set_option("threads_per_pair", tpp)
            got = al.align_arrays(*data)
            assert_batch_equal(got, O.align_batch(_oracle_params(glob), *data, n_threads=8), f"tpp={tpp} glob={glob}")
            al.close()


@pytest.mark.parametrize("team_wgs,solo_max", [(2, 0), (5, 16), (3, 4096)])
def test_team_kernel_small(built, team_wgs, solo_max):
    """wfa_team_kernel (several workgroups per pair, for wide wavefronts) forced onto short pairs: team mode for
    every row (solo_max 0), switches between team and solo mode (16), solo mode after the first barrier (4096);
    global and semi-global, wf-adaptive on and off (the arena retry ladder runs through it too)."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=11, n_pairs=40, length=700, error_rate=0.1)
    for glob in (True, False):
        for ad in ((10, 50, 1), None):
            al = _aligner(glob, ad)
            for k, v in (("packed", 0), ("team_min_len", 1), ("team_wgs", team_wgs), ("team_solo_max", solo_max)):
                al.set_option(k, v)
            got = al.align_arrays(*data)
            want = O.align_batch(_oracle_params(glob, ad), *data, n_threads=8)
            assert_batch_equal(got, want, f"team T={team_wgs} solo_max={solo_max} glob={glob} ad={ad}")
            ref = _aligner(glob, ad)
            ref.set_option("packed", 0), ref.set_option("team_min_len", 0)
            ref.align_arrays(*data)
            assert al.last_timing().cells_stored == ref.last_timing().cells_stored
            ref.close()
            al.close()


def test_long_pair_semiglobal(built):
    """One 20 kbp pair, semi-global + adaptive (wide seeded wavefronts, 256 threads per pair)."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=5, n_pairs=2, length=20000, error_rate=0.05)
    for glob in (True, False):
        al = _aligner(glob, (10, 50, 1))
        got = al.align_arrays(*data)
        assert_batch_equal(got, O.align_batch(_oracle_params(glob), *data, n_threads=2), f"20kbp glob={glob}")
        al.close()


def test_config5_sample(built):
    """BASELINE configs[4] in miniature: 40 kbp pairs @10 %, semi-global + wf-adaptive.  The seeded wavefronts
    are ~8e4 diagonals wide for thousands of scores (the reference's own rules), so this exercises the team
    kernel (several workgroups per pair: team and solo mode, team-wide end-cell search) and the arena retry
    ladder into multi-GB slots; the one-workgroup-per-pair kernel must give the same records."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=5, n_pairs=2, length=40000, error_rate=0.10)
    al = _aligner(False, (10, 50, 1))
    got = al.align_arrays(*data)
    want = O.align_batch(_oracle_params(False), *data, n_threads=2)
    assert_batch_equal(got, want, "C5 sample")
    assert al.last_timing().n_launches >= 2
    cells = al.last_timing().cells_stored
    al.set_option("team_min_len", 0)  # wfa_generic_kernel<16,0>
    assert_batch_equal(al.align_arrays(*data), want, "C5 sample, one workgroup per pair")
    assert al.last_timing().cells_stored == cells
    al.close()


@pytest.mark.parametrize("penalties", [(4, 6, 2), (2, 3, 1), (6, 4, 2), (2, 4, 2), (1, 1, 1), (5, 20, 3), (3, 40, 1)])
def test_team_kernel_wave_mode(built, penalties):
    """The team kernel's wave mode (rows of <= 64 diagonals stepped by one wave out of an LDS ring of the last rows;
    backtrace by a wave over an LDS window of the directory) against the oracle: ring depths 2..64 rows
    (farthest source 1..41 scores back), bands that grow past 64 diagonals and shrink again (adaptive off: wave ->
    solo -> wave), global and semi-global; the same batch with wave mode off must store the same number of cells."""
    import wfa_amd as w
    from oracle import oracle as O
    a = w.generate_pairs(seed=21, n_pairs=16 if max(penalties) >= 20 else 24, length=600, error_rate=0.08)
    # (the two sets with a 20 / 40 gap open reach scores -- rows, and oracle seconds -- several times the others': a smaller second batch)
    heavy = max(penalties) >= 20
    b = w.generate_pairs(seed=22, n_pairs=2 if heavy else 4, length=1800 if heavy else 4000, error_rate=0.12)
    for data in (a, b):
        for glob, ad in ((True, (10, 50, 1)), (False, (10, 50, 1)), (True, None), (False, (4, 5, 1))):
            want = O.align_batch(_oracle_params(glob, ad, penalties), *data, n_threads=8)
            cells = []
            for wave, strict in ((1, 1), (0, 1)):
                al = _aligner(glob, ad, penalties)
                for k, v in (("packed", 0), ("team_min_len", 1), ("team_wgs", 3), ("team_solo_max", 4096), ("team_wave", wave),
                             ("team_strict", strict), ("arena_poison", 1)):  # (poison: a stale arena word cannot pass)
                    al.set_option(k, v)
                got = al.align_arrays(*data)
                assert_batch_equal(got, want, f"wave={wave} strict={strict} pen={penalties} glob={glob} ad={ad}")
                cells.append(al.last_timing().cells_stored)
                al.close()
            assert len(set(cells)) == 1


def test_sub_wave_kernels_read_only_what_they_wrote(built):
    """The blocked / register kernels zero nothing and skip the words nothing reads: over an arena filled with a
    pattern before every forward launch (option arena_poison) every record and CIGAR equals the oracle's -- first
    pass, the wider retry rungs and the short-read instance."""
    import wfa_amd as w
    from oracle import oracle as O
    for length, err, glob, ad in ((1000, 0.05, True, (10, 50, 1)), (400, 0.12, True, None), (150, 0.02, True, None)):
        data = w.generate_pairs(seed=17, n_pairs=3000, length=length, error_rate=err, n_threads=8)
        want = O.align_batch(_oracle_params(glob, ad), *data, n_threads=8)
        al = _aligner(glob, ad)
        al.set_option("arena_poison", 1)
        assert_batch_equal(al.align_arrays(*data), want, f"poisoned arena, {length} bp @{err}")
        al.close()


def test_team_kernel_sees_no_stale_arena_words(built):
    """Rows written by one workgroup of a team are read by the others after a barrier; a read that overtakes the
    write returns what an EARLIER launch left at that address -- the right value when the same batch is run twice,
    which is how such a race hides.  Two different batches alternate through one aligner (each launch finds the other
    batch's rows in the arena), then once more over an arena filled with a pattern: every result equals the
    oracle's."""
    import wfa_amd as w
    from oracle import oracle as O
    batches = [w.generate_pairs(seed=sd, n_pairs=4, length=20000, error_rate=0.10) for sd in (5, 6)]
    wants = [O.align_batch(_oracle_params(False), *b, n_threads=4) for b in batches]
    al = _aligner(False, (10, 50, 1))
    al.set_option("team_min_len", 1)
    for rnd in range(3):
        for i in (0, 1):
            assert_batch_equal(al.align_arrays(*batches[i]), wants[i], f"round {rnd} batch {i}")
    al.set_option("arena_poison", 1)
    for i in (0, 1):
        assert_batch_equal(al.align_arrays(*batches[i]), wants[i], f"poisoned arena, batch {i}")
    al.close()


@pytest.mark.parametrize("penalties", [(4, 6, 2), (5, 20, 3), (2, 3, 1)])
def test_team_kernel_wavefronts_word_for_word(built, penalties):
    """Every stored M / I / D word of the team kernel -- team, solo and wave mode, and the switches between them --
    against the oracle's final wavefronts (the dump entry runs the team kernel when `team_wgs` is set)."""
    import wfa_amd as w
    from oracle import oracle as O
    blob, qo, ql, to, tl = w.generate_pairs(seed=31, n_pairs=3, length=900, error_rate=0.1)
    for glob, ad in ((True, (10, 50, 1)), (False, (10, 50, 1)), (False, (4, 5, 1)), (True, None)):
        oa = O.Aligner(_oracle_params(glob, ad, penalties))
        for wave, solo_max in ((1, 4096), (1, 100), (0, 4096)):
            al = _aligner(glob, ad, penalties)
            for k, v in (("packed", 0), ("team_min_len", 1), ("team_wgs", 3), ("team_solo_max", solo_max), ("team_wave", wave)):
                al.set_option(k, v)
            for i in range(len(ql)):
                q, t = bytes(blob[qo[i]:qo[i] + ql[i]]), bytes(blob[to[i]:to[i] + tl[i]])
                r = oa.align(q, t)
                want = {c: {sc: {lo + j: v for j, v in enumerate(raw) if v} for sc, (lo, hi, raw) in d.items()}
                        for c, d in oa.dump().items()}
                wf, res = al.debug_wavefronts(q, t)
                assert res.Score == r.score
                for c in "MID":
                    assert wf[c] == {sc: row for sc, row in want[c].items() if row}, (penalties, glob, ad, wave, solo_max, i, c)
            al.close()
        oa.close()


@pytest.mark.parametrize("slack", [1024, 3])
def test_teamc_kernel_words_and_results(built, slack):
    """wfa_teamc_kernel (round 5: one backtrace word per diagonal in the arena, the rows the next steps source in LDS stripes or the
    team's exchange rows, reductions travelling with the barrier, the semi-global end cell found in flight) against the oracle:
    EVERY word it leaves in the arena -- the pre-extension offset backTrace recomputes (wfa.go:766-817) and the decisions of
    next() -- for pairs whose rows span several stripes and cross their edges (semi-global seeds: n + m - 1 diagonals;
    wf-adaptive off: bands that grow until the axis has to move), with the axis positioned with 3 diagonals of room (slack = 3: it
    moves every few steps), XBUF / team-stripe / solo-stripe / wave mode and the switches between them, three penalty shapes;
    then the results of 9 kbp pairs through three and five workgroups per team against wfa_team_kernel's and the oracle's."""
    import wfa_amd as w
    from oracle import oracle as O
    blob, qo, ql, to, tl = w.generate_pairs(seed=41, n_pairs=1, length=2100 if slack == 3 else 1300, error_rate=0.08)
    for pen in ((4, 6, 2), (2, 4, 2), (6, 4, 2)):
        for glob, ad in ((False, (10, 50, 1)), (True, None), (False, None), (True, (10, 50, 1))):
            oa = O.Aligner(_oracle_params(glob, ad, pen))
            # (the default penalties see every shape of team; the others the two that differ most)
            # (fast = 0: every step of the stripe modes takes the general step instead of the short one of their steady state)
            # (fast = 2: the short steps without the pipelined ones, which are the default and need wf-adaptive, a team and e/g = 1, x/g >= 2)
            # (slack = 3, the 2 100-base pair: the other penalties see one shape of team -- the run is bound by the word-for-word
            # comparison in Python, and the moving axis does not depend on the penalties)
            for wgs, solo_max, wave, fast in ((((2, 64, 1, 1), (3, 0, 1, 1), (3, 0, 1, 2)) if slack == 3 else
                                               ((2, 64, 1, 1), (2, 4096, 1, 1), (1, 4096, 0, 1), (3, 0, 1, 1), (2, 64, 1, 0), (3, 0, 1, 2))) if pen == (4, 6, 2)
                                              else ((3, 0, 1, 1),) if slack == 3 else ((2, 64, 1, 1), (3, 0, 1, 1))):
                al = _aligner(glob, ad, pen)
                for k, v in (("packed", 0), ("team_min_len", 1), ("team_wgs", wgs), ("team_solo_max", solo_max), ("team_wave", wave),
                             ("team_slack", slack), ("team_fast", min(fast, 1)), ("team_pipe", 0 if fast == 2 else 1), ("arena_poison", 1)):
                    al.set_option(k, v)
                for i in range(len(ql)):
                    q, t = bytes(blob[qo[i]:qo[i] + ql[i]]), bytes(blob[to[i]:to[i] + tl[i]])
                    r = oa.align(q, t)
                    exp = _expected_compact_words(oa.dump(), *pen)
                    rows, res = al.debug_team_compact(q, t)
                    what = (pen, glob, ad, wgs, solo_max, wave, fast, i)
                    assert (res.Score, res.CIGAR(False)) == (r.score, r.cigar), what
                    assert (res.QBegin, res.QEnd, res.TBegin, res.TEnd, res.AlignLen, res.Matches, res.Gaps, res.GapRegions) == (
                        r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps, r.gap_regions), what
                    for (sc, k), (wv, mask) in exp.items():
                        have = rows.get(sc, {}).get(k, 0)
                        assert (have & mask) == (wv & mask), (what, sc, k, hex(have), hex(wv), hex(mask))
                    # ... and nothing else: inside a row's kept band a word is there exactly where the reference holds an M cell
                    want_cells = {(sc, k) for (sc, k) in exp}
                    have_cells = {(sc, k) for sc, row in rows.items() for k, wd in row.items() if wd}
                    assert have_cells == want_cells, (what, sorted(have_cells ^ want_cells)[:8])
                al.close()
            oa.close()
    data = w.generate_pairs(seed=42, n_pairs=6, length=9000, error_rate=0.06)
    want = O.align_batch(_oracle_params(False, (10, 50, 1)), *data, n_threads=6)
    cells = []
    for wgs, compact, fast in ((3, 1, 1), (5, 1, 1), (3, 1, 0), (3, 1, 2), (3, 0, 1)):
        al = _aligner(False, (10, 50, 1))
        for k, v in (("packed", 0), ("team_min_len", 1), ("team_wgs", wgs), ("team_compact", compact), ("team_slack", slack), ("team_fast", min(fast, 1)),
                     ("team_pipe", 0 if fast == 2 else 1), ("team_order", min(fast, 1)), ("arena_poison", 1)):
            al.set_option(k, v)
        assert_batch_equal(al.align_arrays(*data), want, f"9 kbp semi-global, {wgs} workgroups per team, compact={compact}, fast={fast}")
        assert al.last_timing().main_kernel_kind == (17 if compact else 7)
        cells.append(al.last_timing().cells_stored)
        al.close()
    assert len(set(cells)) == 1, cells  # (the census of stored wavefront words is the reference's, whatever the arena holds)


@pytest.mark.parametrize("penalties", [(4, 6, 2), (5, 20, 3)])
def test_generic_kernel_wavefronts_word_for_word(built, penalties):
    """The one-workgroup-per-pair kernel, with its wave mode (rows of <= 64 diagonals stepped by one wave,
    wfa_wave.hpp) and without: every stored M / I / D word against the oracle's final wavefronts -- global pairs run
    in wave mode from the first score, semi-global ones enter it when the band collapses, wf-adaptive off leaves it
    when the band outgrows 64 diagonals."""
    import wfa_amd as w
    from oracle import oracle as O
    blob, qo, ql, to, tl = w.generate_pairs(seed=33, n_pairs=3, length=1200, error_rate=0.08)
    for glob, ad in ((True, (10, 50, 1)), (False, (10, 50, 1)), (True, None), (False, (4, 5, 1))):
        oa = O.Aligner(_oracle_params(glob, ad, penalties))
        for wave in (1, 0):
            al = _aligner(glob, ad, penalties)
            al.set_option("team_wave", wave)
            for i in range(len(ql)):
                q, t = bytes(blob[qo[i]:qo[i] + ql[i]]), bytes(blob[to[i]:to[i] + tl[i]])
                r = oa.align(q, t)
                want = {c: {sc: {lo + j: v for j, v in enumerate(raw) if v} for sc, (lo, hi, raw) in d.items()}
                        for c, d in oa.dump().items()}
                wf, res = al.debug_wavefronts(q, t)
                assert res.Score == r.score
                for c in "MID":
                    assert wf[c] == {sc: row for sc, row in want[c].items() if row}, (penalties, glob, ad, wave, i, c)
            al.close()
        oa.close()


@pytest.mark.timeout(1500)
def test_config5_full_length_pair(built):
    """BASELINE configs[4] at its stated length: the EIGHT 100 kbp pairs @10 % of the bench's --config c5s sample (seed 5),
    semi-global + wf-adaptive 10/50/1 -- eight teams on the paged arena, four of the pairs with 1e5-diagonal wavefronts up to
    score 65 000 -- every record field and every CIGAR op against the oracle, twice over a poisoned pool.  The oracle's results
    come from the committed fixture tests/golden/c5_sample_oracle.json (made by tests/golden/make_c5_golden.py: all record
    fields + the SHA-256 of each pair's op array -- bit-exact, without the minute and a half of a host core and the 30 GB a
    hard pair costs the oracle); WFA_TEST_FULL_ORACLE=1 runs the oracle itself instead."""
    import hashlib
    import wfa_amd as w
    from conftest import load_golden
    data = w.generate_pairs(seed=5, n_pairs=8, length=100_000, error_rate=0.10)
    al = _aligner(False, (10, 50, 1))
    al.set_option("arena_poison", 1)  # no word an earlier launch left in the arena may be read
    got = al.align_arrays(*data)
    again = al.align_arrays(*data)
    if os.environ.get("WFA_TEST_FULL_ORACLE", "0") not in ("", "0"):
        from oracle import oracle as O
        want = O.align_batch(_oracle_params(False), *data, n_threads=8)
        assert_batch_equal(got, want, "C5 full length")
        assert_batch_equal(again, want, "C5 full length, second call")
    else:
        gold = load_golden("c5_sample_oracle.json")
        assert hashlib.sha256(data[0].tobytes()).hexdigest() == gold["input_sha256"], "the generator no longer yields the fixture's dataset"
        for tag, r in (("first call", got), ("second call", again)):
            for i, exp in enumerate(gold["pairs"]):
                for f in FIELDS:
                    assert int(getattr(r, f)[i]) == exp[f], (tag, i, f, int(getattr(r, f)[i]), exp[f])
                ops = np.ascontiguousarray(r.pair_ops(i)).astype("<u8")
                assert hashlib.sha256(ops.tobytes()).hexdigest() == exp["ops_sha256"], (tag, i, "CIGAR ops")
    assert (got.status == 0).all() and got.score.min() > 10_000
    al.close()


def test_failed_retry_pass_fails_the_call(built):
    """A sub-wave retry pass that cannot get its arena (injected: option fail_pass = 5, the 256-diagonal rung) must
    fail the whole call with its error code -- not drop the pairs it was given and report success."""
    import ctypes as C
    import wfa_amd as w
    from wfa_amd import _lib as L
    data = w.generate_pairs(seed=5, n_pairs=300, length=400, error_rate=0.06, n_threads=8)  # bands of ~100 diagonals
    al = _aligner(True, None)
    ok = al.align_arrays(*data)
    assert al.last_timing().n_retried_pairs > 0
    al.set_option("fail_pass", 5)
    with pytest.raises(L.WfaHipError) as ei:
        al.align_arrays(*data)
    assert ei.value.code == L.ERR_OOM
    al.set_option("fail_pass", 0)
    again = al.align_arrays(*data)
    assert_batch_equal(again, ok, "after the injected failure")
    al.close()


def test_full_size_properties(built):
    """Size-independent properties at a BASELINE-sized batch (2e5 x 1 kbp): every CIGAR's gap-affine cost equals
    its score, it consumes exactly both sequences (except where the reference's own off-by-one overshoot,
    SURVEY.md 3.3 `next`, yields a CIGAR one base too long -- about 1 pair in 2e5, reproduced bit-for-bit),
    statistics are consistent with the ops, and a second run is bit-identical (determinism)."""
    import wfa_amd as w
    n = 200_000
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=3, n_pairs=n, length=1000, error_rate=0.05)
    al = _aligner(True, (10, 50, 1))
    a = al.align_arrays(blob, q_off, q_len, t_off, t_len)
    b = al.align_arrays(blob, q_off, q_len, t_off, t_len)
    assert (a.status == 0).all()
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(a.ops, b.ops)
    letters = (a.ops >> np.uint64(32)).astype(np.uint8)
    counts = (a.ops & np.uint64(0xFFFFFFFF)).astype(np.int64)
    pair_of_op = np.repeat(np.arange(n), a.ops_len)
    q_used = np.bincount(pair_of_op, weights=counts * np.isin(letters, list(b"MXDH")), minlength=n).astype(np.int64)
    t_used = np.bincount(pair_of_op, weights=counts * np.isin(letters, list(b"MXI")), minlength=n).astype(np.int64)
    dq, dt = q_used - q_len.astype(np.int64), t_used - t_len.astype(np.int64)
    assert (np.abs(dq) <= 1).all() and (np.abs(dt) <= 1).all()
    assert ((dq != 0) | (dt != 0)).sum() <= n // 10_000
    cost = np.bincount(pair_of_op, weights=(letters == ord("X")) * counts * 4
                       + np.isin(letters, list(b"IDH")) * (6 + 2 * counts), minlength=n)
    assert np.array_equal(cost.astype(np.int64), a.score.astype(np.int64))
    matches = np.bincount(pair_of_op, weights=(letters == ord("M")) * counts, minlength=n)
    assert (a.matches <= matches).all()
    al.close()


def test_full_size_parity_c3(built):
    """BASELINE configs[2] at full size (1e6 x 1 kbp @5 %, global + wf-adaptive) against the oracle run on
    all host cores of the GPU box (seconds there), plus configs[1] (1e5 x 150 bp @2 %, adaptive off)."""
    import os
    import wfa_amd as w
    from oracle import oracle as O
    thr = max(8, (os.cpu_count() or 8) // 2)
    for (n, length, err, seed, ad) in ((1_000_000, 1000, 0.05, 3, (10, 50, 1)), (100_000, 150, 0.02, 2, None)):
        data = w.generate_pairs(seed=seed, n_pairs=n, length=length, error_rate=err, n_threads=32)
        al = _aligner(True, ad)
        got = al.align_arrays(*data)
        cells = al.last_timing().cells_stored
        want = O.align_batch(_oracle_params(True, ad), *data, n_threads=thr)
        assert_batch_equal(got, want, f"full size L={length}")
        # streamed backtrace -- waves of the forward launch walk finished pairs while the others are still aligning;
        # the hand-over between them must hold every time
        al.set_option("bt_stream_single", 1)
        al.set_option("duo", 0)  # (the streaming instance belongs to wfa_blk_kernel<16,1>; batches this large start on wfa_duo_kernel)
        for rep in range(3):
            assert_batch_equal(al.align_arrays(*data), want, f"full size L={length}, streamed, repeat {rep}")
            assert al.last_timing().cells_stored == cells
        al.set_option("bt_stream_single", 0)
        assert_batch_equal(al.align_arrays(*data), want, f"full size L={length}, wfa_blk_kernel<16,1>")
        al.set_option("duo", 1)
        # the retry passes run beside the first pass's backtrace kernel: same records and same cell census as the
        # serial schedule
        al.set_option("tail_overlap", 0)
        again = al.align_arrays(*data)
        assert_batch_equal(again, want, f"full size L={length}, serial schedule")
        assert al.last_timing().cells_stored == cells
        al.set_option("tail_overlap", 1)
        # the counting instance of the forward kernel (option census: off by default) gives the same records; its count
        # is at most what the oracle ever set (wf-adaptive deletes some words afterwards) -- the exact count is checked
        # pair by pair in test_blocked_kernel_arena_word_for_word
        assert cells == 0
        al.set_option("census", 1)
        assert_batch_equal(al.align_arrays(*data), want, f"full size L={length}, census on")
        assert 0 < al.last_timing().cells_stored <= int(want.cells.sum())
        al.close()


def test_device_entry_with_an_unaligned_ops_buffer(built):
    """wfahip_align_batch_device with caller-owned device buffers: the backtrace combines eight CIGAR ops per 64-byte
    store when the ops buffer allows it; a buffer that is only 8-byte aligned must give the same result (single
    stores), and a buffer that is too small must be reported with the capacity that is needed."""
    import ctypes as C
    import torch
    import wfa_amd as w
    from wfa_amd import _lib as L
    from oracle import oracle as O
    n = 20000
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=77, n_pairs=n, length=500, error_rate=0.05, n_threads=8)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), blob, q_off, q_len, t_off, t_len, n_threads=8)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(a).to(dev) for a in (blob, q_off.view(np.int64), q_len.view(np.int32), t_off.view(np.int64),
                                               t_len.view(np.int32))]
    al = _aligner(True, (10, 50, 1))
    prm = al._params()
    cap = int(want.ops_len.astype(np.int64).sum()) * 4 + 16 * n
    d_ops_all = torch.zeros(cap + 8, dtype=torch.int64, device=dev)
    d_rec = torch.zeros((n, L.REC_WORDS), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)  # (torch's fills are on its stream, the calls below on the context's own)
    for shift in (0, 1, 3):  # ops buffer 64-byte aligned / 8 bytes off / 24 bytes off
        d_ops = d_ops_all[shift:shift + cap]
        needed = C.c_uint64()
        rc = L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), d[0].data_ptr(), blob.size, d[1].data_ptr(), d[2].data_ptr(),
                                               d[3].data_ptr(), d[4].data_ptr(), n, 0, d_rec.data_ptr(), d_ops.data_ptr(), cap,
                                               C.byref(needed), None)
        L.check(rc, "wfahip_align_batch_device")
        rec = d_rec.cpu().numpy().view(np.uint32)
        ops = d_ops.cpu().numpy().view(np.uint64)
        assert np.array_equal(rec[:, L.REC_SCORE], want.score) and np.array_equal(rec[:, L.REC_OPS_LEN], want.ops_len), shift
        off = rec[:, L.REC_OPS_OFF_LO].astype(np.uint64) | (rec[:, L.REC_OPS_OFF_HI].astype(np.uint64) << np.uint64(32))
        for i in range(0, n, 97):
            assert np.array_equal(ops[int(off[i]):int(off[i]) + int(rec[i, L.REC_OPS_LEN])], want.pair_ops(i)), (shift, i)
    needed = C.c_uint64()
    rc = L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), d[0].data_ptr(), blob.size, d[1].data_ptr(), d[2].data_ptr(),
                                           d[3].data_ptr(), d[4].data_ptr(), n, 0, d_rec.data_ptr(), d_ops_all.data_ptr(), 1000,
                                           C.byref(needed), None)
    assert rc == L.ERR_OOM and needed.value > 1000, (rc, needed.value)
    al.close()


def test_host_entry_slices_agree_with_the_plain_call(built):
    """wfahip_align_batch aligns a large batch in four slices while the rest of the blob is still uploading, when the
    slices refer to (nearly) disjoint parts of the blob.  Same pairs in shuffled order (every slice then refers to
    the whole blob: plain upload-then-align) and with the overlap switched off: identical results pair by pair."""
    import os
    import wfa_amd as w
    from oracle import oracle as O
    n = 240000
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=91, n_pairs=n, length=300, error_rate=0.05, n_threads=16)
    assert blob.size >= (64 << 20)
    al = _aligner(True, (10, 50, 1))
    a = al.align_arrays(blob, q_off, q_len, t_off, t_len)                      # sliced
    perm = np.random.default_rng(5).permutation(n)
    b = al.align_arrays(blob, q_off[perm], q_len[perm], t_off[perm], t_len[perm])  # not sliced: ranges overlap
    os.environ["WFAHIP_NO_UPLOAD_OVERLAP"] = "1"
    try:
        c = al.align_arrays(blob, q_off, q_len, t_off, t_len)                  # not sliced: switched off
    finally:
        del os.environ["WFAHIP_NO_UPLOAD_OVERLAP"]
    assert_batch_equal(a, c, "sliced vs plain")
    for f in ("status",) + FIELDS:
        assert np.array_equal(getattr(a, f)[perm], getattr(b, f)), f
    for i in range(0, n, 1013):
        assert np.array_equal(a.pair_ops(int(perm[i])), b.pair_ops(i)), i
    sample = slice(0, 20000)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), blob, q_off[sample], q_len[sample], t_off[sample], t_len[sample], n_threads=8)
    assert np.array_equal(a.score[sample], want.score) and np.array_equal(a.ops_len[sample], want.ops_len)
    # the host op array is sized from the first slice; here the later slices have five times as many ops per pair,
    # so what does not fit is fetched at the end into an array of the exact size
    p1 = w.generate_pairs(seed=92, n_pairs=n // 4, length=300, error_rate=0.01, n_threads=16)
    p2 = w.generate_pairs(seed=93, n_pairs=n - n // 4, length=300, error_rate=0.10, n_threads=16)
    off = np.uint64(len(p1[0]))
    mixed = (np.concatenate([p1[0], p2[0]]), np.concatenate([p1[1], p2[1] + off]), np.concatenate([p1[2], p2[2]]),
             np.concatenate([p1[3], p2[3] + off]), np.concatenate([p1[4], p2[4]]))
    d = al.align_arrays(*mixed)
    os.environ["WFAHIP_NO_UPLOAD_OVERLAP"] = "1"
    try:
        e = al.align_arrays(*mixed)
    finally:
        del os.environ["WFAHIP_NO_UPLOAD_OVERLAP"]
    assert_batch_equal(d, e, "sliced vs plain, op-count estimate too low")
    al.close()


def _expected_compact_words(dump, x, o, e):
    """{(s, k): (word, mask)} from the oracle's final wavefronts, in the blocked kernels' encoding (wfa_device.hpp,
    blk_word): which source the M cell took, whether the I / D cells are extensions, and the pre-extension offset exactly
    as the reference's backTrace recomputes it from the UNBOUNDED sources (wfa.go:766-817)."""
    M, I, D = ({s: {lo + i: v for i, v in enumerate(raw) if v} for s, (lo, hi, raw) in dump[c].items()} for c in "MID")

    def get(C, s, k):  # Component.Get: 0 when the score underflows or nothing is stored (wfa_component.go:158-167)
        return (C.get(s, {}).get(k, 0) >> 3) if s >= 0 else 0

    out = {}
    for s, row in M.items():
        for k, raw in row.items():
            tag = raw & 7
            a0, b0 = get(M, s - o - e, k - 1), get(I, s - e, k - 1)
            c0, d0 = get(M, s - o - e, k + 1), get(D, s - e, k + 1)
            x0 = get(M, s - x, k)
            isk = max(a0, b0) + 1 if (a0 or b0) else 0   # wfa.go:768-776 / 790-797
            dsk = max(c0, d0)                             # wfa.go:779-787 / 799-806
            if tag == 2:
                off0 = isk
            elif tag == 4:
                off0 = dsk
            else:
                off0 = max(isk, dsk, x0 + 1) if (isk or dsk or x0) else 0  # wfa.go:808-813 (else: fromItself)
            ic = I.get(s, {}).get(k, 0) & 7               # 0 / InsOpen 1 / InsExt 2
            dt = D.get(s, {}).get(k, 0) & 7               # 0 / DelOpen 3 / DelExt 4
            # (word, mask): the I / D bits only mean something where that cell exists
            if off0 == 0:   # a seed of initComponents: Match / Mismatch in bits 0 / 1, offset field 0
                out[(s, k)] = (1 if tag == 6 else 2, 0xFFFFFFF3)
                continue
            word = (off0 << 4) | ((ic == 2) << 3) | ((dt == 4) << 2) | ((tag == 5) << 1) | (tag in (1, 2))
            mask = 0xFFFFFFF3 | (8 if ic else 0) | (4 if dt else 0)
            if tag == 5:
                mask &= ~1  # (the mismatch won: whether the insertion tied with it is not part of the decision)
            assert tag in (1, 2, 3, 4, 5) and (tag not in (1, 2) or tag == ic) and (tag not in (3, 4) or tag == dt)
            out[(s, k)] = (word, mask)
    return out


def _arena_slot(fmt, i, k, n_words=0):
    if fmt == 10:      # (wfa_duo_kernel, round 6: groups two by two -- [diagonal / 8 & 7][index / 8][diagonal / 4 & 1][index & 7][diagonal & 3])
        return ((k & 56) >> 3) * (n_words // 64) * 8 + 64 * (i >> 3) + ((k & 4) << 3) + ((i & 7) << 2) + (k & 3)
    if fmt == 9:       # (wfa_duo_kernel, round 6: group-major halfwords -- [diagonal / 4 & 15][score index][diagonal & 3], n_words / 64 score indices)
        return ((k & 60) >> 2) * (n_words // 64) * 4 + 4 * i + (k & 3)
    if fmt in (3, 7):  # (fmt 7: the same tiles with 16-bit words -- the caller views the arena as uint16)
        return (i >> 3) * 512 + (((k & 63) >> 2) << 5) + ((i & 7) << 2) + (k & 3)
    if fmt == 8:       # (wfa_lane_kernel: 32 halfwords per score)
        return i * 32 + (k & 31)
    return i * {1: 64, 4: 256, 5: 32, 6: 128}[fmt] + (k & ({1: 63, 4: 255, 5: 31, 6: 127}[fmt]))


@pytest.mark.parametrize("length,err,pen,ad,fmt", [(1000, 0.05, (4, 6, 2), (10, 50, 1), 3), (400, 0.08, (4, 6, 2), (10, 50, 1), 3),
                                                   (300, 0.03, (2, 3, 1), None, 3), (600, 0.05, (8, 12, 4), (4, 10, 1), 3),
                                                   (150, 0.02, (4, 6, 2), None, 5), (120, 0.06, (4, 6, 2), (10, 50, 1), 5)])
@pytest.mark.parametrize("census", [0, 1])
@pytest.mark.parametrize("duo", [0, 1])
def test_blocked_kernel_arena_word_for_word(built, length, err, pen, ad, fmt, census, duo):
    _arena_word_check(length, err, pen, ad, fmt, census, duo, 0)


@pytest.mark.parametrize("length,err,pen,ad", [(150, 0.02, (4, 6, 2), None), (120, 0.04, (4, 6, 2), (10, 50, 1)), (230, 0.03, (2, 3, 1), (5, 20, 1)),
                                               (60, 0.06, (8, 12, 4), None)])
@pytest.mark.parametrize("census", [0, 1])
def test_lane_kernel_arena_word_for_word(built, length, err, pen, ad, census):
    """The same for wfa_lane_kernel (a lane per pair, short reads): rows of 32 halfwords."""
    _arena_word_check(length, err, pen, ad, 8, census, 0, 1)


def _arena_word_check(length, err, pen, ad, fmt, census, duo, lane, min_pairs=None):
    """The dominant kernel's stored state, not only its results: every compact backtrace word wfa_blk_kernel leaves in
    HBM (M tag, I and D tag bits, pre-extension offset) against what the oracle's wavefronts imply -- visited by the
    backtrace or not.  Covers wf-adaptive pruning (deleted cells must not be there with a source role), ragged lengths
    (cells at sequence ends: the exact WF_NEXT with its rejections, where the stored offset is the UNBOUNDED
    recomputation), both tile formats (8 x 64 tiles of the 16-lane instance, 32-word rows of the 8-lane one)."""
    import wfa_amd as w
    from oracle import oracle as O
    n = 160
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=length + int(err * 1000), n_pairs=n, length=length, error_rate=err)
    rng = np.random.default_rng(length)
    cut = rng.integers(0, 4, n) == 0  # a quarter of the pairs lose a piece of one sequence: overhangs, early sequence ends
    q_len = np.where(cut & (np.arange(n) % 2 == 0), np.maximum(1, q_len - rng.integers(1, 25, n)), q_len).astype(np.uint32)
    t_len = np.where(cut & (np.arange(n) % 2 == 1), np.maximum(1, t_len - rng.integers(1, 25, n)), t_len).astype(np.uint32)
    al = _aligner(True, ad, pen)
    al.set_option("census", census)  # (two instances of the kernel: with and without the count of stored words)
    if fmt == 5 and duo:
        pytest.skip("short reads stay on the batched 8-lane instance")
    al.set_option("duo", 2 * duo)  # wfa_duo_kernel (8 or 16 lanes per pair) writes the same arena as wfa_blk_kernel<16,1>
    al.set_option("lane", 2 * lane)
    got = al.align_arrays(blob, q_off, q_len, t_off, t_len)
    assert al.last_timing().main_kernel_kind == (10 if lane else 6 if fmt == 5 else (8 if duo else 3))
    g = np.gcd.reduce([pen[0], pen[1] + pen[2], pen[2]])
    oa = O.Aligner(O.make_params(*pen, global_alignment=True, adaptive=ad))
    checked = pairs = 0
    for i in range(n):
        words, f, meta = al.debug_compact_arena(i)
        assert f == (10 if duo else fmt)
        if f in (7, 8, 9, 10):
            words = words.view(np.uint16)
        if meta[0] != 0:  # handed on to another kernel (band / arena): its slot is not the final state
            continue
        q = bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])])
        t = bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])])
        r = oa.align(q, t)
        assert meta[1] == r.score == int(got.score[i])
        dump = oa.dump()
        exp = _expected_compact_words(dump, *pen)
        # the census: every M, I and D word the reference still holds after the alignment (wf-adaptive deletes some)
        assert meta[3] == (sum(sum(1 for v in raw if v) for c in "MID" for (lo, hi, raw) in dump[c].values()) if census else 0), (i, meta)
        for (s, k), (wv, mask) in exp.items():
            if s > r.score:
                continue
            have = int(words[_arena_slot(f, s // g, k, len(words))])
            assert (have & mask) == (wv & mask), (f"pair {i} score {s} diagonal {k}: arena {have:#x} = off {have >> 4} bits {have & 15:04b}, "
                                                 f"expected off {wv >> 4} bits {wv & 15:04b} (mask {mask & 15:04b})")
            checked += 1
        pairs += 1
    # (the others were handed on: band or arena.  The lane-per-pair kernel hands on rows wider than 30 diagonals, and its
    # short pairs have fewer cells)
    assert pairs >= (min_pairs if min_pairs is not None else (n // 3 if lane else n // 2)) and checked > (20 if lane else 50) * pairs, (pairs, checked)
    al.close()


def test_penalties_the_reference_cannot_align_are_refused(built):
    """Mismatch == 0: the reference's own loop does not terminate when the first bases differ;
    GapOpen + GapExt == 0: M[s-o-e] is the row being written.  Both are refused with an error code, not aligned."""
    import wfa_amd as w
    from wfa_amd import _lib as L
    for pen in ((0, 6, 2), (4, 0, 0)):
        al = _aligner(True, None, pen)
        with pytest.raises(L.WfaHipError) as ei:
            al.Align(b"ACGT", b"CCGT")
        assert ei.value.code == L.ERR_UNSUPPORTED
        al.close()


@pytest.mark.parametrize("opts", [{}, {"prepack": 1}, {"overlap": 1}, {"narrow_long": 1}, {"duo": 2}, {"duo": 0}])
def test_several_chunks_and_optional_paths(built, opts):
    """A pass cut into several chunks (each chunk's arenas are reused by the next: the default for batches that do not fit
    35 % of HBM, forced here with chunk_pairs), and the optional paths kept behind options: the pre-packing kernel, the
    double-buffered chunks, the 8-lanes-per-pair first pass for long reads."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=61, n_pairs=5000, length=600, error_rate=0.05, n_threads=8)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), *data, n_threads=8)
    al = _aligner(True, (10, 50, 1))
    al.set_option("chunk_pairs", 1200)
    for k, v in opts.items():
        al.set_option(k, v)
    for rep in range(2):
        assert_batch_equal(al.align_arrays(*data), want, f"chunks opts={opts} rep={rep}")
    assert al.last_timing().n_main_launches == 5
    al.close()


@pytest.mark.parametrize("length,err,n,ad", [(1000, 0.05, 60000, (10, 50, 1)), (1000, 0.05, 777, (10, 50, 1)), (700, 0.08, 20000, (10, 50, 1)),
                                             (1000, 0.03, 20000, None), (400, 0.10, 20000, (10, 50, 1)), (1500, 0.04, 9000, (20, 100, 1)),
                                             (1000, 0.05, 3, (10, 50, 1)), (150, 0.02, 100000, None), (100, 0.06, 30000, (10, 50, 1)),
                                             (230, 0.03, 7, None), (60, 0.1, 50000, None)])
def test_duo_kernel_batches(built, length, err, n, ad):
    """wfa_duo_kernel (a pair runs on 8 or 16 lanes and changes between the two; pairs are parked in LDS and resumed;
    every wave prefetches its next pair) on batches large enough that every wave widens, narrows, parks and resumes many
    times, plus tiny ones (fewer pairs than halves of one wave).  Twice through the same aligner over a poisoned arena
    (the kernel zeroes nothing), every field and every CIGAR op against the oracle."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=900 + length + n % 97, n_pairs=n, length=length, error_rate=err, n_threads=8)
    want = O.align_batch(_oracle_params(True, ad), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    al = _aligner(True, ad)
    al.set_option("duo", 2)
    al.set_option("duo_short", 2)  # (short reads: eight pairs per fetch)
    al.set_option("arena_poison", 1)
    for rep in range(2):
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == 8
        assert_batch_equal(got, want, f"duo L={length} err={err} n={n} ad={ad} rep={rep}")
    al.close()


def test_duo_kernel_ragged_lengths(built):
    """Unequal lengths (windows that start off-centre and follow the band), overhangs that end at sequence ends (the exact
    WF_NEXT), bands that outgrow a whole row (handed on), 15 % error pairs between easy ones, a small arena."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(4242)
    qs, ts = [], []
    for i in range(6000):
        L = int(rng.integers(250, 1400))
        q = rng.integers(0, 4, L)
        t = list(q)
        for _ in range(int(L * (0.15 if i % 9 == 0 else 0.05))):
            kind_e, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(1, len(t))))
            if kind_e == 0 and t:
                t[pos] = int(rng.integers(0, 4))
            elif kind_e == 1:
                t.insert(pos, int(rng.integers(0, 4)))
            elif t:
                del t[pos]
        if i % 7 == 0:
            extra = list(rng.integers(0, 4, int(rng.integers(1, 120))))
            t = (extra + t) if i % 2 else (t + extra)
        if i % 11 == 0:
            q = q[int(rng.integers(0, 100)):]
        qs.append(bytes(b"ACGT"[c] for c in q))
        ts.append(bytes(b"ACGT"[c] for c in t))
    data = w.make_blob(qs, ts)
    for ad in ((10, 50, 1), None):
        want = O.align_batch(_oracle_params(True, ad), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
        for small in (False, True):
            al = _aligner(True, ad)
            al.set_option("duo", 2)
            if small:
                al.set_option("packed_arena_bytes", 24 * 1024)
            got = al.align_arrays(*data)
            assert al.last_timing().main_kernel_kind == 8
            assert_batch_equal(got, want, f"duo ragged ad={ad} small={small}")
            al.close()


@pytest.mark.timeout(900)
def test_learned_start_level_is_only_a_hint(built):
    """The arena level a context learns for a class of long pairs (option learn, on by default) may speed a call up but
    never changes its result: after a hard batch an easy one of the same class still aligns and -- every
    sixteenth call -- probes a lower start level, so the class is not pinned to large slots for ever; and a batch of the
    same class whose pairs are twice as long (slots scale with the length, the class buckets lengths by powers of two)
    steps the start level down instead of reporting "no memory".  The device is made to look small (mem_limit) so that the
    ladder reaches its one-slot-per-level region with 9-16 kbp pairs."""
    import wfa_amd as w
    from oracle import oracle as O
    hard = w.generate_pairs(seed=71, n_pairs=4, length=9000, error_rate=0.12)
    easy = w.generate_pairs(seed=72, n_pairs=4, length=8400, error_rate=0.01)
    longer = w.generate_pairs(seed=73, n_pairs=3, length=15500, error_rate=0.10)
    op = _oracle_params(False, (10, 50, 1))
    want = {k: O.align_batch(op, *d, n_threads=8) for k, d in (("hard", hard), ("easy", easy), ("longer", longer))}
    al = _aligner(False, (10, 50, 1))
    al.set_option("mem_limit", 3 << 30)
    assert_batch_equal(al.align_arrays(*hard), want["hard"], "hard, first call")
    assert al.last_timing().ladder_start_level == 0  # nothing learned yet
    assert_batch_equal(al.align_arrays(*hard), want["hard"], "hard, second call")
    learned = al.last_timing().ladder_start_level
    starts = []
    for i in range(18):
        assert_batch_equal(al.align_arrays(*easy), want["easy"], f"easy call {i} after the hard one")
        starts.append(al.last_timing().ladder_start_level)
    if learned > 0:
        assert min(starts) < learned and starts[-1] < learned, (learned, starts)  # the hint decays
    assert_batch_equal(al.align_arrays(*hard), want["hard"], "hard again")
    assert_batch_equal(al.align_arrays(*hard), want["hard"], "hard again (start level learned)")
    got = al.align_arrays(*longer)
    assert (got.status == 0).all(), got.status  # never "no memory" straight from a learned start level
    assert_batch_equal(got, want["longer"], "twice as long, same class")
    al.close()


@pytest.mark.parametrize("length,err,ad,n", [(1000, 0.20, (10, 50, 1), 3000), (1000, 0.10, (10, 50, 1), 6000), (600, 0.15, None, 3000),
                                             (1200, 0.25, (10, 50, 1), 1500)])
def test_mid_window_rung_and_learned_start(built, length, err, ad, n):
    """Bands of 60-250 diagonals (1 kbp at 10-25 % error): the 64-diagonal first pass hands most pairs on, the 128-diagonal
    instance (wfa_blk_kernel<32,1>, two pairs per wave) takes what fits, the 256-diagonal one the rest; pairs that run
    out of arena rows (scores above half the read length) are re-run with four times the rows.  From its second call
    on a context starts such a class on the window that took the pairs, with the rows they needed -- the results are the
    same on every call, every field and every CIGAR op against the oracle."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=int(err * 100) + length, n_pairs=n, length=length, error_rate=err, n_threads=8)
    want = O.align_batch(_oracle_params(True, ad), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    al = _aligner(True, ad)
    kinds = []
    for rep in range(4):
        assert_batch_equal(al.align_arrays(*data), want, f"L={length} err={err} ad={ad} call {rep}")
        kinds.append(al.last_timing().main_kernel_kind)
    al.set_option("blk_mid", 0)  # without the 128-diagonal rung
    assert_batch_equal(al.align_arrays(*data), want, f"L={length} err={err} ad={ad} blk_mid=0")
    al.close()
    if ad is not None and err >= 0.2:
        assert kinds[0] == 3 and kinds[-1] in (9, 5), kinds  # the class moved to a wider window


def test_mid_window_arena_word_for_word(built):
    """The 128-diagonal instance's stored words (plain rows of 128, CompactView fmt 6) against the oracle's wavefronts."""
    import wfa_amd as w
    from oracle import oracle as O
    n, pen, ad = 96, (4, 6, 2), (10, 50, 1)
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=2020, n_pairs=n, length=800, error_rate=0.2)
    al = _aligner(True, ad, pen)
    al.align_arrays(blob, q_off, q_len, t_off, t_len)  # first call: the class learns its window (and its rows)
    al.align_arrays(blob, q_off, q_len, t_off, t_len)
    got = al.align_arrays(blob, q_off, q_len, t_off, t_len)
    assert al.last_timing().main_kernel_kind == 9
    oa = O.Aligner(O.make_params(*pen, global_alignment=True, adaptive=ad))
    checked = pairs = 0
    for i in range(n):
        words, f, meta = al.debug_compact_arena(i)
        assert f == 6
        if meta[0] != 0:
            continue
        q = bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])])
        t = bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])])
        r = oa.align(q, t)
        assert meta[1] == r.score == int(got.score[i])
        for (s, k), (wv, mask) in _expected_compact_words(oa.dump(), *pen).items():
            if s > r.score:
                continue
            have = int(words[_arena_slot(f, s // 2, k)])
            assert (have & mask) == (wv & mask), (i, s, k, hex(have), hex(wv))
            checked += 1
        pairs += 1
    assert pairs >= n // 2 and checked > 50 * pairs, (pairs, checked)
    al.close()


@pytest.mark.parametrize("length,err,n", [(150, 0.02, 40000), (150, 0.02, 7), (100, 0.06, 30000), (230, 0.03, 5000), (60, 0.1, 20000),
                                          (150, 0.15, 8000), (30, 0.05, 70), (150, 0.02, 65)])
@pytest.mark.parametrize("ad", [(10, 50, 1), None])
def test_lane_kernel_batches(built, length, err, n, ad):
    """wfa_lane_kernel (short reads, a lane per pair, 64 pairs per wave and generation): full generations, a single
    partial one, error rates at which most pairs outgrow the 28-diagonal rows and are handed on.  Twice through the
    same aligner over a poisoned arena (the kernel zeroes nothing but its LDS rings), every field and CIGAR op."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=77 + length + n % 91, n_pairs=n, length=length, error_rate=err, n_threads=8)
    want = O.align_batch(_oracle_params(True, ad), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    al = _aligner(True, ad)
    al.set_option("lane", 2)
    al.set_option("arena_poison", 1)
    for rep in range(2):
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == 10
        assert_batch_equal(got, want, f"lane L={length} err={err} n={n} ad={ad} rep={rep}")
    al.close()


def test_lane_kernel_ragged_lengths_and_penalties(built):
    """Ragged lengths 1..240, overhangs (the exact WF_NEXT at sequence ends), empty / lowercase / non-ACGT entries inside
    generations, other penalty sets of the 2 : 4 : 1 shape, three wf-adaptive settings; and the default routing: a
    batch of at least lane_min_pairs short pairs starts on the kernel, a smaller one on the 8-lane instance."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    qs, ts = [], []
    for i in range(5000):
        L = int(rng.integers(1, 240))
        q = bytes(b"ACGT"[c] for c in rng.integers(0, 4, L))
        t = bytearray(q)
        for _ in range(int(rng.integers(0, 1 + L // 10))):
            pos, kind = int(rng.integers(0, len(t))), int(rng.integers(0, 3))
            if kind == 0:
                t[pos] = b"ACGT"[int(rng.integers(0, 4))]
            elif kind == 1:
                t.insert(pos, b"ACGT"[int(rng.integers(0, 4))])
            elif len(t) > 1:
                del t[pos]
        if i % 7 == 0:
            extra = bytes(b"ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(1, 60))))
            t = (extra + bytes(t)) if i % 2 else (bytes(t) + extra)
        t = bytes(t[:240])
        if i % 17 == 5:
            q = b""
        elif i % 17 == 9:
            t = t.lower()
        elif i % 17 == 13:
            q = (q[:len(q) // 2] + b"N" + q[len(q) // 2:])[:240]
        elif i % 23 == 7:
            q, t = b"A", b"CA"
        qs.append(q), ts.append(t)
    data = w.make_blob(qs, ts)
    for ad in ((10, 50, 1), None, (5, 10, 1)):
        for pen in ((4, 6, 2), (2, 3, 1), (6, 9, 3)):
            want = O.align_batch(_oracle_params(True, ad, pen), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
            al = _aligner(True, ad, pen)
            al.set_option("lane", 2)
            al.set_option("arena_poison", 1)
            got = al.align_arrays(*data)
            t = al.last_timing()
            assert t.main_kernel_kind == 10 and t.n_retried_pairs > 0  # (bands wider than a row, non-ACGT bytes: handed on)
            assert_batch_equal(got, want, f"lane ragged ad={ad} pen={pen}")
            al.close()
    if os.environ.get("WFA_TEST_OPTS"):
        return  # (forced options: the default routing is not what runs)
    for n, kind in ((40000, 10), (20000, 6)):
        data = w.generate_pairs(seed=3, n_pairs=n, length=150, error_rate=0.02, n_threads=8)
        al = _aligner(True, None)
        al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == kind, (n, al.last_timing().main_kernel_kind)
        al.close()


@pytest.mark.parametrize("seed", list(range(300, 312)))
def test_fuzz_short_reads(built, seed):
    """Seeded random short-read batches through wfa_lane_kernel and what it hands on to: penalties (2:4:1 shapes, which the
    kernel takes, and others, which it must leave alone), wf-adaptive parameters down to a minimum length of 1, lengths
    1-240 with up to 20 % edits, its packing paths (own, packing kernel; aligned and unaligned blobs), arenas too small
    for the pair (rows run out: handed on), small calls with and without the detached backtrace, poisoned arenas."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    pens = [(4, 6, 2), (2, 3, 1), (8, 12, 4), (6, 9, 3), (1, 1, 1), (3, 5, 2)]
    for _ in range(5):
        pen = pens[int(rng.integers(0, len(pens)))]
        ad = None if rng.integers(0, 3) == 0 else (int(rng.integers(1, 24)), int(rng.integers(1, 120)), 1)
        n = int(rng.integers(64, 3000))
        max_l = int(rng.choice([60, 150, 240]))
        parts, q_off, q_len, t_off, t_len = [], [], [], [], []
        pos = 0
        unaligned = bool(rng.integers(0, 3) == 0)
        for i in range(n):
            L = int(rng.integers(1, max_l + 1))
            q = rng.integers(0, 4, L)
            t = list(q)
            for _e in range(int(L * rng.uniform(0, 0.2))):
                kind, p = int(rng.integers(0, 3)), int(rng.integers(0, max(1, len(t))))
                if kind == 0 and t:
                    t[p] = int(rng.integers(0, 4))
                elif kind == 1:
                    t.insert(p, int(rng.integers(0, 4)))
                elif len(t) > 1:
                    del t[p]
            qb, tb = bytes(b"ACGT"[c] for c in q), (bytes(b"ACGT"[c] for c in t) or b"G")[:240]
            for s_, offs, lens in ((qb, q_off, q_len), (tb, t_off, t_len)):
                pad = int(rng.integers(0, 5)) if unaligned else (-pos) % 16
                parts.append(b"#" * pad)
                pos += pad
                offs.append(pos), lens.append(len(s_)), parts.append(s_)
                pos += len(s_)
        parts.append(b"#" * 32)
        data = (np.frombuffer(b"".join(parts), dtype=np.uint8), np.array(q_off, np.uint64), np.array(q_len, np.uint32),
                np.array(t_off, np.uint64), np.array(t_len, np.uint32))
        opts = {"lane": 2, "lane_pack": int(rng.integers(0, 2)), "compact_call_bases": int(rng.choice([0, 50000000])),
                "arena_poison": 1}
        if rng.integers(0, 4) == 0:
            opts["packed_arena_bytes"] = int(rng.choice([1024, 2048, 4096]))
        al = _aligner(True, ad, pen)
        for k, v in opts.items():
            al.set_option(k, v)
        want = O.align_batch(_oracle_params(True, ad, pen), *data, n_threads=8)
        for rep in range(2):
            got = al.align_arrays(*data)
            assert_batch_equal(got, want, f"fuzz short seed={seed} pen={pen} ad={ad} n={n} max_l={max_l} unaligned={unaligned} opts={opts} rep={rep}")
        # (the penalty shapes the register-ring kernels are instantiated for, wfa_amd/csrc/wfa_fwd.hpp: e/g == 1 and
        # x/g : (o+e)/g one of 2:4, 1:3, 1:2, 2:3, 2:2, 3:3 -- here 4/6/2, 2/3/1, 8/12/4, 6/9/3 and, since round 5, 1/1/1; not 3/5/2)
        g = int(np.gcd.reduce([pen[0], pen[1] + pen[2], pen[2]]))
        shaped = pen[2] == g and (pen[0] // g, (pen[1] + pen[2]) // g) in ((2, 4), (1, 3), (1, 2), (2, 3), (2, 2), (3, 3))
        assert (al.last_timing().main_kernel_kind == 10) == shaped, (pen, al.last_timing().main_kernel_kind)
        al.close()


@pytest.mark.parametrize("chunk_pairs", [1000, 4097])
def test_lane_kernel_several_chunks(built, chunk_pairs):
    """wfa_lane_kernel over a pass of several chunks (two arena buffers in turn, the backtrace kernel of a chunk beside the
    forward kernel of the next): every chunk's waves start from their own index again, pairs are addressed from the
    chunk's first."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=41, n_pairs=9000, length=150, error_rate=0.03, n_threads=8)
    for ad in ((10, 50, 1), None):
        want = O.align_batch(_oracle_params(True, ad), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
        al = _aligner(True, ad)
        al.set_option("lane", 2)
        al.set_option("chunk_pairs", chunk_pairs)
        al.set_option("arena_poison", 1)
        for rep in range(2):
            got = al.align_arrays(*data)
            t = al.last_timing()
            assert t.main_kernel_kind == 10 and t.n_main_launches == -(-9000 // chunk_pairs), t
            assert_batch_equal(got, want, f"lane chunks={chunk_pairs} ad={ad} rep={rep}")
        al.close()


# ------------------------------------------------------------------------------------------------------------------
# Round 4: long reads on the sub-wave kernels (wfa_blk_kernel<.., LONG>: sliding 2-bit sequence windows in LDS).
# The rules being windowed are WF_EXTEND's (wfa.go:381-458): whatever the window does, a cell gets its full LCP.
def _mutate(rng, s, rate):
    """s with ~rate of its bases edited (mismatch / insertion / deletion, equally likely)."""
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < rate / 3:
            out.append(rng.choice(b"ACGT".replace(bytes([c]), b"")))
        elif r < 2 * rate / 3:
            out.append(c)
            out.append(rng.choice(b"ACGT"))
        elif r < rate:
            continue
        else:
            out.append(c)
    return bytes(out)


def _long_window_cases(seed):
    import random
    rng = random.Random(seed)
    rnd = lambda n: bytes(rng.choice(b"ACGT") for _ in range(n))
    qs, ts = [], []
    for ln, rate in ((4100, 0.05), (6000, 0.02), (9000, 0.08), (12000, 0.05), (20000, 0.03), (30000, 0.06), (17000, 0.10), (8000, 0.15)):
        q = rnd(ln)
        qs.append(q), ts.append(_mutate(rng, q, rate))
    q = rnd(21000)
    qs.append(q), ts.append(q)                                  # identical: one match run through every window to the END of both sequences
    qs.append(q), ts.append(q[:20000])                          # ... the run ends at the end of the target, the query hangs over
    qs.append(q[:20500]), ts.append(q)                          # ... and the other way round
    qs.append(q), ts.append(q[:9000] + b"T" + q[9001:])         # a run of 9 000, a mismatch in the middle of a window, a run of 12 000
    qs.append(q), ts.append(q[:8192] + _mutate(rng, q[8192:], 0.05))   # a run that ends exactly on a window's 64-base grid
    qs.append(q[:15000] + rnd(40) + q[15000:]), ts.append(q)    # a 40-base insertion far from the start (the band moves 40 diagonals at once)
    qs.append(q), ts.append(q[:5000] + q[5300:])                # a 300-base deletion: band failure of the 64-diagonal window -> 128 -> 256 diagonals
    qs.append(rnd(7000)), ts.append(rnd(7000))                  # unrelated sequences (arena rows run out; the band grows)
    qs.append(q[:4001] + b"N" + q[4002:6000]), ts.append(q[:6000])   # a byte outside ACGT: the byte path takes the pair
    qs.append(q[:5000].lower()), ts.append(q[:5000])            # lowercase against uppercase (wfa.go:408-454 compares raw bytes)
    qs.append(rnd(300)), ts.append(rnd(5000))                   # a short read against a long one
    qs.append(q[:4200]), ts.append(b"")                         # empty (ErrEmptySeq)
    return qs, ts


@pytest.mark.parametrize("adaptive", [(10, 50, 1), None, (4, 10, 1)])
@pytest.mark.parametrize("window_words,first", [(256, 11), (64, 11), (64, 12), (256, 13), (64, 14), (64, 15), (256, 0)])
def test_long_window_kernel_cases(built, adaptive, window_words, first):
    """Pairs of 4-30 kbp through the sliding-window instances: every field and every CIGAR op against the oracle.  The
    cases put match runs across window refills, a run up to the very end of both sequences, refills at a sequence end
    with an overhang, band failures that climb the 128- and 256-diagonal rungs, and pairs the path must hand on.  With 64-word
    windows (1 024 bases, 480 of them usable) every pair repositions its windows dozens of times."""
    from oracle import oracle as O
    import wfa_amd as w
    qs, ts = _long_window_cases(4)
    data = w.make_blob(qs, ts)
    al = _aligner(True, adaptive)
    al.set_option("long_window_words", window_words)
    al.set_option("long_first", first)  # (0: by batch size -- a batch this small starts with a wave per pair, 128 diagonals)
    got = al.align_arrays(*data)
    assert al.last_timing().main_kernel_kind == (first or 15)  # the sliding-window instance asked for was the first pass
    want = O.align_batch(_oracle_params(True, adaptive), *data, n_threads=8)
    assert_batch_equal(got, want, f"long windows {adaptive} {window_words}")
    again = al.align_arrays(*data)
    assert_batch_equal(again, want, "second call")
    al.close()


def test_long_window_kernel_batch(built):
    """A batch that fills waves and queues (600 pairs of 5-12 kbp at 2-8 % error, ragged lengths, a few unaligned blob
    offsets): every wave refills, repositions and finishes pairs at different steps.  Twice over a poisoned arena."""
    import random
    from oracle import oracle as O
    import wfa_amd as w
    rng = random.Random(11)
    qs, ts = [], []
    for i in range(600):
        ln = rng.randint(5000, 12000)
        q = bytes(rng.choice(b"ACGT") for _ in range(ln))
        qs.append(q), ts.append(_mutate(rng, q, rng.choice((0.02, 0.05, 0.08))))
    data = w.make_blob(qs, ts)
    al = _aligner(True, (10, 50, 1))
    al.set_option("arena_poison", 1)
    al.set_option("long_first", 11)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), *data, n_threads=8)
    for rep in range(2):
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == 11
        assert_batch_equal(got, want, f"long batch, pass {rep}")
    al.set_option("long_wave_bt", 0)  # ... and with the lane-per-pair backtrace kernel behind the same forward pass
    assert_batch_equal(al.align_arrays(*data), want, "lane-per-pair backtrace")
    al.set_option("long_wave_bt", 2)  # ... and the wave-per-pair one whatever the batch size
    assert_batch_equal(al.align_arrays(*data), want, "wave-per-pair backtrace")
    al.set_option("long_wave_bt", 1)
    al.set_option("long_mid_lone", 0)  # ... and the leftovers of the 64-diagonal pass on two pairs per wave instead of a wave per pair
    assert_batch_equal(al.align_arrays(*data), want, "leftovers on two pairs per wave")
    al.set_option("long_mid_lone", 1)
    # the same pairs on the plain whole-sequence path (option long = 0): the two paths agree with each other too
    al.set_option("long", 0)
    plain = al.align_arrays(*data)
    assert al.last_timing().main_kernel_kind != 11
    assert_batch_equal(plain, want, "long = 0")
    al.close()


@pytest.mark.parametrize("opts", [{}, {"mem_limit": 4 << 30}, {"team_order": 0}])
def test_team_kernel_scout_pass(built, opts):
    """The scout pass of wfa_teamc_kernel (round 5): the batch first runs with ONE workgroup per pair, which finishes the pairs
    whose band collapses under wf-adaptive and hands the others on (ST_REDO_WIDE) to teams at the same arena level -- forced here
    on ten 20 kbp semi-global pairs (option team_scout; off by default: on configs[4] most pairs are wide and it gains nothing).
    Every field and op against the oracle, twice over a poisoned pool; the launches count as the one kernel."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=77, n_pairs=10, length=20000, error_rate=0.10)
    want = O.align_batch(_oracle_params(False, (10, 50, 1)), *data, n_threads=8)
    al = _aligner(False, (10, 50, 1))
    al.set_option("arena_poison", 1)
    al.set_option("team_scout", 2)
    for k, v in opts.items():
        al.set_option(k, v)
    for rep in range(2):
        got = al.align_arrays(*data)
        tm = al.last_timing()
        assert tm.main_kernel_kind == 17 and tm.n_launches >= 2 and tm.n_main_launches >= 2, tm
        assert_batch_equal(got, want, f"scout pass {opts} pass {rep}")
    al.set_option("team_scout", 0)
    got = al.align_arrays(*data)
    assert_batch_equal(got, want, "without the scout pass")
    al.close()


@pytest.mark.parametrize("opts", [{}, {"mem_limit": 4 << 30}, {"team_xcd": 2}, {"team_xcd": 1}, {"team_paged": 0}])
@pytest.mark.parametrize("compact", [1, 0])
def test_team_kernel_paged_arena(built, opts, compact):
    """The team kernel with its arena as ONE pool of pages shared by up to eight teams (round 4): ten 20 kbp semi-global
    pairs (wide seeded wavefronts, then wave mode) through more teams than the four slots of old, over a poisoned pool;
    on a device made to look small (mem_limit: the pool runs dry, pairs are handed on and re-run with fewer teams); with the
    XCD-local protocol; and with one slot per team as before.  Every field and op against the oracle, twice."""
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=77, n_pairs=10, length=20000, error_rate=0.10)
    want = O.align_batch(_oracle_params(False, (10, 50, 1)), *data, n_threads=8)
    al = _aligner(False, (10, 50, 1))
    al.set_option("arena_poison", 1)
    al.set_option("team_compact", compact)  # (round 5: wfa_teamc_kernel shares the pool and its page protocol)
    for k, v in opts.items():
        al.set_option(k, v)
    for rep in range(2):
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == (17 if compact else 7)  # wfa_teamc_kernel / wfa_team_kernel
        assert_batch_equal(got, want, f"paged arena {opts} pass {rep}")
    al.close()


def _semi_global_cases(rng, n_pairs, lo, hi, flank=None):
    """Semi-global shapes the n + m - 1 seeds care about: a read inside a longer target (any offset), a target inside a
    longer query, equal lengths, overhangs at either end, plus plain edited copies -- lengths in [lo, hi].
    flank: the longest run of extra bases on a side (None: up to hi / 2, and unrelated pairs too).  Under wf-adaptive the
    batches keep |m - n| under MaxDistDiff: with a larger difference the reference's first reduce cuts the final diagonal
    off (wfa.go:235-239 only ever looks at M[s][m - n]) and the alignment runs on for thousands of scores until a gap chain
    finds its way back -- the reference's behaviour, minutes per pair in the oracle, and not what these tests are about."""
    qs, ts = [], []
    for i in range(n_pairs):
        L = int(rng.integers(lo, hi + 1))
        core = rng.integers(0, 4, L)
        edited = list(core)
        for _ in range(int(L * (0.02 + 0.1 * rng.random()))):
            kind_e, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(1, len(edited))))
            if kind_e == 0 and edited:
                edited[pos] = int(rng.integers(0, 4))
            elif kind_e == 1:
                edited.insert(pos, int(rng.integers(0, 4)))
            elif edited:
                del edited[pos]
        a, b = list(core), edited
        shape = i % 6
        fl = hi // 2 if flank is None else flank
        if shape == 0:    # read inside a longer target
            b = list(rng.integers(0, 4, int(rng.integers(1, fl + 1)))) + b + list(rng.integers(0, 4, int(rng.integers(0, fl + 1))))
        elif shape == 1:  # target inside a longer query
            a = list(rng.integers(0, 4, int(rng.integers(1, fl + 1)))) + a + list(rng.integers(0, 4, int(rng.integers(0, fl + 1))))
        elif shape == 2:  # overhangs: suffix of one against prefix of the other
            cut = int(rng.integers(1, max(2, min(L // 2, fl))))
            a, b = a[cut:], b[:max(1, len(b) - cut)]
        elif shape == 3 and flank is None:  # unrelated sequences
            b = list(rng.integers(0, 4, int(rng.integers(lo, hi + 1))))
        a, b = a[:hi], b[:hi]
        if not a:
            a = [0]
        if not b:
            b = [1]
        qs.append(bytes(b"ACGT"[c] for c in a))
        ts.append(bytes(b"ACGT"[c] for c in b))
    return qs, ts


@pytest.mark.parametrize("pen", [(4, 6, 2), (2, 4, 2), (1, 1, 1), (4, 4, 2), (6, 6, 3)])
@pytest.mark.parametrize("ad", [(10, 50, 1), None, (4, 8, 1)])
def test_wide_kernel_semi_global_shapes(built, pen, ad):
    """wfa_wide_kernel (round 6: a wave per pair, the rows in 16-bit LDS rings of any width, the semi-global end cell found in
    flight, one 16-bit backtrace word per diagonal) against the oracle on every field and every CIGAR op: reads inside longer
    targets and the other way round, overhangs, unrelated sequences, lengths from 1 to 700 (tiles that start and end
    anywhere, rows narrower than a tile), five penalty shapes, wf-adaptive on / off / aggressive; twice over a poisoned arena."""
    import wfa_amd as w
    from oracle import oracle as O
    rng = np.random.default_rng(77 + pen[0] * 10 + pen[1])
    qs, ts = _semi_global_cases(rng, 600, 1, 700, flank=None if ad is None else 12)
    qs[:6] = [b"A", b"ACGT", b"A", b"ACGTACGTAC", b"C", b"GATTACA"]
    ts[:6] = [b"A", b"A", b"ACGT", b"ACGTTCGTAC", b"G", b"TTGATTACATT"]
    data = w.make_blob(qs, ts)
    want = O.align_batch(_oracle_params(False, ad, pen), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    al = _aligner(False, ad, pen)
    al.set_option("arena_poison", 1)
    for rep in range(2):
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == 18
        assert_batch_equal(got, want, f"wide kernel pen={pen} ad={ad} rep={rep}")
    al.set_option("wide_waves", 1)  # (rings of this size get four waves per pair by default)
    assert_batch_equal(al.align_arrays(*data), want, f"wide kernel, a wave per pair, pen={pen} ad={ad}")
    al.set_option("wide_waves", 0)
    al.set_option("wide_exact", 1)  # (the interior of a wide row is computed two diagonals per register by default: here every round takes the exact per-cell path)
    assert_batch_equal(al.align_arrays(*data), want, f"wide kernel, exact path only, pen={pen} ad={ad}")
    al.set_option("wide_exact", 0)
    if ad is not None:  # one launch per chunk: every pair runs to its end in the wide rings (no hand-over to the narrow phase)
        al.set_option("wide", 3)
        assert_batch_equal(al.align_arrays(*data), want, f"wide kernel, one phase, pen={pen} ad={ad}")
    # the same batch on the ladder's kernels: the same records
    al.set_option("wide", 0)
    assert_batch_equal(al.align_arrays(*data), want, f"ladder pen={pen} ad={ad}")
    al.close()


@pytest.mark.parametrize("ad", [(10, 50, 1), None])
def test_wide_kernel_semi_global_1kbp(built, ad):
    """1e5 x 1 kbp @5 % pairs, semi-global, wf-adaptive on and off: every record and every CIGAR op
    against the oracle on the host's cores; with a small arena, the pairs that overflow it finish on the ladder."""
    import wfa_amd as w
    from oracle import oracle as O
    n = 100_000 if ad is not None else 20_000  # (adaptive off: 3e5 cells per pair in the oracle too)
    data = w.generate_pairs(seed=63, n_pairs=n, length=1000, error_rate=0.05, n_threads=32)
    want = O.align_batch(_oracle_params(False, ad), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    al = _aligner(False, ad)
    got = al.align_arrays(*data)
    assert al.last_timing().main_kernel_kind == 18
    assert_batch_equal(got, want, f"1 kbp semi-global ad={ad}")
    if ad is not None:
        al.set_option("packed_arena_bytes", 72 * 1024)  # (some pairs narrow late: their rows do not fit, the ladder finishes them)
        got = al.align_arrays(*data)
        assert al.last_timing().n_retried_pairs > 0
        assert_batch_equal(got, want, "1 kbp semi-global, small arena")
    al.close()


def test_wide_kernel_is_the_default_for_short_semi_global_reads(built):
    """Batches of semi-global reads of at most 2 047 bases start on wfa_wide_kernel without any option; longer ones on the ladder."""
    import wfa_amd as w
    from oracle import oracle as O
    for length, n_pairs, kind in ((300, 4000, 18), (1000, 2000, 18), (1900, 300, 18), (2300, 200, 0)):  # (1 900: rings of 48 KB, offsets near the 16-bit words' 2 047)
        data = w.generate_pairs(seed=64, n_pairs=n_pairs, length=length, error_rate=0.06, n_threads=8)
        want = O.align_batch(_oracle_params(False, (10, 50, 1)), *data, n_threads=8)
        al = _aligner(False, (10, 50, 1))
        got = al.align_arrays(*data)
        assert al.last_timing().main_kernel_kind == kind
        assert_batch_equal(got, want, f"semi-global {length} bp, default routing")
        al.close()
