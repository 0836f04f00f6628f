"""GPU: the round-2 entry points of the C-ABI -- pre-packed 2-bit input, per-pair submit / collect, the context set
over several GPUs -- each against the plain batch entry and the oracle."""
import os

import numpy as np
import pytest

from test_parity_gpu import _aligner, _oracle_params, assert_batch_equal

pytestmark = pytest.mark.gpu


def test_packed_entry_matches_byte_entry_and_oracle(built):
    import wfa_amd as w
    from oracle import oracle as O
    data = w.generate_pairs(seed=17, n_pairs=6000, length=300, error_rate=0.06, n_threads=8)
    # ragged: cut some sequences short so that word boundaries and pad words fall everywhere
    blob, q_off, q_len, t_off, t_len = data
    rng = np.random.default_rng(3)
    q_len = np.maximum(1, q_len - rng.integers(0, 200, len(q_len)).astype(np.uint32)).astype(np.uint32)
    t_len = np.maximum(1, t_len - rng.integers(0, 200, len(t_len)).astype(np.uint32)).astype(np.uint32)
    data = (blob, q_off, q_len, t_off, t_len)
    packed = w.pack_pairs(*data)
    for glob, ad in ((True, (10, 50, 1)), (True, None), (False, (10, 50, 1))):
        al = _aligner(glob, ad)
        got = al.align_arrays_packed(packed[0], packed[1], q_len, packed[2], t_len)
        want = O.align_batch(_oracle_params(glob, ad), *data, n_threads=8)
        assert_batch_equal(got, want, f"packed entry glob={glob} ad={ad}")
        assert_batch_equal(al.align_arrays(*data), want, "byte entry")
        al.close()


def test_packed_entry_sliced_upload(built):
    """A batch large enough for the sliced pipeline (upload of slice k+1 and its unpack kernel beside the alignment of
    slice k): same results as the byte entry, pair by pair."""
    import wfa_amd as w
    n = 300_000
    data = w.generate_pairs(seed=23, n_pairs=n, length=1000, error_rate=0.05, n_threads=32)
    packed, q_woff, t_woff = w.pack_pairs(*data, n_threads=32)
    assert packed.size * 16 >= (256 << 20)
    al = _aligner(True, (10, 50, 1))
    a = al.align_arrays(*data)
    b = al.align_arrays_packed(packed, q_woff, data[2], t_woff, data[4])
    assert_batch_equal(b, a, "packed (sliced) vs bytes")
    al.close()


def test_submit_collect_is_the_batch(built):
    import wfa_amd as w
    data = w.generate_pairs(seed=29, n_pairs=500, length=200, error_rate=0.08, n_threads=4)
    blob, q_off, q_len, t_off, t_len = data
    qs = [bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]) for i in range(500)]
    ts = [bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]) for i in range(500)]
    qs[7], ts[9] = b"", b""          # ErrEmptySeq keeps its ticket
    qs[11] = b"ACGTNNNNACGT"         # byte path inside the batch
    al = _aligner(True, (10, 50, 1))
    want, werr = al.AlignBatch(qs, ts)
    assert al.Pending() == 0
    for i in range(500):
        assert al.Submit(qs[i], ts[i]) == i
    assert al.Pending() == 500
    got, gerr = al.Collect()
    assert al.Pending() == 0
    assert len(got) == 500
    for i in range(500):
        assert (gerr[i] is werr[i])
        if werr[i] is None:
            assert got[i].key() == want[i].key(), i
    assert gerr[7] is w.ErrEmptySeq and gerr[9] is w.ErrEmptySeq
    # tickets restart after a collect; an empty collect is fine
    assert al.Submit(b"ACCATACTCG", b"AGGATGCTCG") == 0
    r, e = al.Collect()
    assert e == [None] and r[0].CIGAR(False) == "1M2X2M1X4M"
    assert al.Collect() == ([], [])
    al.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0] * 8, None])  # ([0] * 8: the eight contexts of an 8-GPU node, here on one GPU)
def test_context_set_shards_and_merges_in_pair_order(built, devices):
    """wfahip_create_multi / wfahip_align_batch_multi: contiguous shards balanced by sequence bytes, one host thread
    and one context per shard (here several contexts on the one GPU of the box; None = every device), merged in pair
    order with the op offsets re-based."""
    import wfa_amd as w
    from oracle import oracle as O
    parts = [w.generate_pairs(seed=31, n_pairs=4000, length=100, error_rate=0.05, n_threads=8),
             w.generate_pairs(seed=32, n_pairs=700, length=1500, error_rate=0.05, n_threads=8)]
    qs, ts = [], []
    for blob, q_off, q_len, t_off, t_len in parts:
        for i in range(len(q_len)):
            qs.append(bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]))
            ts.append(bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]))
    qs[0] = b""            # first pair of the first shard
    ts[len(qs) - 1] = b""  # last pair of the last shard
    qs[100] = b"ACGTRYACGT"
    data = w.make_blob(qs, ts)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), *data, n_threads=8)
    m = w.MultiAligner(w.DefaultPenalties, w.DefaultOptions, devices=devices)
    assert m.AdaptiveReduction(w.DefaultAdaptiveOption) is None
    assert m.size() == (len(devices) if devices else max(1, __import__("torch").cuda.device_count()))
    got = m.align_arrays(*data)
    assert_batch_equal(got, want, f"context set {devices}")
    # few pairs: fewer than two per context -> one context takes them all
    small = w.make_blob(qs[1:4], ts[1:4])
    assert_batch_equal(m.align_arrays(*small), O.align_batch(_oracle_params(True, (10, 50, 1)), *small), "small batch")
    results, errors = m.AlignBatch(qs[:3], ts[:3])
    assert errors[0] is w.ErrEmptySeq and errors[1] is None
    m.close()


@pytest.mark.parametrize("n,length,err,first", [(3000, 1000, 0.05, 0), (500, 150, 0.02, 12345), (4, 100_000, 0.10, 2), (64, 37, 0.3, 7)])
def test_device_generator_writes_the_host_generator_s_dataset(built, n, length, err, first):
    """the synthetic pairs generated in HBM are, byte for byte, the ones the host generator
    makes (same seeded splitmix64 stream, same sequential edits) -- offsets, lengths and every sequence byte."""
    import torch
    import wfa_amd as w
    torch.zeros(1, device="cuda:0")  # (torch's own HIP runtime has to come up before the library's: it holds the buffers)
    host = w.generate_pairs(seed=11, n_pairs=n, length=length, error_rate=err, first_index=first, n_threads=8)
    al = w.New(device=0)
    dev = [t.cpu().numpy() for t in w.generate_pairs_device(al, 11, n, length, err, first_index=first)]
    blob_h, q_off, q_len, t_off, t_len = host
    assert np.array_equal(dev[1].view(np.uint64), q_off) and np.array_equal(dev[3].view(np.uint64), t_off)
    assert np.array_equal(dev[2].view(np.uint32), q_len) and np.array_equal(dev[4].view(np.uint32), t_len)
    for i in range(n):
        for off, ln in ((q_off, q_len), (t_off, t_len)):
            a, b = int(off[i]), int(off[i]) + int(ln[i])
            assert np.array_equal(dev[0][a:b], blob_h[a:b]), i
    al.close()


def test_align_pair_lds_arena(built):
    """The lone-pair instance with its arena rows in LDS (option pair_lds, the default): same results as the oracle and as the
    global-memory instance; a pair whose score needs more rows than 160 KB of LDS hold is finished by the global-memory
    instance in a second launch, and the calls after it start there."""
    import wfa_amd as w
    from oracle import oracle as O
    oa = O.Aligner(_oracle_params(True, (10, 50, 1), (4, 6, 2)))
    sets = {}
    for name, length, err, n in (("near", 1000, 0.05, 40), ("far", 3000, 0.30, 3), ("short", 60, 0.1, 20)):
        blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=length + 3, n_pairs=n, length=length, error_rate=err)
        sets[name] = [(bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]), bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]))
                      for i in range(n)]
    key = lambda r: (r.Score, r.CIGAR(False), r.QBegin, r.QEnd, r.TBegin, r.TEnd, r.AlignLen, r.Matches, r.Gaps, r.GapRegions)
    okey = lambda x: (x.score, x.cigar, x.qbegin, x.qend, x.tbegin, x.tend, x.align_len, x.matches, x.gaps, x.gap_regions)
    al = _aligner(True, (10, 50, 1), (4, 6, 2))
    for q, t in sets["near"] + sets["short"]:
        assert key(al.Align(q, t)) == okey(oa.align(q, t))
        tm = al.last_timing()
        assert tm.n_launches == 1 and tm.main_kernel_kind == 16
    # a pair that outgrows the rows in LDS (score > 1 240): two launches, then the global-memory instance first for a while
    q, t = sets["far"][0]
    want = oa.align(q, t)
    assert want.score > 1300
    al.set_option("pair_lds", 1)  # (also clears the skip count)
    assert key(al.Align(q, t)) == okey(want)
    first = al.last_timing().n_launches
    assert key(al.Align(q, t)) == okey(want)
    second = al.last_timing().n_launches
    # (first call: the LDS instance, then the global-memory instance -- and the batch entry behind it when the band outgrows the
    # 64-diagonal window too; second call: the LDS instance is skipped)
    assert first >= 2 and second <= first, (first, second)
    for q, t in sets["far"][1:] + sets["near"][:5]:
        assert key(al.Align(q, t)) == okey(oa.align(q, t))
    # the global-memory instance alone
    al.set_option("pair_lds", 0)
    for q, t in sets["near"][:10] + sets["far"][:1]:
        assert key(al.Align(q, t)) == okey(oa.align(q, t))
    al.close()


def test_align_pair_entry(built):
    """wfahip_align_pair (Aligner.Align): the two-launch path through the mapped pinned block for pairs shaped like the
    reference's defaults, the batch entry behind it for everything else -- semi-global, other penalties, bytes outside ACGT,
    bands wider than 64 diagonals, long pairs -- every result against the oracle; a CIGAR buffer that is too small reports
    the capacity it needs."""
    import ctypes as C
    import wfa_amd as w
    from wfa_amd import _lib as L
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    cases = []
    for length, err, n in ((1000, 0.05, 60), (150, 0.02, 60), (37, 0.1, 30), (1000, 0.25, 12), (5000, 0.03, 6), (12000, 0.02, 2)):
        blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=length + 7, n_pairs=n, length=length, error_rate=err)
        for i in range(n):
            q = bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])])
            t = bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])])
            if i % 9 == 0:
                q = q[int(rng.integers(0, max(1, len(q) // 3))):]  # overhangs: the exact WF_NEXT at sequence ends
            if i % 13 == 0:
                t = t.lower() if i % 2 else t[:len(t) // 2] + b"N" + t[len(t) // 2:]  # bytes outside ACGT: the byte-compare path
            cases.append((q, t))
    cases += [(b"ACCATACTCG", b"AGGATGCTCG"), (b"A", b"CA"), (b"C", b"C"), (b"ACTG", b"ACTGA")]
    for glob, ad, pen in ((True, (10, 50, 1), (4, 6, 2)), (True, None, (4, 6, 2)), (False, (10, 50, 1), (4, 6, 2)), (True, (10, 50, 1), (2, 3, 1)),
                          (True, None, (5, 3, 2))):
        al = _aligner(glob, ad, pen)
        oa = O.Aligner(_oracle_params(glob, ad, pen))
        fast = 0
        for q, t in cases:
            r = al.Align(q, t)
            tm = al.last_timing()
            fast += tm.n_launches == 1 and tm.main_kernel_kind == 16  # (ONE launch of the lone-pair instance: only the fast path does that)
            want = oa.align(q, t)
            assert (r.Score, r.CIGAR(False), r.QBegin, r.QEnd, r.TBegin, r.TEnd, r.AlignLen, r.Matches, r.Gaps, r.GapRegions) == \
                   (want.score, want.cigar, want.qbegin, want.qend, want.tbegin, want.tend, want.align_len, want.matches, want.gaps,
                    want.gap_regions), (glob, ad, pen, len(q), len(t))
        if glob and pen in ((4, 6, 2), (2, 3, 1)):
            # most pairs took the two-launch path (without wf-adaptive the 1 kbp pairs outgrow the 64-diagonal window)
            assert fast > (len(cases) // 2 if ad else len(cases) // 3), fast
        else:
            assert fast == 0
        # round 3's fast path (one launch of the four-pairs-per-wave streaming instance, a lane walking the backtrace) is kept
        al.set_option("pair_fast", 3)
        for q, t in cases[:40]:
            r, want = al.Align(q, t), oa.align(q, t)
            assert (r.Score, r.CIGAR(False), r.QBegin, r.QEnd, r.TBegin, r.TEnd) == (want.score, want.cigar, want.qbegin, want.qend, want.tbegin, want.tend)
        # the same pairs with the fast path switched off: the batch entry alone
        al.set_option("pair_fast", 0)
        for q, t in cases[:40]:
            r, want = al.Align(q, t), oa.align(q, t)
            assert (r.Score, r.CIGAR(False)) == (want.score, want.cigar)
        with pytest.raises(w.ErrEmptySeq.__class__):
            al.Align(b"", b"ACGT")
        al.close()
    # a CIGAR buffer that is too small
    al = _aligner(True, (10, 50, 1))
    q, t = cases[0]
    rec, ops, n_ops = (C.c_uint32 * 16)(), (C.c_uint64 * 4)(), C.c_uint64()
    prm = al._params()
    rc = L.lib().wfahip_align_pair(al._ctx, C.byref(prm), q, len(q), t, len(t), rec, ops, 4, C.byref(n_ops))
    assert rc == L.ERR_OOM and n_ops.value == len(al.Align(q, t).Ops) > 4
    al.close()


def test_host_entry_packs_on_the_fly(built):
    """wfahip_align_batch on a large batch 2-bit packs the sequences on host threads, slice by slice, beside the upload
    (a quarter of the bytes cross PCIe): same results as with the packing switched off; a single byte outside ACGT deep
    inside the batch sends the whole batch down the byte path -- same results again, and that pair equals the oracle's."""
    import wfa_amd as w
    from oracle import oracle as O
    n = 300_000
    data = w.generate_pairs(seed=29, n_pairs=n, length=1000, error_rate=0.05, n_threads=32)
    al = _aligner(True, (10, 50, 1))
    a = al.align_arrays(*data)
    al.set_option("autopack", 0)
    b = al.align_arrays(*data)
    al.set_option("autopack", 1)
    assert_batch_equal(a, b, "packed on the fly vs bytes")
    assert (a.status == 0).all()
    # the oracle on a sample
    idx = np.arange(0, n, 997)
    blob, q_off, q_len, t_off, t_len = data
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), blob, q_off[idx], q_len[idx], t_off[idx], t_len[idx], n_threads=8)
    assert np.array_equal(a.score[idx], want.score) and np.array_equal(a.ops_len[idx], want.ops_len)
    for j, i in enumerate(idx[:50]):
        assert np.array_equal(a.pair_ops(int(i)), want.pair_ops(j))
    # a lowercase base in pair 250 001 (last slice)
    blob2 = blob.copy()
    k = 250_001
    blob2[int(q_off[k]) + 500] = ord("a")
    c = al.align_arrays(blob2, q_off, q_len, t_off, t_len)
    for f in ("status", "score", "ops_len"):
        x, y = getattr(c, f).copy(), getattr(a, f).copy()
        x[k] = y[k] = 0
        assert np.array_equal(x, y), f
    wk = O.align_batch(_oracle_params(True, (10, 50, 1)), blob2, q_off[k:k + 1], q_len[k:k + 1], t_off[k:k + 1], t_len[k:k + 1], n_threads=1)
    assert int(c.score[k]) == int(wk.score[0]) and np.array_equal(c.pair_ops(k), wk.pair_ops(0))
    al.close()


def test_debug_options_need_the_environment_switch(built):
    """wfahip_set_option: the public keys are always accepted; routing experiments and test aids (here "team_strict", which
    drops a release the team kernel needs, and "fail_pass", which injects a failure) are refused with WFAHIP_ERR_UNSUPPORTED
    unless WFAHIP_DEBUG=1 is in the environment (tests/conftest.py sets it for the suite)."""
    import wfa_amd as w
    from wfa_amd import _lib as L
    al = w.New()
    lib = L.lib()
    saved = os.environ.pop("WFAHIP_DEBUG", None)
    try:
        assert lib.wfahip_set_option(al._ctx, b"census", 1) == L.OK
        assert lib.wfahip_set_option(al._ctx, b"census", 0) == L.OK
        assert lib.wfahip_set_option(al._ctx, b"learn", 1) == L.OK
        for key in (b"team_strict", b"fail_pass", b"arena_poison", b"unpack_all", b"duo", b"chunk_pairs"):
            assert lib.wfahip_set_option(al._ctx, key, 0) == L.ERR_UNSUPPORTED, key
        assert b"WFAHIP_DEBUG" in lib.wfahip_last_error(al._ctx)
        os.environ["WFAHIP_DEBUG"] = "0"
        assert lib.wfahip_set_option(al._ctx, b"team_strict", 1) == L.ERR_UNSUPPORTED
        os.environ["WFAHIP_DEBUG"] = "1"
        assert lib.wfahip_set_option(al._ctx, b"team_strict", 1) == L.OK
        assert lib.wfahip_set_option(al._ctx, b"no_such_option", 1) == L.ERR_BAD_ARG
    finally:
        if saved is None:
            os.environ.pop("WFAHIP_DEBUG", None)
        else:
            os.environ["WFAHIP_DEBUG"] = saved
    al.close()


def test_two_contexts_run_team_kernels_on_one_gpu(built):
    """Two contexts on ONE GPU, each aligning 2 x 40 kbp semi-global pairs on its own host thread at the same time (what
    wfahip_create_multi with repeated device ids, or bench.py --share-gpus, produces).  The team kernels spin on barriers
    between workgroups and need their whole launch resident: two such launches side by side would starve each other until
    the barrier timeout.  The library serialises team launches per device (TeamLaunchLock, wfa_host.hip): both calls must
    finish, bit-exact against the oracle, with the team kernel having run in both -- three rounds, over poisoned arenas."""
    import threading
    import wfa_amd as w
    from oracle import oracle as O
    sets = [w.generate_pairs(seed=700 + i, n_pairs=2, length=40000, error_rate=0.08, n_threads=2) for i in range(2)]
    prm = _oracle_params(False, (10, 50, 1))
    wants = [O.align_batch(prm, *d, n_threads=2) for d in sets]
    als = [_aligner(False, (10, 50, 1)) for _ in range(2)]
    for al in als:
        al.set_option("arena_poison", 1)
    for rnd in range(3):
        gots, errs = [None, None], [None, None]

        def run(i):
            try:
                gots[i] = als[i].align_arrays(*sets[i])
            except Exception as e:  # noqa: BLE001 -- reported below, on the test's thread
                errs[i] = e
        ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in ths), "a call did not come back"
        assert errs == [None, None], errs
        for i in range(2):
            assert_batch_equal(gots[i], wants[i], f"two contexts, round {rnd}, context {i}")
    for al in als:
        al.close()


def test_two_contexts_run_the_headline_shape_on_one_gpu(built):
    """Two contexts on ONE GPU, each aligning the headline shape (1e6 x 1 kbp @5 %, global + wf-adaptive: a 32 GiB arena
    per context) on its own host thread at the same time -- a second tenant on the device.  Both must return the records and
    CIGARs a lone context returns for the same batch (which test_full_size_parity_c3 holds against the oracle), in two rounds."""
    import threading
    import wfa_amd as w
    data = w.generate_pairs(seed=3, n_pairs=1_000_000, length=1000, error_rate=0.05, n_threads=32)
    lone = _aligner(True, (10, 50, 1))
    want = lone.align_arrays(*data)
    assert (want.status == 0).all()
    others = [lone, _aligner(True, (10, 50, 1))]
    for rnd in range(2):
        gots, errs = [None, None], [None, None]

        def run(i):
            try:
                gots[i] = others[i].align_arrays(*data)
            except Exception as e:  # noqa: BLE001 -- reported on the test's thread
                errs[i] = e
        ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in ths) and errs == [None, None], errs
        for i in range(2):
            assert_batch_equal(gots[i], want, f"two tenants, round {rnd}, context {i}")
    for al in others:
        al.close()
