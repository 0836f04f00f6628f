"""CPU: the oracle against every known answer the reference publishes for this path."""
import pytest

from oracle import oracle as O


def _align(ka):
    al = O.Aligner(global_alignment=(ka["mode"] == "global"), adaptive=tuple(ka["adaptive"]))
    return al, al.align(ka["q"].encode(), ka["t"].encode())


def test_known_answers(known_answers):
    for ka in known_answers["vectors"]:
        al, r = _align(ka)
        assert r.status == O.OK, ka["id"]
        if ka.get("cigar_exact", True):
            assert r.cigar == ka["cigar"], ka["id"]
        for f in ("score", "qbegin", "qend", "tbegin", "tend", "align_len", "matches", "gaps", "gap_regions"):
            if f in ka:
                assert getattr(r, f) == ka[f], (ka["id"], f)


def _cigar_counts(cigar):
    import re
    c = {}
    for n, o in re.findall(r"(\d+)([A-Z])", cigar):
        c[o] = c.get(o, 0) + int(n)
    return c


def test_ka2_equivalent_alignment(known_answers):
    """KA2's printed CIGAR predates v0.4.0; the oracle's differs only in where an equal-cost insertion sits."""
    ka = [k for k in known_answers["vectors"] if k["id"] == "KA2"][0]
    _, r = _align(ka)
    assert r.cigar == "1I1M1X1M1X2M1I3M1I"
    assert _cigar_counts(r.cigar) == _cigar_counts(ka["cigar"])
    assert cigar_score(r.cigar, ka["q"], ka["t"], semi_global=True) == ka["score"]
    assert cigar_score(ka["cigar"], ka["q"], ka["t"], semi_global=True) == ka["score"]


def cigar_score(cigar, q, t, x=4, o=6, e=2, semi_global=False):
    """Gap-affine cost of a CIGAR; checks M/X columns against the sequences.  Flanking I/H are free in
    semi-global mode (they are clipping)."""
    import re
    ops = [(int(n), c) for n, c in re.findall(r"(\d+)([A-Z])", cigar)]
    first_m = next(i for i, (_, c) in enumerate(ops) if c in "MX")
    last_m = max(i for i, (_, c) in enumerate(ops) if c in "MX")
    v = h = 0
    score = 0
    for i, (n, c) in enumerate(ops):
        inner = first_m <= i <= last_m or not semi_global
        if c == "M":
            assert q[v:v + n] == t[h:h + n]
            v += n
            h += n
        elif c == "X":
            assert all(a != b for a, b in zip(q[v:v + n], t[h:h + n]))
            score += x * n
            v += n
            h += n
        elif c == "I":
            h += n
            if inner:
                score += o + e * n
        elif c in "DH":
            v += n
            if inner:
                score += o + e * n
    assert v == len(q) and h == len(t)
    return score


def test_alignment_text(known_answers):
    """AlignmentText rendering of KA3-KA5 (wfa_cigar.go:259-333) from the oracle's ops."""
    from wfa_amd.aligner import AlignmentResult
    for ka in known_answers["vectors"]:
        if "text" not in ka:
            continue
        _, r = _align(ka)
        ar = AlignmentResult(Ops=r.ops, QBegin=r.qbegin, QEnd=r.qend, TBegin=r.tbegin, TEnd=r.tend)
        Q, A, T = ar.AlignmentText(ka["q"].encode(), ka["t"].encode(), False)
        assert Q.decode() == ka["text"][0]
        assert A.decode().rstrip() == ka["text"][1].rstrip()
        assert T.decode() == ka["text"][2]


def test_ka1_m_table(known_answers):
    """README.md:101-114: every cell of the KA1 M-table (score + arrow), rebuilt from the oracle's
    pre-/post-extension M wavefronts.  Pins next/extend internals cell by cell."""
    ka = known_answers["vectors"][0]
    al = O.Aligner(global_alignment=True, adaptive=tuple(ka["adaptive"]))
    pre = {}

    def hook(phase, s):
        if phase in (O.PH_NEXT, O.PH_INIT):
            for sc in ([s] if phase == O.PH_NEXT else range(0, 16)):
                w = al.wavefront(0, sc)
                if w is not None:
                    lo, hi, raw = w
                    pre[sc] = {lo + i: v for i, v in enumerate(raw) if v}

    al.set_hook(hook)
    al.align(ka["q"].encode(), ka["t"].encode())
    final = {s: {lo + i: v for i, v in enumerate(raw) if v} for s, (lo, hi, raw) in al.dump()["M"].items()}
    kinds = {1: "io", 2: "ie", 3: "do", 4: "de", 5: "mis", 6: "match"}
    table = {}
    for s in sorted(final):
        for k, raw_post in final[s].items():
            raw_pre = pre[s][k]
            h0, tag = raw_pre >> 3, raw_pre & 7
            h1 = raw_post >> 3
            for h in range(h0, h1 + 1):
                v = h - k
                if v < 1 or v > len(ka["q"]) or h > len(ka["t"]):
                    continue
                key = f"{v},{h}"
                if key not in table:  # the first (lowest) score paints the cell (wfa_component_plot.go:97-99)
                    table[key] = [kinds[tag] if h == h0 else "match", s]
    assert table == known_answers["ka1_m_table"]


def test_appendix_a_dump(oracle_vectors):
    """SURVEY.md Appendix A (hand-derived KA1 trace): raw words after next/extend at each score."""
    d = [x for x in oracle_vectors["dumps"] if x["id"] == "KA1"][0]["final"]
    assert d["M"]["0"] == {"0": 14}
    assert d["M"]["4"] == {"0": 21}
    assert d["I"]["8"] == {"1": 17} and d["D"]["8"] == {"-1": 11}
    assert d["M"]["8"] == {"-1": 11, "0": 45, "1": 17}
    assert d["M"]["10"] == {"-2": 12, "2": 26}
    assert d["I"]["12"] == {"1": 25, "3": 34} and d["D"]["12"] == {"-3": 12, "-1": 19}
    assert d["M"]["12"] == {"-3": 12, "-1": 21, "0": 85, "1": 29, "3": 34}


def test_edge_cases():
    al = O.Aligner(global_alignment=True)
    assert al.align(b"", b"ACGT").status == O.ERR_EMPTY
    assert al.align(b"ACGT", b"").status == O.ERR_EMPTY
    r = al.align(b"A", b"CA")  # first cell is always consumed
    assert (r.cigar, r.score) == ("1X1I", 12)
    r = al.align(b"C", b"C")
    assert (r.cigar, r.score, r.qbegin, r.qend, r.tbegin, r.tend) == ("1M", 0, 1, 1, 1, 1)
    r = al.align(b"CG", b"C")
    assert (r.cigar, r.score) == ("1M1D", 8)
    r = al.align(b"ACTG", b"ACTGA")
    assert (r.cigar, r.score) == ("4M1I", 8)


def test_oracle_vectors_stable(ref_pairs, oracle_vectors):
    """The committed oracle vectors are what the oracle produces today (guards accidental edits)."""
    opts = {"global+adaptive": (True, (10, 50, 1)), "global": (True, None),
            "semiglobal+adaptive": (False, (10, 50, 1)), "semiglobal": (False, None)}
    aligners = {k: O.Aligner(global_alignment=g, adaptive=a) for k, (g, a) in opts.items()}
    for ent in oracle_vectors["results"]:
        p = ref_pairs[ent["pair"]]
        if max(len(p["q"]), len(p["t"])) > 300 and ent["options"] != "global+adaptive":
            continue  # keep the CPU suite fast
        r = aligners[ent["options"]].align(p["q"].upper().encode(), p["t"].upper().encode())
        assert (r.status, r.score, r.cigar, r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps,
                r.gap_regions) == (ent["status"], ent["score"], ent["cigar"], ent["qbegin"], ent["qend"],
                                   ent["tbegin"], ent["tend"], ent["align_len"], ent["matches"], ent["gaps"],
                                   ent["gap_regions"])


def test_global_cigars_are_valid_alignments(ref_pairs, oracle_vectors):
    """Size-independent property: every global CIGAR consumes both sequences, its M/X columns agree with
    the bases, and its gap-affine cost equals the reported score."""
    for ent in oracle_vectors["results"]:
        if ent["options"] != "global":
            continue
        p = ref_pairs[ent["pair"]]
        q, t = p["q"].upper(), p["t"].upper()
        assert cigar_score(ent["cigar"], q, t) == ent["score"], p["source"]


def _plot_cells(text):
    return [[c.strip() for c in line.split("\t")] for line in text.rstrip("\n").split("\n")]


def _oracle_wavefronts(ka):
    al = O.Aligner(global_alignment=ka["mode"] == "global", adaptive=tuple(ka["adaptive"]))
    al.align(ka["q"].encode(), ka["t"].encode())
    return {c: {s: {lo + i: v for i, v in enumerate(raw) if v} for s, (lo, hi, raw) in d.items()}
            for c, d in al.dump().items()}


def test_plot_reproduces_readme_tables(known_answers):
    """N3 of SURVEY.md section 8f: the host mirror of (*Aligner).Plot (wfa_component_plot.go:41-209) over the
    oracle's final wavefronts prints the README's M-component tables (tests/golden/plot_tables.json, extracted
    by scripts/make_plot_golden.py).  KA1 (README.md:101-114): every cell.  KA2 (README.md:128-140): the README
    block predates v0.4.0 (caveat 3); the cells that differ are listed and must not grow."""
    import json
    import os
    from wfa_amd.aligner import plot_component
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "plot_tables.json"), encoding="utf-8"))
    ka1, ka2 = known_answers["vectors"][0], known_answers["vectors"][1]
    got1 = _plot_cells(plot_component(ka1["q"].encode(), ka1["t"].encode(), _oracle_wavefronts(ka1)))
    assert got1 == gold["ka1_global"]
    got2 = _plot_cells(plot_component(ka2["q"].encode(), ka2["t"].encode(), _oracle_wavefronts(ka2)))
    want2 = gold["ka2_semiglobal"]
    assert len(got2) == len(want2) and all(len(a) == len(b) for a, b in zip(got2, want2))
    diff = sorted((r - 1, c - 1) for r in range(len(want2)) for c in range(len(want2[r])) if got2[r][c] != want2[r][c])
    assert diff == [tuple(x) for x in PLOT_KA2_STALE], diff


# (query row, target column), 1-based, of the KA2 README cells that v0.4.0's rules do not reproduce
PLOT_KA2_STALE = [(3, 10), (4, 6), (5, 5), (5, 10), (6, 6), (7, 3), (7, 5)]
