#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/.  Run in the BUILD container (it reads the
reference checkout for input vectors; the GPU box never runs this).

  known_answers.json   hand-entered from the reference's published outputs (README.md, wfa_test.go
                       comments): these PIN the oracle.  Each entry cites its source.
  ref_test_pairs.json  the input pairs the reference's own test file holds (wfa_test.go:49-141,
                       wfa-go/seqs.txt) -- inputs only, the reference's test asserts nothing.
  oracle_vectors.json  oracle-generated expectations (results for every pair above under several
                       option sets + per-step wavefront dumps for KA1/KA2).  These are
                       self-consistency vectors for the HIP path, labelled as such.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

KNOWN = [
    dict(id="KA1", source="README.md:101-124", mode="global", adaptive=[10, 50, 1],
         q="ACCATACTCG", t="AGGATGCTCG", cigar="1M2X2M1X4M", score=12, qbegin=1, qend=10, tbegin=1, tend=10,
         align_len=10, matches=7, gaps=0, gap_regions=0, cigar_exact=True),
    dict(id="KA2", source="README.md:128-149", mode="semi-global", adaptive=[10, 50, 1],
         q="ACGATCTCG", t="CAGGCTCCTCGG", cigar="1I1M1X1M1X1M1I4M1I", score=16, qbegin=1, qend=9, tbegin=2,
         tend=11, align_len=10, matches=7, gaps=1, gap_regions=1, cigar_exact=False,
         note="README block predates v0.4.0 (usage text says v0.2.0; its M-table has a cell unreachable under "
              "v0.4.0 rules, SURVEY.md section 4 caveat 3).  v0.4.0's backtrace derives match-run lengths from "
              "stored offsets (wfa.go:833-848), which places the inner insertion one column later: "
              "1I1M1X1M1X2M1I3M1I -- same score, region and statistics.  Only those are pinned for KA2."),
    dict(id="KA3", source="README.md:231-239; wfa_test.go:83", mode="global", adaptive=[10, 50, 1],
         q="AGCTAGTGTCAATGGCTACTTTTCAGGTCCT", t="AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT",
         cigar="1M1X3M1I5M2X8M3I1M1X9M", score=36, qbegin=1, qend=31, tbegin=1, tend=35, align_len=35, matches=27,
         gaps=4, gap_regions=2, cigar_exact=True,
         text=["AGCTA-GTGTCAATGGCTACT---TTTCAGGTCCT", "| ||| |||||  ||||||||   | |||||||||",
               "AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT"]),
    dict(id="KA4", source="README.md:245-253; wfa-go/seqs.txt:1-2", mode="global", adaptive=[10, 50, 1],
         q="ATTGGAAAATAGGATTGGGGTTTGTTTATATTTGGGTTGAGGGATGTCCCACCTTCGTCGTCCTTACGTTTCCGGAAGGGAGTGGTTAGCTCGAAGCCCA",
         t="GATTGGAAAATAGGATGGGGTTTGTTTATATTTGGGTTGAGGGATGTCCCACCTTGTCGTCCTTACGTTTCCGGAAGGGAGTGGTTGCTCGAAGCCCA",
         cigar="1X1I14M1D39M1D31M1D12M", score=36, qbegin=2, qend=100, tbegin=3, tend=98, align_len=99, matches=96,
         gaps=3, gap_regions=3, cigar_exact=True,
         text=["A-TTGGAAAATAGGATTGGGGTTTGTTTATATTTGGGTTGAGGGATGTCCCACCTTCGTCGTCCTTACGTTTCCGGAAGGGAGTGGTTAGCTCGAAGCCCA",
               "  |||||||||||||| ||||||||||||||||||||||||||||||||||||||| ||||||||||||||||||||||||||||||| ||||||||||||",
               "GATTGGAAAATAGGAT-GGGGTTTGTTTATATTTGGGTTGAGGGATGTCCCACCTT-GTCGTCCTTACGTTTCCGGAAGGGAGTGGTT-GCTCGAAGCCCA"]),
    dict(id="KA5", source="README.md:18-27", mode="semi-global", adaptive=[10, 50, 1],
         q="Bioinformatics helps Biology", t="We learn bioinformatics to help biologists",
         cigar="9I1X14M3I4M1D1M1X5M1X3I", score=32, qbegin=2, qend=27, tbegin=11, tend=38, align_len=29, matches=24,
         gaps=4, gap_regions=2, cigar_exact=True,
         text=["---------Bioinformatics ---helps Biology---", "          ||||||||||||||   |||| | |||||   ",
               "We learn bioinformatics to help- biologists"]),
    dict(id="KA6", source="wfa_test.go:94-96 (comment '1X99M')", mode="global", adaptive=[10, 50, 1],
         q="ACTATAAGCGTCCTCTGCGAGACCGGATGCGTTGATGACAGCGAATTGAGTTGAACTCCCTAAGGACACTCAATAATATTGGTCTATGCAAAAAGTCATT",
         t="CCTATAAGCGTCCTCTGCGAGACCGGATGCGTTGATGACAGCGAATTGAGTTGAACTCCCTAAGGACACTCAATAATATTGGTCTATGCAAAAAGTCATT",
         cigar="1X99M", cigar_exact=True),
]

# KA1's M-component table as printed in README.md:101-114 (Plot with notChangeToMatch=false): rows = query
# position v (1-based), cols = target position h; value = [arrow kind, score].  Kinds: match = cell reached by
# extension or the (0,0) match seed, mis = Mismatch, io/ie = insertion open/ext, do/de = deletion open/ext.
KA1_TABLE = {
    "1,1": ["match", 0], "1,2": ["io", 8], "1,3": ["ie", 10], "1,4": ["ie", 12],
    "2,1": ["do", 8], "2,2": ["mis", 4], "2,3": ["mis", 12],
    "3,1": ["de", 10], "3,2": ["mis", 12], "3,3": ["mis", 8],
    "4,1": ["de", 12], "4,4": ["match", 8],
    "5,5": ["match", 8],
    "6,6": ["mis", 12],
    "7,7": ["match", 12], "8,8": ["match", 12], "9,9": ["match", 12], "10,10": ["match", 12],
}


def extract_ref_pairs():
    """(label, q, t) from the commented q/t assignments of wfa_test.go and the seqs.txt pairs."""
    pairs = []
    src = open(os.path.join(REF, "wfa_test.go")).read().splitlines()
    cur_q = cur_t = None
    label = ""
    for ln, line in enumerate(src, 1):
        m = re.match(r'\s*(?://\s*)?([qt]) = \[\]byte\("([^"]*)"\)', line)
        if not m:
            c = re.match(r"\s*//\s*(.+)$", line)
            if c and "byte(" not in line:
                label = c.group(1).strip()
            continue
        if m.group(1) == "q":
            cur_q = (ln, m.group(2))
        else:
            cur_t = (ln, m.group(2))
        if cur_q and cur_t and abs(cur_q[0] - cur_t[0]) == 1:
            pairs.append(dict(source=f"wfa_test.go:{min(cur_q[0], cur_t[0])}", label=label, q=cur_q[1], t=cur_t[1]))
            cur_q = cur_t = None
    seqs = open(os.path.join(REF, "wfa-go", "seqs.txt")).read().splitlines()
    for i in range(0, len(seqs) - 1, 2):
        pairs.append(dict(source=f"wfa-go/seqs.txt:{i + 1}", label="seqs.txt", q=seqs[i][1:], t=seqs[i + 1][1:]))
    return pairs


def main():
    from oracle import oracle as O

    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(dict(vectors=KNOWN, ka1_m_table=KA1_TABLE), f, indent=1)

    pairs = extract_ref_pairs()
    with open(os.path.join(HERE, "ref_test_pairs.json"), "w") as f:
        json.dump(pairs, f, indent=1)

    option_sets = [
        dict(name="global+adaptive", global_alignment=True, adaptive=[10, 50, 1]),
        dict(name="global", global_alignment=True, adaptive=None),
        dict(name="semiglobal+adaptive", global_alignment=False, adaptive=[10, 50, 1]),
        dict(name="semiglobal", global_alignment=False, adaptive=None),
    ]
    out = dict(note="oracle-generated (self-consistency vectors, NOT reference outputs)", results=[], dumps=[])
    for oi, opt in enumerate(option_sets):
        al = O.Aligner(global_alignment=opt["global_alignment"],
                       adaptive=tuple(opt["adaptive"]) if opt["adaptive"] else None)
        for pi, p in enumerate(pairs):
            q, t = p["q"].upper().encode(), p["t"].upper().encode()  # the reference test upper-cases (wfa_test.go:143)
            r = al.align(q, t)
            out["results"].append(dict(pair=pi, options=opt["name"], status=r.status, score=r.score, cigar=r.cigar,
                                       qbegin=r.qbegin, qend=r.qend, tbegin=r.tbegin, tend=r.tend,
                                       align_len=r.align_len, matches=r.matches, gaps=r.gaps,
                                       gap_regions=r.gap_regions))
    # per-step dumps (after next / extend / reduce) for KA1 and KA2
    for ka in KNOWN[:2]:
        al = O.Aligner(global_alignment=(ka["mode"] == "global"), adaptive=tuple(ka["adaptive"]))
        steps = []

        def hook(phase, s, al=al, steps=steps):
            snap = {}
            for ci, name in enumerate("MID"):
                w = al.wavefront(ci, s)
                if w is not None:
                    lo, hi, raw = w
                    snap[name] = {str(lo + i): v for i, v in enumerate(raw) if v}
            steps.append(dict(phase=["init", "next", "extend", "reduce"][phase], s=s, wf=snap))

        al.set_hook(hook)
        al.align(ka["q"].encode(), ka["t"].encode())
        final = {name: {str(s): {str(lo + i): v for i, v in enumerate(raw) if v} for s, (lo, hi, raw) in d.items()}
                 for name, d in al.dump().items()}
        out["dumps"].append(dict(id=ka["id"], steps=steps, final=final))
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"{len(KNOWN)} known answers, {len(pairs)} reference test pairs, {len(out['results'])} oracle results")


if __name__ == "__main__":
    main()
