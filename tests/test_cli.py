"""The wfa-go-equivalent harness (wfa_amd/cli.py): output text identical to what the reference's README shows."""
import io
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

README_KA3 = """query   AGCTA-GTGTCAATGGCTACT---TTTCAGGTCCT
        | ||| |||||  ||||||||   | |||||||||
target  AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT
cigar   1M1X3M1I5M2X8M3I1M1X9M

align-score : 36
match-region: q[1, 31]/31 vs t[1, 35]/35
align-length: 35, matches: 27 (77.14%), gaps: 4, gap regions: 2

"""  # README.md:231-239

README_KA5 = """query   ---------Bioinformatics ---helps Biology---
                  ||||||||||||||   |||| | |||||   
target  We learn bioinformatics to help- biologists
cigar   9I1X14M3I4M1D1M1X5M1X3I

align-score : 32
match-region: q[2, 27]/28 vs t[11, 38]/42
align-length: 29, matches: 24 (82.76%), gaps: 4, gap regions: 2

"""  # README.md:18-27 (the README strips the trailing blanks of the bar line)


def _norm(text):
    return "\n".join(line.rstrip() for line in text.split("\n"))


def test_format_matches_readme_blocks_from_oracle_results():
    """CPU: the formatter + AlignmentText over ORACLE results reproduce the README's printed blocks."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from wfa_amd.aligner import AlignmentResult
    from wfa_amd.cli import format_result, read_pairs
    for q, t, glob, want in ((b"AGCTAGTGTCAATGGCTACTTTTCAGGTCCT", b"AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT", True, README_KA3),
                             (b"Bioinformatics helps Biology", b"We learn bioinformatics to help biologists", False,
                              README_KA5)):
        r = O.Aligner(global_alignment=glob, adaptive=(10, 50, 1)).align(q, t)
        ar = AlignmentResult(Ops=r.ops, Score=r.score, TBegin=r.tbegin, TEnd=r.tend, QBegin=r.qbegin, QEnd=r.qend,
                             AlignLen=r.align_len, Matches=r.matches, Gaps=r.gaps, GapRegions=r.gap_regions)
        assert _norm(format_result(ar, q, t, False)) == _norm(want)
    pairs = read_pairs(os.path.join(ROOT, "tests", "golden", "seqs_format_example.txt"))
    assert pairs == [(b"ACCATACTCG", b"AGGATGCTCG"), (b"ACGATCTCG", b"CAGGCTCCTCGG")]


@pytest.mark.gpu
def test_cli_end_to_end(built, tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "wfa_amd.cli", "AGCTAGTGTCAATGGCTACTTTTCAGGTCCT",
                          "AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    assert _norm(out.stdout) == _norm(README_KA3)
    out = subprocess.run([sys.executable, "-m", "wfa_amd.cli", "-g", "Bioinformatics helps Biology",
                          "We learn bioinformatics to help biologists"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert _norm(out.stdout) == _norm(README_KA5)
    f = tmp_path / "pairs.seq"
    f.write_text(">ACCATACTCG\n<AGGATGCTCG\n>ACGATCTCG\n<CAGGCTCCTCGG\n")
    out = subprocess.run([sys.executable, "-m", "wfa_amd.cli", "-i", str(f)], capture_output=True, text=True, env=env,
                         cwd=ROOT)
    assert out.stdout.count("cigar   ") == 2 and "cigar   1M2X2M1X4M" in out.stdout
    out = subprocess.run([sys.executable, "-m", "wfa_amd.cli", "-N", "-i", str(f)], capture_output=True, text=True,
                         env=env, cwd=ROOT)
    assert out.returncode == 0 and out.stdout == ""
    out = subprocess.run([sys.executable, "-m", "wfa_amd.cli", "-t", "-g", "Bioinformatics helps Biology",
                          "We learn bioinformatics to help biologists"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert "cigar   14M3I4M1D1M1X5M\n" in out.stdout
