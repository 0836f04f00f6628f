"""wfa-go equivalent command line: same flags, same input format and the
same text output as the reference CLI (wfa-go/wfa-go.go:70-78 flags, :125-136 output, :166-178 pair file), with
the alignments computed by the HIP path -- all pairs of the file in ONE batch call.

    python -m wfa_amd.cli [options] <query seq> <target seq>
    python -m wfa_amd.cli [options] -i input.txt          (lines ">query" / "<target")

Options:  -g  do not use global alignment          -a  do not use adaptive reduction
          -N  do not output alignment (benchmark)  -t  only show the aligned region
"""
from __future__ import annotations

import argparse
import sys
from typing import List, Tuple

VERSION = "0.4.0"


def read_pairs(path: str) -> List[Tuple[bytes, bytes]]:
    """wfa-go/wfa-go.go:166-178: line 2i = marker + query, line 2i+1 = marker + target; [1:] strips the marker."""
    pairs = []
    with open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for i in range(0, len(lines) - 1, 2):
        pairs.append((lines[i][1:], lines[i + 1][1:]))
    return pairs


def format_result(r, q: bytes, t: bytes, trim: bool) -> str:
    """The block wfa-go prints per pair (wfa-go/wfa-go.go:125-136), byte for byte."""
    Q, A, T = r.AlignmentText(q, t, trim)
    pct = float(r.Matches) / float(r.AlignLen) * 100 if r.AlignLen else float("nan")
    return (f"query   {Q.decode('latin-1')}\n"
            f"        {A.decode('latin-1')}\n"
            f"target  {T.decode('latin-1')}\n"
            f"cigar   {r.CIGAR(trim)}\n"
            f"\n"
            f"align-score : {r.Score}\n"
            f"match-region: q[{r.QBegin}, {r.QEnd}]/{len(q)} vs t[{r.TBegin}, {r.TEnd}]/{len(t)}\n"
            f"align-length: {r.AlignLen}, matches: {r.Matches} ({pct:.2f}%), gaps: {r.Gaps}, "
            f"gap regions: {r.GapRegions}\n"
            f"\n")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="wfa-hip", add_help=False, description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-h", action="help", help="print help message")
    ap.add_argument("-i", dest="infile", default="", help="input file.")
    ap.add_argument("-g", dest="no_global", action="store_true", help="do not use global alignment")
    ap.add_argument("-a", dest="no_adaptive", action="store_true", help="do not use adaptive reduction")
    ap.add_argument("-N", dest="no_output", action="store_true", help="do not output alignment (for benchmark)")
    ap.add_argument("-t", dest="trim", action="store_true", help="only show the aligned region")
    ap.add_argument("seqs", nargs="*")
    args = ap.parse_args(argv)

    import wfa_amd as wfa
    if args.infile == "":
        if len(args.seqs) != 2:
            print('if flag -i not given, please give me two sequences. type "wfa-hip -h" for help.', file=sys.stderr)
            return 1
        pairs = [(args.seqs[0].encode(), args.seqs[1].encode())]
    else:
        try:
            pairs = read_pairs(args.infile)
        except OSError:
            print(f"failed to read file: {args.infile}", file=sys.stderr)
            return 1

    algn = wfa.New(wfa.DefaultPenalties, wfa.Options(GlobalAlignment=not args.no_global))  # wfa-go.go:96-98
    if not args.no_adaptive:
        algn.AdaptiveReduction(wfa.AdaptiveReductionOption(10, 50, 1))  # wfa-go.go:100-106
    try:
        results, errors = algn.AlignBatch([p[0] for p in pairs], [p[1] for p in pairs])
        out = sys.stdout
        for (q, t), r, err in zip(pairs, results, errors):
            if err is not None:  # checkError: print and exit 1 (wfa-go.go:117-119,185-190)
                print(err, file=sys.stderr)
                return 1
            if not args.no_output:
                out.write(format_result(r, q, t, args.trim))
        out.flush()
    finally:
        wfa.RecycleAligner(algn)
    return 0


if __name__ == "__main__":
    sys.exit(main())
