#!/usr/bin/env python3
"""Oracle results of BASELINE configs[4]'s sample -- the first 8 pairs of the seed-5 dataset (100 kbp @10 %), semi-global,
wf-adaptive 10/50/1 -- as a fixture: per pair every record field and the SHA-256 of its CIGAR op array (op << 32 | n, little
endian), i.e. a bit-exact stand-in for re-running the oracle (a minute and a half of one core and up to 30 GB per pair).
tests/test_parity_gpu.py::test_config5_full_length_pair compares the GPU path with it; WFA_TEST_FULL_ORACLE=1 makes that
test run the oracle itself as before.  Usage: python tests/golden/make_c5_golden.py [threads=2]  (writes c5_sample_oracle.json)"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

FIELDS = ("status", "score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len")


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import wfa_amd as w  # (the dataset generator lives in the library; it needs no GPU)
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(5, 8, 100_000, 0.10)
    want = O.align_batch(O.make_params(global_alignment=False, adaptive=(10, 50, 1)), blob, q_off, q_len, t_off, t_len, n_threads=threads)
    doc = {"workload": "seed 5, pairs 0..7, 100000 bp @10 %, semi-global 4/6/2, wf-adaptive 10/50/1",
           "input_sha256": hashlib.sha256(blob.tobytes()).hexdigest(), "pairs": []}
    for i in range(8):
        rec = {f: int(getattr(want, f)[i]) for f in FIELDS}
        rec["ops_sha256"] = hashlib.sha256(np.ascontiguousarray(want.pair_ops(i)).astype("<u8").tobytes()).hexdigest()
        doc["pairs"].append(rec)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_sample_oracle.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc)[:400])


if __name__ == "__main__":
    main()
