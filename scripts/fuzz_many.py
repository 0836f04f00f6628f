"""GPU box: test_fuzz_configs for a range of seeds.  Usage: scripts/fuzz_many.py [first_seed] [count]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_parity_gpu as T
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    try:
        T.test_fuzz_configs(None, seed)
    except AssertionError as e:
        bad += 1; print("FAIL", seed, str(e)[:200], flush=True)
print("done, seeds", first, "..", first + count - 1, "failures:", bad)
