import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_parity_gpu as T
bad = 0
for seed in range(1000, 1060):
    try:
        T.test_fuzz_configs(None, seed)
    except AssertionError as e:
        bad += 1; print("FAIL", seed, str(e)[:200], flush=True)
print("done, failures:", bad)
