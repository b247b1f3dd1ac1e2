"""more seeds of the randomised differential tests (strict build against the oracle, bit-exact) than the test suite runs.
usage: python tools/stress_random.py [cell|vertex] [first_seed] [last_seed]"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
which = sys.argv[1] if len(sys.argv) > 1 else "cell"
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 140)
if which == "vertex":
    import test_gpu_node_parity as T
else:
    import test_gpu_parity as T
bad = 0
for seed in range(lo, hi):
    try:
        T.test_randomised_configurations_strict_vs_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:300], flush=True)
print(f"{which}: seeds {lo}..{hi - 1}, failures: {bad}")
sys.exit(1 if bad else 0)
