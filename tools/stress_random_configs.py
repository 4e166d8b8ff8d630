"""dev: the randomised parity tests of tests/test_gpu_scale.py on seeds beyond the committed ones (usage: python
tools/stress_random_configs.py FIRST LAST); prints the failing seeds."""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import test_gpu_scale as T

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    for fn in (T.test_kfac_random_configurations_vs_oracle, T.test_diag_lastlayer_jacobians_random_configurations_vs_oracle):
        try:
            fn(seed)
        except Exception as e:  # noqa: BLE001
            bad.append((seed, fn.__name__, repr(e)[:300]))
            traceback.print_exc()
    if seed % 20 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("FAILURES", bad)
