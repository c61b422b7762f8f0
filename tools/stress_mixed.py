"""tools/stress_mixed.py [first_seed] [count] — one-off stress of the fused kernel's mixed-block path on the GPU box: the body of
tests/test_gpu_parity.py::test_random_block_patterns_runs_and_cuts for many seeds and run lengths (one submit == oracle; cut at
random places == uncut, bit for bit)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import test_gpu_parity as t  # noqa: E402


class _Env:
    def setenv(self, k, v):
        os.environ[k] = v


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + count):
    rl = [0, 4, 5, 6, 7, 9, 13][seed % 7]
    os.environ.pop("VSYN_RUN_LEN", None)
    t.test_random_block_patterns_runs_and_cuts(seed, rl, _Env())
    print("seed", seed, "run_len", rl, "ok", flush=True)
print("all ok")
