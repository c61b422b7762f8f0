"""GPU: the N > 1 path of bench.py, rehearsed on ONE GPU — two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ...`), both placed on device 0 and with gloo carrying the
three tiny collectives (RCCL refuses two ranks on one device; see the rehearsal knobs in bench.py). Checks what a multi-GPU node
would otherwise be the first to run: setup broadcast, barriers, max-clock / summed-count reduction, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(extra):
    port = _free_port()
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    return json.loads(lines[0])


def test_two_ranks_default_workload():
    j = _run(["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--parity-gather"])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["packets_per_gpu"] == 65536
    # the parity configuration's PCM gather: both ranks' streams, gathered over the process group, equal the oracle's
    assert j["parity_gather"]["streams"] == 5 and j["parity_gather"]["ranks"] == 2 and j["parity_gather"]["max_abs_err_vs_oracle"] < 1e-5
    # whole-job aggregate: both ranks' packets over the slower rank's clock
    assert abs(j["value"] - 2 * 65536 * 20 / (j["ms_per_step"] * 1e-3 * 20)) / j["value"] < 0.01
    assert j["pcm_max_abs_err_vs_oracle"] is not None and j["pcm_max_abs_err_vs_oracle"] < 1e-5


def test_two_ranks_real_files():
    j = _run(["--steps", "1", "--warmup", "1", "--workload", "config5", "--files-per-gpu", "200", "--host-threads", "4"])
    assert j["n_gpus"] == 2 and j["config"]["packets_per_gpu"] == 200 * 94 and j["replicas_bit_identical"]
    assert j["host_placement"]["pinned"] and j["host_placement"]["rank0_cpus"]  # every rank pinned itself to its share of the cores
