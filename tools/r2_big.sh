timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "synthetic_vs_oracle" > gpurun_out/pytest_big.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_big.log)"; grep -E "^FAILED|^E  " gpurun_out/pytest_big.log | head -6
timeout -k 10 120 python tools/u_debug.py 1 256 4096 111001110011 1 12 2>&1 | grep -v amdgpu | tail -14
