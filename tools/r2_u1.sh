mkdir -p gpurun_out/r2h
VSYN_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "synthetic_vs_oracle" > gpurun_out/r2h/pytest_default.log 2>&1; echo "default rc=$?"; tail -25 gpurun_out/r2h/pytest_default.log
VSYN_U_MIXED=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2h/pytest_umixed.log 2>&1; echo "umixed rc=$?"; tail -15 gpurun_out/r2h/pytest_umixed.log
