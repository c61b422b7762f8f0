mkdir -p gpurun_out/r2i
VSYN_U_MIXED=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2i/pytest_umixed.log 2>&1; echo "umixed rc=$?"; tail -5 gpurun_out/r2i/pytest_umixed.log
for w in config4 config3; do
VSYN_U_MIXED=1 timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --workload $w | python tools/bench_line.py "U_MIXED $w"
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --workload $w | python tools/bench_line.py "default $w"
done
