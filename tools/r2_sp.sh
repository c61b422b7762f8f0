mkdir -p gpurun_out/r2n
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2n/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r2n/pytest.log)"
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10"
$B --workload config4 | python tools/bench_line.py "config4 packed shorts"
$B --workload config4 --no-overlap | python tools/bench_line.py "config4 packed shorts, no overlap"
$B | python tools/bench_line.py "config3"
$B --no-overlap | python tools/bench_line.py "config3, no overlap"
