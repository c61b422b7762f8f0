B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10"
for v in dma; do
cp build_ab/lib_$v.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
echo "== $v"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_host_decoder.py -x -q > gpurun_out/pytest_$v.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_$v.log)"
$B --workload config4 | python tools/bench_line.py "config4"
$B --workload config4 --no-overlap | python tools/bench_line.py "config4, no overlap"
$B | python tools/bench_line.py "config3"
done
