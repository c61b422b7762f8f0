B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10"
cp build_ab/lib_ud.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_decoder.py -x -q > gpurun_out/pytest_ud.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_ud.log)"
$B --workload config3 --blocksizes 128,1024 | python tools/bench_line.py "U config3 128/1024"
$B --workload config4 --blocksizes 128,1024 | python tools/bench_line.py "U config4 128/1024"
$B --workload config3 --blocksizes 512,512 | python tools/bench_line.py "U config3 512/512"
$B --workload config3 --blocksizes 512,4096 --packets-per-stream 512 --steps 30 | python tools/bench_line.py "U config3 512/4096"
