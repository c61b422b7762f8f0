B="timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
for v in uo768; do
  cp build_ab/lib_$v.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
  echo "== $v"
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/pytest_$v.log 2>&1; echo "default rc=$? $(tail -1 gpurun_out/pytest_$v.log)"
  VSYN_U_MIXED=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > gpurun_out/pytestm_$v.log 2>&1; echo "umixed rc=$? $(tail -1 gpurun_out/pytestm_$v.log)"
  VSYN_U_MIXED=1 $B --workload config4 | python tools/bench_line.py "U config4 256/2048"
  $B --workload config3 --blocksizes 128,1024 | python tools/bench_line.py "U config3 128/1024"
  $B --workload config4 --blocksizes 128,1024 | python tools/bench_line.py "U config4 128/1024"
  $B --workload config3 --blocksizes 512,512 | python tools/bench_line.py "U config3 512/512"
done
