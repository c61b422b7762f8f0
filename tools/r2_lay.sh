for f in tests/test_gpu_fullsize.py tests/test_gpu_host_decoder.py tests/test_gpu_multirank.py tests/test_gpu_parity.py tests/test_gpu_pcm_stage.py tests/test_gpu_vq.py; do
  timeout -k 10 300 python -m pytest $f -x -q > gpurun_out/pytest_lay_$(basename $f .py).log 2>&1; echo "$f rc=$? $(tail -1 gpurun_out/pytest_lay_$(basename $f .py).log)"
done
B="timeout -k 10 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10"
for geo in "1 65536" "2 32768" "8 8192" "16 4096" "64 1024" "256 256" "1024 64" "4096 16"; do
  set -- $geo
  $B --streams $1 --packets-per-stream $2 | python tools/bench_line.py "$1 x $2"
done
