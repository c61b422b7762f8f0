mkdir -p gpurun_out/r2d
cp build_ab/lib_k0.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
for R in 32 33 34 35 37 41 48; do
  echo "== VSYN_RUN_LEN=$R" >> gpurun_out/r2d/summary.txt
  for extra in "" "--no-overlap"; do
    VSYN_RUN_LEN=$R timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline $extra 2>> gpurun_out/r2d/err.log | python tools/bench_line.py "$extra" >> gpurun_out/r2d/summary.txt
  done
done
for args in "--packets-per-stream 1000" "--streams 60 --packets-per-stream 1000" "--streams 63 --packets-per-stream 1040"; do
  echo "== $args" >> gpurun_out/r2d/summary.txt
  for extra in "" "--no-overlap"; do
    timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline $args $extra 2>> gpurun_out/r2d/err.log | python tools/bench_line.py "$extra" >> gpurun_out/r2d/summary.txt
  done
done
cat gpurun_out/r2d/summary.txt
