mkdir -p gpurun_out/r2j
B="timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
for v in u512 u768; do
  cp build_ab/lib_$v.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so
  echo "== $v"
  VSYN_U_MIXED=1 $B --workload config4 | python tools/bench_line.py "U config4 256/2048"
  $B --workload config3 --blocksizes 128,1024 | python tools/bench_line.py "U config3 128/1024"
  $B --workload config4 --blocksizes 128,1024 | python tools/bench_line.py "U config4 128/1024"
  $B --workload config3 --blocksizes 512,512 | python tools/bench_line.py "U config3 512/512"
done
VSYN_NO_U=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload config3 --blocksizes 128,1024 | python tools/bench_line.py "staged config3 128/1024"
VSYN_NO_U=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload config4 --blocksizes 128,1024 | python tools/bench_line.py "staged config4 128/1024"
