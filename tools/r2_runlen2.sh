for R in 32 22 16 11 8; do
  for extra in "" "--no-overlap"; do
    VSYN_UNWRAP_KERNEL=1 VSYN_RUN_LEN=$R timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline $extra | python tools/bench_line.py "R=$R $extra"
  done
done
