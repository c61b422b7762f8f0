# tools/ab_variants.sh <tag> <variant...> — A/B of prebuilt library variants (build_ab/lib_<variant>.so) on the GPU box: quick
# parity check, then the default bench (overlapped) and --no-overlap, 200 steps each. Results under gpurun_out/<tag>/.
# AB_EXTRA: further bench argument sets (one word each, e.g. "--workload=config4"); AB_PYTEST=0 skips the parity check.
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  cp build_ab/lib_$v.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so || exit 1
  echo "== $v" >> $OUT/summary.txt
  if [ "${AB_PYTEST:-1}" = "1" ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/pytest_$v.log 2>&1
    echo "  pytest rc=$? $(tail -1 $OUT/pytest_$v.log)" >> $OUT/summary.txt
  fi
  for extra in "" "--no-overlap" ${AB_EXTRA}; do
    timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline $extra > $OUT/b.json 2>> $OUT/err_$v.log
    python tools/bench_line.py "$extra" < $OUT/b.json >> $OUT/summary.txt
  done
done
rm -f $OUT/b.json
cat $OUT/summary.txt
