# tools/vq_ab.sh <tag> <variant...> — A/B of prebuilt library variants (build_ab/lib_<variant>.so) on the residue VQ workloads:
# VQ parity tests, then bench.py --workload config3_vq with the synthetic and the fixture's codebooks. Results: gpurun_out/<tag>/vq.txt
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  cp build_ab/lib_$v.so parseoggvorbis_amd/csrc/libvorbis_synth_hip.so || exit 1
  echo "== $v" >> $OUT/vq.txt
  if [ "${AB_PYTEST:-1}" = "1" ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_vq.py tests/test_gpu_host_decoder.py -x -q > $OUT/pytest_$v.log 2>&1
    echo "  pytest rc=$? $(tail -1 $OUT/pytest_$v.log)" >> $OUT/vq.txt
  fi
  for w in synthetic fixture; do
    timeout -k 10 300 python bench.py --workload config3_vq --vq-books $w --steps 50 --warmup 5 --no-cpu-baseline ${AB_BENCH_EXTRA} 2>> $OUT/err_$v.log | python tools/bench_line.py "$w" >> $OUT/vq.txt
  done
done
cat $OUT/vq.txt
