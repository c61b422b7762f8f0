# tools/final_round.sh <tag> — the round-final measurement set on the GPU box (run from the repo root): PMC passes + kernel stats of the
# default bench, bench lines of every workload, kernel stats over 200 steps and without overlap. Raw output under gpurun_out/<tag>*; copy
# the summaries into profiles/.
TAG=${1:-r03_z}
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG > gpurun_out/$TAG.log 2>&1
echo "profile_round done"
BENCH_ARGS="--workload config3_vq" bash tools/profile_round.sh ${TAG}_vq > gpurun_out/${TAG}_vq.log 2>&1
echo "vq profile_round done"
B="timeout -k 10 400 python bench.py"
( $B --steps 20 --warmup 5 &&
  $B --steps 20 --warmup 5 --no-cpu-baseline &&
  $B --steps 200 --warmup 10 --no-cpu-baseline &&
  $B --steps 200 --warmup 10 --no-cpu-baseline --no-overlap &&
  $B --no-cpu-baseline --workload config4 &&
  $B --workload config2 &&
  $B --no-cpu-baseline --workload config3_vq &&
  $B --no-cpu-baseline --workload config3_vq --vq-books fixture &&
  $B --no-cpu-baseline --pcm-s16 &&
  $B --no-cpu-baseline --feature-taps &&
  $B --no-cpu-baseline --steps 100 --feature-taps --blocksizes 128,1024 &&
  $B --steps 200 --warmup 10 --no-cpu-baseline --hidden-pre-kernels &&
  $B --no-cpu-baseline --workload config4 --hidden-pre-kernels &&
  $B --no-cpu-baseline --steps 100 --workload config3 --blocksizes 128,1024 &&
  $B --no-cpu-baseline --steps 100 --workload config4 --blocksizes 128,1024 &&
  $B --no-cpu-baseline --steps 100 --workload config3 --blocksizes 512,512 &&
  $B --no-cpu-baseline --steps 100 --workload config3 --blocksizes 1024,1024 &&
  $B --no-cpu-baseline --steps 50 --workload config3 --blocksizes 512,4096 --packets-per-stream 512 &&
  $B --no-cpu-baseline --steps 50 --workload config4 --blocksizes 512,4096 --packets-per-stream 512 &&
  $B --no-cpu-baseline --steps 20 --workload config3 --blocksizes 1024,8192 --packets-per-stream 256 &&
  $B --no-cpu-baseline --steps 100 --streams 1 --packets-per-stream 65536 &&
  $B --no-cpu-baseline --steps 100 --streams 4096 --packets-per-stream 16 &&
  $B --workload config5 --steps 3 --warmup 1 ) > gpurun_out/${TAG}_bench.jsonl 2> gpurun_out/${TAG}_bench.err
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_p200 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_p200.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_noov -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_noov.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_drv -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_drv.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_u1024 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --blocksizes 128,1024 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_u1024.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/${TAG}_p200 gpurun_out/${TAG}_noov gpurun_out/${TAG}_u1024 gpurun_out/${TAG}_drv -name "*kernel_stats.csv" | while read f; do cp $f gpurun_out/$(echo $f | cut -d/ -f2)_kernel_stats.csv; done
rm -rf gpurun_out/${TAG}_p200 gpurun_out/${TAG}_noov gpurun_out/${TAG}_u1024 gpurun_out/${TAG}_drv gpurun_out/$TAG/pmc*/ gpurun_out/$TAG/stats/ gpurun_out/${TAG}_vq/pmc*/ gpurun_out/${TAG}_vq/stats/
ls gpurun_out | grep $TAG
