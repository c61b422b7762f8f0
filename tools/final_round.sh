set -e
mkdir -p gpurun_out
bash tools/profile_round.sh r01_k > gpurun_out/r01_k.log 2>&1
( timeout -k 10 400 python bench.py ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config4 ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config2 ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config3_vq ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config3_vq --vq-books fixture ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --pcm-s16 ) > gpurun_out/r01_k_bench.jsonl 2> gpurun_out/r01_k_bench.err
cd /tmp && export TMPDIR=/tmp
# kernel stats over as many launches as bench.py's default run (the 20-step run of profile_round.sh is dominated by cold launches)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_k200 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01_k200.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_k_noov -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/r01_k_noov.log 2>&1
