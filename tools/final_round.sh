# tools/final_round.sh — the round-final measurement set on the GPU box (run from the repo root): PMC passes + kernel stats of the default
# bench, bench lines of every workload, kernel stats over 200 steps and without overlap. Raw output under gpurun_out/<tag>*; copy the
# summaries into profiles/.
set -e
mkdir -p gpurun_out
bash tools/profile_round.sh r01_p > gpurun_out/r01_p.log 2>&1
( timeout -k 10 400 python bench.py ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config4 ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config2 ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config3_vq ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload config3_vq --vq-books fixture ;
  timeout -k 10 300 python bench.py --no-cpu-baseline --pcm-s16 ;
  timeout -k 10 300 python bench.py --workload config5 --steps 3 --warmup 1 ) > gpurun_out/r01_p_bench.jsonl 2> gpurun_out/r01_p_bench.err
cd /tmp && export TMPDIR=/tmp
# kernel stats over as many launches as bench.py's default run (the 20-step run of profile_round.sh is dominated by cold launches)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_p200 -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01_p200.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01_p_noov -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/r01_p_noov.log 2>&1
