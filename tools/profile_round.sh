# tools/profile_round.sh <tag> — rocprofv3 passes over `bench.py` on the GPU box, raw output under gpurun_out/<tag>/,
# summaries (what gets committed under profiles/) written by tools/pmc_summarize.py.  Run from the repo root.
# BENCH_ARGS (environment) is appended to the bench command, e.g. "--workload config3_vq".
# Kernel timing and every PMC group are separate runs (gpurun refuses --pmc combined with tracing).
set -e
TAG=${1:-prof}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $BENCH > $OUT/stats.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o run -- $BENCH > $OUT/pmc$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summarize.py $OUT > $OUT/summary.json
cat $OUT/summary.json
