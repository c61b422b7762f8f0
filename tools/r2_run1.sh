set -x
mkdir -p gpurun_out/r2a
./tools/fft_xchg_bench 200 > gpurun_out/r2a/fftx.log 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_20_5.json 2> gpurun_out/r2a/bench_20_5.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> gpurun_out/r2a/bench_20_5.json 2>> gpurun_out/r2a/bench_20_5.err
timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r2a/bench_200.json 2> gpurun_out/r2a/bench_200.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/trace -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2a/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'vsyn_fused_kernel' in r['Kernel_Name']]
t0=int(rows[0]['Start_Timestamp'])
out=open('gpurun_out/r2a/fused_series.txt','w')
for i,r in enumerate(rows):
    out.write("%d start_us %.1f dur_us %.1f\n"%(i,(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
rm -rf gpurun_out/r2a/trace
