#!/bin/bash
# A/B builds of the fused kernel geometry on the GPU box; prints kernel ms for each
cd $GRAFT_REPO_ROOT
for cfg in "8 4" "10 5" "5 5" "4 4"; do
  set -- $cfg
  VSYN_HIPCC_EXTRA="-DFUSED_WAVES=$1 -DFUSED_MIN_WAVES_PER_SIMD=$2" python -c "import __graft_entry__ as g; g.build_hip(force=True)" 2>&1 | grep -E "error" 
  for r in 0; do
    VSYN_RUN_LEN=$r timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves=$1 minwps=$2 R=$r', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['pcm_max_abs_err_vs_oracle'])"
  done
done
