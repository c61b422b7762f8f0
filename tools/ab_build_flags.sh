# tools/ab_build_flags.sh — A/B of compile-time variants of the HIP library on the GPU box: rebuilds with each VSYN_HIPCC_EXTRA
# value and prints the bench line's value / kernel time. Run from the repo root (restores the default build at the end).
set -e
run() {
  echo "== VSYN_HIPCC_EXTRA='$1'"
  VSYN_HIPCC_EXTRA="$1" python -c "import __graft_entry__ as g; g.build_hip(force=True)" > /dev/null 2>&1
  for i in 1 2; do
    timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
}
for f in "$@"; do run "$f"; done
run ""
