"""stdin: bench.py output; prints one short line per JSON record (used by tools/ab_variants.sh)."""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    err = d.get("pcm_max_abs_err_vs_oracle")
    print("  [%s] %.1f M/s  step %.4f ms  kernel %s ms  frac %s  err %s" % (
        label, d["value"] / 1e6, d["ms_per_step"], r.get("kernel_ms"), r.get("frac"), "%.2e" % err if err is not None else None))
