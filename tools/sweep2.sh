#!/bin/bash
# Run on the GPU box from the repository root: bench.py under a list of tuning settings
# ("VAR=value VAR=value" per line on stdin; X=0 for the defaults), one summary line each.
# BENCH_ARGS holds extra bench arguments.  Stops at the first run that fails.
while read -r line; do
  [ -z "$line" ] && continue
  out=$(env $line timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-scale --no-host-inclusive ${BENCH_ARGS:-} 2>/dev/null | grep '^{') || { echo "$line: failed"; exit 1; }
  python3 - "$line" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d.get("roofline", {})
print("%-44s %.3e cand/s  %.4f ms/step  kernel %.4f ms  frac %.3f  clock %.0f MHz  cyc/tile %.1f  parity %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], r.get("kernel_ms", 0), r.get("frac", 0),
    r.get("clock_mhz_measured") or 0, r.get("cycles_per_tile_measured") or 0, d.get("parity")))
PY
done
