#!/bin/bash
# Run on the GPU box from the repository root: the default bench under a list of tuning
# settings ("VAR=value VAR=value" per line on stdin), one summary line each.
while read -r line; do
  [ -z "$line" ] && continue
  out=$(env $line timeout -k 10 120 python3 bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null) || { echo "$line: failed"; exit 1; }
  python3 - "$line" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d.get("roofline", {})
print("%-44s %.3e cand/s  %.4f ms/step  kernel %.4f ms (pipelined %s)  frac %.3f" % (
    sys.argv[1], d["value"], d["ms_per_step"], r.get("kernel_ms", 0),
    "%.4f" % r["kernel_ms_pipelined"] if r.get("kernel_ms_pipelined") else "-", r.get("frac", 0)))
PY
done
