#!/bin/bash
# Run on the GPU box from the repository root: the candidate loop's efficiency against the Seq2
# length -- large batches of one length each (Seq1 = 10000), enumeration stage only.
for L in 256 512 768 1024 1536 2000 4000; do
  n=$((600000 / L)); [ $n -gt 512 ] && n=512
  out=$(timeout -k 10 150 python3 bench.py --config c3 --nseq2 $n --len-lo $L --len-hi $L --stages enumerate --no-cpu-baseline --no-scale --no-host-inclusive --steps 2 --warmup 1 ${BENCH_ARGS:-} 2>/dev/null | grep '^{') || { echo "L2=$L failed"; exit 1; }
  python3 - "$L" "$n" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
r = d.get("roofline", {})
print("L2=%-5s n=%-4s %.3e cand/s  %.3f ms/step  kernel %.3f ms  frac %.3f  clock %.0f MHz  cyc/tile %.1f  cand/tile %.0f  parity %s" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r.get("kernel_ms", 0), r.get("frac", 0),
    r.get("clock_mhz_measured") or 0, r.get("cycles_per_tile_measured") or 0, r.get("candidates_per_tile") or 0, d.get("parity")))
PY
done
