#!/bin/bash
# Run on the GPU box from the repository root: per-kernel durations of the resolve chain alone
# (4 x Seq2 = 2000 against Seq1 = 3000, one run at a time) and the host-inclusive figures.
# usage: tools/prof_resolve.sh <tag>
set -o pipefail
tag=${1:-res}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --stages resolve --sync-steps --steps 300 --no-cpu-baseline --no-scale --no-host-inclusive --no-boundary --repeats 0 > $out/stats.log 2>&1 || exit 1
cd $R
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_resolve_kernel_stats.csv
cat $f
timeout -k 10 200 python3 bench.py --config c1 --no-cpu-baseline --no-scale > gpurun_out/${tag}_bench_c1.json || exit 1
python3 - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_bench_c1.json").read().strip().splitlines()[-1])
print("c1 ms/step", d["ms_per_step"], "parity", d["parity"], "host_inclusive", json.dumps(d.get("host_inclusive")))
PY
