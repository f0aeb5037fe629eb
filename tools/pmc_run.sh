#!/bin/bash
# Run on the GPU box from the repository root: rocprofv3 --pmc passes of one bench.py workload, one
# pass per quoted counter list, each a run of its own; tools/pmc_summary.py condenses them into
# gpurun_out/prof/<tag>_pmc.json.  Usage: tools/pmc_run.sh <tag> "<bench args>" "<counters>" ["<counters>" ...]
tag=$1; bargs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
cd /tmp || exit 1
i=0; dirs=""
for pmc in "$@"; do
  i=$((i + 1))
  rm -rf $out/${tag}_pmc$i
  timeout -s KILL 150 rocprofv3 --pmc $pmc --output-format csv -d $out/${tag}_pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale --no-host-inclusive $bargs > $out/${tag}_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/${tag}_pmc$i.log; exit 1; }
  dirs="$dirs $out/${tag}_pmc$i"
  echo "pmc pass $i done"
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py --command "rocprofv3 --pmc <one counter list per run> --output-format csv -- python3 bench.py --no-cpu-baseline --no-scale --no-host-inclusive $bargs; counter lists: $*" --workload "$bargs" $dirs > $out/${tag}_pmc.json || exit 1
python3 - $out/${tag}_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, c in d["kernels"].items():
    if "enum_wave" in k:
        print(k[:60], {n: round(v) for n, v in c.items()})
PY
