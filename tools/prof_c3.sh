#!/bin/bash
# Run on the GPU box from the repository root: rocprofv3 kernel statistics of one workload
# (arguments after the tag go to bench.py); the csv lands in gpurun_out/prof/<tag>_kernel_stats.csv.
tag=$1; shift
out=gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale --no-host-inclusive "$@" > $GRAFT_REPO_ROOT/$out/$tag.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
cp $(find $out/$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
cat $out/${tag}_kernel_stats.csv | cut -c1-200
