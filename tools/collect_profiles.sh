#!/bin/bash
# Run on the GPU box from the repository root: bench lines, rocprofv3 kernel statistics and PMC
# passes of the default workload into gpurun_out/prof/; tools/pmc_summary.py condenses the PMC
# passes.  Usage: bash tools/collect_profiles.sh <tag>
set -o pipefail
tag=${1:-r01}
out=gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
for c in c2 c1; do
  timeout -k 10 300 python3 bench.py --config $c > $out/${tag}_bench_$c.json 2> $out/bench_$c.err || exit 1
done
echo benches c2 c1 done
timeout -k 10 300 python3 bench.py --config c2 --sync-steps --no-cpu-baseline > $out/${tag}_bench_c2_sync_steps.json 2>> $out/bench_c2.err || exit 1
timeout -k 10 300 python3 bench.py --config c3 --steps 3 --warmup 1 > $out/${tag}_bench_c3_256seq.json 2> $out/bench_c3.err || exit 1
timeout -k 10 300 python3 bench.py --config c4 --steps 3 --warmup 1 > $out/${tag}_bench_c4_1gpu.json 2> $out/bench_c4.err || exit 1
timeout -k 10 300 python3 bench.py --config c3 --nseq2 4096 --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c3_4096seq.json 2>> $out/bench_c3.err || exit 1
echo benches done
# kernel statistics: the default command (consecutive candidate loops share the GPU inside its
# timed region) and the same with --sync-steps (every launch on its own: the duration the
# roofline block of the bench line quotes)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/stats.log 2>&1 || exit 1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats.csv
cp $(find $out/stats -name "*domain_stats.csv" | head -1) $out/${tag}_c2_domain_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_sync -- python3 bench.py --no-cpu-baseline --sync-steps > $out/stats_sync.log 2>&1 || exit 1
cp $(find $out/stats_sync -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats_sync_steps.csv
python3 tools/timeline.py $out/stats 60 95 > $out/${tag}_c2_timeline.txt
echo stats done
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i + 1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- python3 bench.py --no-cpu-baseline --sync-steps --steps 3 --warmup 1 > $out/pmc$i.log 2>&1 || exit 1
  echo pmc pass $i done
done
python3 tools/pmc_summary.py $out/pmc1 $out/pmc2 $out/pmc3 > $out/${tag}_c2_pmc.json || exit 1
echo collected
