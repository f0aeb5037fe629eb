#!/bin/bash
# Run on the GPU box from the repository root: bench lines, rocprofv3 kernel statistics and
# PMC passes into gpurun_out/prof/ (copy what is to be kept into profiles/).  A step that fails ends the script.
set -o pipefail
tag=r02
out=gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > $out/$name 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }; echo "$name done"; }
run ${tag}_bench_c2.json python3 bench.py
run ${tag}_bench_c1.json python3 bench.py --config c1
run ${tag}_bench_c2_sync_steps.json python3 bench.py --sync-steps --no-cpu-baseline --no-scale
run ${tag}_bench_c3_256seq.json python3 bench.py --config c3 --nseq2 256
run ${tag}_bench_c3_4096seq.json python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline
run ${tag}_bench_c4_1gpu.json python3 bench.py --config c4
run ${tag}_bench_c2_valu_loop.json env PSA_ENUM_MFMA=0 python3 bench.py --no-cpu-baseline --no-scale
run ${tag}_valu_mfma_ceilings.txt tools/valu_bench10
run ${tag}_len_sweep.txt tools/len_sweep.sh
run ${tag}_partition_balance.txt python3 tools/partition_balance.py 8
# kernel statistics: the default command (consecutive candidate loops share the GPU inside its timed
# region) and the same with --sync-steps (every launch on its own: the duration the roofline block quotes)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/bench.py --no-cpu-baseline --no-scale --no-host-inclusive > $R/$out/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_sync -- python3 $R/bench.py --no-cpu-baseline --no-scale --no-host-inclusive --sync-steps > $R/$out/stats_sync.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_c4 -- python3 $R/bench.py --no-cpu-baseline --config c4 > $R/$out/stats_c4.log 2>&1 || exit 1
cd $R
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats.csv
cp $(find $out/stats_sync -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats_sync_steps.csv
cp $(find $out/stats_c4 -name "*kernel_stats.csv" | head -1) $out/${tag}_c4_kernel_stats.csv
python3 tools/timeline.py $out/stats 60 95 > $out/${tag}_c2_timeline.txt
echo stats done
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
if [ -z "$SKIP_PMC" ]; then
  # one run per counter list (FETCH_SIZE and WRITE_SIZE do not fit one pass), the program directly behind `--`
  tools/pmc_run.sh ${tag}_c2 "" "$A" "$B" "FETCH_SIZE" "WRITE_SIZE" || exit 1
  tools/pmc_run.sh ${tag}_c4 "--config c4" "$A" "$B" "FETCH_SIZE" "WRITE_SIZE" || exit 1
  tools/pmc_run.sh ${tag}_len256 "--config c3 --nseq2 512 --len-lo 256 --len-hi 256 --steps 2 --warmup 1" "$A" "$B" || exit 1
  tools/pmc_run.sh ${tag}_len4000 "--config c3 --nseq2 64 --len-lo 4000 --len-hi 4000 --steps 2 --warmup 1" "$A" "$B" || exit 1
fi
# the resolve chain alone (kernel statistics of resolve-only steps) and the chain micro-benchmark
tools/prof_resolve.sh ${tag} > $out/${tag}_resolve.log 2>&1 || exit 1
cp gpurun_out/${tag}_resolve_kernel_stats.csv $out/${tag}_resolve_kernel_stats.csv
run ${tag}_chain_costs.txt tools/chain_bench
echo collected
