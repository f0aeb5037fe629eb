#!/bin/bash
# Run on the GPU box from the repository root: bench lines, rocprofv3 kernel statistics and PMC passes into
# gpurun_out/prof/ (copy what is to be kept into profiles/).  A step that fails ends the script.
# usage: bash tools/collect_profiles_r03.sh [part]   part 1: bench lines and sweeps, 2: kernel statistics and
# timelines, 3: PMC passes, 4: rehearsals of bench.py with 2 and 4 ranks sharing the GPU
set -o pipefail
tag=r03
part=${1:-1}
out=gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 400 "$@" > $out/$name 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }; echo "$name done"; }
if [ $part = 1 ]; then
  run ${tag}_bench_c2.json python3 bench.py
  run ${tag}_bench_c1.json python3 bench.py --config c1
  run ${tag}_bench_c2_sync_steps.json python3 bench.py --sync-steps --no-cpu-baseline --no-scale --no-boundary
  run ${tag}_bench_c3_256seq.json python3 bench.py --config c3 --nseq2 256
  run ${tag}_bench_c3_4096seq.json python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline
  run ${tag}_bench_c4_1gpu.json python3 bench.py --config c4
  run ${tag}_len_sweep.txt tools/len_sweep.sh
  run ${tag}_oneshot_breakdown.txt python3 tools/oneshot_breakdown.py
  run ${tag}_enqueue_pace.txt python3 tools/enqueue_pace.py 40
fi
if [ $part = 2 ]; then
  # kernel statistics: the default command (consecutive candidate loops share the GPU inside its timed
  # region), the same with --sync-steps (every launch on its own: the duration the roofline block quotes),
  # the long pair, a list of 512 Seq2, the resolve chain alone, the block-wide pre-pass forced on a list
  cd /tmp
  B="--no-cpu-baseline --no-scale --no-host-inclusive --no-boundary --repeats 0"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/bench.py $B > $R/$out/stats.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_sync -- python3 $R/bench.py $B --sync-steps > $R/$out/stats_sync.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_c4 -- python3 $R/bench.py $B --config c4 > $R/$out/stats_c4.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_c3 -- python3 $R/bench.py $B --config c3 --nseq2 512 --steps 2 --warmup 1 > $R/$out/stats_c3.log 2>&1 || exit 1
  PSA_PREP_FORM=block timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_c3_block -- python3 $R/bench.py $B --config c3 --nseq2 512 --steps 2 --warmup 1 > $R/$out/stats_c3_block.log 2>&1 || exit 1
  cd $R
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats.csv
  cp $(find $out/stats_sync -name "*kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats_sync_steps.csv
  cp $(find $out/stats_c4 -name "*kernel_stats.csv" | head -1) $out/${tag}_c4_kernel_stats.csv
  cp $(find $out/stats_c3 -name "*kernel_stats.csv" | head -1) $out/${tag}_c3_512seq_kernel_stats.csv
  cp $(find $out/stats_c3_block -name "*kernel_stats.csv" | head -1) $out/${tag}_c3_512seq_kernel_stats_block_prepass.csv
  python3 tools/timeline.py $out/stats 60 95 > $out/${tag}_c2_timeline_pipelined.txt
  python3 tools/timeline.py $out/stats_sync 48 0 > $out/${tag}_c2_timeline.txt
  tools/prof_resolve.sh ${tag} > $out/${tag}_resolve.log 2>&1 || exit 1
  cp gpurun_out/${tag}_resolve_kernel_stats.csv $out/${tag}_resolve_kernel_stats.csv
  echo stats done
fi
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
if [ $part = 3 ]; then
  # one run per counter list (FETCH_SIZE and WRITE_SIZE do not fit one pass), the program directly behind `--`
  tools/pmc_run.sh ${tag}_c2 "--no-boundary --repeats 0" "$A" "$B" "FETCH_SIZE" "WRITE_SIZE" || exit 1
  tools/pmc_run.sh ${tag}_c4 "--no-boundary --config c4 --steps 2 --warmup 1 --prewarm-s 0" "$A" "$B" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" || exit 1
  tools/pmc_run.sh ${tag}_c3 "--no-boundary --config c3 --nseq2 512 --steps 2 --warmup 1 --prewarm-s 0" "$A" "$B" "FETCH_SIZE" "WRITE_SIZE" || exit 1
  PSA_ENUM_BAND=32 tools/pmc_run.sh ${tag}_c4_banded "--no-boundary --config c4 --steps 2 --warmup 1 --prewarm-s 0" "$A" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" || exit 1
  PSA_PREP_FORM=block tools/pmc_run.sh ${tag}_c3_block_prepass "--no-boundary --config c3 --nseq2 512 --steps 2 --warmup 1 --prewarm-s 0" "$B" || exit 1
  echo pmc done
fi
if [ $part = 4 ]; then
  for n in 2 4; do
    timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --backend gloo --share-gpu --steps 5 --warmup 2 --nseq2 256 --no-cpu-baseline --no-boundary > $out/${tag}_rehearsal_n$n.json 2> $out/rehearsal_n$n.err || { echo "rehearsal $n failed"; tail -5 $out/rehearsal_n$n.err; exit 1; }
    echo "rehearsal n=$n done"
  done
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --force-dist --backend nccl --config c4 --steps 2 --no-cpu-baseline --no-boundary > $out/${tag}_nccl_one_rank.json 2> $out/nccl_one_rank.err || { tail -5 $out/nccl_one_rank.err; exit 1; }
  echo rehearsals done
fi
echo collected part $part
