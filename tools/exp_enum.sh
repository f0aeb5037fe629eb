# enumeration-kernel time of the default workload under a few tuning settings
for cfg in "PSA_ENUM_BLOCKS_PER_CU=3" "PSA_ENUM_BLOCKS_PER_CU=4" "PSA_ENUM_BLOCKS_PER_CU=3 PSA_ENUM_NS=2"; do
  for st in enumerate both; do
    env $cfg timeout -k 10 100 python -u bench.py --no-cpu-baseline --stages $st 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', '$st', 'step_ms=%.4f kernel_ms=%.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
