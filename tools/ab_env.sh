#!/bin/bash
# A/B of one environment switch on the headline workload, alternating in one gpurun call:  tools/ab_env.sh VAR value_a value_b [extra bench flags]
VAR=$1; A=$2; B=$3; shift 3
for v in $A $B $A $B; do
  env $VAR=$v python3 bench.py --secondary off --cpu-queries 0 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'], d['parity']['idx_equal'], d['phases_ms_per_step_rank0'])"
done
