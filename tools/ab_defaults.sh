N=${1:-2}
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for CFG in "2" "4 --global-batch 256" "3"; do for i in $(seq 1 $N); do for C in "4 8" "8 88" "44 4"; do set -- $C
  env SEGMM_PL_VAR=$1 SEGMM_TN_VAR=$2 timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 5 $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('config $CFG PL_VAR=$1 TN_VAR=$2', r['value'], 'ms', r['ms_per_step'])"
done; done; done
