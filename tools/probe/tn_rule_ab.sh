# which weight-gradient launches should take gemm_pl_tn4?  (probe library built with -DSEGMM_TN4_RULE_PROBE)
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for i in 1 2; do
  for R in 0 1 2 3 4; do
    SEGMM_LIB=build/probe/libsegmm_tnrule.so SEGMM_TN4_RULE=$R timeout -k 10 300 python bench.py $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('rule $R', r['value'], r['ms_per_step'])"
  done
done
