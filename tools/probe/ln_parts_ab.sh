# LayerNorm backward: workgroups per launch (knob LN_BWD_PARTS; three 4-wave workgroups are resident per CU at d = 768 -> 768 slots).
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for P in 1024 768 1536 2048; do echo "== LN_BWD_PARTS=$P"; SEGMM_LN_BWD_PARTS=$P python tools/rowops_bench.py 20 2>/dev/null | grep -i "bwd"; done
for r in 1 2 3; do for P in 1024 768 1536 2048; do
  SEGMM_LN_BWD_PARTS=$P timeout -k 10 300 python bench.py $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('LN_BWD_PARTS=$P', r['value'], r['ms_per_step'])"
done; done
