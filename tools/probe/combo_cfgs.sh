X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for CFG in "4 --global-batch 256" "3"; do
 for i in 1 2; do
  for C in "8 8" "4 8" "4 4" "8 4"; do
    set -- $C
    env SEGMM_PL_VAR=$1 SEGMM_TN_VAR=$2 timeout -k 10 300 python bench.py --config $CFG --steps 20 --warmup 5 $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); pk={k['kernel'].split(' ')[0]:k for k in r['roofline']['per_kernel']}
print('cfg $CFG NT$1 TN$2', r['value'], 'ms', r['ms_per_step'], '|', ' '.join('%s %.1fus %.3f' % (n, k['avg_us'], k.get('frac', k.get('frac_of_f32_mfma_peak', 0))) for n,k in list(pk.items())[:4]))"
  done
 done
done
