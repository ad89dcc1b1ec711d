N=${1:-2}
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for i in $(seq 1 $N); do for C in "8 88" "44 88" "44 4" "8 4"; do set -- $C
  env SEGMM_PL_VAR=$1 SEGMM_TN_VAR=$2 timeout -k 10 300 python bench.py $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('PL_VAR=$1 TN_VAR=$2', r['value'], 'ms', r['ms_per_step'], '|', ' '.join('%s %.1fus %.3f' % (k['kernel'].split(' ')[0], k['avg_us'], k.get('frac', k.get('frac_of_f32_mfma_peak', 0))) for k in r['roofline']['per_kernel'][:5]))"
done; done
