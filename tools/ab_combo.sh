# In-step A/B of the plane-GEMM kernel generations on ONE box: every (PL_VAR, TN_VAR) combination in turn, N rounds.
# usage (GPU box): bash tools/ab_combo.sh [rounds] ["extra env"]
N=${1:-2}; X="$2"
for i in $(seq 1 $N); do
  for C in "8 8" "4 8" "4 4" "8 4"; do
    set -- $C
    env SEGMM_PL_VAR=$1 SEGMM_TN_VAR=$2 $X timeout -k 10 300 python bench.py --no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); pk={k['kernel'].split(' ')[0]:k for k in r['roofline']['per_kernel']}
print('NT$1 TN$2 $X', r['value'], 'ms', r['ms_per_step'], '|', ' '.join('%s %.1fus %.3f' % (n, k['avg_us'], k.get('frac', k.get('frac_of_f32_mfma_peak', 0))) for n,k in pk.items()))"
  done
done
