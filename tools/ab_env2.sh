# same-box A/B of one environment knob on the default bench: bash tools/ab_env2.sh NAME A_VALUE B_VALUE [bench args]
N=$1; A=$2; B=$3; shift 3
X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg"
for v in $A $B $A $B; do
  env $N=$v python bench.py --steps 20 --warmup 5 $X "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v', d['value'], d['ms_per_step'], 'gemm', d['roofline']['frac'], 'attn', d['roofline_attention']['ms_per_step'], 'loss', d['config']['final_loss'])"
done
