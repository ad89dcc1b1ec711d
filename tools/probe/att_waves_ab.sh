# planes-in attention backward: waves per workgroup (knob ATT_WAVES; key tiles of a block in passes).  3 waves: 40.7 KB of LDS for the
# 100-key block too -> four workgroups per CU instead of three.   bash tools/probe/att_waves_ab.sh
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for W in 4 3 2 4 3; do echo "== ATT_WAVES=$W"; SEGMM_ATT_WAVES=$W python tools/attn_bench.py 20 2>/dev/null | grep "bwd planes-in\|bwd(fused)"; done
for r in 1 2 3; do for W in 4 3; do
  SEGMM_ATT_WAVES=$W timeout -k 10 300 python bench.py $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); a=[k for k in r['roofline']['per_kernel'] if k['kernel']=='attn_bwd4'][0]; print('ATT_WAVES=$W', r['value'], r['ms_per_step'], 'attn_bwd4', a['avg_us'])"
done; done
