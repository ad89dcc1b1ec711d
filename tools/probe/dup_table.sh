# Exposure of the row-kernel families by DUPLICATION (needs build/probe/libsegmm_skip.so): step time with every launch of a family
# enqueued twice minus the plain step = the family's exposed time.   bash tools/probe/dup_table.sh
X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg --steps 20 --warmup 5"
export SEGMM_LIB=$PWD/build/probe/libsegmm_skip.so
run() { env SEGMM_DUP=$1 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %10.1f /s  %.4f ms/step' % ('dup $1', d['value'], d['ms_per_step']))"; }
run none
for n in layernorm_fwd layernorm_bwd l1norm colsum splitk_reduce wsplit adamw loss attn_fwd attn_bwd layernorm_fwd,layernorm_bwd,l1norm,colsum,splitk_reduce; do run $n; done
run none
