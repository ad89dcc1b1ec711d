# Exposure table on the GPU box (needs build/probe/libsegmm_skip.so: tools/probe/build_skip_probe.py).  bash tools/probe/skip_table.sh [bench args]
X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --steps 20 --warmup 5"
export SEGMM_LIB=$PWD/build/probe/libsegmm_skip.so
run() { env SEGMM_SKIP=$1 python bench.py $X "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %10.1f /s  %.4f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
run none "$@"
for n in gemm_nt gemm_tn splitk_reduce attn_fwd attn_bwd layernorm_fwd layernorm_bwd l1norm colsum wsplit adamw loss fixup,scales \
         gemm_nt,gemm_tn,splitk_reduce attn_fwd,attn_bwd layernorm_fwd,layernorm_bwd,l1norm,colsum,wsplit,adamw,loss,fixup,scales; do run $n "$@"; done
run none "$@"
