X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --steps 20 --warmup 5"
export SEGMM_LIB=$PWD/build/probe/libsegmm_skip.so
run() { env SEGMM_SKIP=$1 python bench.py $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %10.1f /s  %.4f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
run none
run splitk_reduce,colsum,wsplit,adamw,loss,fixup,scales
run fixup
run scales
run colsum,splitk_reduce
run none
