# A/B of library builds on one box: bash tools/probe/lib_ab.sh <rounds> lib1 lib2 ...
N=$1; shift
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for i in $(seq 1 $N); do for L in "$@"; do
  SEGMM_LIB=$L timeout -k 10 300 python bench.py $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$L', r['value'], r['ms_per_step'])"
done; done
