mkdir -p gpurun_out/r6
for r in 1 2; do for G in 0 2 4; do echo "== v8 group $G"; ./build/probe/g4_n8g$G 20 8 NT_ | grep "NT_20480x3072\|NT_20480x2048\|NT_51200x1536\|NT_10240x3072"; done; done
X="--no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine --no-probe"
for r in 1 2 3; do for G in 0 2 4; do
  SEGMM_LIB=build/probe/libsegmm_n8g$G.so timeout -k 10 300 python bench.py --config 3 $X 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cfg3 n8g$G', r['value'], r['ms_per_step'])"
done; done
