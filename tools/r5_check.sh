# round-5 checkpoint on the GPU box: bash tools/r5_check.sh <tag> [tests|notests]
TAG=${1:-x}; O=gpurun_out/r5; mkdir -p $O
X="--no-cpu-baseline --no-f32-engine --no-host-fed"
if [ "${2:-tests}" = "tests" ]; then
  python -m pytest tests -x -q -m gpu > $O/gpu_tests_$TAG.log 2>&1; tail -12 $O/gpu_tests_$TAG.log
fi
python bench.py --steps 20 --warmup 5 $X > $O/bench_${TAG}_a.json 2> $O/bench_${TAG}_a.err
SEGMM_ATT_PL=0 python bench.py --steps 20 --warmup 5 $X > $O/bench_${TAG}_b.json 2>/dev/null
python bench.py --steps 20 --warmup 5 $X > $O/bench_${TAG}_a2.json 2>/dev/null
SEGMM_ATT_PL=0 python bench.py --steps 20 --warmup 5 $X > $O/bench_${TAG}_b2.json 2>/dev/null
for f in a b a2 b2; do python - $O/bench_${TAG}_$f.json $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], d["value"], d["ms_per_step"], "gemm", d["roofline"]["frac"], "attn ms", d["roofline_attention"]["ms_per_step"], d["roofline_attention"]["frac"], "ovf", d["config"]["delayed_scale_overflows"], "loss", d["config"]["final_loss"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
tail -5 $O/bench_${TAG}_a.err
