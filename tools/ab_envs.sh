# Same-box sweep of environment settings on the bench:  bash tools/ab_envs.sh <out tag> "<bench args>" "VAR=a" "VAR=b" ...   (two rounds, alternating)
TAG=$1; ARGS=$2; shift 2
X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg"
O=gpurun_out/$TAG.jsonl; : > $O
for i in 1 2; do
  for E in "$@"; do
    env $E timeout -k 10 200 python bench.py --steps 20 --warmup 5 $X $ARGS 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$E" <<'PY' >> $O
import json, sys
try:
    r = json.load(open("/tmp/ab_line.json"))
    print("%-34s value %9.1f (min %9.1f max %9.1f) ms/step %.4f host %.3f gemm frac %s attn %s" % (sys.argv[1], r["value"], r["value_min"], r["value_max"], r["ms_per_step"],
          r["host_enqueue_ms_per_step"], r.get("roofline", {}).get("frac"), r.get("roofline_attention", {}).get("ms_per_step")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
cat $O
