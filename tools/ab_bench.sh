# A/B of bench.py on ONE box: alternates two environments, prints interactions/s, GEMM TF and GEMM-busy ms per run.
# usage (GPU box): bash tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [repeats]
A="$1"; B="$2"; N=${3:-3}
for i in $(seq 1 $N); do
  for E in "$A" "$B"; do
    env $E timeout -k 10 300 python bench.py --no-cpu-baseline --no-sustained --no-index-leg --no-host-fed --no-f32-engine 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', r['value'], r['roofline']['achieved'], r['roofline']['gemm_busy_ms_per_step'], r['roofline_attention']['ms_per_step'])"
  done
done
