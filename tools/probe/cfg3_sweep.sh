# config-3 A/B runs on one box: each line = env, interactions/s, ms/step, GEMM TF, GEMM busy ms, attention ms, host enqueue ms
for E in "$@"; do
  env $E timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-f32-engine --no-host-fed 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$E', round(r['value']), r['ms_per_step'], r['roofline']['achieved'], r['roofline']['gemm_busy_ms_per_step'], r['roofline_attention']['ms_per_step'], r['host_enqueue_ms_per_step'])"
done
