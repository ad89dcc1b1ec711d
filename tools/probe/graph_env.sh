# hipGraph replay of the step under the HIP runtime's graph knobs (process environment only): interactions/s, ms/step, host ms
ARGS="${ARGS:---config 4 --global-batch 256}"
for E in "$@"; do
  env $E timeout -k 10 300 python bench.py $ARGS --graph --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$ARGS', '$E', round(r['value']), r['ms_per_step'], 'host', r['host_enqueue_ms_per_step'])"
done
