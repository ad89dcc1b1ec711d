# same-box A/B of the eager step, the eager device-state step and its hipGraph replay: interactions/s, ms/step, host enqueue ms
for ARGS in "--config 4 --global-batch 256" "--config 2" "--config 3"; do
  for MODE in "" "--device-state" "--graph" "" "--graph"; do
    timeout -k 10 300 python bench.py $ARGS $MODE --no-cpu-baseline --no-f32-engine --no-host-fed 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$ARGS', '$MODE' or 'eager', round(r['value']), r['ms_per_step'], 'host', r['host_enqueue_ms_per_step'], 'loss', r['config']['final_loss'])"
  done
done
