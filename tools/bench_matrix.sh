# Step-mode matrix on one GPU box: configs 2 / 3 / 4 (256 rows) / 5, eager vs recorded, plain and through the forced one-rank DP machinery.
#   bash tools/bench_matrix.sh <out dir under gpurun_out>
O=gpurun_out/${1:-matrix}; mkdir -p $O
X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg"
for cfg in "2" "3" "4 --global-batch 256" "5"; do
  tag=$(echo $cfg | tr -d ' -' ); 
  for mode in "--eager" ""; do
    m=${mode:---recorded}; m=${m#--}
    timeout -k 10 200 python bench.py --config $cfg $mode $X > $O/cfg${tag}_${m}.json 2>> $O/err.txt
    SEGMM_DP_FORCE=1 timeout -k 10 200 python bench.py --config $cfg $mode $X > $O/cfg${tag}_${m}_dpforce.json 2>> $O/err.txt
  done
done
timeout -k 10 200 python bench.py --input index $X > $O/cfg2_index_recorded.json 2>> $O/err.txt
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --windows 2 $X > $O/cfg2_gloo2_recorded.json 2>> $O/err.txt
python - "$O" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        r = json.load(open(f))
        print("%-36s %10.1f /s  %7.3f ms/step  host %6.3f ms  gemm frac %s  attn %s ms  overflows %s" % (
            os.path.basename(f)[:-5], r["value"], r["ms_per_step"], r["host_enqueue_ms_per_step"], r.get("roofline", {}).get("frac"),
            r.get("roofline_attention", {}).get("ms_per_step"), r["config"].get("delayed_scale_overflows")))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
