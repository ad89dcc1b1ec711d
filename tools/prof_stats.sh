# Kernel-trace + per-kernel stats of the bench on the GPU box:  bash tools/prof_stats.sh <tag> [bench args]
# -> gpurun_out/prof_<tag>/ (raw) and gpurun_out/prof_<tag>_kernel_stats.csv (the summary to commit under profiles/)
TAG=${1:-r2}; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --windows 2 --no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg "$@" > $R/gpurun_out/prof_bench_$TAG.log 2>&1
cd $R
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/prof_${TAG}_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mfma_rate_kernel" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-110s %8s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:45]:
    print("%-110s %8s %10.1f %10.1f %6.2f" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
steps = max([int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]] or [1])
print("total kernel time per step (%d steps profiled: warm-up, instrumented pass, recording, timed windows): %.3f ms" % (steps, tot / 1e6 / steps))
PY
