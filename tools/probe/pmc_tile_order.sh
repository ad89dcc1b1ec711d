# HBM-side traffic of gemm_pl_nt4 under the tile orders of P4_NGROUP (probe build: dbg bits 20-23 = group width + 1), one shape.
# usage (GPU box): bash tools/probe/pmc_tile_order.sh [shape] [binary built with -DSEGMM_GEMM_PROBE] [group widths, default "0 3 4 6"]
SHAPE=${1:-NT_20480x3072x768}; BIN=${2:-build/probe/g4_p}; GS=${3:-0 3 4 6}
R=$GRAFT_REPO_ROOT; TAG=r6/tile_order_$SHAPE
rm -rf $R/gpurun_out/$TAG; mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for G in $GS; do
  D=$(( (G + 1) << 20 ))
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | cut -c1-5)
    timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/g${G}_$N -- $R/$BIN 3 4 $SHAPE $D > $R/gpurun_out/$TAG/g${G}_$N.log 2>&1 || exit 1
  done
done
cd $R
python3 - <<PY
import csv, glob, collections
print("shape $SHAPE: per launch of gemm_pl_nt4 (FETCH_SIZE KiB x 2 on gfx950, WRITE_SIZE KiB)")
for G in [int(g) for g in "$GS".split()]:
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/$TAG/g%d_*/**/*counter_collection.csv" % G, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_pl_nt4" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    rd, wr = 2 * 1024 * m.get("FETCH_SIZE", 0), 1024 * m.get("WRITE_SIZE", 0)
    h, mi = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print("group width %d: read %7.1f MB  written %7.1f MB  L2 hit %5.1f %%" % (G, rd / 1e6, wr / 1e6, 100 * h / max(h + mi, 1)))
PY
