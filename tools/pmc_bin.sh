# PMC passes over a native binary (separate rocprofv3 runs, kernel trace only).
# usage (on the GPU box): bash tools/pmc_bin.sh <out tag> <kernel-name substring> <binary> [args...]
TAG=$1; FILT=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
BIN=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_ACTIVE_INST_MISC" \
         "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/p$i -- $BIN "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1
done
cd $R
python3 - "$TAG" "$FILT" <<'PY'
import csv, glob, collections, sys
tag, filt = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
dur = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/%s/p*/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        if filt not in r["Kernel_Name"]:
            continue
        acc.setdefault((r["Kernel_Name"][:60], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for f in sorted(glob.glob("gpurun_out/%s/p1/**/*kernel_trace.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            dur.setdefault(r["Kernel_Name"][:60], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open("gpurun_out/%s/summary.csv" % tag, "w") as o:
    o.write("kernel,counter,dispatches,mean_value_per_dispatch\n")
    for k, v in dur.items():
        o.write("%s,DURATION_NS(profiled pass 1),%d,%.1f\n" % (k.replace(",", ";"), len(v), sum(v) / len(v)))
    for (k, c), v in acc.items():
        o.write("%s,%s,%d,%.1f\n" % (k.replace(",", ";"), c, len(v), sum(v) / len(v)))
print(open("gpurun_out/%s/summary.csv" % tag).read())
PY
