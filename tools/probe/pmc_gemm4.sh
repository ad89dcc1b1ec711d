# PMC passes over the stand-alone plane-GEMM harness (tools/probe/gemm4_bench.hip), one shape, both kernel generations.
# usage (GPU box): bash tools/probe/pmc_gemm4.sh <shape filter> <out tag> [binary]
SHAPE=${1:-NT_20480x3072x768}; TAG=${2:-pmc4}; BIN=${3:-build/probe/gemm4_bench}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_ACTIVE_INST_MISC" \
         "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/p$i -- $R/$BIN 3 0 $SHAPE > $R/gpurun_out/$TAG/p$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/$TAG/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_pl" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].split("(")[0][-24:], r["Counter_Name"])
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
names = sorted({k for k, _ in acc})
ctrs = []
for _, c in acc:
    if c not in ctrs: ctrs.append(c)
with open("gpurun_out/$TAG/summary.csv", "w") as o:
    o.write("counter," + ",".join(names) + "\n")
    for c in ctrs:
        o.write(c + "," + ",".join("%.0f" % (sum(acc[(n, c)]) / len(acc[(n, c)])) if (n, c) in acc else "" for n in names) + "\n")
print(open("gpurun_out/$TAG/summary.csv").read())
PY
