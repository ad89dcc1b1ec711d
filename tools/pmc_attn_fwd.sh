# PMC passes over the attention micro-benchmark for the forward kernels:  bash tools/pmc_attn_fwd.sh <tag>   (env passes through)
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/p$i -- python3 $R/tools/attn_bench.py 3 > $R/gpurun_out/$TAG/p$i.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/%s/p*/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" not in r["Kernel_Name"]:
            continue
        acc.setdefault((r["Kernel_Name"][:48], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in acc.items():
    print("%s,%s,%d,%.1f" % (k.replace(",", ";"), c, len(v), sum(v) / len(v)))
PY
