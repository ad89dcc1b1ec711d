# Round checkpoint on the GPU box: every number the docs quote, in one call.   bash tools/round_profiles.sh <tag> <git sha>
TAG=${1:-r5}; export GIT_SHA=${2:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
X="--no-cpu-baseline --no-f32-engine --no-host-fed --no-probe --no-sustained --no-index-leg"
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --eager --steps 20 --warmup 5 $X > $O/bench_n1_eager.json 2>/dev/null
SEGMM_SCALING=exact python bench.py --steps 20 --warmup 5 $X > $O/bench_n1_exact_scaling.json 2>/dev/null
python bench.py --config 3 --steps 20 --warmup 5 > $O/bench_cfg3.json 2>/dev/null
python bench.py --config 5 --steps 20 --warmup 5 $X > $O/bench_cfg5.json 2>/dev/null
python bench.py --config 4 --gpus 1 --global-batch 256 --steps 20 --warmup 5 $X > $O/bench_cfg4_256rows.json 2>/dev/null
for M in "--eager" "--device-state" ""; do python bench.py --config 4 --gpus 1 --global-batch 256 --steps 20 --warmup 5 $M $X 2>/dev/null | tail -1; done > $O/bench_cfg4_256rows_step_modes.jsonl
python bench.py --input index --steps 20 --warmup 5 $X > $O/bench_index_input.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --windows 2 --no-cpu-baseline --no-probe > $O/bench_gloo2_one_gpu.json 2>/dev/null
SEGMM_DP_FORCE=1 python bench.py --steps 20 --warmup 5 $X > $O/bench_dp_forced_one_rank.json 2>/dev/null
bash tools/bench_matrix.sh ${TAG}_matrix > $O/step_mode_matrix.txt 2>&1
bash tools/prof_stats.sh ${TAG}_cfg2 > $O/prof_cfg2_summary.txt 2>&1
cp gpurun_out/prof_${TAG}_cfg2_kernel_stats.csv $O/bench_cfg2_kernel_stats.csv
python tools/timeline.py gpurun_out/prof_${TAG}_cfg2 --full > $O/step_timeline_cfg2.txt 2>&1
bash tools/prof_stats.sh ${TAG}_cfg3 --config 3 > $O/prof_cfg3_summary.txt 2>&1
cp gpurun_out/prof_${TAG}_cfg3_kernel_stats.csv $O/bench_cfg3_kernel_stats.csv
python tools/timeline.py gpurun_out/prof_${TAG}_cfg3 --full > $O/step_timeline_cfg3.txt 2>&1
bash tools/traffic_pass.sh $TAG > $O/traffic.log 2>&1
cp gpurun_out/traffic_$TAG/hbm_traffic.json $O/hbm_traffic.json; cp gpurun_out/traffic_$TAG/summary.csv $O/hbm_traffic_by_kernel.csv
bash tools/pmc_run.sh ${TAG}_pmc_nt gemm_pl tools/gemm_p_one.py nt 20480 3072 768 > /dev/null 2>&1; cp gpurun_out/${TAG}_pmc_nt/summary.csv $O/gemm_pl_nt_20480x3072x768_pmc.csv
bash tools/pmc_run.sh ${TAG}_pmc_tn gemm_pl tools/gemm_p_one.py tn 3072 768 20480 > /dev/null 2>&1; cp gpurun_out/${TAG}_pmc_tn/summary.csv $O/gemm_pl_tn_3072x768x20480_pmc.csv
bash tools/pmc_run.sh ${TAG}_pmc_attn attn_ tools/attn_bench.py 3 > /dev/null 2>&1; grep -v "dq_kernel\|dkv_kernel\|D_kernel" gpurun_out/${TAG}_pmc_attn/summary.csv > $O/attention_pmc.csv
python tools/attn_bench.py 20 > $O/attention_standalone.txt 2>&1; LQ=100 python tools/attn_bench.py 20 >> $O/attention_standalone.txt 2>&1; python tools/attn_bench_small.py >> $O/attention_standalone.txt 2>&1
python tools/gemm_p_check.py > $O/gemm_p_standalone.txt 2>&1
ls -la $O
python tools/hipblaslt_yardstick.py 20 > $O/hipblaslt_yardstick.txt 2>&1
python tools/gemm_out_variants.py 20 > $O/gemm_out_variants.txt 2>&1
SEGMM_ATT_PL=0 python bench.py --steps 20 --warmup 5 $X > $O/bench_n1_att_pl0.json 2>/dev/null
SEGMM_EU_PLANES_ONLY=0 python bench.py --steps 20 --warmup 5 $X > $O/bench_n1_eu_fp32.json 2>/dev/null
SEGMM_ATT_REPAIR_WALK=0 python bench.py --steps 20 --warmup 5 $X > $O/bench_n1_repair_walk0.json 2>/dev/null
python bench.py --steps 20 --warmup 5 $X > $O/bench_n1_again.json 2>/dev/null
python tools/tn_split_table.py 2>&1 | grep -v amdgpu.ids > $O/tn_split_table.txt
bash tools/pmc_bin.sh ${TAG}_pmc_fwdpl attn_fwd_pl tools/probe/attn_pl_bench 3 > /dev/null 2>&1; cp gpurun_out/${TAG}_pmc_fwdpl/summary.csv $O/attention_fwd_pl_pmc.csv
ls -la $O
