# round-5 second checkpoint: the repair launch of the planes-in backward walks the heads.  bash tools/r5_step2.sh
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_planes_gpu.py -x -q -m gpu -k "repair or input_planes" > $O/t_repair.log 2>&1 || { tail -30 $O/t_repair.log; exit 1; }
tail -3 $O/t_repair.log
python tools/attn_bench.py > $O/attn_bench2.txt 2>&1; grep -E "planes-in|fused" $O/attn_bench2.txt
bash tools/ab_env2.sh SEGMM_ATT_REPAIR_WALK 0 512 | tee $O/ab_repair_walk.txt
