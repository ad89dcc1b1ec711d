timeout -k 10 600 python -m pytest tests/test_planes_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_bench.sh "SEGMM_TN_VAR=88" "SEGMM_TN_VAR=8" 3
