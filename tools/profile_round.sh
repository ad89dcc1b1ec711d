# Round checkpoint on the GPU box: kernel trace of the bench, un-profiled bench lines for every engine, smoke.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench_$TAG.log 2>&1
cd $R
timeout 400 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_${TAG}_f16x3.json
SEGMM_GEMM=bf16x6 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${TAG}_bf16x6.json
SEGMM_GEMM=f32 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${TAG}_f32.json
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
