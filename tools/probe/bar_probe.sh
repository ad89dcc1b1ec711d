# K-loop barrier / DMA-wait timing probes (wrong results by design): builds on the build host, runs on the GPU box.
#   build:  bash tools/probe/bar_probe.sh build      run (GPU box):  bash tools/probe/bar_probe.sh run
R=$(cd $(dirname $0)/../.. && pwd)
if [ "$1" = build ]; then
  mkdir -p $R/build/probe
  for v in 4 7; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSEGMM_PROBE_BAR=$v -I $R/include -o $R/build/probe/libsegmm_bar$v.so $R/segmminterest_amd/csrc/capi.hip &
  done
  wait
else
  for v in 0 1 2 3 4 7; do
    L=$R/segmminterest_amd/libsegmm_hip.so; [ $v != 0 ] && L=$R/build/probe/libsegmm_bar$v.so
    for shape in "nt 20480 768 3072" "nt 20480 3072 768" "tn 3072 768 20480"; do
      echo "probe $v  $shape  $(SEGMM_LIB=$L python $R/tools/probe/time_one.py $shape)"
    done
  done
fi
