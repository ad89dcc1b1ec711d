# Same kernels, random vs all-zero operands (power / clock sensitivity of the plane GEMMs); run on the GPU box.
R=$(cd $(dirname $0)/../.. && pwd)
for var in 8 1; do for z in 0 1; do
  for shape in "nt 20480 768 3072" "nt 20480 3072 768" "nt 71680 768 768" "tn 3072 768 20480"; do
    echo "probe PL_VAR=$var zero=$z  $shape  $(SEGMM_PL_VAR=$var PROBE_ZERO=$z python $R/tools/probe/time_one.py $shape 2>/dev/null)"
  done
done; done
