# step timelines of two kernel combinations on one box: bash tools/probe/tl_ab.sh
for C in "4 8" "4 4"; do
  set -- $C
  SEGMM_PL_VAR=$1 SEGMM_TN_VAR=$2 bash tools/prof_stats.sh tl_$1$2 > /dev/null 2>&1
  python tools/timeline.py gpurun_out/prof_tl_$1$2 --full > gpurun_out/timeline_nt$1_tn$2.txt 2>&1
  head -22 gpurun_out/timeline_nt$1_tn$2.txt
done
