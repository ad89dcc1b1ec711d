# Round-6 additions to the round checkpoint (run on the GPU box AFTER tools/round_profiles.sh r6 <git>):  bash tools/round6_profiles.sh
# Needs the probe builds of this round in build/probe/ (tools/probe/README.md): g4_p, g4_st, libsegmm_{rownt0,store0,policy0}.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; cd $R
./build/probe/g4_p 20 0 "" 0 > $O/gemm4_standalone.txt 2>&1
./build/probe/g4_st 10 0 NT_20480x3072 0 > $O/gemm4_stamps.txt 2>&1; ./build/probe/g4_st 10 0 NT_51200x768x768 0 >> $O/gemm4_stamps.txt 2>&1
bash tools/probe/pmc_gemm4.sh NT_20480x3072x768 r6_pmc4_nt build/probe/g4_p > /dev/null 2>&1; cp gpurun_out/r6_pmc4_nt/summary.csv $O/gemm_pl_nt4_vs_nt8_20480x3072x768_pmc.csv
bash tools/probe/pmc_gemm4.sh TN_3072x768x20480 r6_pmc4_tn build/probe/g4_p > /dev/null 2>&1; cp gpurun_out/r6_pmc4_tn/summary.csv $O/gemm_pl_tn4_vs_tn8_3072x768x20480_pmc.csv
{ echo "# in-step A/B of the kernel generations (PL_VAR 44 / 8 = gemm_pl_nt4 / nt8 everywhere; TN_VAR 4 / 88 = gemm_pl_tn4 / tn8 everywhere), config 2, one box"; bash tools/ab_combo44.sh 2; } > $O/ab_kernel_generations.txt 2>&1
{ echo "# the library's default choice (PL_VAR 4, TN_VAR 8) against the forced generations, configs 2 / 4 (256 rows) / 3"; bash tools/ab_defaults.sh 2; } >> $O/ab_kernel_generations.txt 2>&1
{ echo "# cache policy A/B, config 2, one box: shipped library (nt epilogue stores + streaming row kernels) | row kernels default policy | GEMM stores default policy | both default"; bash tools/probe/lib_ab.sh 3 segmminterest_amd/libsegmm_hip.so build/probe/libsegmm_rownt0.so build/probe/libsegmm_store0.so build/probe/libsegmm_policy0.so; } > $O/ab_cache_policy.txt 2>&1
ls -la $O
