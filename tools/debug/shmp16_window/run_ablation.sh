cd $GRAFT_REPO_ROOT
bash tools/debug/ab_libs.sh BASE LG128_12 LG256_8 LG64_12 > gpurun_out/r5_s_lg_cox2.log 2>&1
bash tools/debug/ab_libs.sh BASE LG128_12 LG256_8 -- --workload msrc_imdb --replicas 8 > gpurun_out/r5_s_lg_msrc.log 2>&1
bash tools/debug/ab_libs.sh BASE LG128_12 LG256_8 -- --workload syn_1827 --replicas 2 > gpurun_out/r5_s_lg_syn.log 2>&1
cat gpurun_out/r5_s_lg_cox2.log gpurun_out/r5_s_lg_msrc.log gpurun_out/r5_s_lg_syn.log
