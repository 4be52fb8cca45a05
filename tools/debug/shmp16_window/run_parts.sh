cd $GRAFT_REPO_ROOT
bash tools/debug/ab_libs.sh BASE PNOTAB PNOSLOTS PNOMFMA PNOSTORE PNOSCALE > gpurun_out/r5_y_parts_cox2.log 2>&1
bash tools/debug/ab_libs.sh BASE PNOTAB PNOSLOTS PNOMFMA PNOSTORE PNOSCALE -- --workload syn_1827 --replicas 2 > gpurun_out/r5_y_parts_syn.log 2>&1
for f in gpurun_out/r5_y_parts_cox2.log gpurun_out/r5_y_parts_syn.log; do echo $f; grep -o "^[A-Z]* step [0-9.]* ms\|shmp_layer16<3,2,f16x3> [0-9.]*\|shmp_layer16<3,0,f16x3> [0-9.]*" $f | paste - - - ; done
