#!/bin/bash
# Ablation builds of the resident SHMP kernel on the GPU box (results of ablated builds are wrong by design):
#   tools/debug/ab_resident.sh <workload> <replicas>
W=${1:-syn_1827}; R=${2:-1}
cd desco_amd/csrc
OBJS="capi.o gemm_f32.o gemm_split.o graph_ops.o gossip.o shmp_layer.o shmp_layer16.o gossip_fused.o train_ops.o partition_dev.o groundtruth_dev.o partition.o groundtruth.o"
for v in 1 2; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w -mllvm -pragma-unroll-threshold=200000 -DRES_ABL=$v -c shmp_resident.hip -o /tmp/res_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdesco_abl$v.so $OBJS /tmp/res_abl$v.o -lgomp
done
cd ../..
if [ "$3" != "prof" ]; then echo "== full"; python tools/bench_resident.py $W $R 2>&1 | grep -E "resident=True|shmp_resident"
echo "== no gathers"; DESCO_LIB=/tmp/libdesco_abl1.so python tools/bench_resident.py $W $R 2>&1 | grep -E "shmp_resident"
echo "== no MFMA"; DESCO_LIB=/tmp/libdesco_abl2.so python tools/bench_resident.py $W $R 2>&1 | grep -E "shmp_resident"
fi
echo "== phase cycles (s_memtime build)"
cd desco_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w -mllvm -pragma-unroll-threshold=200000 -DRES_PROF -c shmp_resident.hip -o /tmp/res_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdesco_prof.so $OBJS /tmp/res_prof.o -lgomp
cd ../..
DESCO_LIB=/tmp/libdesco_prof.so RES_PROF=1 python tools/bench_resident.py $W $R 2>&1 | grep -E "shmp_resident|phase"
