#!/bin/bash
# Builds one small shared object per variant of gossip_f16_var.hip into ./_build (git-ignored; travels with gpurun).
# usage: build_variants.sh            (all variants of VARIANTS below)
set -e
cd "$(dirname "$0")"
mkdir -p _build
python3 materialize.py      # the variant sources are kept as patches (gossip_f16_var.patch, old_0d06b19/*.ed)
CS=../../../desco_amd/csrc
BASE="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -mllvm -pragma-unroll-threshold=200000 -I$CS"
g++ -O2 -fPIC -c stub.cpp -o _build/stub.o
build() {   # name, flags...
  local name=$1; shift
  local src=gossip_f16_var.hip
  case $name in old_*) src=old_0d06b19/gossip_wave_f16.hip;; ship*) src=$CS/gossip_f16.hip;; abl_*|exp_*) src=_build/gossip_f16_abl.hip;; esac
  /opt/rocm/bin/hipcc $BASE "$@" -c $src -o _build/$name.o 2>_build/$name.log || { cat _build/$name.log; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _build/libgf16_$name.so _build/$name.o _build/stub.o
  /opt/rocm/bin/hipcc $BASE "$@" -S --cuda-device-only $src -o _build/$name.s 2>/dev/null
  printf "%-22s v_pk %5d  s_nop %4d  vgprs %s\n" $name $(grep -c "v_pk_" _build/$name.s) $(grep -c s_nop _build/$name.s) \
     "$(grep -m1 -E '^\s+\.vgpr_count' _build/$name.s | tr -d ' ')"
}
NOSLP="-fno-slp-vectorize"
abl() {   # timing ablations of the shipped kernel (make_ablations.py): wrong results by design
  python3 make_ablations.py
  build abl_none
  build abl_nomfma -DABL_NOMFMA
  build abl_noreq  -DABL_NOREQ
  build abl_nop1   -DABL_NOP1
  build abl_noepi  -DABL_NOEPI
  build abl_nohead -DABL_NOHEAD
  build abl_valu   -DABL_NOMFMA -DABL_NOREQ
  build abl_mfma   -DABL_NOP1 -DABL_NOEPI
}
exps() {  # scheduling experiments on the shipped kernel (right results)
  python3 make_ablations.py
  build exp_prio1   -DEXP_PRIO=1
  build exp_prio2   -DEXP_PRIO=2
  build exp_stagger -DEXP_STAGGER
}
if [ -n "$1" ]; then for v in "$@"; do case $v in ship) build ship;; abl) abl;; exps) exps;; *) echo "ship | abl | exps"; exit 1;; esac; done; exit 0; fi
build ship          # the kernel as it ships (desco_amd/csrc/gossip_f16.hip)
build nopk
build pk            -DVAR_PK
build pk_w4         -DVAR_PK -DVAR_WAVES=4
build nopk_w4       -DVAR_WAVES=4
build pk_fence      -DVAR_PK -DVAR_FENCE
build pk_noslp      -DVAR_PK $NOSLP
build pk_p1s        -DVAR_PK $NOSLP -DVAR_P1_SCALAR
build pk_epis       -DVAR_PK $NOSLP -DVAR_EPI_SCALAR
build pk_p1s_epis   -DVAR_PK $NOSLP -DVAR_P1_SCALAR -DVAR_EPI_SCALAR
build pk_nodrain    -DVAR_PK -DVAR_NODRAIN
build nopk_nodrain  -DVAR_NODRAIN
build pk_noasm      -DVAR_PK -DVAR_NOASM
build pk_p1asm      -DVAR_PK -DVAR_P1_ASM
build pk_p1asm_nop0 -DVAR_PK -DVAR_P1_ASM -DVAR_P1_NOP=0
build pk_p1asm_nop3 -DVAR_PK -DVAR_P1_ASM -DVAR_P1_NOP=3
build pk_pad        -DVAR_PK -mllvm -amdgpu-mfma-padding-ratio=100
# the first wave-autonomous version (commit 0d06b19), whose packed build was the wrong one in round 4
build old_nopk
build old_pk        -DVAR_PK
build old_pk_w4     -DVAR_PK -DVAR_WAVES=4
build old_pk_fence  -DVAR_PK -DVAR_FENCE
build old_pk_h1fma  -DVAR_PK -DVAR_H1FMA
build old_pk_noflat -DVAR_PK -DVAR_NOFLAT
build old_pk_vmcnt0 -DVAR_PK -DVAR_VMCNT0
build old_pk_noslp  -DVAR_PK $NOSLP
build old_pk_p1s    -DVAR_PK $NOSLP -DVAR_P1_SCALAR
build old_pk_epis   -DVAR_PK $NOSLP -DVAR_EPI_SCALAR
build old_pk_asmnop -DVAR_PK -DVAR_ASMNOP
build old_pk_noasm  -DVAR_PK -DVAR_NOASM
for m in 0 1 2 4 8 16 31 3 7 15; do build old_pk_m$m -DVAR_PK $NOSLP -DVAR_PKMASK=$m; done
for f in 1 2 3 4 5; do build old_pk_m1f$f -DVAR_PK $NOSLP -DVAR_PKMASK=1 -DVAR_M1FORM=$f; done
build old_pk_pad    -DVAR_PK -mllvm -amdgpu-mfma-padding-ratio=100
