#!/bin/bash
# Timing-only variants of the HEADLINE build (split-fp16, 48-token waves; results WRONG): H3_FFN_EXPERIMENT=<ffn flags> /
# H3_ATTN_EXPERIMENT=<attn flags> applied to the generated in / out / encoder-stack statements, tw_netblock_h3.hip compiled from
# a scratch copy of csrc/ and linked with the in-tree objects into timewarp_amd/lib/ab/libtimewarp_hip_<name>.so (TW_HIP_LIB;
# tools/ab_lib.sh <name> times it against the in-tree library on one box).
# usage: tools/build_h3_experiment.sh <name> <ffn flags> [attn flags]
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; aflags=$3
d=/tmp/h3_exp_$name/a; rm -rf /tmp/h3_exp_$name; mkdir -p $d /tmp/h3_exp_$name/include; cp include/*.h /tmp/h3_exp_$name/include/; cp -r timewarp_amd/csrc $d/csrc
export H3_FFN_EXPERIMENT=$flags H3_ATTN_EXPERIMENT=$aflags
for a in "--shape=in" "--shape=out"; do python tools/gen_h3_ffn_asm.py $a --out-dir=$d/csrc > /dev/null; done
python tools/gen_h3_enc_asm.py --out-dir=$d/csrc > /dev/null
python tools/gen_h3_enc_asm.py --mode=windowed --out-dir=$d/csrc > /dev/null
mkdir -p timewarp_amd/lib/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $d/csrc/tw_netblock_h3.hip -o $d/h3.o 2> $d/err.txt || { tail $d/err.txt; exit 1; }
L=timewarp_amd/lib
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/ab/libtimewarp_hip_$name.so $L/tw_api.o $L/tw_energy.o $L/tw_kernels.o $L/tw_md.o $L/tw_mh_step.o $L/tw_netblock.o $L/tw_netblock_dense.o $d/h3.o
echo built $name
