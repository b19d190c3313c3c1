#!/bin/bash
# Build a variant of libcqlrec.so with extra -D flags for qhead.hip/topk.hip (A/B-ing kernel parameters):
#   tools/build_variant.sh NAME -DQS_FUSED_VALU=48 ...   ->  replay_cql_amd/libcqlrec_NAME.so
# Run a tool against it with CQLREC_LIB=replay_cql_amd/libcqlrec_NAME.so (the other objects come from the last full build).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=replay_cql_amd/build
mkdir -p $B/var_$name
for f in qhead qhead_de qhead_de2 qhead_de3 qhead_argmax2 qhead_fwd2 qhead_fwd3 qhead_topk2 qhead_topk4 topk train; do
  x=""; [ $f != train ] && x="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $x "$@" -c replay_cql_amd/csrc/$f.hip -o $B/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o replay_cql_amd/libcqlrec_$name.so $B/misc.hip.o $B/var_$name/qhead.o $B/var_$name/qhead_de.o $B/var_$name/qhead_de2.o $B/var_$name/qhead_de3.o $B/var_$name/qhead_argmax2.o $B/var_$name/qhead_fwd2.o $B/var_$name/qhead_fwd3.o $B/var_$name/qhead_topk2.o $B/var_$name/qhead_topk4.o $B/var_$name/topk.o $B/gbwd.hip.o $B/prep.hip.o $B/var_$name/train.o
echo replay_cql_amd/libcqlrec_$name.so
