#!/bin/bash
# tools/exp/build_variant.sh NAME [--sed EXPR]... [-DFLAG ...]  ->  tools/exp/lib_NAME.so
# A whole-library variant for A/B runs (tools/exp/ab_multi.py, ab_streams.py).  The shipped sources carry no experiment switches: a variant
# that changes code is built from a scratch COPY of basisu_rs_amd/csrc + include/ edited with the given sed expressions (applied to every file),
# e.g.  build_variant.sh bc7half --sed 's/BuBigShape<BU_TGT_BC7, BU_POLICY_SHARED> : BuShape<256, 4, 1, true, true, 2>/BuBigShape<BU_TGT_BC7, BU_POLICY_SHARED> : BuShape<512, 2, 1, true, true, 2>/'
name=$1; shift
seds=()
while [ "$1" = "--sed" ]; do seds+=(-e "$2"); shift 2; done
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/basisu_rs_amd/csrc
if [ ${#seds[@]} -gt 0 ]; then
  tmp=$(mktemp -d)
  mkdir -p $tmp/basisu_rs_amd $tmp/include
  cp -r $root/basisu_rs_amd/csrc $tmp/basisu_rs_amd/; cp $root/include/*.h $tmp/include/
  sed -i "${seds[@]}" $tmp/basisu_rs_amd/csrc/* $tmp/include/*.h
  src=$tmp/basisu_rs_amd/csrc
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -mllvm -disable-machine-licm -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-kernarg-preload-count=16 "$@" -o $root/tools/exp/lib_$name.so $src/bu_hip.hip
rc=$?
[ -n "$tmp" ] && rm -rf $tmp
exit $rc
