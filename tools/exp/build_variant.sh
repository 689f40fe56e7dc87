#!/bin/bash
# tools/exp/build_variant.sh NAME [-DFLAG ...]  ->  tools/exp/lib_NAME.so
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -mllvm -disable-machine-licm -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-kernarg-preload-count=16 "$@" -o tools/exp/lib_$name.so basisu_rs_amd/csrc/bu_hip.hip
