#!/bin/bash
# tools/exp/build_x.sh NAME WGS BPT WGPCU NT [extra -D...] -> tools/exp/lib_NAME.so : BC7 big-shape experiment (BU_X_* in bu_kernels.hpp)
name=$1; w=$2; b=$3; g=$4; nt=$5; shift 5
exec bash tools/exp/build_variant.sh $name -DBU_X_WGS=$w -DBU_X_BPT=$b -DBU_X_WGPCU=$g -DBU_X_NT=$nt "$@"
