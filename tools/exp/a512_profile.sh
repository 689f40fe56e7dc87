#!/bin/bash
# tools/exp/a512_profile.sh: only the config-5 part of tools/gpu_pmc.sh (the 2^25-block BC7 launch: rocprofv3 kernel trace + stats, FETCH_SIZE and WRITE_SIZE passes)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
sed -n '/^# ---- BASELINE config 5/,$p' tools/gpu_pmc.sh > /tmp/a512_part.sh
bash /tmp/a512_part.sh
