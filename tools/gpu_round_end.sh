#!/bin/bash
# everything the round's committed evidence comes from, in one GPU call: tests, smoke, both bench configs, the rocprofv3
# kernel trace of the headline bench, kernel trace + PMC passes over every kernel
bash tools/gpu_check.sh "$@"
bash tools/gpu_pmc.sh > gpurun_out/pmc_stdout.log 2>&1
tail -3 gpurun_out/pmc_stdout.log
