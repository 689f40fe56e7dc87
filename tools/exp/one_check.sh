#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for nb in 2 8; do for tk in 1 0; do PW_NBUF=$nb BU_TILE_TICKETS=$tk python tools/exp/r6_host.py one 2>&1 | grep -v "amdgpu\|GPU_MAX"; done; done
