#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stamps
timeout -k 10 200 python scratch/stamps_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps/ring.txt
