#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/e2e
timeout -k 10 900 python scratch/e2e_cache.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e2e/e2e.txt
