#!/bin/bash
# phase stamps of the 3x3 weight / input gradient kernels: make -C chexpert_amd/csrc stamps; gpurun -- 'bash scratch/t_stamps.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stamps
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
timeout -k 10 200 python scratch/stamps_strip.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps/strip.txt
timeout -k 10 200 python scratch/stamps_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps/ring.txt
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
