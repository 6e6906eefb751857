#!/bin/bash
# phase stamps of the 3x3 kernels: make -C chexpert_amd/csrc stamps; gpurun -- 'bash scratch/t_stamps.sh [strip|ring]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stamps
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
[ "$1" != "ring" ] && timeout -k 10 200 python scratch/stamps_strip.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps/strip.txt
[ "$1" != "strip" ] && timeout -k 10 200 python scratch/stamps_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps/ring.txt
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
