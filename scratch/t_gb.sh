#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gb; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scratch/gridbar.hip -o /tmp/gridbar || exit 2
timeout -k 10 120 /tmp/gridbar > $O/gridbar.txt 2>&1; echo "gridbar rc=$?"; cat $O/gridbar.txt
