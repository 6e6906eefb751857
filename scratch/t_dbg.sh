#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/dbg
timeout -k 10 300 python scratch/dbg_join.py 1,3,1,1 > gpurun_out/dbg/join.txt 2>&1; echo rc=$?
head -60 gpurun_out/dbg/join.txt
