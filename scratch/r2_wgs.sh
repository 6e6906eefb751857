#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 256 384 512 768 1024; do echo "== WGS=$w"; CX_PW_BWD_WGS=$w timeout -k 10 200 python scratch/bench_pw.py fused 2>&1 | grep fused; done
