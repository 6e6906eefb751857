#!/bin/bash
# A/B of two builds of the library on one box with an arbitrary command: bash scratch/ab_cmd.sh libA libB rounds 'command'   (alternating)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; N=$3; shift 3
cp chexpert_amd/libchexpert_hip.so /tmp/lib_keep.so
for r in $(seq 1 $N); do
  for L in $A $B; do
    cp $L chexpert_amd/libchexpert_hip.so
    echo "== $L"
    bash -c "$*" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/lib_keep.so chexpert_amd/libchexpert_hip.so
