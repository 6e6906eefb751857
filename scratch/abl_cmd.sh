#!/bin/bash
# run one command under several builds of the library on one box: bash scratch/abl_cmd.sh 'command' lib1 lib2 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=$1; shift
cp chexpert_amd/libchexpert_hip.so /tmp/lib_keep.so
for L in "$@"; do
  cp $L chexpert_amd/libchexpert_hip.so
  echo "== $L"
  bash -c "$C" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so chexpert_amd/libchexpert_hip.so
