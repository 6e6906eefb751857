#!/bin/bash
# Which engine switch moves the resnet152 / aaresnet152 smooth fixtures (aaresnet152 4.7e-2 -> 5.9e-2 in round 4): the same test
# under the two-plane stream / fused forward join switches, printed measurements only (the assertions may fail: that is the point)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bisect_aares.txt; : > $O
run() {
  echo "## $*" >> $O
  env "$@" timeout -k 10 280 python -m pytest tests/test_golden_smooth_gpu.py -q -s -k "train_step_matches and (aaresnet152 or resnet152)" 2>&1 | grep -E "x1: (train logits|recorded)|passed|failed" | cut -c1-260 >> $O
}
run CHEXPERT_STREAM_LO=1 CHEXPERT_FWD_JOIN_FUSE=1
run CHEXPERT_STREAM_LO=0
run CHEXPERT_STREAM_LO=1 CHEXPERT_FWD_JOIN_FUSE=0
run CHEXPERT_STREAM_LO_MIN=1
run CHEXPERT_STREAM_LO_MIN=1 CHEXPERT_FWD_JOIN_FUSE=0
cat $O
