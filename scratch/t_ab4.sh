#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { grep -n "Error\|^E \|FAILED" $O/tests.log | head; exit $rc; }
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
bash scratch/ab.sh scratch/libA.so scratch/libB.so
for M in "efficientnet-b4 64 380"; do set -- $M
  for L in scratch/libA.so scratch/libB.so scratch/libA.so scratch/libB.so scratch/libA.so scratch/libB.so; do
    cp $L chexpert_amd/libchexpert_hip.so
    timeout -k 10 300 python bench.py --model $1 --batch $2 --size $3 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$1 $L', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
