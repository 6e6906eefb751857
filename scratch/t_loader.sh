#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/loader; rm -rf $O; mkdir -p $O
nproc > $O/nproc.txt
timeout -k 10 200 python -m chexpert_amd.loader --bench --workers 1,4,8,16 > $O/loader_bench.json 2> $O/err.txt; cat $O/loader_bench.json
python - <<'PY'
from chexpert_amd.loader import make_jpeg_folder
make_jpeg_folder("/tmp/cxdata", n=2048)
PY
export CHEXPERT_NUM_WORKERS=16
timeout -k 10 300 python chexpert.py --train --data_path /tmp/cxdata --model densenet121 --batch_size 256 --resize 320 --n_epochs 4 --fused_optimizer --graph --eval_interval 100000 --log_interval 100000 --output_dir /tmp/out_data > $O/train_data.log 2>> $O/err.txt; grep images_per_sec $O/train_data.log
timeout -k 10 300 python chexpert.py --train --synthetic 2048 --model densenet121 --batch_size 256 --resize 320 --n_epochs 4 --fused_optimizer --graph --eval_interval 100000 --log_interval 100000 --output_dir /tmp/out_syn > $O/train_syn.log 2>> $O/err.txt; grep images_per_sec $O/train_syn.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -q -x -k "cli or CLI or chexpert_py or harness or two_ranks" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
tail -5 $O/err.txt
