#!/bin/bash
# tests of the DenseNet path + bench + PMC traffic of densenet121 (per-kernel bytes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_golden_smooth_gpu.py tests/test_determinism_gpu.py tests/test_kernels_gpu.py -q -x -k "not aa and not resnet and not efficientnet" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'])"
T=$O/t
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $T/fetch -- python bench.py --no-cpu-baseline --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $T/write -- python bench.py --no-cpu-baseline --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
python scratch/pmc_summary.py $(ls $T/fetch/*/*counter_collection.csv) $(ls $T/write/*/*counter_collection.csv) $O/pmc_traffic.json 4 densenet121:bf16:256:320 > $O/pmc.txt
rm -rf $T; head -1 $O/pmc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc/pmc_traffic.json'))['kernels']
for k,v in d.items():
    if any(t in k for t in ('stem','maxpool','unpool','pack_table')): print(k[:60], round(v['fetch_bytes_per_launch']/1e6), round(v['write_bytes_per_launch']/1e6))
PY
