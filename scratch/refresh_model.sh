#!/bin/bash
# bench line + kernel statistics + PMC passes of ONE model: bash scratch/refresh_model.sh efficientnet-b4 64 380
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=$1; B=$2; S=$3; D=bf16
O=gpurun_out/refresh; mkdir -p $O
timeout -k 10 300 python bench.py --model $M --batch $B --size $S --no-cpu-baseline > $O/bench_$M.json 2>> $O/bench.err
T=$O/pmc_$M
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $T/fetch -- python bench.py --model $M --dtype $D --batch $B --size $S --no-cpu-baseline --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $T/write -- python bench.py --model $M --dtype $D --batch $B --size $S --no-cpu-baseline --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
python scratch/pmc_summary.py $(ls $T/fetch/*/*counter_collection.csv) $(ls $T/write/*/*counter_collection.csv) $O/pmc_traffic_$M.json 4 $M:$D:$B:$S > $O/pmc_$M.txt
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $T/sq -- python bench.py --model $M --dtype $D --batch $B --size $S --no-cpu-baseline --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
python scratch/sq_summary.py $(ls $T/sq/*/*counter_collection.csv) $O/sq_counters_$M.json $M:$D:$B:$S 4 > $O/sq_$M.txt
rm -rf $T
timeout -k 10 300 python bench.py --model $M --batch $B --size $S --no-cpu-baseline > $O/bench_$M.json 2>> $O/bench.err
bash scratch/prof_model.sh $M $B $S > gpurun_out/prof_$M.log 2>&1
python -c "
import json; d=json.load(open('$O/bench_$M.json')); print(d['value'], d['ms_per_step'], d['config']['model_hbm_roofline_frac'], d['roofline'])"
