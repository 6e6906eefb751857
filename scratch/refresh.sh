#!/bin/bash
# one gpurun call that refreshes every committed measurement of a round: tests, bench lines, kernel stats, PMC traffic, SQ counters,
# host enqueue cost, input-pipeline rate.  The refresh STOPS when the GPU suite fails or a kernel faults: numbers of a broken
# build are not produced.
# usage: bash scratch/refresh.sh [notests] [A|B]   (then: bash scratch/refresh_copy.sh copies gpurun_out/refresh/* into profiles/<round>_*)
# A gpurun call is at most 20 minutes: part A = suite + headline (bench, fp32, kernel stats, PMC, step table), part B = the other three
# BASELINE configurations + host enqueue + loader; no part argument = both in one call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PART=${2:-AB}; [ "$1" = "A" ] || [ "$1" = "B" ] && PART=$1
O=gpurun_out/refresh; mkdir -p $O
case $PART in *A*) rm -rf $O; mkdir -p $O;; esac
if [ "$1" != "notests" ] && [[ $PART == *A* ]]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?
  tail -3 $O/tests.log
  if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT in the suite: refresh aborted"; exit 3; fi
  if [ $rc -ne 0 ]; then echo "GPU suite rc=$rc: refresh aborted"; grep -h "FAILED\|Error" $O/tests.log | head -10; exit $rc; fi
fi
set -e
if [[ $PART == *A* ]]; then
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python bench.py --dtype fp32 --steps 3 --warmup 1 --no-other-configs > $O/bench_fp32.json 2>> $O/bench.err
CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-other-configs --no-graph --steps 10 --warmup 3 > $O/bench_prof.json 2>> $O/bench.err
cp $(ls $O/stats/*/*_kernel_stats.csv) $O/kernel_stats.csv
python scratch/kstats.py $O/stats 13 40 > $O/kstats_densenet121.txt; rm -rf $O/stats
fi
pmc() {   # model dtype batch size
  local M=$1 D=$2 B=$3 S=$4 T=$O/pmc_$1
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $T/fetch -- python bench.py --model $M --dtype $D --batch $B --size $S --no-cpu-baseline --no-other-configs --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $T/write -- python bench.py --model $M --dtype $D --batch $B --size $S --no-cpu-baseline --no-other-configs --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
  python scratch/pmc_summary.py $(ls $T/fetch/*/*counter_collection.csv) $(ls $T/write/*/*counter_collection.csv) $O/pmc_traffic_$M.json 4 $M:$D:$B:$S > $O/pmc_$M.txt
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $T/sq -- python bench.py --model $M --dtype $D --batch $B --size $S --no-cpu-baseline --no-other-configs --no-graph --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
  python scratch/sq_summary.py $(ls $T/sq/*/*counter_collection.csv) $O/sq_counters_$M.json $M:$D:$B:$S 4 > $O/sq_$M.txt
  rm -rf $T
}
if [[ $PART == *A* ]]; then
pmc densenet121 bf16 256 320
bash scratch/r4_trace.sh > /dev/null 2>> $O/bench.err && cp gpurun_out/r4_trace/step.txt $O/step_table.txt
fi
if [[ $PART == *B* ]]; then
for spec in "aadensenet121 128 320" "resnet152 128 320" "efficientnet-b4 64 380"; do
  set -- $spec
  timeout -k 10 300 python bench.py --model $1 --batch $2 --size $3 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$1.json 2>> $O/bench.err
  CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --model $1 --batch $2 --size $3 --no-cpu-baseline --no-graph --steps 4 --warmup 1 > /dev/null 2>> $O/bench.err
  python scratch/kstats.py $O/stats 5 30 > $O/kstats_$1.txt; rm -rf $O/stats
  pmc $1 bf16 $2 $3
done
timeout -k 10 300 python scratch/host_rate.py > $O/host_enqueue.txt 2>> $O/bench.err
timeout -k 10 300 python -m chexpert_amd.loader --bench > $O/loader_bench.json 2>> $O/bench.err
fi
if grep -q "Memory access fault" $O/bench.err; then echo "GPU FAULT"; exit 3; fi
cat $O/bench.json; cat $O/sq_densenet121.txt; cat $O/host_enqueue.txt $O/loader_bench.json
