#!/bin/bash
# Round 5: PMC passes over k_conv_stack_x3 at the launch sizes the timed window uses (VERDICT r4 Next 3), one counter group per pass
# (rocprofv3 cannot hold FETCH_SIZE and WRITE_SIZE in one pass; gpurun refuses --pmc together with trace domains), leaves fed as
# 16-byte board records like the product.  CONV_BOARDS = boards the launch covers; CONV_RANGE = rows that hold leaves (device-side
# range of a dense-rows step: the grid of a 1 365-slot part is 683 workgroups, ~440 of them find rows).  Then an un-overlapped
# kernel trace of the current code (one stream) and the three-stream steady state under the FETCH_SIZE pass.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05i}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES")
NAMES=(fetch write busy)
for cfg in "256 0" "512 0" "880 0" "1024 0" "1365 880" "1365 0" "2048 0" "4096 0"; do
  set -- $cfg; B=$1; RG=$2
  for i in 0 1 2; do
    D=$O/pmc_${B}_${RG}_${NAMES[$i]}
    if [ "$RG" != "0" ]; then export CONV_RANGE=$RG; else unset CONV_RANGE; fi
    CONV_MODES=f16x3 CONV_INPUT=boards CONV_BOARDS=$B timeout 300 rocprofv3 --output-format csv --pmc ${PASSES[$i]} -d $D -o pmc -- python $R/tools/conv_pmc.py > /dev/null 2>> $O/err.txt
  done
  python $R/tools/pmc_summary.py $O/pmc_${B}_${RG}_* | sed "s/^/${B},${RG},/" >> $O/pmc_conv_by_launch_size.csv
  rm -rf $O/pmc_${B}_${RG}_*
done
unset CONV_RANGE
# the steady-state step on three streams under the FETCH_SIZE / WRITE_SIZE passes (counter collection serialises the dispatches:
# per-dispatch values of real steps' launches, not an overlap measurement)
COMMON="--steps 100 --warmup 20 --preroll 2500 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0"
for i in 0 1; do
  timeout 900 rocprofv3 --output-format csv --pmc ${PASSES[$i]} -d $O/step_${NAMES[$i]} -o pmc -- python $R/bench.py $COMMON > $O/bench_pmc_${NAMES[$i]}.json 2>> $O/err.txt
done
python $R/tools/pmc_summary.py --last=150 $O/step_fetch $O/step_write > $O/pmc_step_three_streams.csv 2>> $O/err.txt
rm -rf $O/step_fetch $O/step_write
# un-overlapped kernel trace: one stream
timeout 900 rocprofv3 --output-format csv --kernel-trace -d $O/trace_one -o t -- python $R/bench.py --steps 300 --warmup 50 --preroll 4000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 20 --no-split > $O/bench_one_stream_profiled.json 2>> $O/err.txt
python $R/tools/kernel_stats.py $O/trace_one > $O/kernel_stats_one_stream.csv 2>> $O/err.txt
rm -rf $O/trace_one
# three streams, trace (for the committed timeline) + the default-size launch alone via HIP events (bench's kernel_alone)
timeout 900 rocprofv3 --output-format csv --kernel-trace -d $O/trace_three -o t -- python $R/bench.py --steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 20 > $O/bench_three_streams_profiled.json 2>> $O/err.txt
python $R/tools/kernel_stats.py $O/trace_three > $O/kernel_stats_three_streams.csv 2>> $O/err.txt
python $R/tools/step_timeline.py $O/trace_three --steps 50 --print-steps 3 > $O/step_timeline_fp32_three_streams.txt 2>> $O/err.txt
rm -rf $O/trace_three
timeout 900 rocprofv3 --output-format csv --kernel-trace -d $O/trace_bf16 -o t -- python $R/bench.py --nn-dtype bf16 --steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0 > $O/bench_bf16_profiled.json 2>> $O/err.txt
python $R/tools/step_timeline.py $O/trace_bf16 --steps 50 --print-steps 3 > $O/step_timeline_bf16_three_streams.txt 2>> $O/err.txt
rm -rf $O/trace_bf16
CKR_TORCH_STREAMS=1 CKR_SPLIT_PARTS=4 timeout 900 rocprofv3 --output-format csv --kernel-trace -d $O/trace_pool4 -o t -- python $R/bench.py --steps 300 --warmup 50 --preroll 6000 --no-complete --extra-steps 0 --cpu-seconds 0 --profile-steps 0 > $O/bench_pool4_profiled.json 2>> $O/err.txt
python $R/tools/step_timeline.py $O/trace_pool4 --steps 50 --print-steps 3 > $O/step_timeline_fp32_four_parts_on_pool_streams_shared_queue.txt 2>> $O/err.txt
rm -rf $O/trace_pool4
cat $O/pmc_conv_by_launch_size.csv
tail -5 $O/err.txt
