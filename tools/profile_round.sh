#!/bin/bash
# Round profile set (run on the GPU box through gpurun from the repo root):  bash tools/profile_round.sh r02
# Writes gpurun_out/<tag>_*: rocprofv3 kernel stats of bench.py at config 2, config 5 (HALS) and the T/8 shard, and three
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ+GRBM) summarised by tools/pmc_summary.py.  Counters are collected in
# their own runs with --kernel-trace only (never combined with --stats / other trace domains).
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
B="python3 bench.py --cpu-seconds 0 --no-extras --sustain 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o t -- $B --steps 10 --warmup 2 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
echo "stats done"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o t -- $B --steps 4 --warmup 1 > /dev/null 2> $out/${tag}_pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o t -- $B --steps 4 --warmup 1 > /dev/null 2> $out/${tag}_pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $out/${tag}_pmc_sq -o t -- $B --steps 4 --warmup 1 > /dev/null 2> $out/${tag}_pmc_sq.err
echo "sq done"
python3 tools/pmc_summary.py $out/${tag}_pmc_summary.json $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_sq > $out/${tag}_pmc_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_hals_stats -o t -- python3 bench.py --config 5 --cpu-seconds 0 --steps 5 --warmup 1 > $out/${tag}_hals_bench_under_rocprof.json 2> $out/${tag}_hals_stats.err
python3 tools/trace_gaps.py $out/${tag}_hals_stats hals_w_sweep > $out/${tag}_hals_timeline.txt   # (the chasing conv launch starts INSIDE the pipeline's span)
echo "hals done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_shard8_stats -o t -- $B --T 6250 --steps 40 --warmup 3 > $out/${tag}_shard8_bench_under_rocprof.json 2> $out/${tag}_shard8_stats.err
python3 tools/trace_gaps.py $out/${tag}_shard8_stats > $out/${tag}_shard8_timeline.txt
echo "shard8 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_gram_shard8_stats -o t -- python3 tools/gram_shard_profile.py 6250 50 1 > $out/${tag}_gram_shard8.txt 2> $out/${tag}_gram_shard8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_gram_stats -o t -- python3 tools/gram_shard_profile.py 50000 30 1 > $out/${tag}_gram.txt 2> $out/${tag}_gram.err
if [ -x tools/bin/valu_latency ]; then tools/bin/valu_latency > $out/${tag}_valu_latency.txt; fi
echo "all done"
