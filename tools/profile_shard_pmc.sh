#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ+GRBM, each its own run with --kernel-trace only) of the T/8 shard of config 2:
#   bash tools/profile_shard_pmc.sh <tag> [T=6250]
set -e
tag=${1:-r03_shard8}
T=${2:-6250}
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
B="python3 bench.py --cpu-seconds 0 --no-extras --sustain 0 --T $T --steps 20 --warmup 3"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o t -- $B > /dev/null 2> $out/${tag}_pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o t -- $B > /dev/null 2> $out/${tag}_pmc_write.err
echo "write done"
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $out/${tag}_pmc_sq -o t -- $B > /dev/null 2> $out/${tag}_pmc_sq.err
echo "sq done"
python3 tools/pmc_summary.py $out/${tag}_pmc_summary.json $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_sq > $out/${tag}_pmc_summary.txt
cat $out/${tag}_pmc_summary.txt
