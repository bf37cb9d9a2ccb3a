#!/bin/bash
# Few-component kernels (csrc/cmf_small_k.h) at the reference's protocol shape (N=250 T=50000 K=5 L=20): rocprofv3 kernel stats and
# the three --pmc passes (each its own run, --kernel-trace only) of tools/time_small_k.py, which runs the few-component kernels
# (small_k = 1) and the general ones (small_k = 0) back to back:   bash tools/profile_small_k.sh r04_smallk [N T K L]
set -e
tag=${1:-r04_smallk}
shape="${2:-250} ${3:-50000} ${4:-5} ${5:-20}"
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
B="python3 tools/time_small_k.py $shape"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o t -- $B > $out/${tag}_times.txt 2> $out/${tag}_stats.err
echo "stats done"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o t -- $B > /dev/null 2> $out/${tag}_pmc_fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o t -- $B > /dev/null 2> $out/${tag}_pmc_write.err
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $out/${tag}_pmc_sq -o t -- $B > /dev/null 2> $out/${tag}_pmc_sq.err
python3 tools/pmc_summary.py $out/${tag}_pmc_summary.json $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_sq > $out/${tag}_pmc_summary.txt
cat $out/${tag}_pmc_summary.txt
