#!/bin/bash
# rocprofv3 kernel stats of the Gram form (option gram = 1) at a shard's size:  bash tools/profile_gram_shard.sh <tag> [T=6250]
set -e
tag=${1:-r03_gram}
T=${2:-6250}
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag} -o t -- python3 tools/gram_shard_profile.py $T 50 1 > $out/${tag}.txt 2>&1
tail -1 $out/${tag}.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/${tag}/t_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:72]:72s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
