for dbg in none nogate nopull; do echo "debug=$dbg"; CMF_HALS_DEBUG=$dbg python bench.py --config 5 --cpu-seconds 0 --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['pipeline_span_ms'])" || exit 1; done
