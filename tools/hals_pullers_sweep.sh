for P in 4 5 6 7; do
  CMF_HALS_PMAX=$P timeout -k 10 200 python bench.py --config 5 --steps 6 --warmup 2 --cpu-seconds 0 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('P=$P', 'ms_per_step', round(d['ms_per_step'],3), 'pipeline_span_ms', round(r['pipeline_span_ms'],3), 'reruns', d.get('hals_pipeline_reruns'))"
done
