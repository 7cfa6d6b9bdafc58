O=gpurun_out/r3_e; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_e/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stages_ms'])
r=d['roofline']; print({k:r[k] for k in r if k not in ('others','launch_form_of_the_same_pass')})
print('others', json.dumps(r['others']))
print('fed', json.dumps(d['fed_gpu'], indent=0)[:3500])
print('pp', d['ms_per_scan']['point_processor_incl_h2d_d2h'], 'cpu', d['cpu_baseline']['value'])
PY
