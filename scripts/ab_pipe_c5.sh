# split (0) against hybrid (2) pipeline on large synthetic scenes
for n in 1000000 10000000; do
  for p in 2 0; do
    RTAMD_PIPELINE=$p timeout -k 10 280 python bench.py --workload c5_synth10m --synth-n $n --width 1920 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/e.log || { tail -5 /tmp/e.log; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/o.json'));r=d['roofline']
print('n=$n pipeline $p', d['value'], 'Mrays/s', d['ms_per_step'],'ms', r['kernel'], r.get('avg_launch_ms'), 'trace', r.get('trace_ms_per_step'), 'shade', r.get('shade_ms_per_step'), 'bounce', r.get('bounce_ms_per_step'))"
  done
done
