# binary against four-child walk by scene size (synthetic spheres, 1920x1080, 16 spp)
for n in 300000 1000000 3000000; do
  for w in 2 4; do
    RTAMD_BVH_WIDTH=$w timeout -k 10 280 python bench.py --workload c5_synth10m --synth-n $n --width 1920 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/e.log || { tail -5 /tmp/e.log; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/o.json'));r=d['roofline']
print('n=$n width $w', d['value'], 'Mrays/s', d['ms_per_step'],'ms', r['kernel'], r.get('avg_launch_ms'))"
  done
done
