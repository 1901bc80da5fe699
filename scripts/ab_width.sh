for w in 2 4; do
  export RTAMD_BVH_WIDTH=$w
  for cfg in "c5_synth10m --width 1920 --spp 16" "c3_random_world --spp 64" "final_scene --spp 64"; do
    timeout -k 10 280 python bench.py --workload $cfg --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/e.log || { tail -5 /tmp/e.log; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/o.json'));r=d['roofline']
print('width $w', d['config']['workload'], d['value'], 'Mrays/s', d['ms_per_step'],'ms', r['kernel'], r.get('avg_launch_ms'))"
  done
done
