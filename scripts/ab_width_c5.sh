# four-child walk on the C5-like scene: LDS stack entries and inner steps per refill check
run() {
  timeout -k 10 280 python bench.py --workload c5_synth10m --width 1920 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/e.log || { tail -5 /tmp/e.log; exit 1; }
  python -c "
import json;d=json.load(open('/tmp/o.json'));r=d['roofline']
print('$1', d['value'], 'Mrays/s', d['ms_per_step'],'ms', r['kernel'], r.get('avg_launch_ms'))"
}
RTAMD_BVH_WIDTH=2 run "width2"
RTAMD_BVH_WIDTH=4 run "width4"
RTAMD_BVH_WIDTH=4 RTAMD_STACK_LDS=12 run "width4 lds12"
RTAMD_BVH_WIDTH=4 RTAMD_STACK_LDS=16 run "width4 lds16"
RTAMD_BVH_WIDTH=4 RTAMD_INNER_STEPS=2 run "width4 inner2"
RTAMD_BVH_WIDTH=4 RTAMD_INNER_STEPS=8 run "width4 inner8"
RTAMD_BVH_WIDTH=4 RTAMD_STACK_LDS=12 RTAMD_INNER_STEPS=2 run "width4 lds12 inner2"
