python - <<'PY'
import sys; sys.path.insert(0, 'raytracer-rs_amd')
import rtamd
ctx = rtamd.Context(0)
for name, n in (("random_world", 0), ("final_scene", 0), ("synth_spheres", 30000), ("synth_spheres", 300000), ("synth_spheres", 1000000), ("synth_spheres", 10000000)):
    i = ctx.load_scene(name, seed=1, n_synth=n)
    print(name, n, "nodes", i["n_nodes"], "stack", i["stack_entries"], "width", i["bvh_width"], flush=True)
PY
for cfg in "final_scene --spp 64" "c3_random_world --spp 64" "c5_synth10m --synth-n 300000 --width 1920 --spp 16" "c5_synth10m --synth-n 30000 --width 1920 --spp 16"; do
  for p in 2 0; do
    RTAMD_PIPELINE=$p timeout -k 10 280 python bench.py --workload $cfg --steps 3 --warmup 1 --no-cpu-baseline > /tmp/o.json 2>/tmp/e.log || { tail -5 /tmp/e.log; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/o.json'));r=d['roofline']
print('$cfg pipeline $p', d['value'], 'Mrays/s', d['ms_per_step'],'ms')"
  done
done
