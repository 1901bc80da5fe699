import sys, time
sys.path.insert(0, 'raytracer-rs_amd'); sys.path.insert(0, 'tests')
import numpy as np, rtamd, orc
ctx = rtamd.Context(0)
for name in ["random_world", "cornell_box", "cornell_smoke", "final_scene", "earth"]:
    info = ctx.load_scene(name, seed=1)
    print(name, info, flush=True)
    osc = orc.Scene(name, seed=1)
    w = 160; aspect = 1.0; h = 160; spp = 8
    cam = ctx.preset_camera(aspect)
    ocam = orc.camera_new(osc.cam_origin, osc.cam_target, [0,1,0], np.float32(osc.vfov_deg)*np.float32(np.pi)/np.float32(180.0), aspect, osc.aperture, 10.0, 0.0, 1.0)
    print(" cam diff", np.abs(cam.as_array() - ocam).max())
    t0 = time.time(); acc, st = ctx.render(w, h, cam, spp, seed=1); dt = time.time() - t0
    ref, rays = osc.render(w, h, aspect, spp, seed=1, mode=1, threads=8)
    d = acc - ref
    print(" gpu rays", st["rays"], "oracle rays", rays, "render_ms", st["render_ms"], "batches", st["batches"], "launches", st["bounce_launches"])
    print(" mean gpu", acc.mean()/spp, "mean ref", ref.mean()/spp, "rmse", np.sqrt((d*d).mean())/spp, "max", np.abs(d).max()/spp, "frac pixels differing", (np.abs(d).max(axis=2) > 1e-4*spp).mean(), flush=True)
