"""Generates the committed golden vectors from the CPU oracle (the reference itself cannot run here:
Rust, no toolchain, and it holds no fixtures of its own).

  python tests/golden/make_golden.py

Writes tests/golden/rays_<scene>.npz  (fixed ray batches: inputs + oracle hit records)
   and tests/golden/image_<scene>.npz (small oracle renders: per-pixel radiance sums + ray count).
The vectors are data only; they pin the oracle against regressions (CPU suite) and give the GPU
suite inputs/expected outputs that do not depend on the oracle library being rebuilt identically.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import orc  # noqa: E402

RAY_SCENES = ["random_world", "earth", "simple_light", "cornell_box", "cornell_smoke", "final_scene"]
IMAGE_SCENES = ["random_world", "two_spheres", "two_perlin_spheres", "earth", "simple_light", "cornell_box",
                "cornell_smoke", "final_scene"]
RAY_W = RAY_H = 32
IMG_W = IMG_H = 40
IMG_SPP = 4
SCENE_SEED, RAY_SEED, HIT_SEED, IMG_SEED = 1, 7, 3, 1


def main():
    for name in RAY_SCENES:
        sc = orc.Scene(name, seed=SCENE_SEED)
        parts = []
        for which in (0, 1):
            rays, valid = sc.gen_rays(RAY_W, RAY_H, 1.0, seed=RAY_SEED, which=which)
            parts.append(rays[:, valid])
        rays = np.ascontiguousarray(np.concatenate(parts, axis=1))
        ref = sc.intersect(rays, seed=HIT_SEED, mode=0)
        np.savez_compressed(HERE / f"rays_{name}.npz", rays=rays, **ref)
        print(name, rays.shape[1], "rays,", int((ref["prim"] != -1).sum()), "hits")
    for name in IMAGE_SCENES:
        sc = orc.Scene(name, seed=SCENE_SEED)
        img, nrays = sc.render(IMG_W, IMG_H, 1.0, IMG_SPP, depth=50, seed=IMG_SEED, mode=1)
        np.savez_compressed(HERE / f"image_{name}.npz", sum_rgb=img, rays=np.int64(nrays), spp=np.int32(IMG_SPP))
        print(name, "image mean", float(img.mean()) / IMG_SPP, "rays", nrays)


if __name__ == "__main__":
    main()
