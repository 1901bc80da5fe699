"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_LIB_PATH = ROOT / "oracle" / "liboracle.so"

c_f = C.c_float
c_fp = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)
c_i64p = C.POINTER(C.c_int64)

_FAST_PATH = ROOT / "oracle" / "liboracle_fast.so"  # -O3 build without the visit counters: the timed CPU baseline

_lib = None
_libs = {}


def lib(path=None):
    """The checker build (default), or the oracle built at `path` (the fast build / one built for this host)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    path = Path(path) if path is not None else _LIB_PATH
    if str(path) in _libs:
        return _libs[str(path)]
    if not path.exists():
        raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(str(path))
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_int64]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_scene_info.argtypes = [C.c_void_p, c_i64p]
    L.orc_scene_preset.argtypes = [C.c_void_p, c_fp]
    L.orc_scene_spheres.restype = C.c_int64
    L.orc_scene_spheres.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_camera_new.argtypes = [c_fp, c_fp, c_fp, c_f, c_f, c_f, c_f, c_f, c_f, c_fp]
    L.orc_intersect_batch.restype = C.c_int
    L.orc_intersect_batch.argtypes = [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 7 + [c_f, c_f, C.c_uint32] + [C.c_void_p] * 8 + [C.c_int]
    L.orc_gen_rays.restype = C.c_int
    L.orc_gen_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, c_f, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_uint32,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_resolve_u8.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.orc_resolve_u8_sqrt.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.orc_rng_u32.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.orc_rng_f32.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.orc_rng_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, c_f, c_f, C.c_int, C.c_void_p]
    L.orc_rng_medium_xi.restype = C.c_double
    L.orc_rng_medium_xi.argtypes = [C.c_uint32] * 5
    L.orc_kat_sphere.argtypes = [c_fp, c_f, c_fp, c_f, c_f, c_fp]
    L.orc_kat_moving_sphere.argtypes = [c_fp, c_fp, c_f, c_f, c_f, c_fp, c_f, c_f, c_fp]
    L.orc_kat_rect.argtypes = [C.c_int, c_f, c_f, c_f, c_f, c_f, c_fp, c_f, c_f, c_fp]
    L.orc_kat_instanced_box.argtypes = [c_f, c_f, c_f, c_f, c_fp, c_fp, c_f, c_f, c_fp, c_fp]
    L.orc_kat_aabb_hit.restype = C.c_int
    L.orc_kat_aabb_hit.argtypes = [c_fp, c_fp, c_fp, c_f, c_f]
    L.orc_kat_medium_sphere.argtypes = [c_fp, c_f, c_f, C.c_double, c_fp, c_f, c_f, c_fp]
    L.orc_kat_reflect.argtypes = [c_fp, c_fp, c_fp]
    L.orc_kat_refract.argtypes = [c_fp, c_fp, c_f, c_fp]
    L.orc_kat_reflectance.restype = c_f
    L.orc_kat_reflectance.argtypes = [c_f, c_f]
    L.orc_kat_quantise.restype = C.c_uint8
    L.orc_kat_quantise.argtypes = [c_f]
    L.orc_kat_image_value.argtypes = [C.c_void_p, C.c_int, C.c_int, c_f, c_f, c_fp]
    L.orc_kat_perlin_noise.restype = c_f
    L.orc_kat_perlin_noise.argtypes = [C.c_void_p, C.c_void_p, c_fp]
    L.orc_kat_perlin_turb.restype = c_f
    L.orc_kat_perlin_turb.argtypes = [C.c_void_p, C.c_void_p, c_fp, C.c_int]
    L.orc_kat_checker.argtypes = [c_fp, c_fp, c_fp, c_fp]
    L.orc_kat_sphere_uv.argtypes = [c_fp, c_fp]
    L.orc_kat_scatter.argtypes = [C.c_int, c_fp, c_f, c_fp, c_fp, c_fp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, c_fp]
    L.orc_kat_camera_ray.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, c_fp]
    _libs[str(path)] = L
    if path == _LIB_PATH:
        _lib = L
    return L


def fa(*vals):
    """float array for ctypes from scalars / sequences."""
    flat = []
    for v in vals:
        if isinstance(v, (list, tuple, np.ndarray)):
            flat.extend(float(x) for x in v)
        else:
            flat.append(float(v))
    return (C.c_float * len(flat))(*flat)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def load_ppm(path):
    """Binary P6 reader -> (h, w, 3) uint8."""
    raw = Path(path).read_bytes()
    parts = raw.split(b"\n", 3)
    assert parts[0] == b"P6"
    w, h = (int(x) for x in parts[1].split())
    assert parts[2] == b"255"
    return np.frombuffer(parts[3], dtype=np.uint8, count=w * h * 3).reshape(h, w, 3)


_EARTH = None


def earthmap():
    global _EARTH
    if _EARTH is None:
        _EARTH = np.ascontiguousarray(load_ppm(ROOT / "tests" / "golden" / "earthmap.ppm"))
    return _EARTH


def image_height(width, aspect):
    """`(image_width as f32 / aspect_ratio) as u32`  (src/main.rs:383)."""
    return int(np.float32(width) / np.float32(aspect))


DEFAULT_THREADS = min(16, os.cpu_count() or 1)


class Scene:
    def __init__(self, name, seed=1, n_synth=0, lib_path=None):
        L = self._L = lib(lib_path)
        img = earthmap() if name in ("earth", "final_scene") else None
        ih, iw = (img.shape[0], img.shape[1]) if img is not None else (0, 0)
        self._img = img
        self.h = L.orc_scene_create(name.encode(), seed, _ptr(img), iw, ih, n_synth)
        if not self.h:
            raise ValueError(f"oracle: cannot build scene {name}")
        self.name = name
        info = (C.c_int64 * 4)()
        L.orc_scene_info(self.h, info)
        self.n_prims, self.n_instances, self.n_media, self.n_top = (int(x) for x in info)
        cam = (C.c_float * 11)()
        L.orc_scene_preset(self.h, cam)
        self.cam_origin = np.array(cam[0:3], dtype=np.float32)
        self.cam_target = np.array(cam[3:6], dtype=np.float32)
        self.vfov_deg = float(cam[6])
        self.aperture = float(cam[7])
        self.background = np.array(cam[8:11], dtype=np.float32)

    def close(self):
        if self.h:
            self._L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def spheres(self):
        """(n_top, 4) array cx,cy,cz,r of the top-level plain spheres (NaN rows for other objects)."""
        out = np.empty((self.n_top, 4), dtype=np.float32)
        self._L.orc_scene_spheres(self.h, self.n_top, _ptr(out))
        return out

    def gen_rays(self, w, h, aspect, seed=1, which=0):
        rays = np.zeros((7, w * h), dtype=np.float32)
        valid = np.zeros(w * h, dtype=np.uint8)
        rc = self._L.orc_gen_rays(self.h, w, h, aspect, seed, which, _ptr(rays), _ptr(valid))
        assert rc == 0
        return rays, valid.astype(bool)

    def intersect(self, rays, tmin=0.005, tmax=np.inf, seed=1, mode=0, threads=DEFAULT_THREADS, want_counters=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[1]
        out = dict(
            prim=np.empty(n, np.int32), inst=np.empty(n, np.int32), t=np.empty(n, np.float32),
            normal=np.empty((n, 3), np.float32), front=np.empty(n, np.uint8), uv=np.empty((n, 2), np.float32),
            point=np.empty((n, 3), np.float32))
        counters = np.zeros(3, np.uint64)
        rc = self._L.orc_intersect_batch(
            self.h, mode, n, *[_ptr(rays[i]) for i in range(7)], tmin, tmax, seed,
            _ptr(out["prim"]), _ptr(out["inst"]), _ptr(out["t"]), _ptr(out["normal"]), _ptr(out["front"]),
            _ptr(out["uv"]), _ptr(out["point"]), _ptr(counters), threads)
        assert rc == 0
        if want_counters:
            out["counters"] = counters
        return out

    def render(self, w, h, aspect, spp, depth=50, seed=1, mode=1, tile=None, s0=0, sum_rgb=None,
               threads=DEFAULT_THREADS, want_counters=False):
        """Returns (sum_rgb (h,w,3) f32, rays)."""
        if sum_rgb is None:
            sum_rgb = np.zeros((h, w, 3), dtype=np.float32)
        x0, y0, x1, y1 = tile if tile is not None else (0, 0, w, h)
        rays = C.c_uint64(0)
        counters = np.zeros(3, np.uint64)
        rc = self._L.orc_render(self.h, mode, w, h, aspect, s0, s0 + spp, depth, seed, x0, y0, x1, y1,
                              _ptr(sum_rgb), C.byref(rays), _ptr(counters), threads)
        assert rc == 0
        if want_counters:
            return sum_rgb, int(rays.value), counters
        return sum_rgb, int(rays.value)


def resolve_u8(sum_rgb, spp, sqrt=False):
    out = np.empty(sum_rgb.shape, dtype=np.uint8)
    fn = lib().orc_resolve_u8_sqrt if sqrt else lib().orc_resolve_u8
    fn(_ptr(np.ascontiguousarray(sum_rgb)), sum_rgb.size // 3, spp, _ptr(out))
    return out


def camera_new(origin, target, up, vfov_rad, aspect, aperture, focus, t0, t1):
    out = (C.c_float * 24)()
    lib().orc_camera_new(fa(origin), fa(target), fa(up), vfov_rad, aspect, aperture, focus, t0, t1, out)
    return np.array(out[:], dtype=np.float32)
