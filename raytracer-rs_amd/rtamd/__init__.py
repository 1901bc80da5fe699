"""Python binding (ctypes) of the MI355X render path.

Two shared libraries, built in-tree by ``make -C raytracer-rs_amd``:

* ``lib/librtamd.so``  -- HIP kernels + the C ABI of ``include/rtamd.h``
* ``lib/librthost.so`` -- C++ mirror of the reference's scene API and its scene builders

There is no CPU fallback: importing works anywhere (so that symbols can be checked), but every
work call needs a GPU and raises ``RtError`` otherwise.  Arrays may be numpy arrays (host) or
torch CUDA tensors (device pointers are passed straight through).
"""
import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent.parent
REPO_ROOT = PKG_DIR.parent
LIB_DIR = Path(__import__("os").environ.get("RTAMD_LIB_DIR", PKG_DIR / "lib"))
EARTHMAP = REPO_ROOT / "tests" / "golden" / "earthmap.ppm"

DEFAULT_TILE = 16  # RT_DEFAULT_TILE of include/rtamd.h: side of the tiles a frame is dealt to the ranks in
SCENES = ("random_world", "two_spheres", "two_perlin_spheres", "earth", "simple_light", "cornell_box",
          "cornell_smoke", "final_scene", "nested_instances", "medium_cluster", "synth_spheres")

RT_OK = 0
STATUS = {0: "RT_OK", -1: "RT_ERR_INVALID", -2: "RT_ERR_HIP", -3: "RT_ERR_UNSUPPORTED", -4: "RT_ERR_NOT_COMMITTED",
          -5: "RT_ERR_NO_DEVICE", -6: "RT_ERR_IO"}

# every symbol include/rtamd.h declares
ABI_SYMBOLS = (
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_ctx_set_stream", "rt_ctx_set_max_paths",
    "rt_ctx_synchronize", "rt_ctx_set_pipeline", "rt_ctx_set_tail_paths", "rt_ctx_reserve", "rt_ctx_set_builder", "rt_ctx_set_bvh_width", "rt_ctx_set_fuse_boxes", "rt_scene_read_bvh", "rt_scene_clear", "rt_scene_set_spheres", "rt_scene_set_rects", "rt_scene_set_groups",
    "rt_scene_set_materials", "rt_scene_set_textures", "rt_scene_set_perlin", "rt_scene_set_image",
    "rt_scene_commit", "rt_scene_get_info", "rt_intersect_batch", "rt_intersect_batch_noalloc", "rt_render", "rt_profile_enable",
    "rt_resolve_u8", "rt_resolve_u8_transfer", "rt_image_write", "rt_image_read", "rt_accum_write", "rt_accum_read", "rt_accum_write_tagged",
    "rt_accum_read_tagged", "rt_owned_pixels", "rt_comm_unique_id", "rt_comm_create", "rt_comm_attach", "rt_comm_destroy",
    "rt_gather_frame", "rt_frame_block_pixels", "rt_pack_owned", "rt_scatter_gathered")


class RtError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class rt_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float), ("time_start", C.c_float), ("time_end", C.c_float)]

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class rt_render_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sample0", C.c_uint32), ("sample1", C.c_uint32),
                ("max_depth", C.c_uint32), ("background", C.c_float * 3), ("seed", C.c_uint32),
                ("tile_size", C.c_uint32), ("rank", C.c_uint32), ("nranks", C.c_uint32), ("reserved", C.c_uint32)]


class rt_render_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("batches", C.c_uint32), ("bounce_launches", C.c_uint32),
                ("render_ms", C.c_double), ("trace_ms", C.c_double), ("shade_ms", C.c_double),
                ("raygen_ms", C.c_double), ("accumulate_ms", C.c_double), ("trace_launches", C.c_uint64),
                ("shade_launches", C.c_uint64), ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64),
                ("bounce_ms", C.c_double), ("bounce_kernel_launches", C.c_uint64), ("pipeline", C.c_uint32),
                ("tail_launches", C.c_uint32), ("tail_rays", C.c_uint64), ("tail_ms", C.c_double),
                ("paths_ms", C.c_double), ("paths_launches", C.c_uint64), ("settled_rays", C.c_uint64),
                ("settled_ms", C.c_double), ("settled_launches", C.c_uint64), ("settled_entries", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class rt_scene_info(C.Structure):
    _fields_ = [("n_prims", C.c_uint64), ("n_nodes", C.c_uint64), ("n_groups", C.c_uint64), ("n_media", C.c_uint64),
                ("max_depth", C.c_uint32), ("stack_entries", C.c_uint32), ("device_bytes", C.c_uint64),
                ("build_seconds", C.c_double), ("tree_seconds", C.c_double), ("builder", C.c_uint32),
                ("bvh_width", C.c_uint32), ("crown_nodes", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tex", C.c_uint32), ("albedo", C.c_float * 3), ("fuzz", C.c_float),
                ("ior", C.c_float), ("reserved", C.c_uint32)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("odd", C.c_uint32), ("even", C.c_uint32),
                ("freq", C.c_float), ("table", C.c_uint32)]


class rt_group(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("has_xform", C.c_uint32), ("m", C.c_float * 12), ("minv", C.c_float * 12),
                ("neg_inv_density", C.c_double), ("mat", C.c_uint32), ("parent", C.c_uint32)]


_core = None
_host = None


def _load(name):
    path = LIB_DIR / name
    if not path.exists():
        raise RtError(-5, f"{path} is missing: build it with `make -C {PKG_DIR}` (needs hipcc); there is no CPU fallback")
    return C.CDLL(str(path), mode=C.RTLD_GLOBAL)


def core():
    """librtamd.so with argtypes set."""
    global _core
    if _core is not None:
        return _core
    L = _load("librtamd.so")
    vp, u32, u64, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    L.rt_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_ctx_destroy.argtypes = [vp]
    L.rt_ctx_destroy.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_ctx_set_stream.argtypes = [vp, vp]
    L.rt_ctx_set_max_paths.argtypes = [vp, u64]
    L.rt_ctx_synchronize.argtypes = [vp]
    L.rt_ctx_set_pipeline.argtypes = [vp, C.c_int]
    L.rt_ctx_set_tail_paths.argtypes = [vp, u64]
    L.rt_ctx_reserve.argtypes = [vp, u64]
    L.rt_ctx_set_builder.argtypes = [vp, C.c_int]
    L.rt_ctx_set_bvh_width.argtypes = [vp, C.c_int]
    L.rt_ctx_set_fuse_boxes.argtypes = [vp, C.c_int]
    L.rt_scene_read_bvh.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u32)]
    L.rt_scene_clear.argtypes = [vp]
    L.rt_scene_set_spheres.argtypes = [vp, u32] + [vp] * 7
    L.rt_scene_set_rects.argtypes = [vp, u32] + [vp] * 6
    L.rt_scene_set_groups.argtypes = [vp, u32, vp]
    L.rt_scene_set_materials.argtypes = [vp, u32, vp]
    L.rt_scene_set_textures.argtypes = [vp, u32, vp]
    L.rt_scene_set_perlin.argtypes = [vp, u32, vp, vp]
    L.rt_scene_set_image.argtypes = [vp, u32, u32, u32, vp]
    L.rt_scene_commit.argtypes = [vp, f, f]
    L.rt_scene_get_info.argtypes = [vp, C.POINTER(rt_scene_info)]
    L.rt_intersect_batch.argtypes = [vp, u64] + [vp] * 7 + [f, f, u32] + [vp] * 7
    L.rt_intersect_batch_noalloc.argtypes = [vp, u64] + [vp] * 7 + [f, f, u32] + [vp] * 7
    L.rt_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_render_params), vp, C.POINTER(rt_render_stats)]
    L.rt_profile_enable.argtypes = [vp, C.c_int]
    L.rt_resolve_u8.argtypes = [vp, vp, u64, u32, vp]
    L.rt_resolve_u8_transfer.argtypes = [vp, vp, u64, u32, C.c_int, vp]
    L.rt_image_write.argtypes = [C.c_char_p, u32, u32, vp, C.c_int]
    L.rt_image_read.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp, u64]
    L.rt_accum_write.argtypes = [C.c_char_p, u32, u32, u32, u32, vp]
    L.rt_accum_read.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), vp, u64]
    L.rt_accum_write_tagged.argtypes = [C.c_char_p, u32, u32, u32, u32, u64, vp]
    L.rt_accum_read_tagged.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), vp, u64]
    L.rt_owned_pixels.argtypes = [u32, u32, u32, u32, u32, vp, u64, C.POINTER(u64)]
    L.rt_comm_unique_id.argtypes = [vp]
    L.rt_comm_create.argtypes = [vp, vp, u32, u32]
    L.rt_comm_attach.argtypes = [vp, vp, u32, u32]
    L.rt_comm_destroy.argtypes = [vp]
    L.rt_gather_frame.argtypes = [vp, u32, u32, u32, vp, vp]
    L.rt_frame_block_pixels.argtypes = [u32, u32, u32, u32, C.POINTER(u64)]
    L.rt_pack_owned.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp]
    L.rt_scatter_gathered.argtypes = [vp, u32, u32, u32, u32, vp, vp]
    for s in ABI_SYMBOLS:
        fn = getattr(L, s)
        if s not in ("rt_ctx_destroy", "rt_last_error"):
            fn.restype = C.c_int
    _core = L
    return L


def host():
    """librthost.so (scene builders) with argtypes set."""
    global _host
    if _host is not None:
        return _host
    core()
    H = _load("librthost.so")
    vp = C.c_void_p
    H.rth_scene_build.restype = vp
    H.rth_scene_build.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_int64, C.c_char_p, C.c_int]
    H.rth_scene_destroy.argtypes = [vp]
    H.rth_scene_destroy.restype = None
    H.rth_scene_upload.argtypes = [vp, vp, C.c_char_p, C.c_int]
    H.rth_scene_upload.restype = C.c_int
    H.rth_scene_flatten_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    H.rth_scene_flatten_counts.restype = C.c_int
    H.rth_scene_preset.argtypes = [vp, C.POINTER(C.c_float)]
    H.rth_scene_preset.restype = None
    H.rth_scene_flattened_image.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp, C.c_uint64]
    H.rth_scene_flattened_image.restype = C.c_int
    H.rth_scene_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    H.rth_scene_counts.restype = None
    H.rth_scene_camera.argtypes = [vp, C.c_float, C.POINTER(rt_camera)]
    H.rth_scene_camera.restype = None
    H.rth_camera_new.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float] * 6 + [C.POINTER(rt_camera)]
    H.rth_camera_new.restype = None
    _host = H
    return H


def image_height(width, aspect):
    """``(image_width as f32 / aspect_ratio) as u32`` (main.rs:383)."""
    return int(np.float32(width) / np.float32(aspect))


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a):
    """Raw pointer of a numpy array / torch tensor (or None)."""
    if a is None:
        return None
    if _is_torch(a):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(a.data_ptr())
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)


def _check_array(a, what, dtype_name, shape, device=None):
    """The C ABI takes raw pointers: a buffer of another dtype, shape, layout or GPU would be written out of
    bounds, so it is refused here.  shape entries of None match anything."""
    if _is_torch(a):
        ok_dtype = str(a.dtype) == "torch." + dtype_name
        contiguous = a.is_contiguous()
        if a.is_cuda and device is not None and a.device.index != device:
            raise ValueError(f"{what}: tensor lives on cuda:{a.device.index}, the context on cuda:{device}")
    else:
        if not isinstance(a, np.ndarray):
            raise ValueError(f"{what}: numpy array or torch tensor expected, got {type(a).__name__}")
        ok_dtype = a.dtype == np.dtype(dtype_name)
        contiguous = a.flags["C_CONTIGUOUS"]
    if not ok_dtype:
        raise ValueError(f"{what}: dtype must be {dtype_name}, got {a.dtype}")
    if len(a.shape) != len(shape) or any(s is not None and int(d) != s for d, s in zip(a.shape, shape)):
        raise ValueError(f"{what}: shape must be {tuple('*' if s is None else s for s in shape)}, got {tuple(a.shape)}")
    if not contiguous:
        raise ValueError(f"{what}: must be C-contiguous")


def camera_new(origin, target, up, vfov_rad, aspect, aperture, focus, t0, t1):
    """Camera::new (camera.rs:28-52)."""
    cam = rt_camera()
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    host().rth_camera_new(f3(origin), f3(target), f3(up), vfov_rad, aspect, aperture, focus, t0, t1, C.byref(cam))
    return cam


def owned_pixels(width, height, tile_size=DEFAULT_TILE, rank=0, nranks=1):
    n = C.c_uint64(0)
    core().rt_owned_pixels(width, height, tile_size, rank, nranks, None, 0, C.byref(n))
    out = np.empty(n.value, dtype=np.uint32)
    core().rt_owned_pixels(width, height, tile_size, rank, nranks, _ptr(out), n.value, C.byref(n))
    return out


def frame_block_pixels(width, height, tile_size=DEFAULT_TILE, nranks=1):
    """Pixels of the largest share of any rank: the size of one rank's block in the frame exchange."""
    n = C.c_uint64(0)
    rc = core().rt_frame_block_pixels(width, height, tile_size, nranks, C.byref(n))
    if rc:
        raise RtError(rc, "rt_frame_block_pixels")
    return n.value


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes it, every rank passes the same bytes to Context.comm_create)."""
    buf = (C.c_uint8 * 128)()
    rc = core().rt_comm_unique_id(buf)
    if rc:
        raise RtError(rc, "rt_comm_unique_id (is an RCCL library available?)")
    return bytes(buf)


def flatten_counts(name, seed=1, n_synth=0, earthmap=None):
    """Build a named scene with the C++ mirror API and flatten it on the host (no GPU needed).
    Returns a dict of what would go through rt_scene_set_*."""
    H = host()
    err = C.create_string_buffer(512)
    h = H.rth_scene_build(name.encode(), seed, str(earthmap or EARTHMAP).encode(), n_synth, err, len(err))
    if not h:
        raise RtError(-1, err.value.decode())
    try:
        info = (C.c_int64 * 12)()
        ok = C.c_int(0)
        rc = H.rth_scene_flatten_counts(C.c_void_p(h), info, C.byref(ok), err, len(err))
        if rc != RT_OK:
            raise RtError(rc, err.value.decode())
        keys = ("prims", "instances", "media", "top_level", "spheres", "rects", "groups", "materials", "textures",
                "perlin_tables", "images", "moving_spheres")
        out = dict(zip(keys, (int(x) for x in info)))
        out["ids_ok"] = bool(ok.value)
        return out
    finally:
        H.rth_scene_destroy(C.c_void_p(h))


def flattened_image(name, index=0, seed=1, earthmap=None):
    """The index-th image texture of a named scene as the C++ mirror decoded it: (h, w, 3) uint8 (no GPU needed)."""
    H = host()
    err = C.create_string_buffer(512)
    h = H.rth_scene_build(name.encode(), seed, str(earthmap or EARTHMAP).encode(), 0, err, len(err))
    if not h:
        raise RtError(-1, err.value.decode())
    try:
        w, hh = C.c_uint32(), C.c_uint32()
        rc = H.rth_scene_flattened_image(C.c_void_p(h), index, C.byref(w), C.byref(hh), None, 0)
        if rc:
            raise RtError(rc, "rth_scene_flattened_image")
        a = np.empty((hh.value, w.value, 3), dtype=np.uint8)
        rc = H.rth_scene_flattened_image(C.c_void_p(h), index, C.byref(w), C.byref(hh), a.ctypes.data, a.nbytes)
        if rc:
            raise RtError(rc, "rth_scene_flattened_image")
        return a
    finally:
        H.rth_scene_destroy(C.c_void_p(h))


def write_image(path, rgb8, fmt="png"):
    """(h, w, 3) uint8, row 0 = top -> PNG or binary PPM (stands in for main.rs:486-497)."""
    a = np.ascontiguousarray(rgb8.cpu().numpy() if _is_torch(rgb8) else rgb8, dtype=np.uint8)
    h, w, ch = a.shape
    if ch != 3:
        raise ValueError("write_image wants (h, w, 3)")
    rc = core().rt_image_write(str(path).encode(), w, h, a.ctypes.data, {"png": 0, "ppm": 1}[fmt])
    if rc:
        raise RtError(rc, f"rt_image_write({path})")


def read_image(path):
    """Baseline JPEG or binary PPM -> (h, w, 3) uint8, row 0 = top: what ImageTexture::new(path) holds
    (texture.rs:89-93).  Decoded by the library on the host."""
    w, h = C.c_uint32(), C.c_uint32()
    rc = core().rt_image_read(str(path).encode(), C.byref(w), C.byref(h), None, 0)
    if rc:
        raise RtError(rc, f"rt_image_read({path})")
    a = np.empty((h.value, w.value, 3), dtype=np.uint8)
    rc = core().rt_image_read(str(path).encode(), C.byref(w), C.byref(h), a.ctypes.data, a.nbytes)
    if rc:
        raise RtError(rc, f"rt_image_read({path})")
    return a


def save_accum(path, accum, samples_done, seed, tag=0):
    """Checkpoint of a progressive render: f32 sample sums (h, w, 3) + how many samples they hold + the seed + a
    caller's 64-bit tag (hash of scene, camera, depth ...).  Written to a temporary file and renamed."""
    a = np.ascontiguousarray(accum.cpu().numpy() if _is_torch(accum) else accum, dtype=np.float32)
    h, w, _ = a.shape
    rc = core().rt_accum_write_tagged(str(path).encode(), w, h, samples_done, seed, int(tag), a.ctypes.data)
    if rc:
        raise RtError(rc, f"rt_accum_write_tagged({path})")


def load_accum(path, with_tag=False):
    """-> (accum (h, w, 3) float32, samples_done, seed[, tag]); continue with render(..., sample0=samples_done, accum=accum)."""
    L = core()
    w, h, done, seed, tag = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    rc = L.rt_accum_read_tagged(str(path).encode(), C.byref(w), C.byref(h), C.byref(done), C.byref(seed), C.byref(tag), None, 0)
    if rc:
        raise RtError(rc, f"rt_accum_read_tagged({path})")
    a = np.empty((h.value, w.value, 3), np.float32)
    rc = L.rt_accum_read_tagged(str(path).encode(), None, None, None, None, None, a.ctypes.data, a.size)
    if rc:
        raise RtError(rc, f"rt_accum_read_tagged({path})")
    return (a, done.value, seed.value, tag.value) if with_tag else (a, done.value, seed.value)


class Context:
    """One GPU, one stream, one committed scene."""

    def __init__(self, device=0, max_paths=None):
        self.L = core()
        h = C.c_void_p()
        rc = self.L.rt_ctx_create(device, C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, "rt_ctx_create failed (no GPU visible?)")
        self.h = h
        self.device = device
        self.scene_handle = None
        self.scene_name = None
        if max_paths:
            self._check(self.L.rt_ctx_set_max_paths(self.h, int(max_paths)))

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, (self.L.rt_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "scene_handle", None):
            host().rth_scene_destroy(self.scene_handle)
            self.scene_handle = None
        if getattr(self, "h", None):
            self.L.rt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._check(self.L.rt_ctx_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    def set_pipeline(self, pipeline):
        """-1 pick per scene, 0 split (raygen + trace + shade), 1 fused (one kernel per bounce)."""
        self._check(self.L.rt_ctx_set_pipeline(self.h, pipeline))

    def set_bvh_width(self, width):
        """Children per inner node the kernels walk from the next commit on: 2, 4, or -1 = by tree size (same hits either way)."""
        self._check(self.L.rt_ctx_set_bvh_width(self.h, width))

    def set_fuse_boxes(self, on):
        """From the next commit on: keep the six rects of a Box2 as one record and leaf (default) or as six rects (same hits)."""
        self._check(self.L.rt_ctx_set_fuse_boxes(self.h, 1 if on else 0))

    def set_builder(self, builder):
        """World-level BVH builder of the next commit: -1 by size, 0 host (binned SAH), 1 device (linear BVH)."""
        self._check(self.L.rt_ctx_set_builder(self.h, builder))

    def read_bvh(self):
        """(nodes, prims, root_ref): the committed tree as numpy structured arrays (see rt_scene_read_bvh)."""
        info = self.info()
        node_t = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3), ("c0", "<u4"),
                           ("c1", "<u4"), ("pad", "<u4", 2)])
        prim_t = np.dtype([("a", "<f4", 4), ("b0", "<f4"), ("info", "<u4"), ("id", "<u4"), ("aux", "<u4")])
        nodes = np.zeros(info["n_nodes"], node_t)
        prims = np.zeros(info["n_prims"], prim_t)
        root = C.c_uint32(0)
        self._check(self.L.rt_scene_read_bvh(self.h, nodes.ctypes.data, nodes.nbytes, prims.ctypes.data, prims.nbytes, C.byref(root)))
        return nodes, prims, root.value

    def reserve(self, n_paths):
        """Allocate the path queues for renders of up to n_paths camera samples ahead of the first render."""
        self._check(self.L.rt_ctx_reserve(self.h, int(n_paths)))

    def set_tail_paths(self, n):
        """Live-path count at or below which one launch finishes a batch (0 = one launch per bounce to the end)."""
        self._check(self.L.rt_ctx_set_tail_paths(self.h, n))

    def synchronize(self):
        self._check(self.L.rt_ctx_synchronize(self.h))

    def profile(self, events=True, count_visits=False):
        self._check(self.L.rt_profile_enable(self.h, (1 if events else 0) | (2 if count_visits else 0)))

    # ---- scenes ------------------------------------------------------------------
    def load_scene(self, name, seed=1, n_synth=0, earthmap=None):
        """Build one of the reference's scenes through the C++ mirror API, flatten, upload, commit."""
        H = host()
        if self.scene_handle:
            H.rth_scene_destroy(self.scene_handle)
            self.scene_handle = None
        err = C.create_string_buffer(512)
        path = str(earthmap or EARTHMAP).encode()
        h = H.rth_scene_build(name.encode(), seed, path, n_synth, err, len(err))
        if not h:
            raise RtError(-1, err.value.decode())
        self.scene_handle = C.c_void_p(h)
        rc = H.rth_scene_upload(self.scene_handle, self.h, err, len(err))
        if rc != RT_OK:
            raise RtError(rc, err.value.decode())
        self.scene_name = name
        cam = (C.c_float * 11)()
        H.rth_scene_preset(self.scene_handle, cam)
        self.cam_origin = np.array(cam[0:3], np.float32)
        self.cam_target = np.array(cam[3:6], np.float32)
        self.vfov_deg = float(cam[6])
        self.aperture = float(cam[7])
        self.background = np.array(cam[8:11], np.float32)
        cnt = (C.c_int64 * 4)()
        H.rth_scene_counts(self.scene_handle, cnt)
        self.n_prims, self.n_instances, self.n_media, self.n_top = (int(x) for x in cnt)
        return self.info()

    def set_scene(self, spheres=None, rects=None, groups=(), materials=(), textures=(), perlin=None, images=None,
                  shutter=(0.0, 1.0)):
        """Upload a scene given directly as the flat arrays of rt_scene_set_* (what a foreign caller of the C
        ABI sends) and commit it.

        spheres: dict c0r (n,4) [, c1 (n,3), t01 (n,2), moving (n,)], mat (n,), id (n,) [, group (n,)]
        rects:   dict axis (n,), k (n,), bounds (n,4), mat (n,), id (n,) [, group (n,)]
        groups / materials / textures: sequences of rt_group / rt_material / rt_texture
        perlin:  {table: (vec (256,3) f32, perm (3,256) i32)} ; images: {table: (h,w,3) u8}
        """
        L = self.L
        keep = []  # keep converted arrays alive across the calls

        def arr(d, k, dt, shape=None):
            if d is None or d.get(k) is None:
                return None
            a = np.ascontiguousarray(d[k], dtype=dt)
            if shape is not None:
                a = a.reshape(shape)
            keep.append(a)
            return a

        def carr(seq, ctype):
            a = (ctype * max(len(seq), 1))(*seq)
            keep.append(a)
            return a

        self._check(L.rt_scene_clear(self.h))
        self._check(L.rt_scene_set_textures(self.h, len(textures), carr(list(textures), rt_texture)))
        self._check(L.rt_scene_set_materials(self.h, len(materials), carr(list(materials), rt_material)))
        for t, (vec, perm) in (perlin or {}).items():
            v = np.ascontiguousarray(vec, np.float32).reshape(256, 3)
            p = np.ascontiguousarray(perm, np.int32).reshape(3, 256)
            self._check(L.rt_scene_set_perlin(self.h, t, _ptr(v), _ptr(p)))
        for t, img in (images or {}).items():
            im = np.ascontiguousarray(img, np.uint8)
            self._check(L.rt_scene_set_image(self.h, t, im.shape[1], im.shape[0], _ptr(im)))
        self._check(L.rt_scene_set_groups(self.h, len(groups), carr(list(groups), rt_group)))
        n = 0 if spheres is None else len(spheres["id"])
        self._check(L.rt_scene_set_spheres(
            self.h, n, _ptr(arr(spheres, "c0r", np.float32, (n, 4))), _ptr(arr(spheres, "c1", np.float32, (n, 3))),
            _ptr(arr(spheres, "t01", np.float32, (n, 2))), _ptr(arr(spheres, "moving", np.uint8)),
            _ptr(arr(spheres, "mat", np.uint32)), _ptr(arr(spheres, "id", np.uint32)), _ptr(arr(spheres, "group", np.uint32))))
        n = 0 if rects is None else len(rects["id"])
        self._check(L.rt_scene_set_rects(
            self.h, n, _ptr(arr(rects, "axis", np.uint8)), _ptr(arr(rects, "k", np.float32)),
            _ptr(arr(rects, "bounds", np.float32, (n, 4))), _ptr(arr(rects, "mat", np.uint32)),
            _ptr(arr(rects, "id", np.uint32)), _ptr(arr(rects, "group", np.uint32))))
        self._check(L.rt_scene_commit(self.h, shutter[0], shutter[1]))
        self.scene_name = "<raw>"
        self.background = np.zeros(3, np.float32)
        return self.info()

    def preset_camera(self, aspect):
        cam = rt_camera()
        host().rth_scene_camera(self.scene_handle, aspect, C.byref(cam))
        return cam

    def info(self):
        inf = rt_scene_info()
        self._check(self.L.rt_scene_get_info(self.h, C.byref(inf)))
        return inf.as_dict()

    # ---- work ----------------------------------------------------------------------
    def intersect(self, rays, tmin=0.005, tmax=float("inf"), seed=1, outputs=None, noalloc=False):
        """rays: (7, n) float32 numpy array or torch CUDA tensor.  Returns a dict of arrays of the same kind.
        noalloc: go through rt_intersect_batch_noalloc (at most 65536 rays, no device allocation inside the call)."""
        if not _is_torch(rays):
            rays = np.ascontiguousarray(rays, dtype=np.float32)
        _check_array(rays, "rays", "float32", (7, None), self.device)
        n = rays.shape[1]
        if _is_torch(rays):
            import torch
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=rays.device)
            i32, f32, u8 = torch.int32, torch.float32, torch.uint8
        else:
            mk = lambda shape, dt: np.empty(shape, dtype=dt)
            i32, f32, u8 = np.int32, np.float32, np.uint8
        out = dict(prim=mk((n,), i32), inst=mk((n,), i32), t=mk((n,), f32), normal=mk((n, 3), f32), front=mk((n,), u8),
                   uv=mk((n, 2), f32), point=mk((n, 3), f32))
        if outputs is not None:
            out = {k: v for k, v in out.items() if k in outputs}
        g = lambda k: _ptr(out.get(k))
        fn = self.L.rt_intersect_batch_noalloc if noalloc else self.L.rt_intersect_batch
        self._check(fn(self.h, n, *[_ptr(rays[i]) for i in range(7)], tmin, tmax, seed,
                                              g("prim"), g("inst"), g("t"), g("normal"), g("front"), g("uv"), g("point")))
        return out

    def render(self, width, height, cam, spp, depth=50, seed=1, background=None, sample0=0, accum=None,
               tile_size=DEFAULT_TILE, rank=0, nranks=1):
        """Adds samples [sample0, sample0+spp) into accum ((h,w,3) f32; numpy or torch CUDA) and returns
        (accum, stats dict)."""
        if accum is None:
            accum = np.zeros((height, width, 3), dtype=np.float32)
        _check_array(accum, "accum", "float32", (height, width, 3), self.device)
        p = rt_render_params()
        p.width, p.height = width, height
        p.sample0, p.sample1 = sample0, sample0 + spp
        p.max_depth = depth
        bg = self.background if background is None else background
        p.background = (C.c_float * 3)(*[float(x) for x in bg])
        p.seed = seed
        p.tile_size, p.rank, p.nranks = tile_size, rank, nranks
        st = rt_render_stats()
        self._check(self.L.rt_render(self.h, C.byref(cam), C.byref(p), _ptr(accum), C.byref(st)))
        return accum, st.as_dict()

    # ---- frame exchange ------------------------------------------------------------
    def comm_create(self, unique_id, rank, nranks):
        """Collective: every rank calls it with the same 128 bytes (comm_unique_id() of rank 0)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.L.rt_comm_create(self.h, buf, rank, nranks))

    def comm_destroy(self):
        self._check(self.L.rt_comm_destroy(self.h))

    def gather_frame(self, accum, tile_size=DEFAULT_TILE, out=None):
        """One collective per frame: (h, w, 3) with this rank's tiles filled in -> the complete frame on every rank."""
        _check_array(accum, "accum", "float32", (None, None, 3), self.device)
        h, w = int(accum.shape[0]), int(accum.shape[1])
        if out is None:
            if _is_torch(accum):
                import torch
                out = torch.empty_like(accum)
            else:
                out = np.empty_like(accum)
        _check_array(out, "out", "float32", (h, w, 3), self.device)
        self._check(self.L.rt_gather_frame(self.h, w, h, tile_size, _ptr(accum), _ptr(out)))
        return out

    def pack_owned(self, accum, packed, tile_size, rank, nranks):
        h, w = int(accum.shape[0]), int(accum.shape[1])
        self._check(self.L.rt_pack_owned(self.h, w, h, tile_size, rank, nranks, _ptr(accum), _ptr(packed)))

    def scatter_gathered(self, gathered, frame, tile_size, nranks):
        h, w = int(frame.shape[0]), int(frame.shape[1])
        self._check(self.L.rt_scatter_gathered(self.h, w, h, tile_size, nranks, _ptr(gathered), _ptr(frame)))

    def resolve_u8(self, accum, spp, transfer=0):
        """transfer 0: linear (main.rs:307-313); 1: square root first (the commented-out write_color, main.rs:35-42)."""
        _check_array(accum, "accum", "float32", (None, None, 3), self.device)
        if _is_torch(accum):
            import torch
            out = torch.empty(accum.shape, dtype=torch.uint8, device=accum.device)
        else:
            out = np.empty(accum.shape, dtype=np.uint8)
        n_pixels = int(np.prod(accum.shape)) // 3
        self._check(self.L.rt_resolve_u8_transfer(self.h, _ptr(accum), n_pixels, spp, transfer, _ptr(out)))
        return out
