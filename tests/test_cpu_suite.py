"""CPU suite (no GPU): oracle against the committed golden vectors, the C ABI surface, host logic
(scene flattening through the C++ mirror API, tile ownership, BVH builder invariants) and the
multi-rank framebuffer gather over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import orc
import rtamd

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
RAY_SCENES = ["random_world", "earth", "simple_light", "cornell_box", "cornell_smoke", "final_scene"]
IMAGE_SCENES = ["random_world", "two_spheres", "two_perlin_spheres", "earth", "simple_light", "cornell_box",
                "cornell_smoke", "final_scene"]


# ---- oracle vs golden vectors ------------------------------------------------------------------
@pytest.mark.parametrize("name", RAY_SCENES)
def test_oracle_reproduces_golden_ray_batches(name):
    g = np.load(GOLDEN / f"rays_{name}.npz")
    sc = orc.Scene(name, seed=1)
    for mode in (0, 1):
        out = sc.intersect(g["rays"], seed=3, mode=mode)
        assert np.array_equal(out["prim"], g["prim"]) and np.array_equal(out["inst"], g["inst"])
        assert np.array_equal(out["t"], g["t"]) and np.array_equal(out["front"], g["front"])
        assert np.array_equal(out["normal"], g["normal"]) and np.array_equal(out["point"], g["point"])
        assert np.array_equal(out["uv"], g["uv"], equal_nan=True)


@pytest.mark.parametrize("name", IMAGE_SCENES)
def test_oracle_reproduces_golden_images(name):
    g = np.load(GOLDEN / f"image_{name}.npz")
    sc = orc.Scene(name, seed=1)
    h, w, _ = g["sum_rgb"].shape
    img, rays = sc.render(w, h, 1.0, int(g["spp"]), depth=50, seed=1, mode=1)
    assert rays == int(g["rays"])
    assert np.array_equal(img, g["sum_rgb"])
    # the linear Scene path renders the same image: same closest hits, same RNG streams
    img0, rays0 = sc.render(w, h, 1.0, int(g["spp"]), depth=50, seed=1, mode=0)
    assert rays0 == rays and np.array_equal(img0, img)


def test_oracle_render_is_partition_and_sample_split_invariant():
    sc = orc.Scene("cornell_smoke", seed=1)
    w = h = 24
    full, rays = sc.render(w, h, 1.0, 6, seed=2, mode=1)
    parts = np.zeros_like(full)
    r2 = 0
    for tile in [(0, 0, 12, 24), (12, 0, 24, 10), (12, 10, 24, 24)]:
        _, r = sc.render(w, h, 1.0, 6, seed=2, mode=1, tile=tile, sum_rgb=parts)
        r2 += r
    assert r2 == rays and np.array_equal(parts, full)
    split = np.zeros_like(full)
    sc.render(w, h, 1.0, 2, seed=2, mode=1, s0=0, sum_rgb=split)
    sc.render(w, h, 1.0, 4, seed=2, mode=1, s0=2, sum_rgb=split)
    assert np.array_equal(split, full)


@pytest.mark.parametrize("name", ["final_scene", "random_world", "cornell_smoke"])
def test_fast_oracle_build_gives_the_checker_builds_bits(name):
    """bench.py times the oracle built with -O3 and without the visit counters (oracle/Makefile: liboracle_fast.so) as
    the CPU baseline; the tests check against the -O2 build.  Both are -ffp-contract=off, so images, ray counts and hit
    records must be identical -- otherwise the timed program would not be the checked one."""
    w, h, spp = 96, 96, 3
    a = orc.Scene(name, seed=1)
    b = orc.Scene(name, seed=1, lib_path=orc._FAST_PATH)
    for mode in (0, 1):
        img_a, rays_a = a.render(w, h, 1.0, spp, depth=50, seed=3, mode=mode)
        img_b, rays_b = b.render(w, h, 1.0, spp, depth=50, seed=3, mode=mode)
        assert rays_a == rays_b and np.array_equal(img_a, img_b)
    rays, valid = a.gen_rays(64, 64, 1.0, seed=2, which=1)
    ha, hb = a.intersect(rays, seed=2), b.intersect(rays, seed=2)
    for k in ("prim", "inst", "t", "normal", "front", "uv", "point"):
        assert np.array_equal(ha[k][valid], hb[k][valid], equal_nan=True), k
    a.close()
    b.close()


def test_earthmap_fixture():
    import hashlib
    img = orc.earthmap()
    assert img.shape == (512, 1024, 3)
    assert hashlib.sha256(img.tobytes()).hexdigest() == "a8cdc92a168d554ddc693785d31f5e251063724f44099571d7fbce3b43d44c45"


# ---- C ABI surface (no compute calls) -------------------------------------------------------------
def declared_functions():
    text = (ROOT / "include" / "rtamd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(rtamd.ABI_SYMBOLS) == names, "rtamd.ABI_SYMBOLS and include/rtamd.h disagree"
    lib = C.CDLL(str(rtamd.LIB_DIR / "librtamd.so"))
    for n in names:
        assert hasattr(lib, n), f"librtamd.so does not export {n}"


def _header_text():
    text = (ROOT / "include" / "rtamd.h").read_text()
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _c_kind(ctype):
    """A C parameter or field type reduced to what its Rust counterpart must be."""
    t = " ".join(ctype.replace("const", " ").split())
    ptr = t.count("*")
    base = t.replace("*", "").strip()
    scalar = {"int": "i32", "uint8_t": "u8", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "double": "f64",
              "char": "c_char", "void": "c_void", "size_t": "usize"}.get(base, base)
    return ("*" * ptr) + scalar


def _rust_kind(rtype):
    t = rtype.replace("*const", "*").replace("*mut", "*").replace(" ", "")
    return t.replace("c_int", "i32")


def test_rust_declarations_match_the_header():
    """bindings/rust/src/ffi.rs cannot be compiled here (no Rust toolchain), so it is checked as text against
    include/rtamd.h: the same functions, each with the same number of parameters of the same kinds in the same order,
    and the same struct fields in the same order with the same element types and array lengths."""
    header = _header_text()
    rust = re.sub(r"//[^\n]*", "", (ROOT / "bindings" / "rust" / "src" / "ffi.rs").read_text())
    # ---- functions
    c_funcs = {}
    for ret, name, params in re.findall(r"\b(int|void|const char\s*\*)\s+(rt_\w+)\s*\(([^;{]*?)\)\s*;", header):
        kinds = []
        for prm in (x.strip() for x in params.split(",")):
            if prm in ("", "void"):
                continue
            m = re.match(r"(.*?)(\w+)\s*(\[\d*\])?$", prm)
            kinds.append(_c_kind(m.group(1) + ("*" if m.group(3) else "")))
        c_funcs[name] = kinds
    r_funcs = {}
    for name, params in re.findall(r"pub fn (rt_\w+)\s*\(([^)]*)\)", rust):
        r_funcs[name] = [_rust_kind(x.split(":", 1)[1]) for x in params.split(",") if ":" in x]
    assert sorted(c_funcs) == declared_functions()
    assert sorted(r_funcs) == sorted(c_funcs), f"only in the header: {sorted(set(c_funcs) - set(r_funcs))}; only in ffi.rs: {sorted(set(r_funcs) - set(c_funcs))}"
    for name, kinds in c_funcs.items():
        assert r_funcs[name] == kinds, f"{name}: header {kinds} against ffi.rs {r_funcs[name]}"
    # ---- structs
    c_structs = {}
    for body, name in re.findall(r"typedef struct \w+\s*\{(.*?)\}\s*(rt_\w+)\s*;", header, flags=re.S):
        fields = []
        for decl in (d.strip() for d in body.split(";")):
            if not decl:
                continue
            ctype, rest = decl.split(None, 1)
            for item in rest.split(","):
                m = re.match(r"\s*(\w+)\s*(?:\[(\d+)\])?\s*$", item)
                fields.append((m.group(1), _c_kind(ctype), int(m.group(2) or 0)))
        c_structs[name] = fields
    r_structs = {}
    for name, body in re.findall(r"pub struct (rt_\w+)\s*\{(.*?)\}", rust, flags=re.S):
        fields = []
        for fname, ftype in re.findall(r"pub (\w+)\s*:\s*([^,\n]+)", body):
            m = re.match(r"\[(\w+);\s*(\d+)\]", ftype.strip())
            fields.append((fname, m.group(1), int(m.group(2))) if m else (fname, ftype.strip(), 0))
        r_structs[name] = fields
    assert set(c_structs) >= {"rt_material", "rt_texture", "rt_group", "rt_scene_info", "rt_camera", "rt_render_params", "rt_render_stats"}
    for name, fields in c_structs.items():
        assert r_structs.get(name) == fields, f"{name}: header {fields} against ffi.rs {r_structs.get(name)}"
    # ---- constants the two sides must agree on
    for cname in ("RT_NOALLOC_MAX_RAYS", "RT_DEFAULT_TILE"):
        c_val = int(re.search(rf"#define {cname} (\d+)u", header).group(1))
        assert int(re.search(rf"pub const {cname}: u32 = (\d+);", rust).group(1)) == c_val
    for cname, val in re.findall(r"\b(RT_(?:ERR_\w+|OK))\s*=\s*(-?\d+)", header):
        assert re.search(rf"pub const {cname}: c_int = {val};", rust), cname


def test_struct_layouts_match_header(tmp_path):
    """The header compiles as plain C and the ctypes mirrors have the sizes the C compiler gives."""
    names = ["rt_camera", "rt_material", "rt_texture", "rt_group", "rt_render_params", "rt_render_stats", "rt_scene_info"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rtamd.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), "-o", str(exe), str(src)], check=True)
    sizes = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for n in names:
        assert C.sizeof(getattr(rtamd, n)) == int(sizes[n]), n
    assert int(sizes["rt_material"]) == 32 and int(sizes["rt_texture"]) == 32 and int(sizes["rt_camera"]) == 96


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(rtamd.RtError) as e:
        rtamd.Context(0)
    assert e.value.code == -5  # RT_ERR_NO_DEVICE
    assert rtamd.core().rt_scene_clear(None) == -1  # RT_ERR_INVALID on a null context
    assert rtamd.core().rt_last_error(None) == b"null context"


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer-rs_amd/ or include/ may name it."""
    for path in list((ROOT / "raytracer-rs_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if path.is_file() and path.suffix in (".py", ".cpp", ".hpp", ".h", ".hip", "") and "lib" not in path.parts and "build" not in path.parts:
            text = path.read_text(errors="ignore")
            assert "liboracle" not in text and "rt_oracle" not in text and "import orc" not in text, path


# ---- host logic ---------------------------------------------------------------------------------------
def test_owned_pixels_partition_the_frame():
    w, h, ts = 100, 70, 32
    for world in (1, 2, 3, 8):
        seen = np.zeros(w * h, np.int32)
        counts = []
        for r in range(world):
            px = rtamd.owned_pixels(w, h, ts, r, world)
            seen[px] += 1
            counts.append(len(px))
            tiles = (px // w // ts) * ((w + ts - 1) // ts) + (px % w) // ts
            assert np.all(tiles % world == r)
        assert np.all(seen == 1)
        assert sum(counts) == w * h
    # tile-major order: the first tile's rows come first
    px = rtamd.owned_pixels(w, h, ts, 0, 1)
    assert px[:3].tolist() == [0, 1, 2] and px[32] == w and px[32 * 32] == 32


@pytest.mark.parametrize("name", rtamd.SCENES[:-1])
def test_flattened_scene_matches_oracle_inventory(name):
    sc = orc.Scene(name, seed=4)
    fc = rtamd.flatten_counts(name, seed=4)
    assert fc["ids_ok"]
    assert (fc["prims"], fc["instances"], fc["media"], fc["top_level"]) == (sc.n_prims, sc.n_instances, sc.n_media, sc.n_top)
    assert fc["spheres"] + fc["rects"] == fc["prims"]


def test_flattened_scene_details():
    fc = rtamd.flatten_counts("cornell_smoke")
    assert fc["rects"] == 18 and fc["groups"] == 2 and fc["spheres"] == 0
    fc = rtamd.flatten_counts("final_scene")
    assert fc["rects"] == 2401 and fc["spheres"] == 1008 and fc["groups"] == 3 and fc["moving_spheres"] == 1
    assert fc["perlin_tables"] == 1 and fc["images"] == 1
    fc = rtamd.flatten_counts("random_world", seed=9)
    sc = orc.Scene("random_world", seed=9)
    assert fc["prims"] == sc.n_prims and 0.6 * fc["prims"] < fc["moving_spheres"] < 0.95 * fc["prims"]
    fc = rtamd.flatten_counts("synth_spheres", n_synth=1000)
    assert fc["spheres"] == 1001 and fc["materials"] == 1001
    with pytest.raises(rtamd.RtError):
        rtamd.flatten_counts("no_such_scene")
    with pytest.raises(rtamd.RtError):
        rtamd.flatten_counts("earth", earthmap="/nonexistent/earthmap.ppm")  # texture.rs:91 panics; here an error


def test_camera_mirror_matches_oracle():
    for name in ("random_world", "cornell_box", "final_scene", "simple_light"):
        sc = orc.Scene(name)
        for aspect in (1.0, 16.0 / 9.0):
            want = orc.camera_new(sc.cam_origin, sc.cam_target, (0, 1, 0), np.float32(sc.vfov_deg) * np.float32(np.pi) / np.float32(180),
                                  aspect, sc.aperture, 10.0, 0.0, 1.0)
            got = rtamd.camera_new(sc.cam_origin, sc.cam_target, (0, 1, 0), np.float32(sc.vfov_deg) * np.float32(np.pi) / np.float32(180),
                                   aspect, sc.aperture, 10.0, 0.0, 1.0).as_array()
            assert np.array_equal(got, want)


@pytest.fixture(scope="module")
def bvh_check(tmp_path_factory):
    exe = tmp_path_factory.mktemp("bvh") / "bvh_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "cpp" / "bvh_check.cpp"),
                    str(ROOT / "raytracer-rs_amd" / "csrc" / "rt_bvh.cpp")], check=True)
    return exe


@pytest.mark.parametrize("n,seed,leaf", [(1, 1, 4), (2, 1, 4), (7, 2, 4), (1000, 3, 1), (20001, 5, 4), (50000, 7, 16)])
def test_bvh_builder_invariants(bvh_check, n, seed, leaf):
    r = subprocess.run([str(bvh_check), str(n), str(seed), str(leaf)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), r.stdout[-500:]


def test_nesting_the_device_does_not_take_is_refused_with_a_message(tmp_path):
    """TransformedObject<T> / ConstantMedium<T> take any Hittable in the reference (shapes.rs:26-41, 624-646); the device
    takes up to four stacked transforms and top-level media.  The mirror API must flatten the former into parent-linked
    groups and refuse anything else, naming the reason."""
    exe = tmp_path / "nesting_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "cpp" / "nesting_check.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "OK", r.stdout[-500:]


# ---- N > 1: framebuffer gather over gloo, world size 2 -----------------------------------------------
GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["RT_PKG"])
import numpy as np, torch, torch.distributed as dist
import rtamd
from rtamd.dist import TileGather
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h, ts = 70, 45, 16
frame = torch.arange(w * h * 3, dtype=torch.float32).reshape(h, w, 3) * 0.5 + 1.0   # what a 1-rank render would hold
mine = torch.zeros_like(frame)
idx = torch.from_numpy(rtamd.owned_pixels(w, h, ts, rank, world).astype(np.int64))
mine.view(-1, 3)[idx] = frame.view(-1, 3)[idx]                                          # this rank rendered only its tiles
g = TileGather(w, h, rank, world, "cpu", ts)
full = g.gather(mine)
assert torch.equal(full, frame), "gathered frame differs from the single-rank frame"
full2 = g.gather(mine)                                                                  # reusable across frames
assert torch.equal(full2, frame)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_tile_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, RT_PKG=str(ROOT / "raytracer-rs_amd"), MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-2000:]


# ---- output files (host only) -------------------------------------------------------------------------
def test_image_write_png_and_ppm_round_trip(tmp_path):
    """rt_image_write stands in for main.rs:486-497.  The PNG is read back with Pillow, the PPM by hand."""
    from PIL import Image
    rng = np.random.default_rng(1)
    for (h, w) in ((1, 1), (7, 13), (225, 400)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[h // 2:, :, :] = 77  # something compressible too
        rtamd.write_image(tmp_path / "a.png", img, "png")
        back = np.asarray(Image.open(tmp_path / "a.png"))
        assert back.shape == img.shape and back.dtype == np.uint8 and np.array_equal(back, img)
        rtamd.write_image(tmp_path / "a.ppm", img, "ppm")
        raw = (tmp_path / "a.ppm").read_bytes()
        head = f"P6\n{w} {h}\n255\n".encode()
        assert raw.startswith(head) and np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3), img)
    with pytest.raises(rtamd.RtError):
        rtamd.write_image(tmp_path / "no_such_dir" / "a.png", img, "png")


# ---- texture files (host only) -------------------------------------------------------------------------
def _smooth_noise(rng, h, w):
    """Something photograph-like: low-frequency colour plus a little noise (pure noise clips everywhere)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 100 * np.sin(x / 7.0 + c) * np.cos(y / 5.0 - c) for c in range(3)], axis=-1)
    return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(1, 1), (2, 2), (16, 1), (3, 130), (21, 37), (48, 64), (45, 77)])
def test_image_read_decodes_baseline_jpeg_as_libjpeg_does(tmp_path, size):
    """rt_image_read stands in for image::open(path).into_rgb8() (texture.rs:89-93).  Files written by Pillow at every
    chroma sampling the decoder takes, with and without restart markers and optimised tables, greyscale too: the
    decode equals Pillow's (libjpeg's accurate IDCT and triangle upsampling) byte for byte."""
    from PIL import Image
    h, w = size
    rng = np.random.default_rng(h * 1000 + w)
    im = Image.fromarray(_smooth_noise(rng, h, w))
    path = tmp_path / "t.jpg"
    for sub in (0, 1, 2):
        for q in (25, 75, 95):
            for extra in ({}, {"restart_marker_blocks": 3}, {"restart_marker_rows": 1}, {"optimize": True}):
                im.save(path, quality=q, subsampling=sub, **extra)
                want = np.asarray(Image.open(path).convert("RGB"))
                got = rtamd.read_image(path)
                assert got.shape == want.shape and np.array_equal(got, want), (size, sub, q, extra)
    im.convert("L").save(path, quality=80)
    assert np.array_equal(rtamd.read_image(path), np.asarray(Image.open(path).convert("RGB")))


def test_image_read_errors_and_ppm(tmp_path):
    """A missing file is an I/O error, a file outside the decoder's formats unsupported (the reference panics on both,
    texture.rs:91); binary PPM with comments in the header reads back as written."""
    from PIL import Image
    rng = np.random.default_rng(5)
    img = _smooth_noise(rng, 19, 23)
    with pytest.raises(rtamd.RtError) as e:
        rtamd.read_image(tmp_path / "nothing.jpg")
    assert e.value.code == -6
    Image.fromarray(img).save(tmp_path / "p.jpg", progressive=True)
    Image.fromarray(img).save(tmp_path / "p.png")
    (tmp_path / "junk.jpg").write_bytes(b"\xff\xd8\xff")
    for name in ("p.jpg", "p.png", "junk.jpg"):
        with pytest.raises(rtamd.RtError) as e:
            rtamd.read_image(tmp_path / name)
        assert e.value.code == -3, name
    (tmp_path / "c.ppm").write_bytes(b"P6\n# a comment\n23 19\n# another\n255\n" + img.tobytes())
    assert np.array_equal(rtamd.read_image(tmp_path / "c.ppm"), img)
    rtamd.write_image(tmp_path / "w.ppm", img, "ppm")
    assert np.array_equal(rtamd.read_image(tmp_path / "w.ppm"), img)
    # a file cut inside its scan still decodes to the right size (missing data reads as zero bits, as libjpeg
    # does); cut inside its headers it is refused
    Image.fromarray(img).save(tmp_path / "whole.jpg")
    raw = (tmp_path / "whole.jpg").read_bytes()
    scan = raw.index(b"\xff\xda")
    (tmp_path / "cut.jpg").write_bytes(raw[:scan + 60])
    assert rtamd.read_image(tmp_path / "cut.jpg").shape == img.shape
    (tmp_path / "cut2.jpg").write_bytes(raw[:scan - 20])
    with pytest.raises(rtamd.RtError):
        rtamd.read_image(tmp_path / "cut2.jpg")
    # a frame header that claims 65535 x 65535 pixels is refused at the header, not attempted (12 GB of planes)
    sof = raw.index(b"\xff\xc0")
    huge = bytearray(raw)
    huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"
    (tmp_path / "huge.jpg").write_bytes(bytes(huge))
    with pytest.raises(rtamd.RtError) as e:
        rtamd.read_image(tmp_path / "huge.jpg")
    assert e.value.code == -3


def test_mirror_image_texture_reads_a_jpeg(tmp_path):
    """ImageTexture::new("earthmap.jpg") (main.rs:153,286) through the C++ mirror: the earth scene built from a JPEG
    holds the texels libjpeg decodes.  With the reference's own file at hand, they are the committed fixture's."""
    from PIL import Image
    rng = np.random.default_rng(6)
    Image.fromarray(_smooth_noise(rng, 64, 128)).save(tmp_path / "map.jpg", quality=85, subsampling=2)
    got = rtamd.flattened_image("earth", earthmap=tmp_path / "map.jpg")
    assert np.array_equal(got, np.asarray(Image.open(tmp_path / "map.jpg").convert("RGB")))
    assert np.array_equal(rtamd.flattened_image("final_scene"), orc.earthmap())  # the PPM fixture, as before
    ref_file = Path("/root/reference/raytracer-rs/earthmap.jpg")
    if ref_file.exists():
        assert np.array_equal(rtamd.read_image(ref_file), orc.earthmap())
        assert np.array_equal(rtamd.flattened_image("earth", earthmap=ref_file), orc.earthmap())


def test_accum_checkpoint_round_trip(tmp_path):
    rng = np.random.default_rng(2)
    acc = rng.uniform(0, 50, (9, 17, 3)).astype(np.float32)
    acc[0, 0] = [np.inf, 0.0, 1e-30]
    rtamd.save_accum(tmp_path / "c.acc", acc, 37, 12345)
    back, done, seed = rtamd.load_accum(tmp_path / "c.acc")
    assert done == 37 and seed == 12345 and back.dtype == np.float32 and np.array_equal(back, acc)
    (tmp_path / "bad.acc").write_bytes(b"not a checkpoint")
    with pytest.raises(rtamd.RtError):
        rtamd.load_accum(tmp_path / "bad.acc")
    with pytest.raises(rtamd.RtError):
        rtamd.load_accum(tmp_path / "missing.acc")
    # the tag travels with the file; an overwrite goes through "<path>.tmp" + rename and leaves no temporary behind
    rtamd.save_accum(tmp_path / "c.acc", acc * 2, 38, 12345, tag=0xDEADBEEFCAFEF00D)
    back, done, seed, tag = rtamd.load_accum(tmp_path / "c.acc", with_tag=True)
    assert done == 38 and tag == 0xDEADBEEFCAFEF00D and np.array_equal(back, acc * 2)
    assert sorted(q.name for q in tmp_path.iterdir()) == ["bad.acc", "c.acc"]
    # a file of the first format (no tag field) still loads, with tag 0
    import struct
    (tmp_path / "v1.acc").write_bytes(b"RTACCUM1" + struct.pack("<4I", 17, 9, 5, 77) + acc.tobytes())
    back, done, seed, tag = rtamd.load_accum(tmp_path / "v1.acc", with_tag=True)
    assert (done, seed, tag) == (5, 77, 0) and np.array_equal(back, acc)
    # a write that cannot be completed leaves the previous checkpoint alone
    with pytest.raises(rtamd.RtError):
        rtamd.save_accum(tmp_path / "no_such_dir" / "c.acc", acc, 1, 1)


def test_binding_refuses_buffers_of_the_wrong_kind():
    """The C ABI takes raw pointers, so the binding checks dtype, shape and layout before it passes one on."""
    for bad in (np.zeros((4, 4, 3), np.float64), np.zeros((4, 4, 4), np.float32), np.zeros((4, 8, 3), np.float32)[:, ::2], [1.0]):
        with pytest.raises(ValueError):
            rtamd._check_array(bad, "accum", "float32", (4, 4, 3))
    rtamd._check_array(np.zeros((4, 4, 3), np.float32), "accum", "float32", (4, 4, 3))
    rtamd._check_array(np.zeros((7, 5), np.float32), "rays", "float32", (7, None))
    with pytest.raises(ValueError):
        rtamd._check_array(np.zeros((6, 5), np.float32), "rays", "float32", (7, None))
    import torch
    with pytest.raises(ValueError):
        rtamd._check_array(torch.zeros((4, 4, 3), dtype=torch.float64), "accum", "float32", (4, 4, 3))
    with pytest.raises(ValueError):
        rtamd._check_array(torch.zeros((4, 3, 4)).permute(0, 2, 1), "accum", "float32", (4, 4, 3))
    rtamd._check_array(torch.zeros((4, 4, 3)), "accum", "float32", (4, 4, 3))
