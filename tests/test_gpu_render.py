"""GPU parity tests of the render path (all through the C ABI): known answers on raw scenes, golden
vectors, image parity with the CPU oracle at equal spp, and size-independent properties at
BASELINE.json's full frame sizes.

Stated tolerances
  * hit ids bit-exact, t / normal within 1e-5 relative (north star);
  * images: the GPU follows the oracle's arithmetic and RNG streams, so almost every pixel agrees to
    rounding.  What differs are libm-vs-device transcendentals (sin, acos, atan2, pow, f64 log) and the
    order of the throughput product, which can flip a rare path at a checker edge, a texel boundary or a
    Schlick test.  Bar: per-sample RMSE <= 5e-3 (Monte Carlo noise at these sample counts is > 1e-1) and
    at least 99 % of the pixels within 1e-4 of the oracle.
"""
import ctypes as C
import math

import numpy as np
import pytest

import orc
import rtamd

pytestmark = pytest.mark.gpu

GOLDEN = orc.ROOT / "tests" / "golden"
ALL_SCENES = ["random_world", "two_spheres", "two_perlin_spheres", "earth", "simple_light", "cornell_box",
              "cornell_smoke", "final_scene"]
INF = float("inf")


def assert_images_close(gpu_sum, ref_sum, spp, name):
    d = (gpu_sum.astype(np.float64) - ref_sum.astype(np.float64)) / spp
    rmse = math.sqrt(float(np.mean(d * d)))
    close = np.abs(d).max(axis=2) <= 1e-4
    assert np.isfinite(gpu_sum).all(), name
    assert rmse <= 5e-3, f"{name}: per-sample RMSE {rmse}"
    assert close.mean() >= 0.99, f"{name}: only {close.mean():.4f} of the pixels within 1e-4"
    return rmse, close.mean()


# ---- known answers on raw scenes sent through rt_scene_set_* ----------------------------------------
def solid_lambertian():
    t = rtamd.rt_texture(kind=0, color=(C.c_float * 3)(0.5, 0.5, 0.5))
    m = rtamd.rt_material(kind=0, tex=0)
    return [m], [t]


def one_ray(o, d, time=0.0):
    return np.array([[o[0]], [o[1]], [o[2]], [d[0]], [d[1]], [d[2]], [time]], np.float32)


def test_kat_sphere_through_raw_abi(gpu_ctx):
    mats, texs = solid_lambertian()
    gpu_ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0]), materials=mats, textures=texs)
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)))  # K1
    assert r["prim"][0] == 0 and r["t"][0] == 4.0 and r["front"][0] == 1
    assert r["normal"][0].tolist() == [0.0, 0.0, -1.0] and r["point"][0].tolist() == [0.0, 0.0, -1.0]
    assert r["uv"][0] == pytest.approx([0.75, 0.5], abs=1e-6)
    r = gpu_ctx.intersect(one_ray((0, 0, 0), (0, 0, 1)))  # K2: from inside
    assert r["t"][0] == 1.0 and r["front"][0] == 0 and r["normal"][0].tolist() == [0.0, 0.0, -1.0]
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 2)))  # K3
    assert r["t"][0] == 2.0
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)), tmax=4.0)  # inclusive bound
    assert r["prim"][0] == 0
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)), tmax=3.9)
    assert r["prim"][0] == -1 and np.isinf(r["t"][0])
    r = gpu_ctx.intersect(one_ray((0, 3, -5), (0, 0, 1)))
    assert r["prim"][0] == -1


def test_kat_rects_moving_sphere_and_tie_rule(gpu_ctx):
    mats, texs = solid_lambertian()
    # K5: the Cornell light (main.rs:187); K6: moving sphere at mid shutter
    gpu_ctx.set_scene(
        rects=dict(axis=[1, 2], k=[554.0, 3.0], bounds=[[213, 227, 343, 332], [0, 10, 5, 20]], mat=[0, 0], id=[0, 1]),
        spheres=dict(c0r=[[100, 0, 0, 1]], c1=[[100, 2, 0]], t01=[[0, 1]], moving=[1], mat=[0], id=[2]),
        materials=mats, textures=texs)
    r = gpu_ctx.intersect(one_ray((278, 0, 279), (0, 1, 0)))
    assert r["prim"][0] == 0 and r["t"][0] == 554.0 and r["front"][0] == 0 and r["normal"][0].tolist() == [0.0, -1.0, 0.0]
    assert r["uv"][0] == pytest.approx([(278 - 213) / 130, (279 - 227) / 105], rel=1e-6)
    r = gpu_ctx.intersect(one_ray((0, 2, 15), (1, 0, 0)))
    assert r["prim"][0] == 1 and r["t"][0] == 3.0 and r["normal"][0].tolist() == [-1.0, 0.0, 0.0]
    r = gpu_ctx.intersect(one_ray((100, 1, -5), (0, 0, 1), 0.5))
    assert r["prim"][0] == 2 and r["t"][0] == 4.0
    # exact tie: two coincident rects, the later id wins (scene.rs:32-38)
    gpu_ctx.set_scene(rects=dict(axis=[0, 0, 0], k=[2.0, 2.0, 2.0], bounds=[[0, 0, 1, 1]] * 3, mat=[0, 0, 0], id=[0, 1, 2]),
                      materials=mats, textures=texs)
    r = gpu_ctx.intersect(one_ray((0.5, 0.5, 0), (0, 0, 1)))
    assert r["prim"][0] == 2 and r["t"][0] == 2.0


def box_rects(w, h, d, first_id, group):
    """Box2::full_box side order (shapes.rs:459-466)."""
    axis = [0, 0, 1, 1, 2, 2]
    k = [0.0, d, 0.0, h, 0.0, w]
    b = [[0, 0, w, h], [0, 0, w, h], [0, 0, w, d], [0, 0, w, d], [0, 0, h, d], [0, 0, h, d]]
    return dict(axis=axis, k=k, bounds=b, mat=[0] * 6, id=list(range(first_id, first_id + 6)), group=[group] * 6)


def test_kat_instance_and_medium(gpu_ctx):
    mats, texs = solid_lambertian()
    mats.append(rtamd.rt_material(kind=4, tex=0))
    g = rtamd.rt_group(kind=1, has_xform=1)
    g.m = (C.c_float * 12)(1, 0, 0, 10, 0, 1, 0, 0, 0, 0, 1, 0)      # translate x by 10
    g.minv = (C.c_float * 12)(1, 0, 0, -10, 0, 1, 0, 0, 0, 0, 1, 0)
    gpu_ctx.set_scene(rects=box_rects(2, 2, 2, 0, 1), groups=[g], materials=mats, textures=texs)
    r = gpu_ctx.intersect(one_ray((11, 1, -5), (0, 0, 1)))  # K7
    assert r["prim"][0] == 0 and r["inst"][0] == 0 and r["t"][0] == 5.0 and r["point"][0].tolist() == [11.0, 1.0, 0.0]
    assert r["front"][0] == 1 and r["normal"][0].tolist() == [0.0, 0.0, -1.0]
    r = gpu_ctx.intersect(one_ray((11, 1, 1), (0, 0, 1)))  # from inside: front_face still true (shapes.rs:75-86)
    assert r["prim"][0] == 1 and r["t"][0] == 1.0 and r["front"][0] == 1 and r["normal"][0].tolist() == [0.0, 0.0, -1.0]
    # K13: medium over a sphere; xi comes from stream (seed, pixel = ray index, sample 0), bounce 0, medium 0
    density = 0.5
    m = rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1.0 / density, mat=1)
    gpu_ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 2]], mat=[0], id=[0], group=[1]), groups=[m], materials=mats, textures=texs)
    rays = np.repeat(one_ray((0, 0, -5), (0, 0, 1)), 64, axis=1)
    r = gpu_ctx.intersect(rays, seed=11)
    for i in range(64):
        xi = orc.lib().orc_rng_medium_xi(11, i, 0, 0, 0)
        hit = -math.log(xi) / density
        if hit > 4.0:
            assert r["prim"][i] == -1
        else:
            assert r["prim"][i] == -2 and r["t"][i] == pytest.approx(3.0 + hit, rel=1e-6)
            assert r["normal"][i].tolist() == [1.0, 0.0, 0.0] and r["front"][i] == 1
    assert (r["prim"] == -2).sum() > 40


def test_kat_instance_inside_instance(gpu_ctx):
    """TransformedObject<TransformedObject<Sphere>> through the raw ABI (shapes.rs:26-41 takes any Hittable): the
    outer group shifts x by 10, the inner one (parent = outer) shifts y by 3, the unit sphere sits at the origin of
    the inner space, i.e. at (10, 3, 0) in the world.  Power-of-two-free but exact arithmetic: every value below is
    representable, so the answers are exact."""
    mats, texs = solid_lambertian()
    outer = rtamd.rt_group(kind=1, has_xform=1)
    outer.m = (C.c_float * 12)(1, 0, 0, 10, 0, 1, 0, 0, 0, 0, 1, 0)
    outer.minv = (C.c_float * 12)(1, 0, 0, -10, 0, 1, 0, 0, 0, 0, 1, 0)
    inner = rtamd.rt_group(kind=1, has_xform=1, parent=1)
    inner.m = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 3, 0, 0, 1, 0)
    inner.minv = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, -3, 0, 0, 1, 0)
    # a second sphere directly in the outer group, and one in the world, so that the walk has to come back out
    spheres = dict(c0r=[[0, 0, 0, 1], [0, -4, 0, 1], [0, 0, 20, 1]], mat=[0, 0, 0], id=[0, 1, 2], group=[2, 1, 0])
    info = gpu_ctx.set_scene(spheres=spheres, groups=[outer, inner], materials=mats, textures=texs)
    assert info["n_groups"] == 2
    r = gpu_ctx.intersect(one_ray((10, 3, -5), (0, 0, 1)))
    assert r["prim"][0] == 0 and r["inst"][0] == 0 and r["t"][0] == 4.0 and r["point"][0].tolist() == [10.0, 3.0, -1.0]
    assert r["front"][0] == 1 and r["normal"][0].tolist() == [0.0, 0.0, -1.0]
    r = gpu_ctx.intersect(one_ray((10, -4, -5), (0, 0, 1)))  # the sphere of the outer group alone
    assert r["prim"][0] == 1 and r["inst"][0] == 0 and r["t"][0] == 4.0 and r["point"][0].tolist() == [10.0, -4.0, -1.0]
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)))  # misses both instances' boxes, hits the world sphere
    assert r["prim"][0] == 2 and r["inst"][0] == -1 and r["t"][0] == 24.0
    r = gpu_ctx.intersect(one_ray((10, 3, 5), (0, 0, 1)))  # through nothing but the world (starts behind the instance)
    assert r["prim"][0] == -1
    # a ray that enters the outer box, the inner box, misses the inner sphere, leaves both and hits the world sphere:
    # it passes (10.9, 3.9, 0), a corner region of the inner box 1.27 away from the sphere's centre, on its way to (0, 0, 20)
    o, tgt = np.array([13.625, 4.875, -5.0]), np.array([0.0, 0.0, 20.0])
    r = gpu_ctx.intersect(one_ray(tuple(o), tuple(tgt - o)))
    assert r["prim"][0] == 2 and r["inst"][0] == -1
    cam = rtamd.camera_new((10, 0, -12), (10, 0, 0), (0, 1, 0), 1.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    imgs = []
    for pipe in (0, 1, 2):
        gpu_ctx.set_pipeline(pipe)
        img, st = gpu_ctx.render(32, 32, cam, 8, background=(1.0, 1.0, 1.0))
        imgs.append(img)
    gpu_ctx.set_pipeline(-1)
    assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2]) and imgs[0].min() < 8.0 * 0.9
    # a medium inside the outer instance (TransformedObject<ConstantMedium<Sphere>>): boundary sphere of radius 2 at the
    # origin of the instance's space, i.e. at (10, 0, 0); extremely dense, so it scatters where the ray enters
    mats.append(rtamd.rt_material(kind=4, tex=0))
    fog = rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1e-3, mat=1, parent=1)
    gpu_ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 2]], mat=[0], id=[0], group=[2]), groups=[outer, fog], materials=mats, textures=texs)
    r = gpu_ctx.intersect(one_ray((10, 0, -5), (0, 0, 1)))
    assert r["prim"][0] == -2 and r["inst"][0] == 0 and r["t"][0] == pytest.approx(3.0, abs=0.02)
    # the medium's own record says normal (1, 0, 0), front face; the instance around it face-sets it again against the
    # ray (shapes.rs:86): dot((0,0,1), (1,0,0)) = 0 is not < 0, so the normal is flipped and the face is "back"
    assert r["point"][0] == pytest.approx([10.0, 0.0, -2.0], abs=0.02) and r["normal"][0].tolist() == [-1.0, 0.0, 0.0] and r["front"][0] == 0
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)))  # where the sphere would be without the instance's shift
    assert r["prim"][0] == -1


def test_empty_medium_group_is_never_hit(gpu_ctx):
    """A ConstantMedium whose boundary holds no primitive (group 1) next to a normal one (group 2): the empty one
    must not be evaluated -- it has no record in the device's media list -- and the other keeps medium index 1,
    i.e. its RNG salt and the id it reports."""
    mats, texs = solid_lambertian()
    mats.append(rtamd.rt_material(kind=4, tex=0))
    density = 0.5
    empty = rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1.0 / density, mat=1)
    full = rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1.0 / density, mat=1)
    info = gpu_ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 2]], mat=[0], id=[0], group=[2]), groups=[empty, full], materials=mats, textures=texs)
    assert info["n_media"] == 2
    rays = np.repeat(one_ray((0, 0, -5), (0, 0, 1)), 256, axis=1)
    r = gpu_ctx.intersect(rays, seed=11)
    n_hit = 0
    for i in range(256):
        xi = orc.lib().orc_rng_medium_xi(11, i, 0, 0, 1)
        hit = -math.log(xi) / density
        if hit > 4.0:
            assert r["prim"][i] == -1
        else:
            n_hit += 1
            assert r["prim"][i] == -3 and r["t"][i] == pytest.approx(3.0 + hit, rel=1e-6)
    assert n_hit > 150
    cam = rtamd.camera_new((0, 0, -5), (0, 0, 0), (0, 1, 0), 0.5, 1.0, 0.0, 10.0, 0.0, 1.0)
    for pipe in (0, 1):  # rt_render walks the same media list (k_shade and the fused kernel)
        gpu_ctx.set_pipeline(pipe)
        img, st = gpu_ctx.render(16, 16, cam, 4, background=(1.0, 1.0, 1.0))
        assert np.isfinite(img).all() and st["rays"] > 16 * 16 * 4
    gpu_ctx.set_pipeline(-1)
    # only empty media: nothing to hit at all
    gpu_ctx.set_scene(spheres=dict(c0r=[[0, 0, 9, 1]], mat=[0], id=[0]), groups=[empty], materials=mats, textures=texs)
    r = gpu_ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)))
    assert r["prim"][0] == 0 and r["t"][0] == 13.0


def test_error_paths(gpu_ctx):
    ctx = rtamd.Context(0)
    with pytest.raises(rtamd.RtError) as e:
        ctx.intersect(one_ray((0, 0, 0), (0, 0, 1)))
    assert e.value.code == -4  # RT_ERR_NOT_COMMITTED
    mats, texs = solid_lambertian()
    with pytest.raises(rtamd.RtError) as e:
        ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[5], id=[0]), materials=mats, textures=texs)
    assert e.value.code == -1 and "material" in str(e.value)
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0]), materials=mats, textures=texs)
    cam = rtamd.camera_new((0, 0, -5), (0, 0, 0), (0, 1, 0), 0.5, 1.0, 0.0, 10.0, 0.0, 1.0)
    with pytest.raises(rtamd.RtError):
        ctx.render(1, 1, cam, 1)  # width - 1 divides the jitter (main.rs:471)
    with pytest.raises(rtamd.RtError):
        ctx.render(8, 8, cam, 1, depth=0)
    with pytest.raises(rtamd.RtError):
        ctx.render(8, 8, cam, 1, depth=254)  # bounce counters 254 and 255 belong to the lens and the camera
    with pytest.raises(rtamd.RtError) as e:
        ctx.render(65536, 2, cam, 1)  # pixel coordinates are packed in 16 bits each
    assert "size" in str(e.value)
    # gen_range(time_start..time_end) panics on an empty range (camera.rs:63); on the device the redraw
    # loop would never end, so the call must be refused
    still = rtamd.camera_new((0, 0, -5), (0, 0, 0), (0, 1, 0), 0.5, 1.0, 0.0, 10.0, 0.5, 0.5)
    with pytest.raises(rtamd.RtError) as e:
        ctx.render(8, 8, still, 1)
    assert "time" in str(e.value)
    with pytest.raises(ValueError):
        ctx.render(8, 8, cam, 1, accum=np.zeros((8, 8, 3), np.float64))
    with pytest.raises(ValueError):
        ctx.render(8, 8, cam, 1, accum=np.zeros((8, 8, 4), np.float32))
    # a static scene takes any ray-time interval; with a moving sphere the interval must lie inside the shutter
    # interval of the commit, whose boxes cover nothing else (shapes.rs:236-249)
    late = rtamd.camera_new((0, 0, -5), (0, 0, 0), (0, 1, 0), 0.5, 1.0, 0.0, 10.0, 1.0, 2.0)
    ctx.render(8, 8, late, 1)
    moving = dict(c0r=[[0, 0, 0, 1]], c1=[[1, 0, 0]], t01=[[0, 1]], moving=[1], mat=[0], id=[0])
    ctx.set_scene(spheres=moving, materials=mats, textures=texs, shutter=(0.0, 1.0))
    with pytest.raises(rtamd.RtError) as e:
        ctx.render(8, 8, late, 1)
    assert e.value.code == -1 and "shutter" in str(e.value)
    ctx.set_scene(spheres=moving, materials=mats, textures=texs, shutter=(0.0, 2.0))
    ctx.render(8, 8, late, 1)
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0]), materials=mats, textures=texs)
    # a medium boundary of any number of primitives is taken (ConstantMedium<T> takes any T, shapes.rs:624-646; the
    # device evaluates the boundary as one primitive list) up to a limit that is there to catch mistakes
    mats2 = mats + [rtamd.rt_material(kind=4, tex=0)]
    many = dict(c0r=[[float(i), 0, 0, 0.4] for i in range(17)], mat=[0] * 17, id=list(range(17)), group=[1] * 17)
    fog = rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1e-3, mat=1)
    ctx.set_scene(spheres=many, groups=[fog], materials=mats2, textures=texs)
    r = ctx.intersect(one_ray((16, 0, -5), (0, 0, 1)))  # through the last of the 17 spheres, extremely dense: scatters at its surface
    assert r["prim"][0] == -2 and r["t"][0] == pytest.approx(4.6, abs=0.02)
    r = ctx.intersect(one_ray((16.9, 0, -5), (0, 0, 1)))  # past all of them
    assert r["prim"][0] == -1
    n_big = 65537
    huge = dict(c0r=np.tile(np.array([[0, 0, 0, 0.4]], np.float32), (n_big, 1)), mat=[0] * n_big, id=list(range(n_big)), group=[1] * n_big)
    with pytest.raises(rtamd.RtError) as e:
        ctx.set_scene(spheres=huge, groups=[fog], materials=mats2, textures=texs)
    assert e.value.code == -3 and "65536" in str(e.value)
    with pytest.raises(rtamd.RtError):
        ctx.intersect(one_ray((0, 0, 0), (0, 0, 1)))  # the failed commit left no scene behind
    # groups inside groups (rt_group.parent): instances up to four deep; anything else is refused by name
    one = dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0], group=[1])
    inst = lambda parent=0: rtamd.rt_group(kind=1, has_xform=0, parent=parent)
    with pytest.raises(rtamd.RtError) as e:
        ctx.set_scene(spheres=one, groups=[inst(2), inst()], materials=mats, textures=texs)  # parent after child
    assert e.value.code == -1 and "parent" in str(e.value)
    with pytest.raises(rtamd.RtError) as e:  # a medium inside a medium
        fogs = [rtamd.rt_group(kind=2, neg_inv_density=-1.0, mat=0), rtamd.rt_group(kind=2, neg_inv_density=-1.0, mat=0, parent=1)]
        ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0], group=[2]), groups=fogs, materials=mats, textures=texs)
    assert e.value.code == -3 and "medium" in str(e.value)
    with pytest.raises(rtamd.RtError) as e:
        ctx.set_scene(spheres=one, groups=[rtamd.rt_group(kind=2, neg_inv_density=-1.0, mat=0), inst(1)], materials=mats, textures=texs)
    assert e.value.code == -3 and "medium" in str(e.value)
    with pytest.raises(rtamd.RtError) as e:
        ctx.set_scene(spheres=one, groups=[inst(), inst(1), inst(2), inst(3), inst(4)], materials=mats, textures=texs)
    assert e.value.code == -3 and "4 deep" in str(e.value)
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0], group=[4]), groups=[inst(), inst(1), inst(2), inst(3)], materials=mats, textures=texs)
    assert ctx.intersect(one_ray((0, 0, -5), (0, 0, 1)))["t"][0] == 4.0
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0]), materials=mats, textures=texs)
    img, st = ctx.render(8, 8, cam, 2, depth=253)
    assert st["rays"] >= 128 and np.isfinite(img).all()
    ctx.reserve(1 << 20)  # queues allocated ahead of time: same image
    img2, st2 = ctx.render(8, 8, cam, 2, depth=253)
    assert st2["rays"] == st["rays"] and np.array_equal(img2, img)
    ctx.close()


# ---- golden vectors -------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["random_world", "earth", "simple_light", "cornell_box", "cornell_smoke", "final_scene"])
def test_golden_ray_batches(gpu_ctx, name):
    g = np.load(GOLDEN / f"rays_{name}.npz")
    gpu_ctx.load_scene(name, seed=1)
    out = gpu_ctx.intersect(g["rays"], seed=3)
    assert np.array_equal(out["prim"], g["prim"]) and np.array_equal(out["inst"], g["inst"])
    hit = g["prim"] != -1
    assert np.all(np.abs(out["t"][hit] - g["t"][hit]) <= 1e-5 * np.abs(g["t"][hit]))
    assert np.abs(out["normal"] - g["normal"]).max() <= 1e-5
    assert np.array_equal(out["front"], g["front"])


@pytest.mark.parametrize("name", ALL_SCENES)
def test_golden_images(gpu_ctx, name):
    g = np.load(GOLDEN / f"image_{name}.npz")
    h, w, _ = g["sum_rgb"].shape
    spp = int(g["spp"])
    gpu_ctx.load_scene(name, seed=1)
    img, st = gpu_ctx.render(w, h, gpu_ctx.preset_camera(1.0), spp, depth=50, seed=1)
    assert abs(st["rays"] - int(g["rays"])) <= max(2, 0.002 * int(g["rays"]))
    assert st["paths"] == w * h * spp
    assert_images_close(img, g["sum_rgb"], spp, name)


# ---- image parity with the oracle at equal spp -------------------------------------------------------
@pytest.mark.parametrize("name,aspect", [("random_world", 16 / 9), ("two_spheres", 16 / 9), ("two_perlin_spheres", 16 / 9),
                                         ("earth", 16 / 9), ("simple_light", 16 / 9), ("cornell_box", 1.0),
                                         ("cornell_smoke", 1.0), ("final_scene", 1.0), ("nested_instances", 1.0),
                                         ("medium_cluster", 1.0)])
def test_image_matches_oracle_at_equal_spp(gpu_ctx, name, aspect):
    w, spp = 160, 8
    h = rtamd.image_height(w, aspect)
    osc = orc.Scene(name, seed=2)
    gpu_ctx.load_scene(name, seed=2)
    ref, rays = osc.render(w, h, aspect, spp, depth=50, seed=5, mode=1)
    img, st = gpu_ctx.render(w, h, gpu_ctx.preset_camera(aspect), spp, depth=50, seed=5)
    assert abs(st["rays"] - rays) <= 0.002 * rays
    assert_images_close(img, ref, spp, name)
    # quantised output (main.rs:307-313): device resolve == oracle resolve on the same sums, and the two
    # images differ on a vanishing share of the bytes
    q_gpu = gpu_ctx.resolve_u8(img, spp)
    assert np.array_equal(q_gpu, orc.resolve_u8(img, spp))
    assert (q_gpu != orc.resolve_u8(ref, spp)).mean() <= 0.01


def test_c1_config_full_size_against_oracle(gpu_ctx):
    """BASELINE configs[0]: random world 400x225, 10 spp, depth 50 -- the whole frame against the oracle."""
    w, aspect, spp = 400, 16 / 9, 10
    h = rtamd.image_height(w, aspect)
    assert h == 225
    osc = orc.Scene("random_world", seed=1)
    gpu_ctx.load_scene("random_world", seed=1)
    ref, rays = osc.render(w, h, aspect, spp, depth=50, seed=1, mode=1)
    img, st = gpu_ctx.render(w, h, gpu_ctx.preset_camera(aspect), spp, depth=50, seed=1)
    assert st["paths"] == 900000 and abs(st["rays"] - rays) <= 0.001 * rays
    assert_images_close(img, ref, spp, "c1")


def test_resolve_u8_edge_values(gpu_ctx):
    vals = np.array([0.0, 0.5, 1.0, 0.999, np.nan, -1.0, 2.0, 1 / 256, np.inf, -np.inf, 0.998, 1e-9], np.float32)
    sums = np.tile(vals, 3)[: 12 * 3].reshape(12, 1, 3).copy()
    assert np.array_equal(gpu_ctx.resolve_u8(sums, 1), orc.resolve_u8(sums, 1))
    assert gpu_ctx.resolve_u8(sums, 1).reshape(-1)[:8].tolist() == [0, 128, 255, 255, 0, 0, 255, 1]
    rng = np.random.default_rng(0)
    big = (rng.uniform(-0.2, 1.3, (64, 64, 3)) * 7).astype(np.float32)
    assert np.array_equal(gpu_ctx.resolve_u8(big, 7), orc.resolve_u8(big, 7))


def test_resolve_sqrt_transfer_matches_oracle(gpu_ctx):
    """The gamma-2 write_color the reference keeps commented out (main.rs:35-42), offered as a switch."""
    rng = np.random.default_rng(3)
    sums = (rng.uniform(-0.1, 1.2, (50, 70, 3)) * 9).astype(np.float32)
    sums[0, 0] = [np.nan, np.inf, 0.0]
    assert np.array_equal(gpu_ctx.resolve_u8(sums, 9, transfer=1), orc.resolve_u8(sums, 9, sqrt=True))
    assert np.array_equal(gpu_ctx.resolve_u8(sums, 9, transfer=0), orc.resolve_u8(sums, 9))
    vals = np.float32([[[0.25, 1.0, 0.0]]])
    assert gpu_ctx.resolve_u8(vals, 1, transfer=1).reshape(-1).tolist() == [128, 255, 0]


def test_checkpointed_render_continues_bit_exactly(gpu_ctx, tmp_path):
    """Stop after 3 samples, write the checkpoint, read it back, continue to 8: equals the uninterrupted render."""
    name, w, h = "cornell_smoke", 96, 96
    gpu_ctx.load_scene(name, seed=1)
    cam = gpu_ctx.preset_camera(1.0)
    full, _ = gpu_ctx.render(w, h, cam, 8, seed=6)
    part, _ = gpu_ctx.render(w, h, cam, 3, seed=6)
    rtamd.save_accum(tmp_path / "r.acc", part, 3, 6)
    acc, done, seed = rtamd.load_accum(tmp_path / "r.acc")
    assert (done, seed) == (3, 6)
    gpu_ctx.render(w, h, cam, 8 - done, seed=seed, accum=acc, sample0=done)
    assert np.array_equal(acc, full)
    rtamd.write_image(tmp_path / "r.png", gpu_ctx.resolve_u8(acc, 8), "png")
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "r.png")), orc.resolve_u8(full, 8))


# ---- properties that do not depend on size ---------------------------------------------------------------
def test_rank_split_sample_split_and_batch_size_change_no_bit(gpu_ctx):
    name, w, h, spp = "final_scene", 200, 120, 6
    gpu_ctx.load_scene(name, seed=1)
    cam = gpu_ctx.preset_camera(w / h)
    full, st = gpu_ctx.render(w, h, cam, spp, seed=4)
    # 3 ranks, each into the shared buffer (disjoint pixels)
    merged = np.zeros_like(full)
    rays = 0
    for r in range(3):
        _, s = gpu_ctx.render(w, h, cam, spp, seed=4, accum=merged, rank=r, nranks=3, tile_size=16)
        rays += s["rays"]
    assert rays == st["rays"] and np.array_equal(merged, full)
    # samples 0..1 then 2..5 into one buffer
    split = np.zeros_like(full)
    gpu_ctx.render(w, h, cam, 2, seed=4, accum=split, sample0=0)
    gpu_ctx.render(w, h, cam, 4, seed=4, accum=split, sample0=2)
    assert np.array_equal(split, full)
    # tiny wavefront batches: pixel chunks of 4096 paths, one sample at a time
    small = rtamd.Context(0, max_paths=4096)
    small.load_scene(name, seed=1)
    tiny, s2 = small.render(w, h, cam, spp, seed=4)
    assert s2["batches"] > spp and s2["rays"] == st["rays"] and np.array_equal(tiny, full)
    small.close()


@pytest.mark.parametrize("name", ["random_world", "cornell_smoke", "final_scene", "two_perlin_spheres", "earth", "cornell_box",
                                  "nested_instances", "medium_cluster"])
def test_fused_and_split_pipelines_give_the_same_bits(name):
    w, h, spp = 150, 100, 5
    ctx = rtamd.Context(0)
    ctx.set_tail_paths(2048)
    ctx.load_scene(name, seed=1)
    cam = ctx.preset_camera(w / h)
    out = {}
    for pipe in (0, 1, 2, 3):  # split, fused, hybrid (fused camera rays, split scattered rays), whole paths
        ctx.set_pipeline(pipe)
        img, st = ctx.render(w, h, cam, spp, seed=9)
        assert st["pipeline"] == pipe
        out[pipe] = (img, st["rays"])
    ctx.close()
    assert out[0][1] == out[1][1] == out[2][1] == out[3][1]
    for pipe in (1, 2, 3):
        assert np.array_equal(out[0][0], out[pipe][0])


def test_exact_pixels_light_from_behind_background_and_depth_limit():
    """Three facts of ray_color (main.rs:44-63) that fix pixels exactly, with no sampling noise: a DiffuseLight
    emits from both faces (materials.rs:176-184), a miss returns the flat background (main.rs:60-62), and a path
    that runs out of depth contributes nothing (main.rs:45-47)."""
    texs = [rtamd.rt_texture(kind=0, color=(C.c_float * 3)(4.0, 2.0, 1.0)), rtamd.rt_texture(kind=0, color=(C.c_float * 3)(0.5, 0.5, 0.5))]
    mats = [rtamd.rt_material(kind=3, tex=0), rtamd.rt_material(kind=0, tex=1)]
    # an XY rect at z = 0 facing +z (shapes.rs:279-306), looked at from z = -10, i.e. from behind
    rects = dict(axis=np.uint8([0]), k=np.float32([0.0]), bounds=np.float32([[-1, -1, 1, 1]]), mat=np.uint32([0]), id=np.uint32([0]))
    cam = rtamd.camera_new((0, 0, -10), (0, 0, 0), (0, 1, 0), math.radians(40), 1.0, 0.0, 10.0, 0.0, 1.0)
    bg = np.float32([0.25, 0.5, 0.75])
    w = h = 64
    spp = 4
    ctx = rtamd.Context(0)
    for pipe in (0, 1, 2, 3):
        ctx.set_pipeline(pipe)
        ctx.set_scene(rects=rects, materials=mats, textures=texs)
        img, st = ctx.render(w, h, cam, spp, depth=50, seed=3, background=bg)
        px = img.reshape(-1, 3)
        lit = np.all(px == np.float32([4.0, 2.0, 1.0]) * spp, axis=1)
        sky = np.all(px == bg * spp, axis=1)
        edge = ~(lit | sky)  # pixels whose samples fall on both sides of the rect's outline
        assert lit.sum() > 200 and sky.sum() > 2000 and edge.sum() < 4 * 80
        assert st["rays"] == w * h * spp  # neither a light nor a miss scatters
        # a grey Lambertian sphere with depth 1: the scattered ray is never followed, hit pixels are black
        sph = dict(c0r=np.float32([[0, 0, 0, 1]]), mat=np.uint32([1]), id=np.uint32([0]))
        ctx.set_scene(spheres=sph, materials=mats, textures=texs)
        img, st = ctx.render(w, h, cam, spp, depth=1, seed=3, background=bg)
        px = img.reshape(-1, 3)
        black = np.all(px == 0.0, axis=1)
        sky = np.all(px == bg * spp, axis=1)
        assert black.sum() > 150 and sky.sum() > 2000 and (~(black | sky)).sum() < 4 * 80
        assert st["rays"] == w * h * spp
    ctx.close()


def test_contexts_render_concurrently_from_host_threads():
    """Contexts are independent: three of them, each on its own HIP stream and driven by its own host thread, render
    different scenes at the same time and give the images and ray counts that they give one after the other."""
    from concurrent.futures import ThreadPoolExecutor
    w, h, spp = 200, 120, 6
    jobs = []
    for name in ("two_spheres", "two_perlin_spheres", "final_scene"):
        ctx = rtamd.Context(0)
        ctx.load_scene(name, seed=1)
        jobs.append((ctx, ctx.preset_camera(w / h)))
    alone = [ctx.render(w, h, cam, spp, seed=21) for ctx, cam in jobs]
    with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
        for _ in range(3):
            together = list(pool.map(lambda j: j[0].render(w, h, j[1], spp, seed=21), jobs))
            for (img_a, st_a), (img_t, st_t) in zip(alone, together):
                assert st_a["rays"] == st_t["rays"] and np.array_equal(img_a, img_t)
    for ctx, _ in jobs:
        ctx.close()


@pytest.mark.parametrize("name", ["random_world", "cornell_smoke", "final_scene", "earth", "nested_instances"])
def test_tail_kernel_changes_no_bit(name):
    """The last live paths of a batch are walked to their end by one launch (k_finish) instead of one launch per
    bounce.  Where the hand-over happens must not matter: never, after the first bounce, or somewhere in between."""
    w, h, spp = 160, 100, 6
    ctx = rtamd.Context(0)
    ctx.load_scene(name, seed=1)
    cam = ctx.preset_camera(w / h)
    out = []
    for tail in (0, 3000, 1 << 30):
        ctx.set_tail_paths(tail)
        for pipe in (1, 2):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, spp, seed=11)
            if tail != 3000:  # (a closed box can keep more than 3000 paths alive down to the depth limit)
                assert (st["tail_launches"] > 0) == (tail > 0) and (st["tail_rays"] > 0) == (tail > 0)
            if tail == 1 << 30:  # everything but the camera rays went through the tail
                assert st["bounce_launches"] == 1 and st["tail_rays"] == st["rays"] - st["paths"]
            out.append((img, st["rays"]))
    ctx.close()
    for img, rays in out[1:]:
        assert rays == out[0][1] and np.array_equal(img, out[0][0])


@pytest.mark.parametrize("name", ["final_scene", "medium_cluster"])
def test_settling_the_first_medium_ahead_changes_no_bit(name, monkeypatch):
    """k_shade may find out for the ray it has just made that the first medium takes it (primitives near the boundary, then
    the medium's own test) and queue the path where the trace kernel does not look; k_settled then follows such paths in
    registers, bounce after bounce, for as long as their next hit can be settled the same way.  Every answer is the one the
    next bounce would have found after the tree walk, so image and ray count must not change: with the paths joining from one
    lane per wave on, with the default, with the step specialised for sphere-shaped media off (final_scene qualifies for it,
    medium_cluster does not), with its waiting lanes served at once or only in large groups, with short runs and early
    hand-over between launches, and with the tail kernel taking over queues that hold such paths."""
    w, h, spp = 200, 200, 24
    out = []
    cases = [dict(RTAMD_SETTLE_AHEAD="0"), dict(RTAMD_SETTLE_ENTRY="1"), dict(), dict(RTAMD_SETTLED_FAST="0"),
             dict(RTAMD_SETTLED_SLOW_MIN="1"), dict(RTAMD_SETTLED_SLOW_MIN="40", RTAMD_SETTLED_REFILL="4"),
             dict(RTAMD_SETTLED_MIN_RUN="64", RTAMD_SETTLED_DRAIN="60"), dict(RTAMD_SETTLED_DRAIN="1", RTAMD_SETTLE_NEAR_HITS="0", RTAMD_SETTLED_LATE_ROUNDS="200"),
             dict(RTAMD_SETTLED_LATE_ROUNDS="1"),
             dict(tail=20000)]
    for case in cases:
        for k in ("RTAMD_SETTLE_AHEAD", "RTAMD_SETTLE_ENTRY", "RTAMD_SETTLED_FAST", "RTAMD_SETTLED_SLOW_MIN", "RTAMD_SETTLED_REFILL",
                  "RTAMD_SETTLED_MIN_RUN", "RTAMD_SETTLED_DRAIN", "RTAMD_SETTLE_NEAR_HITS", "RTAMD_SETTLED_LATE_ROUNDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in case.items():
            if k.startswith("RTAMD_"):
                monkeypatch.setenv(k, v)  # (a context reads its knobs when it is created)
        ctx = rtamd.Context(0)
        ctx.load_scene(name, seed=1)
        ctx.set_tail_paths(case.get("tail", 0))
        cam = ctx.preset_camera(w / h)
        for pipe in (0, 2):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, spp, seed=5)
            if name == "final_scene":  # (its first medium is the dense sphere that holds paths for many bounces)
                settle = case.get("RTAMD_SETTLE_AHEAD", "1") == "1"
                assert (st["settled_rays"] > 0.1 * st["rays"]) if settle else st["settled_rays"] == 0
            out.append((img, st["rays"]))
        ctx.close()
    for img, rays in out[1:]:
        assert rays == out[0][1] and np.array_equal(img, out[0][0])


@pytest.mark.parametrize("name, env, width, fits", [
    ("random_world", {}, 2, True),                                       # a few hundred nodes: the whole tree
    ("random_world", {"RTAMD_CROWN_PARTIAL": "1", "RTAMD_CROWN_NODES": "100"}, 2, False),
    ("random_world", {"RTAMD_CROWN_PARTIAL": "1"}, 4, False),            # four-child records, the top of the tree
    ("final_scene", {"RTAMD_CROWN_PARTIAL": "1"}, 2, False),             # world tree + the top of the instance's own tree
    ("final_scene", {"RTAMD_CROWN_PARTIAL": "1", "RTAMD_CROWN_NODES": "40"}, 4, False),
    ("cornell_smoke", {"RTAMD_FLAT_MAX": "0"}, 2, True),                 # instances and media, traced through trees: all of it
])
def test_tree_nodes_kept_in_lds_change_no_bit(name, env, width, fits, monkeypatch):
    """The persistent trace kernel may keep inner nodes in LDS (by default: the whole tree of a scene where that fits, which is
    what pays; for this test also the top of trees that do not fit): copies of the same records, child references inside the
    copies pointing at copies, instance roots entered through a table.  Same boxes in the same order, so image and ray count
    must be those of the walk that reads every node from memory."""
    w, h, spp = 160, 120, 12
    out = []
    for case in (dict(RTAMD_CROWN_NODES="0"), env):
        for k in ("RTAMD_CROWN_NODES", "RTAMD_CROWN_PARTIAL", "RTAMD_FLAT_MAX"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("RTAMD_CROWN_MIN_BLOCKS", "0")  # (small frames here: take the crown at any launch size)
        if "RTAMD_FLAT_MAX" in env:
            monkeypatch.setenv("RTAMD_FLAT_MAX", env["RTAMD_FLAT_MAX"])
        for k, v in case.items():
            monkeypatch.setenv(k, v)  # (a context reads its knobs when it is created)
        ctx = rtamd.Context(0)
        ctx.set_bvh_width(width)
        ctx.set_tail_paths(0)
        ctx.load_scene(name, seed=3)
        info = ctx.info()
        if case is env:
            assert info["bvh_width"] == width
            assert (info["crown_nodes"] == info["n_nodes"]) if fits else (0 < info["crown_nodes"] < info["n_nodes"])
        else:
            assert info["crown_nodes"] == 0
        cam = ctx.preset_camera(w / h)
        for pipe in (0, 2):  # the trace kernel from the camera rays on / after a fused first bounce
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, spp, seed=9)
            out.append((img, st["rays"]))
        ctx.close()
    for img, rays in out[1:]:
        assert rays == out[0][1] and np.array_equal(img, out[0][0])


@pytest.mark.parametrize("glass, n_media", [(False, 1), (False, 3), (True, 2), (True, 4)])
def test_settled_paths_in_raw_scenes_of_sphere_shaped_media(glass, n_media, monkeypatch):
    """The step of k_settled that is specialised for sphere-shaped media of plain colour, on scenes sent through the raw ABI:
    a dense ball of medium among sixty diffuse spheres (a tree, not a flat scene), with or without a shell of glass of the
    same radius around it, alone or together with a thin fog around everything and further balls of medium elsewhere (up
    to four media).  Settling ahead on or off, the specialised step on or off: same image bits, same ray count; and with
    it on, a good share of the rays never sees the tree."""
    rng = np.random.default_rng(7)
    c0r, mat, group = [], [], []
    for _ in range(60):  # diffuse spheres on a ring around the ball, clear of it
        ang, rad = rng.uniform(0, 2 * math.pi), rng.uniform(60, 120)
        c0r.append([rad * math.cos(ang), rng.uniform(-20, 20), rad * math.sin(ang), rng.uniform(2, 6)])
        mat.append(0)
        group.append(0)
    texs = [rtamd.rt_texture(kind=0, color=(C.c_float * 3)(0.7, 0.6, 0.5)), rtamd.rt_texture(kind=0, color=(C.c_float * 3)(0.2, 0.4, 0.9)),
            rtamd.rt_texture(kind=0, color=(C.c_float * 3)(1, 1, 1)), rtamd.rt_texture(kind=0, color=(C.c_float * 3)(0.9, 0.3, 0.3))]
    mats = [rtamd.rt_material(kind=0, tex=0), rtamd.rt_material(kind=2, ior=1.5), rtamd.rt_material(kind=4, tex=1),
            rtamd.rt_material(kind=4, tex=2), rtamd.rt_material(kind=4, tex=3)]
    groups = []
    media = [([0, 0, 0, 30], 0.3, 2), ([0, 0, 0, 800], 0.002, 3), ([90, 0, 0, 12], 0.5, 4), ([-90, 5, 10, 15], 0.2, 4)][:n_media]
    if glass:
        c0r.append([0, 0, 0, 30])
        mat.append(1)
        group.append(0)
    for k, (sph, density, m) in enumerate(media):
        groups.append(rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1.0 / density, mat=m))
        c0r.append(sph)
        mat.append(0)
        group.append(k + 1)
    spheres = dict(c0r=c0r, mat=mat, id=list(range(len(c0r))), group=group)
    cam = rtamd.camera_new((0, 10, -150), (0, 0, 0), (0, 1, 0), math.radians(35), 1.0, 0.0, 10.0, 0.0, 1.0)
    w = h = 160
    spp = 16
    out = []
    for case in (dict(RTAMD_SETTLE_AHEAD="0"), dict(), dict(RTAMD_SETTLED_FAST="0"), dict(RTAMD_SETTLED_SLOW_MIN="1", RTAMD_SETTLED_MIN_RUN="64")):
        for k in ("RTAMD_SETTLE_AHEAD", "RTAMD_SETTLED_FAST", "RTAMD_SETTLED_SLOW_MIN", "RTAMD_SETTLED_MIN_RUN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in case.items():
            monkeypatch.setenv(k, v)
        ctx = rtamd.Context(0)
        info = ctx.set_scene(spheres=spheres, groups=groups, materials=mats, textures=texs)
        assert info["n_nodes"] > 20 and info["n_media"] == n_media
        ctx.set_tail_paths(0)  # (a frame this small would otherwise be finished by the tail kernel after the first bounce)
        for pipe in (0, 2):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, spp, background=(0.6, 0.7, 0.9), seed=3)
            assert ((st["settled_rays"] > 0.03 * st["rays"]) if case.get("RTAMD_SETTLE_AHEAD", "1") == "1" else st["settled_rays"] == 0), (st["settled_rays"], st["rays"])
            out.append((img, st["rays"]))
        ctx.close()
    assert np.isfinite(out[0][0]).all() and out[0][0].sum() > 0
    for img, rays in out[1:]:
        assert rays == out[0][1] and np.array_equal(img, out[0][0])


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke", "two_spheres", "simple_light"])
def test_flat_and_tree_tracing_give_the_same_bits(name, monkeypatch):
    """Scenes of a few dozen primitives are traced without the tree (uniform loops over the primitives).  The
    closest hit does not depend on the order of the tests, so forcing the tree walk must change no bit: hit
    ids, t, both pipelines' images and ray counts."""
    w, h, spp = 120, 90, 4
    g = np.load(GOLDEN / f"rays_{name}.npz") if (GOLDEN / f"rays_{name}.npz").exists() else None
    out = {}
    for flat_max in ("48", "0"):
        monkeypatch.setenv("RTAMD_FLAT_MAX", flat_max)
        ctx = rtamd.Context(0)
        ctx.set_tail_paths(2048)
        info = ctx.load_scene(name, seed=1)
        assert info["stack_entries"] == 1 or flat_max == "0"  # a flat scene never pushes
        cam = ctx.preset_camera(w / h)
        res = []
        if g is not None:
            hit = ctx.intersect(g["rays"], seed=3)
            res += [hit["prim"], hit["inst"], hit["t"], hit["normal"]]
        for pipe in (0, 1):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, spp, seed=5)
            res += [img, np.array([st["rays"]])]
        ctx.close()
        out[flat_max] = res
    for a, b in zip(out["48"], out["0"]):
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f")


@pytest.mark.parametrize("name", ["two_perlin_spheres", "two_spheres", "cornell_smoke", "final_scene"])
def test_lds_staged_tables_change_no_bit(name, monkeypatch):
    """The shading kernels copy Perlin table 0 (any scene with a noise texture) and, for small scenes, the
    primitive / material / texture records into LDS.  Where a record is read from must not change a bit."""
    w, h, spp = 128, 96, 4
    out = []
    for staged in ("1", "0"):
        monkeypatch.setenv("RTAMD_LDS_TABLES", staged)
        ctx = rtamd.Context(0)
        ctx.set_tail_paths(2048)
        ctx.load_scene(name, seed=1)
        res = []
        for pipe in (0, 1):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, ctx.preset_camera(w / h), spp, seed=13)
            res += [img, np.array([st["rays"]])]
        ctx.close()
        out.append(res)
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_device_and_host_accumulators_agree(gpu_ctx):
    import torch
    gpu_ctx.load_scene("cornell_box", seed=1)
    cam = gpu_ctx.preset_camera(1.0)
    host, _ = gpu_ctx.render(96, 96, cam, 4, seed=3)
    dev = torch.zeros((96, 96, 3), dtype=torch.float32, device="cuda")
    gpu_ctx.render(96, 96, cam, 4, seed=3, accum=dev)
    assert np.array_equal(dev.cpu().numpy(), host)
    assert np.array_equal(gpu_ctx.resolve_u8(dev, 4).cpu().numpy(), gpu_ctx.resolve_u8(host, 4))


def test_c2_full_frame_properties(gpu_ctx):
    """BASELINE configs[1] at its full 1920x1080 frame (reduced spp): an 8-rank split reproduces the 1-rank
    frame bit for bit, and pixels whose samples all miss hold exactly spp sequential f32 adds of the background."""
    import torch
    w, aspect, spp = 1920, 16 / 9, 3
    h = rtamd.image_height(w, aspect)
    for name in ("earth", "two_spheres", "two_perlin_spheres"):
        gpu_ctx.load_scene(name, seed=1)
        cam = gpu_ctx.preset_camera(aspect)
        one = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        _, st = gpu_ctx.render(w, h, cam, spp, seed=1, accum=one)
        assert st["paths"] == w * h * spp
        eight = torch.zeros_like(one)
        for r in range(8):
            gpu_ctx.render(w, h, cam, spp, seed=1, accum=eight, rank=r, nranks=8)
        assert torch.equal(one, eight)
        if name == "earth":  # the sphere is in the middle; the corners see only background
            bg = np.float32([0.7, 0.8, 1.0])
            acc = np.zeros(3, np.float32)
            for _ in range(spp):
                acc = acc + bg
            corner = one[:8, :8].cpu().numpy().reshape(-1, 3)
            assert np.all(corner == acc)
            assert float(one.mean()) / spp < 1.0


def test_c5_like_deep_bvh_properties(gpu_ctx):
    """The synthetic deep-BVH scene (reduced from 10M to 1M spheres for test time) at a 3840x2160 frame, 1 spp:
    rank split reproduces the single-rank frame bit for bit and every path ends with finite radiance."""
    import torch
    info = gpu_ctx.load_scene("synth_spheres", seed=1, n_synth=1_000_000)
    assert info["n_prims"] == 1_000_001 and 16 < info["stack_entries"] <= 64  # a deep tree: more than the 16 entries small scenes need
    w, h = 3840, 2160
    cam = gpu_ctx.preset_camera(16 / 9)
    one = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    _, st = gpu_ctx.render(w, h, cam, 1, seed=1, accum=one)
    assert st["paths"] == w * h and st["rays"] >= st["paths"]
    assert bool(torch.isfinite(one).all()) and float(one.min()) >= 0.0 and float(one.max()) <= 1.0
    four = torch.zeros_like(one)
    for r in range(4):
        gpu_ctx.render(w, h, cam, 1, seed=1, accum=four, rank=r, nranks=4)
    assert torch.equal(one, four)


def test_c5_ten_million_spheres_properties():
    """BASELINE configs[4] at its full 10M spheres and its full 3840x2160 frame (1 spp): the scene builds on the device,
    takes the four-child walk and the split pipeline, a 2-rank split reproduces the 1-rank frame bit for bit, every
    pixel is finite and inside [0, 1], and rays >= paths."""
    import torch
    ctx = rtamd.Context(0)
    info = ctx.load_scene("synth_spheres", seed=1, n_synth=10_000_000)
    assert info["n_prims"] == 10_000_001 and info["builder"] == 1 and info["bvh_width"] == 4 and info["stack_entries"] <= 64
    w, h = 3840, 2160
    cam = ctx.preset_camera(16 / 9)
    one = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    _, st = ctx.render(w, h, cam, 1, seed=1, accum=one)
    assert st["pipeline"] == 0 and st["paths"] == w * h and st["rays"] >= st["paths"]
    assert bool(torch.isfinite(one).all()) and float(one.min()) >= 0.0 and float(one.max()) <= 1.0
    assert float(one.mean()) > 0.05  # not a black frame
    two = torch.zeros_like(one)
    for r in range(2):
        ctx.render(w, h, cam, 1, seed=1, accum=two, rank=r, nranks=2)
    assert torch.equal(one, two)
    ctx.close()


@pytest.mark.parametrize("name, w, aspect", [("random_world", 1920, 16 / 9), ("cornell_box", 2048, 1.0), ("cornell_smoke", 2048, 1.0)])
def test_c3_and_c4_full_frames(name, w, aspect):
    """BASELINE configs[2] (random world with moving spheres, 1920x1080) and configs[3] (both Cornell scenes, 2048x2048)
    at their full frame sizes, 2 spp: the frame dealt over eight ranks equals the one-rank frame bit for bit, every pixel
    is finite, and a 64x64 crop from the middle of the frame agrees with the oracle's render of that crop."""
    import torch
    h = rtamd.image_height(w, aspect)
    spp = 2
    ctx = rtamd.Context(0)
    ctx.load_scene(name, seed=1)
    cam = ctx.preset_camera(aspect)
    one = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    _, st = ctx.render(w, h, cam, spp, seed=1, accum=one)
    assert st["paths"] == w * h * spp and st["rays"] >= st["paths"] and bool(torch.isfinite(one).all())
    eight = torch.zeros_like(one)
    rays = 0
    for r in range(8):
        rays += ctx.render(w, h, cam, spp, seed=1, accum=eight, rank=r, nranks=8, tile_size=8)[1]["rays"]
    assert rays == st["rays"] and torch.equal(one, eight)
    ctx.close()
    x0, y0 = w // 2 - 32, h // 2 - 32
    osc = orc.Scene(name, seed=1)
    ref, _ = osc.render(w, h, aspect, spp, depth=50, seed=1, mode=1, tile=(x0, y0, x0 + 64, y0 + 64))
    osc.close()
    crop = one[y0:y0 + 64, x0:x0 + 64].cpu().numpy()
    assert crop.sum() > 0
    assert_images_close(crop, ref[y0:y0 + 64, x0:x0 + 64], spp, f"{name} crop")


def test_cpp_example_renders_the_same_image_as_the_python_binding(tmp_path):
    """examples/render_main.cpp builds simple_light with the C++ mirror of the reference's constructors (as
    main.rs:163-175 does), renders through the C ABI and writes a PNG; the named scene of the host library rendered
    through the Python binding must give the same bytes."""
    import subprocess
    from PIL import Image
    exe = orc.ROOT / "raytracer-rs_amd" / "lib" / "render_main"  # built by the Makefile next to the libraries
    assert exe.exists(), "run make -C raytracer-rs_amd (or __graft_entry__.build()) first"
    out = tmp_path / "test.png"
    r = subprocess.run([str(exe), "160", "6", str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "Time elapsed" in r.stdout, r.stderr
    ctx = rtamd.Context(0)
    ctx.load_scene("simple_light", seed=1)
    img, _ = ctx.render(160, 90, ctx.preset_camera(16 / 9), 6, seed=1)
    ref = ctx.resolve_u8(img, 6)
    ctx.close()
    assert np.array_equal(np.asarray(Image.open(out)), ref)


def test_bench_line_keeps_its_contract():
    """bench.py prints ONE JSON line; the keys the harness reads, the roofline and the CPU baseline object are there,
    and the rates follow from the counts in the same line (small frame, one step)."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(orc.ROOT / "bench.py"), "--workload", "c2_textures", "--width", "320", "--spp", "4", "--steps", "1",
                        "--warmup", "1", "--cpu-baseline-seconds", "3"],
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    cfg = d["config"]
    assert cfg["workload"] == "c2_textures" and cfg["width"] == 320 and cfg["spp"] == 4
    assert cfg["paths_per_step"] == 3 * 320 * 180 * 4 and cfg["rays_per_step"] > cfg["paths_per_step"]
    assert d["value"] == pytest.approx(cfg["rays_per_step"] / d["ms_per_step"] / 1e3, rel=1e-3)
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "valu") and roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and 0 < roof["frac"] <= 1.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3, abs=2e-5) and roof["avg_launch_ms"] > 0
    assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-3)
    for k in roof.get("other_kernels", []):
        assert 0 < k["frac"] <= 1.0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == "Mrays/s" and cpu["sample"]
    assert d["vs_cpu_baseline"] == pytest.approx(d["value"] / cpu["value"], rel=1e-2)


def test_bench_default_workload_walks_the_tree():
    """Without --workload the bench renders the reference binary's own scene (final_scene) through the tree-walking
    kernels: trace time is not zero, the ray count per path is that of the scene, no fraction exceeds 1."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(orc.ROOT / "bench.py"), "--width", "400", "--spp", "16", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["workload"] == "final_scene" and d["config"]["frames"] == ["final_scene"] and d["config"]["max_depth"] == 50
    roof = d["roofline"]
    assert roof["trace_ms_per_step"] > 0 and roof["kernel"] in ("k_trace", "k_shade", "k_bounce", "k_settled") and 0 < roof["frac"] <= 1.0
    assert roof["nodes_per_ray"][0] > 5
    assert 3.0 < d["config"]["rays_per_step"] / d["config"]["paths_per_step"] < 7.0
