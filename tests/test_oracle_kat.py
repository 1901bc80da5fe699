"""Known-answer tests that pin the CPU oracle.

The reference (Rust, raytracer-rs/src/*.rs) has no tests, golden vectors or fixtures, and cannot be
built here, so the oracle is pinned by answers derived by hand from the cited reference lines.
"""
import ctypes as C
import math

import numpy as np
import pytest

import orc
from orc import fa

INF = float("inf")


def out11():
    return (C.c_float * 11)()


def ray(o, d, time=0.0):
    return fa(o, d, time)


# ---- K1-K3: Sphere::intersect (shapes.rs:117-156), get_sphere_uv (shapes.rs:19-24) ----------------
def test_k1_sphere_front_hit():
    o = out11()
    orc.lib().orc_kat_sphere(fa(0, 0, 0), 1.0, ray((0, 0, -5), (0, 0, 1)), 0.005, INF, o)
    # a=1, half_b=-5, c=24, disc=1, root=(5-1)/1=4
    assert o[0] == 1.0 and o[1] == 4.0
    assert list(o[2:5]) == [0.0, 0.0, -1.0]
    assert list(o[5:8]) == [0.0, 0.0, -1.0] and o[8] == 1.0
    # theta=acos(-0)=pi/2 -> v=0.5 ; phi=atan2(1,0)+pi=3pi/2 -> u=0.75
    assert o[9] == pytest.approx(0.75, abs=1e-7) and o[10] == pytest.approx(0.5, abs=1e-7)


def test_k2_sphere_from_inside_flips_normal():
    o = out11()
    orc.lib().orc_kat_sphere(fa(0, 0, 0), 1.0, ray((0, 0, 0), (0, 0, 1)), 0.005, INF, o)
    # first root -1 < t_min -> second root +1; outward (0,0,1), d.n>0 -> front=false, normal flipped
    assert o[0] == 1.0 and o[1] == 1.0 and o[8] == 0.0
    assert list(o[5:8]) == [0.0, 0.0, -1.0]


def test_k3_unnormalised_direction_scales_t():
    o = out11()
    orc.lib().orc_kat_sphere(fa(0, 0, 0), 1.0, ray((0, 0, -5), (0, 0, 2)), 0.005, INF, o)
    assert o[1] == 2.0


def test_sphere_range_is_inclusive_and_misses():
    o = out11()
    L = orc.lib()
    L.orc_kat_sphere(fa(0, 0, 0), 1.0, ray((0, 0, -5), (0, 0, 1)), 0.005, 4.0, o)  # root == t_max passes
    assert o[0] == 1.0 and o[1] == 4.0
    L.orc_kat_sphere(fa(0, 0, 0), 1.0, ray((0, 0, -5), (0, 0, 1)), 0.005, 3.9, o)  # both roots out of range
    assert o[0] == 0.0
    L.orc_kat_sphere(fa(0, 0, 0), 1.0, ray((0, 3, -5), (0, 0, 1)), 0.005, INF, o)  # disc < 0
    assert o[0] == 0.0


# ---- K4: AABB::hit ignores the caller's range (aabb.rs:21-33) -----------------------------------
def test_k4_aabb_ignores_t_range_and_rejects_behind():
    L = orc.lib()
    assert L.orc_kat_aabb_hit(fa(-1, -1, -1), fa(1, 1, 1), ray((0, 0, -5), (0, 0, 1)), 0.005, 1.0) == 1
    assert L.orc_kat_aabb_hit(fa(-1, -1, -1), fa(1, 1, 1), ray((0, 0, 5), (0, 0, 1)), 0.005, INF) == 0  # t_max < 0
    # flat box: t_min >= t_max rejects (why the rects pad their boxes by 1e-4)
    assert L.orc_kat_aabb_hit(fa(-1, -1, 0), fa(1, 1, 0), ray((0, 0, -5), (0, 0, 1)), 0.005, INF) == 0


# ---- K5: XZRect (shapes.rs:343-370) with the Cornell light (main.rs:187) ------------------------
def test_k5_xzrect_light_from_below():
    o = out11()
    orc.lib().orc_kat_rect(1, 213.0, 227.0, 343.0, 332.0, 554.0, ray((278, 0, 279), (0, 1, 0)), 0.005, INF, o)
    assert o[0] == 1.0 and o[1] == 554.0
    assert o[9] == pytest.approx((278 - 213) / 130, rel=1e-6) and o[10] == pytest.approx((279 - 227) / 105, rel=1e-6)
    assert list(o[5:8]) == [0.0, -1.0, 0.0] and o[8] == 0.0  # +Y outward, ray along +Y -> back face, flipped


def test_rect_axes_and_bounds():
    o = out11()
    L = orc.lib()
    # YZRect: min.x..max.x is y, min.y..max.y is z (shapes.rs:415-418)
    L.orc_kat_rect(2, 0.0, 10.0, 5.0, 20.0, 3.0, ray((0, 2, 15), (1, 0, 0)), 0.005, INF, o)
    # outward normal +X, ray along +X: d.n > 0 -> back face, normal flipped
    assert o[0] == 1.0 and o[1] == 3.0 and list(o[5:8]) == [-1.0, 0.0, 0.0] and o[8] == 0.0
    L.orc_kat_rect(2, 0.0, 10.0, 5.0, 20.0, 3.0, ray((0, 6, 15), (1, 0, 0)), 0.005, INF, o)  # y outside
    assert o[0] == 0.0
    # XYRect bound is inclusive
    L.orc_kat_rect(0, 0.0, 0.0, 1.0, 1.0, 2.0, ray((1, 1, 0), (0, 0, 1)), 0.005, INF, o)
    assert o[0] == 1.0 and o[1] == 2.0 and o[9] == 1.0 and o[10] == 1.0


# ---- K6: MovingSphere::center (shapes.rs:189-191) ---------------------------------------------
def test_k6_moving_sphere_midpoint():
    o = out11()
    # centre moves (0,0,0)->(0,2,0) over [0,1]; at time 0.5 it is (0,1,0): ray along z at y=1 hits at t=4
    orc.lib().orc_kat_moving_sphere(fa(0, 0, 0), fa(0, 2, 0), 1.0, 0.0, 1.0, ray((0, 1, -5), (0, 0, 1), 0.5), 0.005, INF, o)
    assert o[0] == 1.0 and o[1] == 4.0 and list(o[5:8]) == [0.0, 0.0, -1.0]
    orc.lib().orc_kat_moving_sphere(fa(0, 0, 0), fa(0, 2, 0), 1.0, 0.0, 1.0, ray((0, 1, -5), (0, 0, 1), 0.0), 0.005, INF, o)
    assert o[0] == 1.0 and o[1] == 5.0  # grazes the top of the sphere at its start position


# ---- K7: TransformedObject (shapes.rs:64-97) ---------------------------------------------------
def test_k7_instance_shares_t_and_reports_front_face_true():
    L = orc.lib()
    o = out11()
    box = (C.c_float * 6)()
    # translation only: world hit = local hit shifted, t unchanged
    L.orc_kat_instanced_box(2.0, 2.0, 2.0, 0.0, fa(10, 0, 0), ray((11, 1, -5), (0, 0, 1)), 0.005, INF, o, box)
    assert o[0] == 1.0 and o[1] == 5.0 and list(o[2:5]) == [11.0, 1.0, 0.0]
    assert list(box) == [10.0, 0.0, 0.0, 12.0, 2.0, 2.0]
    assert list(o[5:8]) == [0.0, 0.0, -1.0] and o[8] == 1.0
    # from inside the box the local record is a back face; set_face_normal runs again on the already
    # flipped normal (shapes.rs:75-86), so front_face comes out true
    L.orc_kat_instanced_box(2.0, 2.0, 2.0, 0.0, fa(10, 0, 0), ray((11, 1, 1), (0, 0, 1)), 0.005, INF, o, None)
    assert o[0] == 1.0 and o[1] == 1.0 and o[8] == 1.0 and list(o[5:8]) == [0.0, 0.0, -1.0]
    # rotation about y by 90 degrees maps local +x to world -z: the box spans x in [0,2] -> z in [-2,0]
    L.orc_kat_instanced_box(2.0, 2.0, 2.0, 90.0, fa(0, 0, 0), ray((1, 1, -5), (0, 0, 1)), 0.005, INF, o, box)
    assert o[0] == 1.0 and o[1] == pytest.approx(3.0, abs=1e-5)
    assert box[2] == pytest.approx(-2.0, abs=1e-6) and box[5] == pytest.approx(0.0, abs=1e-6)


# ---- K8/K9: dielectric helpers (materials.rs:109-114, vec3_helpers.rs:62-75) --------------------
def test_k8_reflectance():
    L = orc.lib()
    assert L.orc_kat_reflectance(1.0, 1.5) == pytest.approx(0.04, abs=1e-7)       # r0 = ((1-1.5)/(1+1.5))^2
    assert L.orc_kat_reflectance(0.0, 1.5) == pytest.approx(1.0, abs=1e-7)        # grazing
    assert L.orc_kat_reflectance(0.5, 1.0 / 1.5) == pytest.approx(0.04 + 0.96 * 0.5 ** 5, abs=1e-6)


def test_k9_reflect_refract():
    L = orc.lib()
    o = (C.c_float * 3)()
    L.orc_kat_reflect(fa(1, -1, 0), fa(0, 1, 0), o)
    assert list(o) == [1.0, 1.0, 0.0]
    L.orc_kat_refract(fa(0, -1, 0), fa(0, 1, 0), 1.0 / 1.5, o)  # normal incidence goes straight through
    assert list(o) == [0.0, -1.0, 0.0]
    s = math.sin(math.radians(30))
    L.orc_kat_refract(fa(s, -math.cos(math.radians(30)), 0), fa(0, 1, 0), 1.0 / 1.5, o)  # Snell: sin t = sin i / 1.5
    assert o[0] == pytest.approx(s / 1.5, abs=1e-6) and o[1] == pytest.approx(-math.sqrt(1 - (s / 1.5) ** 2), abs=1e-6)


def test_dielectric_scatter_total_internal_reflection_draws_nothing():
    o = out11()
    # inside glass (front=false -> ratio 1.5), 60 degrees off the normal: 1.5*sin > 1 -> reflect, no draw
    d = (math.sin(math.radians(60)), -math.cos(math.radians(60)), 0.0)
    orc.lib().orc_kat_scatter(2, fa(0, 0, 0), 1.5, ray((0, 1, 0), d), fa(0, 0, 0), fa(0, 1, 0), 0, 1, 2, 3, 0, o)
    assert o[0] == 1.0 and o[10] == 0.0 and list(o[7:10]) == [1.0, 1.0, 1.0]
    assert o[4] == pytest.approx(d[0], abs=1e-6) and o[5] == pytest.approx(-d[1], abs=1e-6)
    # normal incidence from outside draws exactly one number (the Schlick test)
    orc.lib().orc_kat_scatter(2, fa(0, 0, 0), 1.5, ray((0, 1, 0), (0, -1, 0)), fa(0, 0, 0), fa(0, 1, 0), 1, 1, 2, 3, 0, o)
    assert o[10] == 1.0


def test_metal_draws_even_at_zero_fuzz_and_absorbs_below_horizon():
    o = out11()
    orc.lib().orc_kat_scatter(1, fa(0.7, 0.6, 0.5), 0.0, ray((0, 1, 0), (1, -1, 0)), fa(0, 0, 0), fa(0, 1, 0), 1, 1, 2, 3, 0, o)
    assert o[0] == 1.0 and o[10] >= 3.0  # materials.rs:89: random_in_unit_sphere is evaluated regardless of fuzz
    assert o[4] == pytest.approx(math.sqrt(0.5), abs=1e-6) and o[5] == pytest.approx(math.sqrt(0.5), abs=1e-6)
    assert [round(x, 6) for x in o[7:10]] == [0.7, 0.6, 0.5]


def test_lambertian_and_isotropic_scatter():
    o = out11()
    orc.lib().orc_kat_scatter(0, fa(0.1, 0.2, 0.3), 0.0, ray((0, 1, 0), (0, -1, 0), 0.25), fa(1, 2, 3), fa(0, 1, 0), 1, 9, 8, 7, 0, o)
    assert o[0] == 1.0 and list(o[1:4]) == [1.0, 2.0, 3.0] and o[10] % 3 == 0
    dirv = np.array(o[4:7]) - np.array([0, 1, 0])
    assert np.linalg.norm(dirv) == pytest.approx(1.0, abs=1e-6)  # normal + unit vector
    orc.lib().orc_kat_scatter(4, fa(0.1, 0.2, 0.3), 0.0, ray((0, 1, 0), (0, -1, 0)), fa(1, 2, 3), fa(1, 0, 0), 1, 9, 8, 7, 0, o)
    assert np.linalg.norm(np.array(o[4:7])) == pytest.approx(1.0, abs=1e-6)


# ---- K10: quantise (main.rs:307-313) -----------------------------------------------------------
@pytest.mark.parametrize("f,q", [(0.0, 0), (0.5, 128), (1.0, 255), (0.999, 255), (float("nan"), 0), (-1.0, 0), (2.0, 255),
                                 (1 / 256, 1)])
def test_k10_quantise(f, q):
    assert orc.lib().orc_kat_quantise(f) == q


# ---- K11: ImageTexture::value (texture.rs:96-117) -------------------------------------------------
def test_k11_image_texture_corners_and_v_flip():
    w, h = 4, 2
    img = np.arange(w * h * 3, dtype=np.uint8).reshape(h, w, 3)
    o = (C.c_float * 3)()
    L = orc.lib()
    L.orc_kat_image_value(img.ctypes.data, w, h, 0.0, 0.0, o)  # u=0 -> i=0 ; v=0 -> flipped to 1 -> j clamps to h-1
    assert list(o) == pytest.approx(list(img[h - 1, 0] / np.float32(255.0)))
    L.orc_kat_image_value(img.ctypes.data, w, h, 1.0, 1.0, o)  # u=1 -> i clamps to w-1 ; v=1 -> j=0
    assert list(o) == pytest.approx(list(img[0, w - 1] / np.float32(255.0)))
    L.orc_kat_image_value(img.ctypes.data, w, h, 0.26, 0.74, o)  # i=int(1.04)=1, j=int(0.52)=0
    assert list(o) == pytest.approx(list(img[0, 1] / np.float32(255.0)))


# ---- K12: Perlin::noise / turb (perlin.rs:46-102) -------------------------------------------------
def perlin_tables(seed=0):
    rng = np.random.default_rng(seed)
    vec = rng.uniform(-1, 1, (256, 3)).astype(np.float32)
    perm = np.stack([rng.permutation(256) for _ in range(3)]).astype(np.int32)
    return vec, perm


def perlin_noise_py(vec, perm, p):
    """Straight transcription of perlin.rs:46-88 in float64 (independent of the oracle's code)."""
    fl = np.floor(p)
    u, v, w = p - fl
    i, j, k = fl.astype(int)
    uu, vv, ww = (t * t * (3 - 2 * t) for t in (u, v, w))
    acc = 0.0
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                c = vec[perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255]]
                wt = np.array([u - di, v - dj, w - dk])
                acc += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * wt.dot(c)
    return acc


def test_k12_perlin_noise():
    vec, perm = perlin_tables(1)
    L = orc.lib()
    # at lattice points the offset vector of the only corner with weight 1 is zero
    for p in [(0, 0, 0), (3, -2, 7), (-5, 4, 255), (256, 1, -256)]:
        assert L.orc_kat_perlin_noise(vec.ctypes.data, perm.ctypes.data, fa(p)) == 0.0
    rng = np.random.default_rng(2)
    for p in rng.uniform(-300, 300, (50, 3)):
        p32 = p.astype(np.float32)
        got = L.orc_kat_perlin_noise(vec.ctypes.data, perm.ctypes.data, fa(p32))
        assert got == pytest.approx(perlin_noise_py(vec.astype(np.float64), perm, p32.astype(np.float64)), abs=2e-5)
    p = np.array([1.3, -2.7, 0.4], np.float32)
    turb = abs(sum(0.5 ** o * perlin_noise_py(vec.astype(np.float64), perm, p.astype(np.float64) * 2 ** o) for o in range(7)))
    assert L.orc_kat_perlin_turb(vec.ctypes.data, perm.ctypes.data, fa(p), 7) == pytest.approx(turb, abs=2e-5)


def test_checker_uses_hit_point():
    o = (C.c_float * 3)()
    L = orc.lib()
    L.orc_kat_checker(fa(1, 0, 0), fa(0, 1, 0), fa(0.1, 0.1, 0.1), o)   # sin(1)^3 > 0 -> even
    assert list(o) == [0.0, 1.0, 0.0]
    L.orc_kat_checker(fa(1, 0, 0), fa(0, 1, 0), fa(0.1, 0.1, -0.1), o)  # one negative factor -> odd
    assert list(o) == [1.0, 0.0, 0.0]


# ---- K13: ConstantMedium (shapes.rs:648-701) ------------------------------------------------------
def test_k13_medium_free_flight():
    L = orc.lib()
    o = out11()
    density, xi = 0.5, 0.25
    # unit direction through the centre of a radius-2 sphere: r1=3, r2=7, inside=4 ; hit=-ln(xi)/density
    L.orc_kat_medium_sphere(fa(0, 0, 0), 2.0, density, xi, ray((0, 0, -5), (0, 0, 1)), 0.005, INF, o)
    hit = -math.log(xi) / density
    assert o[0] == 1.0 and o[1] == pytest.approx(3 + hit, rel=1e-6)
    assert list(o[5:8]) == [1.0, 0.0, 0.0] and o[8] == 1.0 and o[9] == 0.0 and o[10] == 0.0
    # a sample that lands beyond the chord misses
    L.orc_kat_medium_sphere(fa(0, 0, 0), 2.0, density, 0.1, ray((0, 0, -5), (0, 0, 1)), 0.005, INF, o)
    assert -math.log(0.1) / density > 4 and o[0] == 0.0
    # shapes.rs:676-683 divide by the ray length twice: with |d| = 2, r1=1.5, r2=3.5, inside=(3.5-1.5)/2=1,
    # and t = r1 + (hit/2)
    xi2 = 0.7
    hit2 = -math.log(xi2) / density
    L.orc_kat_medium_sphere(fa(0, 0, 0), 2.0, density, xi2, ray((0, 0, -5), (0, 0, 2)), 0.005, INF, o)
    assert hit2 < 1 and o[0] == 1.0 and o[1] == pytest.approx(1.5 + hit2 / 2, rel=1e-6)
    # xi = 0 -> ln = -inf -> hit distance +inf -> miss
    L.orc_kat_medium_sphere(fa(0, 0, 0), 2.0, density, 0.0, ray((0, 0, -5), (0, 0, 1)), 0.005, INF, o)
    assert o[0] == 0.0
    # origin inside the boundary: r1 clamps to t_min
    L.orc_kat_medium_sphere(fa(0, 0, 0), 2.0, density, 0.9, ray((0, 0, 0), (0, 0, 1)), 0.005, INF, o)
    assert o[0] == 1.0 and o[1] == pytest.approx(0.005 + -math.log(0.9) / density, rel=1e-5)


# ---- camera (camera.rs:28-65, main.rs:471-474) -----------------------------------------------------
def test_camera_new_and_get_ray():
    cam = orc.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), math.radians(90), 2.0, 0.0, 1.0, 0.0, 1.0)
    # h=tan(45deg)=1: viewport 4 x 2 at focus 1, w=(0,0,1), u=(1,0,0), v=(0,1,0), llc=(-2,-1,-1)
    assert cam[3:6] == pytest.approx([-2, -1, -1], abs=1e-6)
    assert cam[6:9] == pytest.approx([4, 0, 0], abs=1e-6) and cam[9:12] == pytest.approx([0, 2, 0], abs=1e-6)
    assert cam[21] == 0.0
    r = (C.c_float * 7)()
    w, h = 5, 3
    orc.lib().orc_kat_camera_ray(cam.ctypes.data_as(C.POINTER(C.c_float)), w, h, 2, 1, 1, 0, r)
    jit = np.empty(2, np.float32)
    orc.lib().orc_rng_f32(1, 1 * w + 2, 0, 0xFF000000, 2, jit.ctypes.data)  # jitter counters
    u = (2 + jit[0]) / (w - 1)
    v = ((h - 1 - 1) + jit[1]) / (h - 1)
    assert list(r[0:3]) == [0.0, 0.0, 0.0]  # lens radius 0 (the disk sampler still runs, camera.rs:55)
    assert r[3] == pytest.approx(-2 + 4 * u, abs=1e-6) and r[4] == pytest.approx(-1 + 2 * v, abs=1e-6) and r[5] == -1.0
    tm = np.empty(1, np.float32)
    orc.lib().orc_rng_range(1, 1 * w + 2, 0, 0xFF000002, 0.0, 1.0, 1, tm.ctypes.data)  # time follows the jitter
    assert r[6] == tm[0] and 0.0 <= r[6] < 1.0


def test_camera_lens_draws_have_their_own_counters():
    # aperture 0.5 -> lens radius 0.25; u=(1,0,0), v=(0,1,0).  The lens point is the first pair from counter
    # 0xFE000000 on with x^2+y^2 >= 1 (vec3_helpers.rs:47-60, sic); jitter and time do not move with it.
    cam = orc.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), math.radians(90), 2.0, 0.5, 1.0, 0.0, 1.0)
    w, h = 5, 3
    for px in range(w * h):
        i, j = px % w, px // w
        r = (C.c_float * 7)()
        orc.lib().orc_kat_camera_ray(cam.ctypes.data_as(C.POINTER(C.c_float)), w, h, i, j, 7, 3, r)
        d = np.empty(64, np.float32)
        orc.lib().orc_rng_range(7, px, 3, 0xFE000000, -1.0, 1.0, 64, d.ctypes.data)
        k = next(k for k in range(32) if np.float32(d[2 * k] * d[2 * k]) + np.float32(d[2 * k + 1] * d[2 * k + 1]) >= 1.0)
        assert r[0] == np.float32(0.25) * d[2 * k] and r[1] == np.float32(0.25) * d[2 * k + 1] and r[2] == 0.0
        tm = np.empty(1, np.float32)
        orc.lib().orc_rng_range(7, px, 3, 0xFF000002, 0.0, 1.0, 1, tm.ctypes.data)
        assert r[6] == tm[0]


# ---- the RNG specification, restated in Python ---------------------------------------------------------
M32 = 0xFFFFFFFF


def mix32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def stream_key(seed, pixel, sample):
    h = mix32(seed ^ 0x51ED270B)
    k0 = mix32(pixel ^ h)
    k1 = mix32(k0 ^ mix32((sample + 0x68E31DA4) & M32))
    return k0, k1


def test_rng_matches_specification():
    seed, pixel, sample = 12345, 777, 42
    k0, k1 = stream_key(seed, pixel, sample)
    want = [mix32(((k0 + n) & M32) ^ k1) for n in range(16)]
    got = np.empty(16, np.uint32)
    orc.lib().orc_rng_u32(seed, pixel, sample, 0, 16, got.ctypes.data)
    assert got.tolist() == want
    first = 0xFE000000  # any starting counter, wrapping included
    orc.lib().orc_rng_u32(seed, pixel, sample, first, 16, got.ctypes.data)
    assert got.tolist() == [mix32(((k0 + first + n) & M32) ^ k1) for n in range(16)]
    f = np.empty(16, np.float32)
    orc.lib().orc_rng_f32(seed, pixel, sample, 0, 16, f.ctypes.data)
    assert f.tolist() == [np.float32(x >> 8) * np.float32(2.0 ** -24) for x in want]
    r = np.empty(16, np.float32)
    orc.lib().orc_rng_range(seed, pixel, sample, 0, -1.0, 1.0, 16, r.ctypes.data)
    exp = [np.float32((np.array([0x3F800000 | (x >> 9)], np.uint32).view(np.float32)[0] - np.float32(1.0)) * np.float32(2.0) + np.float32(-1.0)) for x in want]
    assert r.tolist() == exp
    salt = mix32(3 + 0x3C6EF372)
    hi = mix32(((k0 + 0x80000000 + 2 * 5) & M32) ^ k1 ^ salt)
    lo = mix32(((k0 + 0x80000001 + 2 * 5) & M32) ^ k1 ^ salt)
    xi = np.array([(((hi << 32) | lo) >> 12) | 0x3FF0000000000000], np.uint64).view(np.float64)[0] - 1.0
    assert orc.lib().orc_rng_medium_xi(seed, pixel, sample, 5, 3) == xi


def test_rng_is_uniform_enough():
    n = 1 << 16
    f = np.empty(n, np.float32)
    orc.lib().orc_rng_f32(1, 0, 0, 0, n, f.ctypes.data)
    assert abs(f.mean() - 0.5) < 0.005 and abs(f.var() - 1 / 12) < 0.002 and 0.0 <= f.min() and f.max() < 1.0
    # different pixels / samples give unrelated streams
    g = np.empty(n, np.float32)
    orc.lib().orc_rng_f32(1, 1, 0, 0, n, g.ctypes.data)
    assert abs(np.corrcoef(f, g)[0, 1]) < 0.02


# ---- K14: the oracle's BVHNode path against its linear Scene path ------------------------------------
@pytest.mark.parametrize("name", ["random_world", "cornell_box", "cornell_smoke", "final_scene", "simple_light"])
def test_k14_bvh_equals_linear_scan(name):
    sc = orc.Scene(name, seed=1)
    for which in (0, 1):
        rays, valid = sc.gen_rays(64, 64, 1.0, seed=5, which=which)
        rays = np.ascontiguousarray(rays[:, valid])
        a = sc.intersect(rays, seed=9, mode=0)
        b = sc.intersect(rays, seed=9, mode=1)
        assert np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["t"], b["t"])
        assert np.array_equal(a["normal"], b["normal"]) and np.array_equal(a["front"], b["front"])


def test_scene_inventory_matches_reference_builders():
    # main.rs:177-238: 6 walls + 2 blocks of 6 rects ; main.rs:240-305: 400 boxes*6 + 1 + 5 spheres + 2 boundaries
    # + earth + noise + 1000 spheres
    cb = orc.Scene("cornell_box")
    assert (cb.n_prims, cb.n_instances, cb.n_media, cb.n_top) == (18, 2, 0, 8)
    cs = orc.Scene("cornell_smoke")
    assert (cs.n_prims, cs.n_instances, cs.n_media, cs.n_top) == (18, 2, 2, 8)
    fs = orc.Scene("final_scene")
    assert (fs.n_prims, fs.n_instances, fs.n_media, fs.n_top) == (2400 + 1 + 4 + 2 + 2 + 1000, 1, 2, 11)
    rw = orc.Scene("random_world", seed=3)
    assert rw.n_top == rw.n_prims and 4 + 22 * 22 * 0.9 < rw.n_prims <= 4 + 22 * 22
    assert list(rw.background) == pytest.approx([0.7, 0.8, 1.0]) and rw.aperture == pytest.approx(0.1)


def test_image_height_presets():
    # main.rs:383 with the presets of main.rs:356-381
    assert orc.image_height(400, 16 / 9) == 225 and orc.image_height(1920, 16 / 9) == 1080
    assert orc.image_height(3840, 16 / 9) == 2160 and orc.image_height(1600, 3 / 2) == 1066
    assert orc.image_height(800, 1.0) == 800
