"""Statistical known answers, derived from the reference's source lines and from nothing else.

The oracle and the device path share their deliberate departures from the reference (media evaluated last, the tie
rule, a counter-based generator in place of thread_rng), so a reading mistake made once would pass every comparison of
the two.  The checks below do not compare the two with each other: each states what the cited lines of the reference
imply as a probability or an expectation, and tests the oracle's entry points (CPU) and the C ABI (GPU, raw scenes sent
through rt_scene_set_*) against that number, within four standard deviations of the sample mean.

  transmittance of a constant medium      shapes.rs:675-681   P(pass) = exp(-density * chord)  (unit direction)
  Fresnel reflection at normal incidence  materials.rs:109-148 P(reflect) = ((1 - 1.5) / (1 + 1.5))^2 = 0.04
  Lambertian lobe                         materials.rs:45-61   direction = normal + unit vector: cosine-weighted, mean cosine
                                                              2/3, P(angle to the normal < theta) = sin^2 theta
  lens sampler                            vec3_helpers.rs:47-60 returns the first point with |p|^2 >= 1 (sic): outside the disk
  throughput times background             main.rs:44-63        one diffuse bounce off a convex body: albedo * background
"""
import ctypes as C
import math

import numpy as np
import pytest

import orc
import rtamd

N_ORACLE = 1 << 15


def four_sigma(p, n):
    return 4.0 * math.sqrt(p * (1.0 - p) / n)


def ray7(o, d, time=0.0):
    return orc.fa(o, d, time)


# ---- the oracle (CPU) ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("density, scale", [(0.4, 1.0), (0.25, 1.0), (0.4, 2.0)])
def test_oracle_medium_transmittance(density, scale):
    """A ray through the centre of a medium of radius 2: chord 4.  shapes.rs:675-681: the flight is -ln(xi) / density and the
    ray passes when that exceeds (t2 - t1) / |d| -- the reference DIVIDES the t-interval by the ray's length, so for a
    direction of length s the interval is 4 / s and the medium sees 4 / s^2."""
    L = orc.lib()
    out = (C.c_float * 11)()
    ray = ray7((0, 0, -5), (0, 0, scale))
    passed = 0
    for i in range(N_ORACLE):
        xi = L.orc_rng_medium_xi(21, i, 0, 0, 0)
        L.orc_kat_medium_sphere(orc.fa(0, 0, 0), 2.0, density, xi, ray, 0.005, float("inf"), out)
        passed += out[0] == 0.0
    p = math.exp(-density * 4.0 / (scale * scale))
    assert abs(passed / N_ORACLE - p) < four_sigma(p, N_ORACLE)


def test_oracle_dielectric_reflects_four_per_cent_at_normal_incidence():
    L = orc.lib()
    out = (C.c_float * 11)()
    reflected = 0
    for i in range(N_ORACLE):
        L.orc_kat_scatter(2, orc.fa(0, 0, 0), 1.5, ray7((0, 0, -5), (0, 0, 1)), orc.fa(0, 0, -1), orc.fa(0, 0, -1), 1, 5, i, 0, 0, out)
        assert out[0] == 1.0 and list(out[7:10]) == [1.0, 1.0, 1.0]  # always scatters, attenuation 1 (materials.rs:118, 147)
        reflected += out[6] < 0.0
    assert abs(reflected / N_ORACLE - 0.04) < four_sigma(0.04, N_ORACLE)


def test_oracle_lambertian_lobe_is_cosine_weighted():
    L = orc.lib()
    out = (C.c_float * 11)()
    n = np.array([0.0, 1.0, 0.0])
    cos = np.empty(N_ORACLE)
    for i in range(N_ORACLE):
        L.orc_kat_scatter(0, orc.fa(0.8, 0.5, 0.25), 0.0, ray7((0, 1, 0), (0.3, -1, 0.1)), orc.fa(0, 0, 0), orc.fa(*n), 1, 9, i, 0, 0, out)
        d = np.array(out[4:7], dtype=np.float64)
        assert out[0] == 1.0 and [round(float(x), 6) for x in out[7:10]] == [0.8, 0.5, 0.25]  # attenuation = albedo (materials.rs:57)
        cos[i] = d @ n / np.linalg.norm(d)
    assert cos.min() >= -1e-6
    assert abs(cos.mean() - 2.0 / 3.0) < 4.0 * math.sqrt((0.5 - 4.0 / 9.0) / N_ORACLE)  # E[cos^2] = 1/2 under the cosine law
    p = 0.36  # within the cone of sin(theta) = 0.6 around the normal
    assert abs((cos > 0.8).mean() - p) < four_sigma(p, N_ORACLE)


def test_oracle_lens_points_lie_outside_the_unit_disk():
    """camera.rs:55-57 with vec3_helpers.rs:47-60: offset = u * (r * x) + v * (r * y) with (x, y) in [-1, 1)^2 and
    x^2 + y^2 >= 1 -- the corners of the square, never the disk the book samples."""
    cam = orc.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), math.radians(40), 1.0, 2.0, 10.0, 0.0, 1.0)  # lens radius 1
    r = (C.c_float * 7)()
    w = h = 16
    rr = np.empty((w * h * 8, 2))
    k = 0
    for px in range(w * h):
        for s in range(8):
            orc.lib().orc_kat_camera_ray(cam.ctypes.data_as(C.POINTER(C.c_float)), w, h, px % w, px // w, 3, s, r)
            rr[k] = (r[0], r[1])
            k += 1
    assert (np.abs(rr) <= 1.0).all() and ((rr ** 2).sum(axis=1) >= 1.0 - 1e-6).all()
    # uniform over the square minus the disk: each of the four corners holds a quarter
    for sx in (-1, 1):
        for sy in (-1, 1):
            share = ((rr[:, 0] * sx > 0) & (rr[:, 1] * sy > 0)).mean()
            assert abs(share - 0.25) < four_sigma(0.25, len(rr))


# ---- the C ABI (GPU) -------------------------------------------------------------------------------------------------
def solid(color):
    return rtamd.rt_texture(kind=0, color=(C.c_float * 3)(*color))


@pytest.fixture()
def ctx():
    c = rtamd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("density, scale", [(0.4, 1.0), (0.25, 1.0), (0.4, 2.0)])
def test_abi_medium_transmittance(ctx, density, scale):
    n = 1 << 18
    mats = [rtamd.rt_material(kind=0, tex=0), rtamd.rt_material(kind=4, tex=0)]
    m = rtamd.rt_group(kind=2, has_xform=0, neg_inv_density=-1.0 / density, mat=1)
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 2]], mat=[0], id=[0], group=[1]), groups=[m], materials=mats, textures=[solid((1, 1, 1))])
    rays = np.repeat(np.array([[0], [0], [-5], [0], [0], [scale], [0]], np.float32), n, axis=1)
    r = ctx.intersect(rays, seed=21)
    assert set(np.unique(r["prim"])) <= {-1, -2}
    p = math.exp(-density * 4.0 / (scale * scale))
    assert abs((r["prim"] == -1).mean() - p) < four_sigma(p, n)
    # the hits are spread as the exponential law says: final_t = t1 + flight / |d| (shapes.rs:683), and half of all flights
    # are shorter than ln 2 / density
    flight = (r["t"][r["prim"] == -2] - 3.0 / scale) * scale
    med, inside = math.log(2.0) / density, 4.0 / (scale * scale)
    assert flight.min() >= 0.0 and flight.max() <= inside * (1 + 1e-5)
    if med < inside:
        p_med = 0.5 / (1.0 - p)  # P(flight < median | flight < inside)
        assert abs((flight < med).mean() - p_med) < four_sigma(p_med, len(flight))


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1])
def test_abi_dielectric_reflects_four_per_cent_at_normal_incidence(ctx, pipeline):
    """A ball of glass (ior 1.5) with a lamp at its centre, seen from far away through a 0.2 degree lens: every camera ray
    meets the glass head-on, is turned back with probability 0.04 (black background) and otherwise goes straight on to
    the lamp -- the pixel mean is 0.96 of the lamp's emission."""
    E = (1.0, 2.0, 4.0)
    mats = [rtamd.rt_material(kind=2, ior=1.5), rtamd.rt_material(kind=3, tex=0)]
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1], [0, 0, 0, 0.5]], mat=[0, 1], id=[0, 1]), materials=mats, textures=[solid(E)])
    ctx.set_pipeline(pipeline)
    cam = rtamd.camera_new((0, 0, -10), (0, 0, 0), (0, 1, 0), math.radians(0.2), 1.0, 0.0, 10.0, 0.0, 1.0)
    w = h = 32
    spp = 256
    img, st = ctx.render(w, h, cam, spp, background=(0, 0, 0), seed=4)
    mean = img.reshape(-1, 3).sum(axis=0) / (w * h * spp)
    for k in range(3):
        assert abs(mean[k] / E[k] - 0.96) < four_sigma(0.04, w * h * spp)
    assert st["rays"] == 2 * w * h * spp  # glass, then either the lamp or nothing


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1])
def test_abi_lambertian_lobe_is_cosine_weighted(ctx, pipeline):
    """A diffuse floor under a round lamp of radius 3 whose centre hangs 5 above the point the camera looks at: the lamp
    fills the cone of sin(theta) = 3/5 around the normal, which a cosine-weighted direction falls into with probability
    sin^2(theta) = 0.36 (a uniform one: 0.2).  Pixel mean = albedo * emission * 0.36."""
    a, E = (0.8, 0.5, 0.25), 2.0
    mats = [rtamd.rt_material(kind=0, tex=0), rtamd.rt_material(kind=3, tex=1)]
    texs = [solid(a), solid((E, E, E))]
    ctx.set_scene(rects=dict(axis=[0], k=[0.0], bounds=[[-50, -50, 50, 50]], mat=[0], id=[0]),
                  spheres=dict(c0r=[[0, 0, 5, 3]], mat=[1], id=[1]), materials=mats, textures=texs)
    ctx.set_pipeline(pipeline)
    cam = rtamd.camera_new((0, 0, 1), (0, 0, 0), (0, 1, 0), math.radians(0.1), 1.0, 0.0, 1.0, 0.0, 1.0)
    w = h = 32
    spp = 256
    img, st = ctx.render(w, h, cam, spp, background=(0, 0, 0), seed=6)
    mean = img.reshape(-1, 3).sum(axis=0) / (w * h * spp)
    for k in range(3):
        assert abs(mean[k] / (a[k] * E) - 0.36) < four_sigma(0.36, w * h * spp)


@pytest.mark.gpu
def test_abi_lens_points_lie_outside_the_unit_disk(ctx):
    """A small lamp half way to the focus plane, seen through a lens of radius 1: a camera ray through the focus-plane point
    P from the lens point o passes the lamp's plane at (o + P) / 2, so the lamp shows where P = -o: the picture IS the
    distribution of lens points.  vec3_helpers.rs:47-60 keeps the first point with |p|^2 >= 1: the four corners of the
    square are lit and the disk in the middle is black -- exactly black, no sample may fall there."""
    mats = [rtamd.rt_material(kind=3, tex=0)]
    ctx.set_scene(spheres=dict(c0r=[[0, 0, -5, 0.05]], mat=[0], id=[0]), materials=mats, textures=[solid((1, 1, 1))])
    half = 1.2  # half width of the window on the focus plane (distance 10)
    cam = rtamd.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), 2.0 * math.atan(half / 10.0), 1.0, 2.0, 10.0, 0.0, 1.0)
    w = h = 96
    img, _ = ctx.render(w, h, cam, 64, background=(0, 0, 0), seed=2)
    lum = img[..., 0]
    c = (np.arange(w) + 0.5) / w * 2.0 * half - half
    X, Y = np.meshgrid(c, c)
    R = np.hypot(X, Y)
    assert lum[R < 0.8].sum() == 0.0                                   # inside the disk (the lamp blurs the edge by 0.1)
    corners = (np.abs(X) > 0.8) & (np.abs(X) < 0.95) & (np.abs(Y) > 0.8) & (np.abs(Y) < 0.95)
    assert (lum[corners] > 0).mean() > 0.9                             # the corners of the square
    assert lum[(np.abs(X) > 1.12) | (np.abs(Y) > 1.12)].sum() == 0.0   # nothing beyond the square [-1, 1]^2
    # density: uniform over the square minus the disk, area 4 - pi; the lamp covers pi * 0.1^2 of it
    expect = math.pi * 0.1 ** 2 / (4.0 - math.pi)
    assert lum[corners].mean() / 64 == pytest.approx(expect, rel=0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 3])
def test_abi_one_diffuse_bounce_returns_albedo_times_background(ctx, pipeline):
    """ray_color (main.rs:44-63): hit -> attenuation * ray_color(scattered); miss -> background.  A ray scattered off a convex
    body leaves it, so a path that hits the ball carries albedo * background.  (Not every one in f32: normal + unit vector
    is not normalised (materials.rs:47), and when the two nearly cancel the ray is so short that t_min = 0.005 no longer
    clears the surface it starts on -- about one path in 20 000 meets the ball again from inside and dies at the depth
    limit, in the reference's arithmetic as here.  Such a pixel can only be darker.)"""
    a, B = (0.8, 0.5, 0.25), (0.5, 1.0, 2.0)
    ctx.set_scene(spheres=dict(c0r=[[0, 0, 0, 1]], mat=[0], id=[0]), materials=[rtamd.rt_material(kind=0, tex=0)], textures=[solid(a)])
    ctx.set_pipeline(pipeline)
    cam = rtamd.camera_new((0, 0, -10), (0, 0, 0), (0, 1, 0), math.radians(2.0), 1.0, 0.0, 10.0, 0.0, 1.0)
    w = h = 24
    spp = 32
    img, st = ctx.render(w, h, cam, spp, background=B, seed=8)
    for k in range(3):
        want = np.float32(a[k]) * np.float32(B[k])
        got = img[..., k] / spp
        exact = np.abs(got - want) <= 1e-5 * want
        assert exact.mean() >= 0.99 and (got[~exact] < want).all()
    assert 2 * w * h * spp <= st["rays"] <= 2 * w * h * spp * 1.01
