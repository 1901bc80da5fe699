"""Device-built world BVH (Morton-sorted linear BVH): structural invariants of the tree read back from the
device, and equivalence with the host-built tree -- any valid BVH must give the same hits, bit for bit."""
import numpy as np
import pytest

import rtamd

pytestmark = pytest.mark.gpu

REF_EMPTY = 0xFFFFFFFF


def prim_boxes(prims):
    """Boxes of sphere records (the synthetic scene holds static spheres only), without the widening."""
    assert np.all((prims["info"] & 7) == 0)
    c, r = prims["a"][:, :3].astype(np.float64), np.abs(prims["a"][:, 3]).astype(np.float64)[:, None]
    return c - r, c + r


def check_tree(nodes, prims, root, prim_lo, prim_hi):
    """Every primitive is referenced by exactly one leaf; every child box contains what lies below it."""
    n_nodes = len(nodes)
    assert root >> 30 == 0 and root < n_nodes
    seen_prim = np.zeros(len(prims), np.int64)
    seen_node = np.zeros(n_nodes, np.int64)
    seen_node[root] = 1
    # bottom-up boxes: process nodes level by level from the root to get an order, then fold in reverse
    order, frontier = [], np.array([root], np.int64)
    while len(frontier):
        order.append(frontier)
        kids = np.concatenate([nodes["c0"][frontier], nodes["c1"][frontier]])
        kids = kids[kids != REF_EMPTY]
        inner = kids[(kids >> 30) == 0].astype(np.int64)
        np.add.at(seen_node, inner, 1)
        leaves = kids[(kids >> 30) == 1]
        first, cnt = (leaves & 0x03FFFFFF).astype(np.int64), ((leaves >> 26) & 15).astype(np.int64) + 1
        for k in range(int(cnt.max()) if len(cnt) else 0):
            np.add.at(seen_prim, first[cnt > k] + k, 1)
        frontier = inner
    assert np.all(seen_node == 1), "a node is unreachable or has two parents"
    assert np.all(seen_prim == 1), "a primitive is missing from the leaves or referenced twice"
    box_lo = np.full((n_nodes, 3), np.inf)
    box_hi = np.full((n_nodes, 3), -np.inf)
    for level in reversed(order):
        for side, (lo_f, hi_f) in (("c0", ("lo0", "hi0")), ("c1", ("lo1", "hi1"))):
            ref = nodes[side][level]
            lo, hi = nodes[lo_f][level].astype(np.float64), nodes[hi_f][level].astype(np.float64)
            is_leaf, is_inner = (ref >> 30) == 1, ((ref >> 30) == 0) & (ref != REF_EMPTY)
            # leaf: the stored box contains the primitive's exact box
            li = level[is_leaf]
            first = (ref[is_leaf] & 0x03FFFFFF).astype(np.int64)
            assert np.all(((ref[is_leaf] >> 26) & 15) == 0)  # one primitive per leaf
            assert np.all(lo[is_leaf] <= prim_lo[first]) and np.all(hi[is_leaf] >= prim_hi[first])
            # inner child: the stored box equals the union of that child's two boxes
            ci = ref[is_inner].astype(np.int64)
            assert np.array_equal(lo[is_inner], box_lo[ci]) and np.array_equal(hi[is_inner], box_hi[ci])
            box_lo[level] = np.minimum(box_lo[level], lo)
            box_hi[level] = np.maximum(box_hi[level], hi)
            del li
    return len(order)


def test_device_tree_invariants():
    ctx = rtamd.Context(0)
    ctx.set_builder(1)
    info = ctx.load_scene("synth_spheres", seed=3, n_synth=100_000)
    assert info["builder"] == 1 and info["n_prims"] == 100_001 and info["n_nodes"] == 100_000
    nodes, prims, root = ctx.read_bvh()
    lo, hi = prim_boxes(prims)
    levels = check_tree(nodes, prims, root, lo, hi)
    assert levels <= info["max_depth"] <= 64
    assert sorted(prims["id"].tolist()) == list(range(100_001))
    ctx.close()


@pytest.mark.parametrize("name,n_synth,w", [("synth_spheres", 60_000, 320), ("random_world", 0, 200), ("final_scene", 0, 160)])
def test_device_and_host_trees_give_the_same_bits(name, n_synth, w):
    """Hits of a ray batch (ids, t, normals) and rendered images must not depend on which builder made the tree.
    random_world has moving spheres (boxes over the shutter interval); final_scene puts the device-built world
    tree under a host-built top level together with its instances."""
    h = w * 3 // 4
    rng = np.random.default_rng(5)
    out = []
    for builder in (0, 1):
        ctx = rtamd.Context(0)
        ctx.set_tail_paths(2048)
        ctx.set_builder(builder)
        info = ctx.load_scene(name, seed=2, n_synth=n_synth)
        assert info["builder"] == builder
        cam = ctx.preset_camera(w / h)
        res = []
        # rays from the camera position into a cone around the view direction, plus rays from inside the scene
        o = np.array(cam.as_array()[:3], np.float32)
        n = 20000
        target = np.zeros(3, np.float32) if name != "final_scene" else np.array([278, 278, 0], np.float32)
        d = (target - o)[None, :] + rng.normal(0, 0.25 * np.linalg.norm(target - o), (n, 3)).astype(np.float32)
        rays = np.concatenate([np.repeat(o[:, None], n, 1), d.T.astype(np.float32), rng.uniform(0, 1, (1, n)).astype(np.float32)])
        hit = ctx.intersect(rays.astype(np.float32), seed=3)
        res += [hit["prim"], hit["inst"], hit["t"], hit["normal"], hit["front"]]
        for pipe in (0, 1):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, 3, seed=7)
            res += [img, np.array([st["rays"]])]
        ctx.close()
        rng = np.random.default_rng(5)
        out.append(res)
    assert (out[0][0] >= 0).mean() > 0.05  # the batch does hit things
    for a, b in zip(*out):
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f")


@pytest.mark.parametrize("name,n_synth,w,builder", [("synth_spheres", 60_000, 320, 1), ("synth_spheres", 3000, 240, 0),
                                                    ("random_world", 0, 200, 0), ("random_world", 0, 200, 1),
                                                    ("final_scene", 0, 160, 0), ("final_scene", 0, 160, 1)])
def test_four_child_walk_gives_the_same_bits(name, n_synth, w, builder):
    """The kernels walk the tree two children at a time, or four (the grandchildren of every node in one 128-byte
    record).  Hits of a ray batch and images of every tree-walking pipeline must be the same; only the number of
    node fetches may differ (about half)."""
    h = w * 3 // 4
    out = []
    for width in (2, 4):
        rng = np.random.default_rng(9)
        ctx = rtamd.Context(0)
        ctx.set_tail_paths(2048)
        ctx.set_builder(builder)
        ctx.set_bvh_width(width)
        info = ctx.load_scene(name, seed=2, n_synth=n_synth)
        assert info["bvh_width"] == width and info["builder"] == builder
        cam = ctx.preset_camera(w / h)
        o = np.array(cam.as_array()[:3], np.float32)
        n = 20000
        target = np.zeros(3, np.float32) if name != "final_scene" else np.array([278, 278, 0], np.float32)
        d = (target - o)[None, :] + rng.normal(0, 0.25 * np.linalg.norm(target - o), (n, 3)).astype(np.float32)
        rays = np.concatenate([np.repeat(o[:, None], n, 1), d.T.astype(np.float32), rng.uniform(0, 1, (1, n)).astype(np.float32)])
        hit = ctx.intersect(rays.astype(np.float32), seed=3)
        res = [hit["prim"], hit["inst"], hit["t"], hit["normal"], hit["front"]]
        for pipe in (0, 1, 2):
            ctx.set_pipeline(pipe)
            img, st = ctx.render(w, h, cam, 3, seed=7)
            res += [img, np.array([st["rays"]])]
        ctx.set_pipeline(0)
        ctx.profile(events=False, count_visits=True)
        _, st = ctx.render(w, h, cam, 1, seed=7)
        visits = st["node_visits"] / st["rays"]
        ctx.close()
        out.append((res, visits))
    assert (out[0][0][0] >= 0).mean() > 0.05
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f")
    assert out[1][1] < 0.75 * out[0][1]  # four-child fetches per ray against two-child fetches


def test_four_child_form_is_refused_for_other_widths_and_skipped_for_flat_scenes():
    ctx = rtamd.Context(0)
    with pytest.raises(rtamd.RtError):
        ctx.set_bvh_width(3)
    ctx.set_bvh_width(4)
    assert ctx.load_scene("cornell_box", seed=1)["bvh_width"] == 2  # traced without a tree
    ctx.set_bvh_width(-1)  # by size: four-wide once the nodes outgrow one XCD's L2
    assert ctx.load_scene("random_world", seed=1)["bvh_width"] == 2
    assert ctx.load_scene("synth_spheres", seed=1, n_synth=70_000)["bvh_width"] == 4
    ctx.close()


# ---- degenerate inputs ------------------------------------------------------------------------------------
def _raw_scene(ctx, builder, c0r, rects=None, moving=None):
    import ctypes as C
    t = rtamd.rt_texture(kind=0, color=(C.c_float * 3)(0.5, 0.5, 0.5))
    m = rtamd.rt_material(kind=0, tex=0)
    n = len(c0r)
    sph = dict(c0r=c0r, mat=np.zeros(n, np.uint32), id=np.arange(n, dtype=np.uint32))
    if moving is not None:
        sph.update(c1=moving[0], t01=moving[1], moving=moving[2])
    if rects is not None:
        rects = dict(rects, mat=np.zeros(len(rects["k"]), np.uint32), id=np.arange(n, n + len(rects["k"]), dtype=np.uint32))
    ctx.set_builder(builder)
    ctx.set_scene(spheres=sph, rects=rects, materials=[m], textures=[t])
    return ctx.info()


def _ray_fan(rng, n, origin, spread):
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    o = (np.asarray(origin, np.float32)[None, :] + rng.normal(0, spread, (n, 3))).astype(np.float32)
    return np.concatenate([o.T, d.T, rng.uniform(0, 1, (1, n)).astype(np.float32)]).astype(np.float32)


@pytest.mark.parametrize("case", ["two", "identical", "collinear", "mixed_moving_rects"])
def test_device_builder_on_degenerate_worlds(case, monkeypatch):
    """Worlds that stress the Morton sort: two primitives, thousands of identical ones (all keys equal, ties broken by
    index), centres on one line (two axes without extent), and a mix of rects and moving spheres.  The device tree
    must give the hits of the host tree (flat tracing is off, so both really walk a tree)."""
    monkeypatch.setenv("RTAMD_FLAT_MAX", "0")
    rng = np.random.default_rng(7)
    rects = moving = None
    if case == "two":
        c0r = np.float32([[0, 0, 0, 1], [3, 0, 0, 1]])
    elif case == "identical":
        c0r = np.tile(np.float32([[1, 2, 3, 0.5]]), (3000, 1))
    elif case == "collinear":
        c0r = np.stack([np.linspace(-50, 50, 4000), np.zeros(4000), np.zeros(4000), np.full(4000, 0.01)], 1).astype(np.float32)
    else:
        n = 3000
        c0r = np.concatenate([rng.uniform(-20, 20, (n, 3)), rng.uniform(0.05, 0.6, (n, 1))], 1).astype(np.float32)
        c1 = (c0r[:, :3] + rng.uniform(-1, 1, (n, 3))).astype(np.float32)
        moving = (c1, np.tile(np.float32([[0.0, 1.0]]), (n, 1)), (rng.uniform(0, 1, n) < 0.5).astype(np.uint8))
        m = 2000
        lo = rng.uniform(-20, 19, (m, 2))
        rects = dict(axis=rng.integers(0, 3, m).astype(np.uint8), k=rng.uniform(-20, 20, m).astype(np.float32),
                     bounds=np.concatenate([lo, lo + rng.uniform(0.1, 1.0, (m, 2))], 1).astype(np.float32))
    rays = _ray_fan(rng, 30000, c0r[:, :3].mean(0), 2.0 * c0r[:, :3].std(0).max() + 1.0)
    out = []
    for builder in (0, 1):
        ctx = rtamd.Context(0)
        info = _raw_scene(ctx, builder, c0r, rects, moving)
        assert info["builder"] == builder and info["stack_entries"] <= 64
        hit = ctx.intersect(rays, seed=1)
        out.append([hit["prim"], hit["t"], hit["normal"], hit["front"]])
        ctx.close()
    assert (out[0][0] >= 0).any()
    if case == "identical":  # exact ties everywhere: the highest id wins (scene.rs:32-38, "later shape wins")
        assert set(np.unique(out[1][0]).tolist()) <= {-1, 2999}
    for a, b in zip(*out):
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f")


def test_empty_world_renders_the_background():
    """A Scene with no shapes: every ray misses (main.rs:60-62), every pixel holds spp sequential adds of the background."""
    ctx = rtamd.Context(0)
    ctx.set_scene()
    assert ctx.info()["n_prims"] == 0
    hit = ctx.intersect(_ray_fan(np.random.default_rng(1), 100, (0, 0, 0), 1.0))
    assert np.all(hit["prim"] == -1) and np.all(np.isinf(hit["t"]))
    cam = rtamd.camera_new((0, 0, -5), (0, 0, 0), (0, 1, 0), 0.5, 1.5, 0.0, 10.0, 0.0, 1.0)
    bg = np.float32([0.25, 0.5, 0.75])
    img, st = ctx.render(33, 22, cam, 5, background=bg)
    acc = np.zeros(3, np.float32)
    for _ in range(5):
        acc = acc + bg
    assert st["rays"] == st["paths"] == 33 * 22 * 5 and np.all(img == acc)
    ctx.close()


# ---- random raw scenes through every pipeline ------------------------------------------------------------------
@pytest.mark.parametrize("seed,n_sph,n_rect", [(1, 5, 3), (2, 20, 12), (3, 400, 150), (4, 3000, 0)])
def test_random_raw_scenes_agree_across_pipelines_and_builders(seed, n_sph, n_rect):
    """Random worlds sent through rt_scene_set_* (spheres of all five material kinds, some moving, rects of all three
    axes): split, fused, hybrid and whole-path pipelines, with and without the tail kernel, host- and device-built
    trees must all give one image and one ray count."""
    import ctypes as C
    rng = np.random.default_rng(seed)
    texs = [rtamd.rt_texture(kind=0, color=(C.c_float * 3)(*rng.uniform(0.2, 0.9, 3))) for _ in range(4)]
    texs.append(rtamd.rt_texture(kind=1, odd=0, even=1))  # checker of two solids
    mats = [rtamd.rt_material(kind=0, tex=0), rtamd.rt_material(kind=0, tex=4),
            rtamd.rt_material(kind=1, albedo=(C.c_float * 3)(0.8, 0.7, 0.6), fuzz=0.3),
            rtamd.rt_material(kind=2, ior=1.5), rtamd.rt_material(kind=3, tex=2), rtamd.rt_material(kind=4, tex=3)]
    c0r = np.concatenate([rng.uniform(-8, 8, (n_sph, 3)), rng.uniform(0.2, 1.2, (n_sph, 1))], 1).astype(np.float32)
    sph = dict(c0r=c0r, c1=(c0r[:, :3] + rng.uniform(-0.5, 0.5, (n_sph, 3))).astype(np.float32),
               t01=np.tile(np.float32([[0.0, 1.0]]), (n_sph, 1)), moving=(rng.uniform(0, 1, n_sph) < 0.3).astype(np.uint8),
               mat=rng.integers(0, len(mats), n_sph).astype(np.uint32), id=np.arange(n_sph, dtype=np.uint32))
    rects = None
    if n_rect:
        lo = rng.uniform(-8, 6, (n_rect, 2))
        rects = dict(axis=rng.integers(0, 3, n_rect).astype(np.uint8), k=rng.uniform(-8, 8, n_rect).astype(np.float32),
                     bounds=np.concatenate([lo, lo + rng.uniform(0.5, 3.0, (n_rect, 2))], 1).astype(np.float32),
                     mat=rng.integers(0, len(mats), n_rect).astype(np.uint32), id=np.arange(n_sph, n_sph + n_rect, dtype=np.uint32))
    cam = rtamd.camera_new((0, 2, -25), (0, 0, 0), (0, 1, 0), 0.7, 1.5, 0.2, 25.0, 0.0, 1.0)
    w, h, spp = 150, 100, 4
    out = []
    for builder in (0, 1):
        ctx = rtamd.Context(0)
        ctx.set_builder(builder)
        ctx.set_scene(spheres=sph, rects=rects, materials=mats, textures=texs)
        for pipe in (0, 1, 2, 3):
            for tail in (0, 1 << 30):
                ctx.set_pipeline(pipe)
                ctx.set_tail_paths(tail)
                img, st = ctx.render(w, h, cam, spp, depth=20, seed=seed, background=np.float32([0.6, 0.7, 0.9]))
                assert np.isfinite(img).all()
                out.append((img, st["rays"]))
        ctx.close()
    assert out[0][1] > w * h * spp  # something was hit
    for img, rays in out[1:]:
        assert rays == out[0][1] and np.array_equal(img, out[0][0])
