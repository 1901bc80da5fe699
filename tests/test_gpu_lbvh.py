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
