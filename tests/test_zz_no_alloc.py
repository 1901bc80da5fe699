"""rt_intersect_batch_noalloc -- the entry point that allocates nothing -- against rt_intersect_batch and against the oracle,
through the C ABI.

Its kernels take the overflow flag, and for host callers the staging of rays and results, from a static device array
whose address the library looks up once per device (hipGetSymbolAddress); every call on a GPU shares that array, which
keeps the last call's contents.  Hence: two calls in a row, the second with a smaller input, each compared bit for
bit with the allocating entry point; host and device arrays; the size limit; and no growth of device memory."""
import numpy as np
import pytest

import orc
import rtamd

pytestmark = pytest.mark.gpu

KEYS = ("prim", "inst", "t", "normal", "front", "uv", "point")


def _same(a, b, what):
    for k in KEYS:
        x, y = (v.cpu().numpy() if rtamd._is_torch(v) else v for v in (a[k], b[k]))
        assert np.array_equal(x, y, equal_nan=True), f"{what}: {k} differs"


@pytest.mark.parametrize("name", ["final_scene", "cornell_smoke"])
def test_no_alloc_entry_point_matches_the_allocating_one(name):
    import torch
    ctx = rtamd.Context(0)
    ctx.load_scene(name, seed=1)
    rays, valid = orc.Scene(name, seed=1).gen_rays(256, 256, 1.0, seed=5, which=1)
    rays = np.ascontiguousarray(rays[:, valid])[:, :60000]
    n_big, n_small = rays.shape[1], 1777
    assert 30000 < n_big <= 65536
    free_before = torch.cuda.mem_get_info()[0]
    for n in (n_big, n_small, n_big):  # the second call is smaller: whatever the first left in the scratch must not show
        r = np.ascontiguousarray(rays[:, :n])
        ref = ctx.intersect(r, seed=3)
        _same(ctx.intersect(r, seed=3, noalloc=True), ref, f"{name}/host/{n}")
        dev = torch.from_numpy(r).cuda()
        _same(ctx.intersect(dev, seed=3, noalloc=True), ref, f"{name}/device/{n}")
        # a subset of the outputs (the others NULL)
        part = ctx.intersect(r, seed=3, noalloc=True, outputs=("prim", "t"))
        assert np.array_equal(part["prim"], ref["prim"]) and np.array_equal(part["t"], ref["t"], equal_nan=True)
    assert (ref["prim"] != -1).mean() > 0.3
    # ... and against the oracle directly (not only against the allocating entry point): ids bit for bit, t and normals to 1e-5
    oref = orc.Scene(name, seed=1).intersect(np.ascontiguousarray(rays[:, :n_big]), seed=3, mode=1)
    got = ctx.intersect(np.ascontiguousarray(rays[:, :n_big]), seed=3, noalloc=True)
    assert np.array_equal(got["prim"], oref["prim"]) and np.array_equal(got["inst"], oref["inst"])
    hit = oref["prim"] != -1
    np.testing.assert_allclose(got["t"][hit], oref["t"][hit], rtol=1e-5)
    np.testing.assert_allclose(got["normal"][hit], oref["normal"][hit], atol=1e-5)
    # repeated calls leave the device's free memory where it was (torch's own tensors aside: same sizes every time)
    for _ in range(3):
        ctx.intersect(rays[:, :n_small].copy(), seed=3, noalloc=True)
    assert torch.cuda.mem_get_info()[0] >= free_before - (64 << 20)
    # the limit of the static scratch: refused with a message, and the context stays usable
    too_many = np.ascontiguousarray(np.tile(rays[:, :1], (1, 65537)))
    with pytest.raises(rtamd.RtError) as e:
        ctx.intersect(too_many, seed=3, noalloc=True)
    assert e.value.code == -1 and "65536" in str(e.value)
    exact = np.ascontiguousarray(np.tile(rays[:, :4], (1, 16384)))
    _same(ctx.intersect(exact, seed=3, noalloc=True), ctx.intersect(exact, seed=3), f"{name}/65536")
    ctx.close()


def test_no_alloc_entry_point_before_commit_and_empty():
    ctx = rtamd.Context(0)
    with pytest.raises(rtamd.RtError) as e:
        ctx.intersect(np.zeros((7, 4), np.float32), noalloc=True)
    assert e.value.code == -4
    ctx.load_scene("two_spheres", seed=1)
    out = ctx.intersect(np.zeros((7, 0), np.float32), noalloc=True)
    assert out["prim"].shape == (0,)
    ctx.close()
