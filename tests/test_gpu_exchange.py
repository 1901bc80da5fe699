"""Frame exchange behind the C ABI (rt_pack_owned, rt_scatter_gathered, rt_gather_frame, rt_comm_*).

One GPU is all a test box has, so the N-rank flow is checked with virtual ranks: every rank's share of a frame is
rendered into its own buffer, packed with rt_pack_owned, the blocks are laid side by side exactly as ncclAllGather
would leave them, and rt_scatter_gathered must give the single-rank frame bit for bit.  The collective itself is
exercised with a communicator of one rank (RCCL refuses two ranks on one device)."""
import numpy as np
import pytest

import rtamd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,tile,nranks", [(200, 150, 32, 3), (96, 96, 32, 8), (130, 70, 16, 5), (64, 64, 32, 1), (33, 31, 32, 4)])
def test_virtual_ranks_pack_and_scatter_reproduce_the_single_rank_frame(gpu_ctx, w, h, tile, nranks):
    import torch
    gpu_ctx.load_scene("cornell_box", seed=1)
    cam = gpu_ctx.preset_camera(w / h)
    one = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    gpu_ctx.render(w, h, cam, 3, seed=5, accum=one, tile_size=tile)
    block = rtamd.frame_block_pixels(w, h, tile, nranks)
    owned = [rtamd.owned_pixels(w, h, tile, r, nranks) for r in range(nranks)]
    assert block == max(len(o) for o in owned) and sum(len(o) for o in owned) == w * h
    gathered = torch.full((nranks, block, 3), float("nan"), dtype=torch.float32, device="cuda")
    for r in range(nranks):
        mine = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        gpu_ctx.render(w, h, cam, 3, seed=5, accum=mine, tile_size=tile, rank=r, nranks=nranks)
        gpu_ctx.pack_owned(mine, gathered[r], tile, r, nranks)
        gpu_ctx.synchronize()
        # the block is the rank's pixels in rt_owned_pixels order
        want = one.reshape(-1, 3)[torch.from_numpy(owned[r].astype(np.int64)).cuda()]
        assert torch.equal(gathered[r, : len(owned[r])], want)
    frame = torch.zeros_like(one)
    gpu_ctx.scatter_gathered(gathered, frame, tile, nranks)
    gpu_ctx.synchronize()
    assert torch.equal(frame, one)


def test_gather_frame_single_rank_host_and_device(gpu_ctx):
    import torch
    rng = np.random.default_rng(4)
    acc = rng.uniform(0, 9, (75, 101, 3)).astype(np.float32)
    assert np.array_equal(gpu_ctx.gather_frame(acc), acc)  # host in, host out
    dev = torch.from_numpy(acc).cuda()
    assert torch.equal(gpu_ctx.gather_frame(dev, tile_size=16), dev)
    with pytest.raises(ValueError):
        gpu_ctx.gather_frame(acc.astype(np.float64))


def test_communicator_of_one_rank_runs_the_collective_path():
    """rt_comm_unique_id + rt_comm_create load RCCL at run time and build a real communicator; with one rank the
    gather is the identity, through the code that N ranks run, the ncclAllGather call on the communicator included
    (host and device frames, two tile sizes, twice in a row)."""
    import torch
    ctx = rtamd.Context(0)
    try:
        uid = rtamd.comm_unique_id()
    except rtamd.RtError as e:
        ctx.close()
        pytest.skip(f"no RCCL library on this box: {e}")
    assert len(uid) == 128 and any(uid)
    ctx.comm_create(uid, 0, 1)
    ctx.load_scene("two_spheres", seed=1)
    w, h = 160, 90
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    ctx.render(w, h, ctx.preset_camera(16 / 9), 2, seed=1, accum=acc)
    assert torch.equal(ctx.gather_frame(acc), acc)
    assert torch.equal(ctx.gather_frame(acc, tile_size=8), acc)
    host = acc.cpu().numpy()
    assert np.array_equal(ctx.gather_frame(host), host)
    ctx.comm_destroy()
    assert torch.equal(ctx.gather_frame(acc), acc)  # (and without a communicator again)
    with pytest.raises(rtamd.RtError):
        ctx.comm_create(uid, 3, 2)  # rank >= nranks
    ctx.close()
