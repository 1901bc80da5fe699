"""Framebuffer exchange between ranks: the one collective of the render path.

Every rank renders the tiles it owns (tile index % nranks == rank, see rt_render) into a full-size
accumulation buffer.  Ownership is disjoint, so the complete frame is an all_gather of each rank's
packed pixels (w*h*12/N bytes per rank, each peer's share arrives over its own xGMI link) followed by
a scatter to frame order -- not a ring all-reduce of the whole buffer, which would be bound by one link.
Works on any torch.distributed backend (RCCL on GPUs, gloo on CPU for tests).
"""
import numpy as np
import torch

from . import owned_pixels


class TileGather:
    def __init__(self, width, height, rank, world, device, tile=16, owned_fn=owned_pixels):
        self.w, self.h, self.rank, self.world = width, height, rank, world
        owned = [owned_fn(width, height, tile, r, world) for r in range(world)]
        total = sum(len(o) for o in owned)
        if total != width * height:
            raise ValueError("tile ownership does not cover the frame exactly once")
        self.pad = max(len(o) for o in owned)
        self.n_mine = len(owned[rank])
        self.my_idx = torch.from_numpy(owned[rank].astype(np.int64)).to(device)
        # destination pixel of every gathered slot; padding slots land one past the frame
        dst = np.full((world, self.pad), width * height, dtype=np.int64)
        for r, o in enumerate(owned):
            dst[r, : len(o)] = o
        self.dst_idx = torch.from_numpy(dst.reshape(-1)).to(device)
        self.packed = torch.zeros((self.pad, 3), dtype=torch.float32, device=device)
        self.gathered = torch.empty((world * self.pad, 3), dtype=torch.float32, device=device)
        self.full = torch.zeros((height * width + 1, 3), dtype=torch.float32, device=device)

    def gather(self, accum):
        """accum: (h, w, 3) float32 with this rank's pixels filled in.  Returns the full (h, w, 3) frame."""
        self.packed[: self.n_mine] = accum.reshape(-1, 3)[self.my_idx]
        if self.world > 1:
            torch.distributed.all_gather_into_tensor(self.gathered, self.packed)
        else:
            self.gathered.copy_(self.packed)
        self.full.index_copy_(0, self.dst_idx, self.gathered)
        return self.full[:-1].view(self.h, self.w, 3)
