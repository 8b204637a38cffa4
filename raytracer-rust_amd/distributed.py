"""Row-strip partition of one image across the GPUs of a node + the final gather.

Rows are the reference's unit of independence (own RNG stream, disjoint output slice,
src/renderer.rs:87-91).  Strips of `strip_rows` rows are dealt round-robin to ranks
((y // strip_rows) % world == rank) -- per-row cost is very uneven (sky rows cost one ray per
sample), so contiguous bands would load-balance badly.  Every rank renders its strips with the
full scene resident in its own HBM (no data-path collective) and the only exchange step is ONE
collective over the packed rows (all_gather_into_tensor: RCCL over xGMI when the backend is "nccl", the
same call under gloo in the CPU tests), followed by a single index_select on rank 0 that de-interleaves the strips.  Because the RNG is keyed by the absolute row y,
the gathered image is bit-identical to the 1-GPU image.

Backend-agnostic on purpose: the same code runs under gloo on CPU tensors (tests/test_distributed.py)
and under nccl (= RCCL) on GPU tensors (bench.py).
"""
from dataclasses import dataclass
from typing import List

import torch
import torch.distributed as dist


@dataclass
class StripPlan:
    height: int
    width: int
    world: int
    strip_rows: int
    rows: List[List[int]]          # rows[r] = absolute rows rendered by rank r, ascending
    max_rows: int                  # padded per-rank row count (gather needs equal sizes)
    perm: torch.Tensor             # [height] position of absolute row y in the rank-major padded buffer

    def perm_on(self, device):
        """`perm` on `device`, uploaded once (the gather runs every step)."""
        cache = self.__dict__.setdefault("_perm_cache", {})
        key = str(device)
        if key not in cache:
            cache[key] = self.perm.to(device)
        return cache[key]

    def options_for(self, abi, rank, **kw):
        return abi.Options.make(strip_rows=self.strip_rows, n_parts=self.world, part=rank, **kw)


def choose_strip_rows(height, world, preferred=4):
    """Largest strip <= preferred that deals every rank the same number of rows (no padding), else preferred."""
    for s in range(preferred, 0, -1):
        if height % (s * world) == 0:
            return s
    return preferred


def make_plan(height, width, world, strip_rows=None):
    if strip_rows is None:
        strip_rows = choose_strip_rows(height, world)
    rows = [[y for y in range(height) if (y // strip_rows) % world == r] for r in range(world)]
    max_rows = max(len(r) for r in rows)
    perm = torch.empty(height, dtype=torch.long)
    for r, rr in enumerate(rows):
        for i, y in enumerate(rr):
            perm[y] = r * max_rows + i
    return StripPlan(height, width, world, strip_rows, rows, max_rows, perm)


def collective_name(on_host=False):
    """What gather_image() runs by default."""
    return "all-gather" + (" (gloo, host tensors)" if on_host or dist.get_backend() != "nccl" else " over xGMI")


def gather_image(local_rows: torch.Tensor, plan: StripPlan, rank: int, dst: int = 0, group=None, collective: str = "all_gather",
                 always: bool = False):
    """local_rows: [plan.max_rows, width] tensor (rows beyond this rank's share are padding).
    Returns the de-interleaved [height, width] image on `dst`, None elsewhere.

    The exchange is ONE all_gather_into_tensor: under RCCL a single ring kernel over xGMI that lands rank-major in one
    buffer (1.9 MB for 800x600, 8.3 MB for 1920x1080 -- latency-bound either way), instead of the world-1 point-to-point
    receives torch composes a rooted gather from.  The same call runs under gloo, so the CPU tests exercise exactly the
    code path of the 8-GPU run.  collective="gather" keeps the rooted form (only `dst` receives)."""
    assert local_rows.shape[0] == plan.max_rows and local_rows.shape[1] == plan.width
    if plan.world == 1 and not always:                          # (always: run the collective even for one rank -- the one-GPU check of the RCCL branch)
        return local_rows                                       # one rank owns every row, already in order
    if collective == "all_gather":
        stacked = torch.empty((plan.world * plan.max_rows, plan.width), dtype=local_rows.dtype, device=local_rows.device)
        dist.all_gather_into_tensor(stacked, local_rows.contiguous(), group=group)
        return stacked.index_select(0, plan.perm_on(local_rows.device)) if rank == dst else None
    if rank == dst:
        stacked = torch.empty((plan.world * plan.max_rows, plan.width), dtype=local_rows.dtype, device=local_rows.device)
        parts = list(stacked.chunk(plan.world, dim=0))          # contiguous views: gather lands in place, rank-major
        dist.gather(local_rows, gather_list=parts, dst=dst, group=group)
        return stacked.index_select(0, plan.perm_on(local_rows.device))
    dist.gather(local_rows, gather_list=None, dst=dst, group=group)
    return None
