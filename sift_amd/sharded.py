"""Multi-GPU match: queries stay sharded, reference descriptors are all-gathered.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  The only collective on the path is the
all-gather of the byte descriptors; every rank then sees all references, so its
top-2 per query is complete and no reduction is needed.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the contiguous shard of `rank`: sizes differ by at most one."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def frames_of_rank(nframes: int, world: int, rank: int) -> range:
    """Frames are independent (every octave restarts from its own image, Sift.cpp:145-149):
    frame k goes to rank k mod world; no collective on the extract path."""
    return range(rank, nframes, world)


def gather_feature_counts(counts_local: torch.Tensor, nframes: int, group=None) -> torch.Tensor:
    """Host-side bookkeeping after a sharded batch: every rank's per-frame feature counts (int32,
    in the order of frames_of_rank) -> the counts of all nframes frames in frame order, on every
    rank.  The feature lists themselves stay where they were produced."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    out = torch.zeros(nframes, dtype=torch.int32)
    if world == 1:
        out[:] = counts_local.cpu()
        return out
    per = (nframes + world - 1) // world
    padded = torch.full((per,), -1, dtype=torch.int32, device=counts_local.device)
    padded[: counts_local.numel()] = counts_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    for r in range(world):
        idx = list(frames_of_rank(nframes, world, r))
        out[idx] = parts[r][: len(idx)].cpu()
    assert rank >= 0
    return out


def all_gather_refs(r_local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenates every rank's (n_i x 128) uint8 references in rank order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return r_local
    n_local = torch.tensor([r_local.shape[0]], dtype=torch.int64, device=r_local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if all(c == nmax for c in counts):
        out = torch.empty((world * nmax, 128), dtype=torch.uint8, device=r_local.device)
        dist.all_gather_into_tensor(out, r_local.contiguous(), group=group)
        return out
    padded = torch.zeros((nmax, 128), dtype=torch.uint8, device=r_local.device)
    padded[: r_local.shape[0]] = r_local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def sharded_match(match_fn: Callable, q_local: torch.Tensor, r_local: torch.Tensor, ratio: float, ratio_form: int = 0,
                  group=None):
    """match_fn(q, r_all, ratio, ratio_form) -> (best, d1, d2) for this rank's queries
    against all references; `best` indexes the rank-ordered concatenation."""
    r_all = all_gather_refs(r_local, group)
    return match_fn(q_local, r_all, ratio, ratio_form)


def hip_match_fn(ctx):
    """match_fn over a SiftHip context for CUDA tensors."""

    def fn(q, r_all, ratio, ratio_form):
        nq, nr = q.shape[0], r_all.shape[0]
        best = torch.empty(nq, dtype=torch.int32, device=q.device)
        d1 = torch.empty(nq, dtype=torch.int32, device=q.device)
        d2 = torch.empty(nq, dtype=torch.int32, device=q.device)
        torch.cuda.current_stream(q.device).synchronize()  # the all-gather ran on torch's stream
        ctx.match_u8_dev(q.contiguous(), nq, r_all.contiguous(), nr, ratio, ratio_form, best, d1, d2)
        ctx.synchronize()
        return best, d1, d2

    return fn
