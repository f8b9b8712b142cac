"""Multi-GPU sampling: independent latent volumes shard across ranks; the only collective is one weight broadcast.

The reference wraps everything in a single-process ``tf.distribute.MirroredStrategy`` (main_conditional_dm.py:87, 197)
whose only cross-device traffic is the training gradient reduction.  Sampling chains are independent (inference
BatchNorm uses stored statistics, conditional_dm3d.py:559-573), so here each GPU gets its own process and its own
slice of the batch; rank 0's weights travel once, as a single flat float32 bucket, through ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  No per-step collective exists.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Volumes [lo, hi) of a global batch that ``rank`` denoises: contiguous, sizes differ by at most one."""
    if total < 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(base_seed: int, rank: int) -> int:
    """Philox key per rank (seed 1234 + rank), so shards draw disjoint noise streams."""
    return int(base_seed) + int(rank)


def broadcast_state(state: Optional[Dict[str, np.ndarray]], spec: Dict[str, tuple], src: int = 0,
                    device: Optional[torch.device] = None) -> Dict[str, np.ndarray]:
    """One broadcast of every weight as a single flat bucket (spec order).  ``state`` is needed on ``src`` only."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        if state is None:
            raise ValueError("state is required when not running distributed")
        return state
    names = list(spec)
    sizes = [int(np.prod(spec[n])) for n in names]
    total = int(sum(sizes))
    dev = device if device is not None else torch.device("cpu")
    if dist.get_rank() == src:
        if state is None:
            raise ValueError("the source rank must hold the weights")
        flat = torch.from_numpy(np.concatenate([np.asarray(state[n], np.float32).reshape(-1) for n in names])).to(dev)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=dev)
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out, off = {}, 0
    for n, sz in zip(names, sizes):
        out[n] = host[off:off + sz].reshape(spec[n]).copy()
        off += sz
    return out


def generate_sharded(model, global_shape, last_step: int = 0, context_value=None, *, seed: int = 1234, gather: bool = True,
                     **generate_kw) -> torch.Tensor:
    """``model.generate`` for a global batch sharded over the ranks of the default process group (replaces the
    single-process MirroredStrategy scope of main_conditional_dm.py:197).  Rank r denoises volumes ``shard_range(B, r, R)`` with
    Philox key ``rank_seed(seed, r)``; there is no per-step collective.  ``context_value``: one id (broadcast) or one per volume
    of the GLOBAL batch (each rank takes its slice).  ``gather=True``: one all_gather at the end, every rank returns the whole
    ``global_shape`` tensor in rank order; otherwise the local shard ([hi-lo, ...]; may be empty)."""
    shape = tuple(int(s) for s in global_shape)
    dist_on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist_on else (0, 1)
    lo, hi = shard_range(shape[0], rank, world)
    ctx = context_value
    if ctx is not None:
        ids = np.asarray(ctx.detach().cpu() if torch.is_tensor(ctx) else ctx).reshape(-1)
        if ids.size == shape[0] and ids.size > 1:
            ctx = ids[lo:hi]
        elif ids.size != 1:
            raise ValueError(f"context_value must hold one id or one per volume of the global batch ({shape[0]}), got {ids.size}")
    dev = getattr(model, "device", torch.device("cpu"))
    if hi > lo:
        local = model.generate((hi - lo,) + shape[1:], last_step, ctx, seed=rank_seed(seed, rank), **generate_kw)
    else:
        local = torch.empty((0,) + shape[1:], dtype=torch.float32, device=dev)
    if not gather or not dist_on:
        return local
    # shards differ by at most one volume: pad to the largest, one all_gather, trim
    most = shard_range(shape[0], 0, world)[1]
    pad = torch.zeros((most,) + shape[1:], dtype=torch.float32, device=local.device)
    pad[: hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: shard_range(shape[0], r, world)[1] - shard_range(shape[0], r, world)[0]] for r in range(world)], 0)


def state_digest(state: Dict[str, np.ndarray]) -> str:
    """sha256 over every weight's bytes in name order: ranks compare it after the broadcast (bench.py reports it)."""
    import hashlib
    h = hashlib.sha256()
    for n in sorted(state):
        h.update(n.encode())
        h.update(np.ascontiguousarray(state[n], dtype=np.float32).tobytes())
    return h.hexdigest()[:16]


def gather_strings(value: str) -> list:
    """every rank's string, in rank order (single process: [value])."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, value)
    return out


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
