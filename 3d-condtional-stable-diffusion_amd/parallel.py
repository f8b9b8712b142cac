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


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
