"""Multi-GPU sharding of the VO hot path: one process per GPU, independent sequences (or windows)
sharded across ranks, one all-gather of the resulting pose blocks per batch step.

The reference has no multi-GPU path at all (file-static device buffers, default stream).  Inside
one window the cameras are sequentially dependent and EM iterations are sequential, so the only
parallel axis is across independent sequences: no data-path collective, just the exchange of the
results.  A pose block is [n_registered | poses N x 6 | covar N x 36] = 1 + 42 N floats (211 floats
= 844 B for N = 5): latency-bound, one RCCL all-gather (backend "nccl" on ROCm) over xGMI; the same
code runs over gloo on CPU tensors in the tests.
"""
from __future__ import annotations

import numpy as np


def block_len(n_flows: int) -> int:
    return 1 + 6 * n_flows + 36 * n_flows


def shard(n_items: int, rank: int, world: int):
    """Contiguous, balanced shard of range(n_items) for `rank` (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def pack_pose_block(out: dict, n_flows: int) -> np.ndarray:
    """pyvoldor.voldor() result -> flat float32 block of block_len(n_flows)."""
    blk = np.zeros(block_len(n_flows), np.float32)
    n = int(out["n_registered"])
    blk[0] = n
    blk[1:1 + 6 * n] = np.asarray(out["poses"], np.float32).reshape(-1)[:6 * n]
    blk[1 + 6 * n_flows:1 + 6 * n_flows + 36 * n] = np.asarray(out["poses_covar"], np.float32).reshape(-1)[:36 * n]
    return blk


def unpack_pose_block(blk, n_flows: int) -> dict:
    blk = np.asarray(blk, np.float32)
    n = int(round(float(blk[0])))
    poses = blk[1:1 + 6 * n_flows].reshape(n_flows, 6)[:n].copy()
    covar = blk[1 + 6 * n_flows:].reshape(n_flows, 6, 6)[:n].copy()
    return {"n_registered": n, "poses": poses, "poses_covar": covar}


def allgather_pose_blocks(blk, group=None, device=None):
    """All-gather one pose block per rank. Returns a [world, block_len] float32 numpy array on every rank.
    `blk` is a numpy array (`device` then selects where the collective runs: "cuda" for RCCL, None/"cpu" for gloo) or a torch
    tensor that already lives there (no host hop: vk_voldor_device_block packs the record on the device)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if torch.is_tensor(blk):  # already where the collective runs, e.g. the device record of pyvoldor.voldor_device(pose_block_out=...)
        send = blk.reshape(-1)
    else:
        send = torch.from_numpy(np.ascontiguousarray(blk, np.float32))
        if device is not None and str(device) != "cpu":
            send = send.to(device, non_blocking=True)
    recv = torch.empty(world * send.numel(), dtype=torch.float32, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.view(world, -1).cpu().numpy()


def run_sharded(sequences, run_window, n_flows: int, group=None, device=None):
    """Process `sequences` (a list; item i is passed to run_window) sharded over the ranks, then exchange
    the pose blocks so that every rank holds the result of every sequence, in sequence order.
    `run_window(item) -> dict` is pyvoldor.voldor-like. Ranks with fewer items pad with empty blocks."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = list(shard(len(sequences), rank, world))
    steps = -(-len(sequences) // world)
    results = [None] * len(sequences)
    for s in range(steps):
        blk = np.zeros(block_len(n_flows), np.float32)
        blk[0] = -1.0  # marks "no sequence in this slot"
        if s < len(mine):
            blk = pack_pose_block(run_window(sequences[mine[s]]), n_flows)
        allb = allgather_pose_blocks(blk, group, device)
        for r in range(world):
            sh = list(shard(len(sequences), r, world))
            if s < len(sh) and allb[r, 0] >= 0:
                results[sh[s]] = unpack_pose_block(allb[r], n_flows)
    return results
