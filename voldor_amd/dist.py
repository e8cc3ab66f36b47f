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

VK_DIST_FAILED = -2  # include/voldor_hip.h: n_registered slot of a rank whose window failed (slot 1 = its error code); -1 = no sequence in the slot


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
    `run_window(item) -> dict` is pyvoldor.voldor-like; if the dict carries "pose_block" (the device record that
    pyvoldor.voldor_device(pose_block_out=...) / vk_voldor_device_block packed: a torch tensor of block_len(n_flows) floats
    where the collective runs) that record IS the send buffer -- no host packing; otherwise the block is packed from the host
    arrays.  Ranks with fewer items pad with empty blocks.  Front end over torch.distributed (RCCL: backend "nccl"; gloo in the
    CPU tests); the same step below the C-ABI is vk_voldor_sharded (capi_* functions of this module)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = list(shard(len(sequences), rank, world))
    steps = -(-len(sequences) // world)
    results = [None] * len(sequences)
    failed = []  # (sequence, rank, error code) of windows that failed on their rank: raised after the last step, when every rank has left every collective
    # where the collective runs: the caller's `device`, else what the backend needs (nccl = RCCL: this process's current device), else
    # wherever the previous record of this rank lived -- an idle rank must not feed a CPU tensor to a collective whose peers send
    # device tensors (it would error or hang on the last, uneven step)
    coll_dev = device
    if coll_dev is None and dist.get_backend(group) == "nccl":
        coll_dev = torch.device("cuda", torch.cuda.current_device())
    for s in range(steps):
        blk = None
        if s < len(mine):
            out = run_window(sequences[mine[s]])
            blk = out.get("pose_block")
            if blk is None:
                blk = pack_pose_block(out, n_flows)
            elif torch.is_tensor(blk) and coll_dev is None:
                coll_dev = blk.device
        if blk is None:
            blk = np.zeros(block_len(n_flows), np.float32)
            blk[0] = -1.0  # marks "no sequence in this slot"
            if coll_dev is not None and str(coll_dev) != "cpu":
                blk = torch.from_numpy(blk).to(coll_dev)
        allb = allgather_pose_blocks(blk, group, coll_dev)
        for r in range(world):
            sh = list(shard(len(sequences), r, world))
            if s < len(sh) and allb[r, 0] >= 0:
                results[sh[s]] = unpack_pose_block(allb[r], n_flows)
            elif s < len(sh) and int(allb[r, 0]) == VK_DIST_FAILED:  # that rank's window FAILED (vk_voldor_sharded's marker: error code in slot 1): not an empty slot
                failed.append((sh[s], r, int(allb[r, 1])))
    if failed:
        raise RuntimeError("sharded VO: " + "; ".join(f"sequence {q} failed on rank {r} (error {code})" for q, r, code in failed))
    return results


# ---- the same exchange below the C-ABI: RCCL driven by libvoldor_hip.so itself (include/voldor_hip.h section D) ----------------
def capi_init(rank: int, world: int, store=None, path: str | None = None, key: str = "voldor_hip_rccl_id"):
    """vk_dist_init on the current device.  The 128-byte ncclUniqueId of rank 0 travels through `store` (anything with
    set(key, bytes) / get(key) -> bytes: torch.distributed.TCPStore / FileStore -- rendezvous plumbing only) or, with `path`,
    through a file every rank can see (vk_dist_init_file: no Python in the exchange at all)."""
    import ctypes as C
    from . import capi
    lib = capi.lib()
    if path is not None:
        capi.check(lib.vk_dist_init_file(int(rank), int(world), str(path).encode(), 120), "vk_dist_init_file")
        return
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        capi.check(lib.vk_dist_get_unique_id(buf), "vk_dist_get_unique_id")
        if world > 1:
            store.set(key, bytes(buf))
    else:
        raw = bytes(store.get(key))
        assert len(raw) == 128, len(raw)
        buf = (C.c_ubyte * 128).from_buffer_copy(raw)
    capi.check(lib.vk_dist_init(int(rank), int(world), buf), "vk_dist_init")


def capi_finalize():
    from . import capi
    capi.lib().vk_dist_finalize()


def capi_barrier():
    from . import capi
    capi.check(capi.lib().vk_dist_barrier(), "vk_dist_barrier")


def capi_max(value: float) -> float:
    """max over the ranks of a host double (vk_dist_allreduce_max); also a barrier"""
    import ctypes as C
    from . import capi
    v = C.c_double(float(value))
    capi.check(capi.lib().vk_dist_allreduce_max(C.byref(v)), "vk_dist_allreduce_max")
    return float(v.value)


def capi_allgather_stats(reset: bool = False):
    """(dev_us, host_us): latency of every all-gather the library has issued since the last reset (vk_dist_allgather_stats) -- HIP events
    around ncclAllGather on the communicator's stream, and the host's wall clock from issue to completion -- as numpy arrays."""
    import ctypes as C
    from . import capi
    cap = 65536
    dev, host = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    n = C.c_int(0)
    capi.check(capi.lib().vk_dist_allgather_stats(capi.fp(dev), capi.fp(host), cap, C.byref(n), int(bool(reset))), "vk_dist_allgather_stats")
    return dev[:n.value].copy(), host[:n.value].copy()


def capi_allgather(send, recv):
    """ncclAllGather of send.numel() floats per rank between two torch float32 device tensors (vk_dist_allgather)."""
    import ctypes as C
    import torch
    from . import capi
    assert send.is_cuda and recv.is_cuda and send.dtype == recv.dtype == torch.float32 and send.is_contiguous() and recv.is_contiguous()
    torch.cuda.current_stream().synchronize()  # the library's communicator has its own stream
    PF = C.POINTER(C.c_float)
    capi.check(capi.lib().vk_dist_allgather(C.cast(send.data_ptr(), PF), C.cast(recv.data_ptr(), PF), C.c_int(send.numel())), "vk_dist_allgather")


def capi_run_sharded(sequences, run_step, n_flows: int):
    """run_sharded over the C-ABI: `run_step(item_or_None) -> [world, block_len] array` is one vk_voldor_sharded call
    (pyvoldor.voldor_sharded) -- the window of this rank's item (None: no item in this step) and the all-gather, both inside the
    library.  Returns the per-sequence results on every rank, in sequence order."""
    from . import capi
    lib = capi.lib()
    rank, world = lib.vk_dist_rank(), lib.vk_dist_world()
    assert world >= 1, "capi_init first"
    mine = list(shard(len(sequences), rank, world))
    steps = -(-len(sequences) // world)
    results = [None] * len(sequences)
    failed = []  # (sequence, rank, error code) of windows that failed on their rank: raised after the last step, when every rank has left every collective
    for s in range(steps):
        allb = run_step(sequences[mine[s]] if s < len(mine) else None)
        for r in range(world):
            sh = list(shard(len(sequences), r, world))
            if s < len(sh) and allb[r, 0] >= 0:
                results[sh[s]] = unpack_pose_block(allb[r], n_flows)
            elif s < len(sh) and int(allb[r, 0]) == VK_DIST_FAILED:  # that rank's window FAILED (vk_voldor_sharded's marker: error code in slot 1): not an empty slot
                failed.append((sh[s], r, int(allb[r, 1])))
    if failed:
        raise RuntimeError("sharded VO: " + "; ".join(f"sequence {q} failed on rank {r} (error {code})" for q, r, code in failed))
    return results
