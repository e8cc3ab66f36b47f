"""The N>1 path on CPU: two processes, gloo, sharded sequences + pose all-gather."""
import os
import socket

import numpy as np
import pytest


def _fake_window(seed, n_flows=5):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(0, n_flows + 1))
    return {"n_registered": n, "poses": rng.normal(size=(n, 6)).astype(np.float32),
            "poses_covar": rng.normal(size=(n, 6, 6)).astype(np.float32)}


def _fake_window_with_record(seed, n_flows=5):
    """what pyvoldor.voldor_device(pose_block_out=...) hands over: the record already packed where the collective runs (here: a CPU
    tensor for gloo; on the GPU the library packs it on the device) -- run_sharded must send it as it is"""
    import torch
    from voldor_amd import dist as vd
    w = _fake_window(seed, n_flows)
    return {"n_registered": w["n_registered"], "pose_block": torch.from_numpy(vd.pack_pose_block(w, n_flows))}  # no host arrays: packing them again is impossible


def _worker(rank, world, port, n_seq, q, record=False):
    import torch.distributed as dist
    from voldor_amd import dist as vd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = vd.run_sharded(list(range(100, 100 + n_seq)), (lambda s: _fake_window_with_record(s)) if record else (lambda s: _fake_window(s)), 5)
        ok = True
        for i, r in enumerate(res):
            ref = _fake_window(100 + i)
            ok &= r is not None and r["n_registered"] == ref["n_registered"]
            ok &= np.array_equal(r["poses"], ref["poses"]) and np.array_equal(r["poses_covar"], ref["poses_covar"])
        q.put((rank, bool(ok), len(res)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("n_seq,record", [(2, False), (5, False), (5, True), (3, True)])
def test_two_rank_pose_allgather(n_seq, record):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_seq, q, record)) for r in range(2)]
    for p in procs: p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in out) and all(n == n_seq for _, _, n in out), out


def test_block_roundtrip_and_sharding():
    from voldor_amd import dist as vd
    w = _fake_window(3)
    b = vd.pack_pose_block(w, 5)
    assert b.shape == (vd.block_len(5),) == (211,)
    u = vd.unpack_pose_block(b, 5)
    assert u["n_registered"] == w["n_registered"] and np.array_equal(u["poses"], w["poses"])
    for n, world in ((8, 8), (5, 2), (3, 4), (0, 2)):
        parts = [list(vd.shard(n, r, world)) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(n)) and max(map(len, parts)) - min(map(len, parts)) <= 1


def test_capi_run_sharded_bookkeeping():
    """capi_run_sharded (the vk_voldor_sharded front end) places every rank's record at its sequence: checked with a stand-in for the
    library call that plays both ranks (the RCCL path itself runs in tests/test_gpu_dist_nccl.py)."""
    from voldor_amd import dist as vd

    class FakeLib:
        def __init__(self, rank, world): self.r, self.w = rank, world
        def vk_dist_rank(self): return self.r
        def vk_dist_world(self): return self.w

    import voldor_amd.capi as capi
    seqs = list(range(200, 205))
    world = 2
    real = capi.lib
    try:
        for rank in range(world):
            capi.lib = lambda rank=rank: FakeLib(rank, world)
            step = [0]
            def run_step(item, rank=rank):
                s = step[0]; step[0] += 1
                blocks = np.zeros((world, vd.block_len(5)), np.float32)
                for r in range(world):
                    sh = list(vd.shard(len(seqs), r, world))
                    blocks[r] = vd.pack_pose_block(_fake_window(seqs[sh[s]]), 5) if s < len(sh) else np.r_[-1.0, np.zeros(210)].astype(np.float32)
                assert (item is None) == (s >= len(list(vd.shard(len(seqs), rank, world))))
                return blocks
            res = vd.capi_run_sharded(seqs, run_step, 5)
            for i, r in enumerate(res):
                ref = _fake_window(seqs[i])
                assert r["n_registered"] == ref["n_registered"] and np.array_equal(r["poses"], ref["poses"]) and np.array_equal(r["poses_covar"], ref["poses_covar"])
    finally:
        capi.lib = real
