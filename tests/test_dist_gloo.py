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


def _worker(rank, world, port, n_seq, q):
    import torch.distributed as dist
    from voldor_amd import dist as vd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = vd.run_sharded(list(range(100, 100 + n_seq)), lambda s: _fake_window(s), 5)
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


@pytest.mark.parametrize("n_seq", [2, 5])
def test_two_rank_pose_allgather(n_seq):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_seq, q)) for r in range(2)]
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
