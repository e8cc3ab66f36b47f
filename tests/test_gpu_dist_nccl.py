"""The RCCL branch of the multi-GPU path (bench.py: init_process_group("nccl"), the device pose block of vk_voldor_device_block fed
straight into all_gather_into_tensor, barrier + max-over-ranks timing) must have run before the driver's 8-GPU SCALE run does:
with ONE rank on any GPU box (a one-rank RCCL communicator still goes through every call), and with two ranks where the box has two
GPUs.  The gloo twin of the exchange runs on CPU in tests/test_dist_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("frontend", ["capi", "torch"])
def test_nccl_branch_with_one_rank(frontend):
    """capi: the communicator and the ncclAllGather live inside libvoldor_hip.so (vk_dist.hip; bench.py's default for N > 1);
    torch: the same records through torch.distributed."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541" if frontend == "capi" else "29543")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras", "--force-dist", "--dist-frontend", frontend],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["value"] > 10 and j["n_registered"] == 5
    assert ("below the C-ABI" in j["config"]["parallelism"]) == (frontend == "capi"), j["config"]["parallelism"]


def test_capi_front_end_under_torchrun_with_one_rank():
    """The launch the driver uses for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), here with one rank: the
    ncclUniqueId store comes from torchrun's own rendezvous (the elastic agent owns MASTER_PORT: a second TCPStore server there
    would fail), the ranks agree on the front end through it, the library's communicator does the exchange."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29545",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-extras", "--force-dist"],
                       capture_output=True, text=True, cwd=ROOT, env=dict(os.environ), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["n_registered"] == 5 and "below the C-ABI" in j["config"]["parallelism"] and "fell back" not in j["config"]["parallelism"], j["config"]


@pytest.mark.parametrize("frontend", ["capi", "torch"])
def test_nccl_two_ranks_all_gather(frontend):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on the box")
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        "29542" if frontend == "capi" else "29544", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras",
                        "--dist-frontend", frontend], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 20


_STANDALONE = r"""
import ctypes as C, os, sys, tempfile
import numpy as np
sys.path.insert(0, {root!r})
from voldor_amd import capi, synth
lib = capi.lib()                      # no torch in this process: the library binds the system RCCL against the system HIP runtime
capi.check(lib.vk_set_device(0), "vk_set_device")
assert lib.vk_dist_world() == 0 and lib.vk_dist_rank() == -1
path = os.path.join(tempfile.mkdtemp(), "rccl_id")
capi.check(lib.vk_dist_init_file(0, 1, path.encode(), 30), "vk_dist_init_file")   # rendezvous through a file, world of one
assert lib.vk_dist_world() == 1 and lib.vk_dist_rank() == 0 and lib.vk_dist_rccl_version() > 0
sc = synth.make_scene(w=160, h=120, n_flows=3, fx=80, fy=80, cx=80, cy=60, seed=233)
N, h, w = sc["flows"].shape[:3]
cfg = b"--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2"
def run(fn, *tail):
    poses = np.zeros((N, 6), np.float32); covar = np.zeros((N, 36), np.float32); n = C.c_int(0)
    lib.vk_set_rand_epoch(0)
    capi.check(fn(capi.fp(sc["flows"]), None, None, None, None, None, C.c_float(80), C.c_float(80), C.c_float(80), C.c_float(60), C.c_float(0), N, 0, w, h, cfg,
                  C.byref(n), capi.fp(poses), capi.fp(covar), None, None, *tail), "window")
    return n.value, poses, covar
blocks = np.zeros((1, 1 + 42 * N), np.float32)
n1, p1, c1 = run(lib.vk_voldor_sharded, capi.fp(blocks))
n0, p0, c0 = run(lib.vk_voldor_device)
assert n0 == n1 == 3 and np.array_equal(p0, p1) and np.array_equal(c0, c1)
assert blocks[0, 0] == n1 and np.array_equal(blocks[0, 1:1 + 6 * N], p1.reshape(-1)) and np.array_equal(blocks[0, 1 + 6 * N:], c1.reshape(-1))
# a step in which this rank has no sequence: the record says so
capi.check(lib.vk_voldor_sharded(None, None, None, None, None, None, C.c_float(80), C.c_float(80), C.c_float(80), C.c_float(60), C.c_float(0), N, 0, w, h, cfg,
                                 None, None, None, None, None, capi.fp(blocks)), "empty step")
assert blocks[0, 0] == -1 and not blocks[0, 1:].any()
v = C.c_double(3.5); capi.check(lib.vk_dist_allreduce_max(C.byref(v)), "max"); assert v.value == 3.5
capi.check(lib.vk_dist_barrier(), "barrier")
lib.vk_dist_finalize()
assert lib.vk_dist_world() == 0
print("STANDALONE_OK")
"""


def test_capi_dist_without_torch_in_the_process():
    """The C-ABI exchange needs no Python framework: a process that never imports torch initialises the communicator through the
    file rendezvous (vk_dist_init_file), runs one sharded step (vk_voldor_sharded) and gets, in the gathered record, exactly what
    vk_voldor_device returns for the same window; an empty step is marked -1."""
    r = subprocess.run([sys.executable, "-c", _STANDALONE.format(root=ROOT)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "STANDALONE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


_TWO_RANK = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r})
rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
from voldor_amd import capi, synth
from voldor_amd import dist as vd
lib = capi.lib()
capi.check(lib.vk_set_device(0), "vk_set_device")            # both ranks on GPU 0: the stand-in stages through the host
capi.check(lib.vk_dist_init_file(rank, world, path.encode(), 60), "vk_dist_init_file")
assert lib.vk_dist_world() == world and lib.vk_dist_rank() == rank and lib.vk_dist_rccl_version() == 999
N, h, w = 3, 120, 160
cfg = b"--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2"
seeds = [233, 234, 235, 236, 237]                              # five sequences over two ranks: rank 0 gets three, rank 1 two + an empty step
scenes = {{s: synth.make_scene(w=w, h=h, n_flows=N, fx=80, fy=80, cx=80, cy=60, seed=s) for s in seeds}}
def window(fn, seed, *tail):
    poses = np.zeros((N, 6), np.float32); covar = np.zeros((N, 36), np.float32); n = C.c_int(0)
    lib.vk_set_rand_epoch(0)
    fl = scenes[seed]["flows"] if seed is not None else None
    capi.check(fn(capi.fp(fl), None, None, None, None, None, C.c_float(80), C.c_float(80), C.c_float(80), C.c_float(60), C.c_float(0), N, 0, w, h, cfg,
                  C.byref(n), capi.fp(poses), capi.fp(covar), None, None, *tail), "window")
    return n.value, poses, covar
def step(seed):
    blocks = np.zeros((world, 1 + 42 * N), np.float32)
    window(lib.vk_voldor_sharded, seed, capi.fp(blocks))
    return blocks
res = vd.capi_run_sharded(seeds, step, N)
assert len(res) == len(seeds) and all(r is not None for r in res)
for s, r in zip(seeds, res):                                    # every rank holds every sequence's result = the one-at-a-time window
    n0, p0, c0 = window(lib.vk_voldor_device, s)
    assert r["n_registered"] == n0 == N and np.array_equal(r["poses"], p0[:n0]) and np.array_equal(r["poses_covar"].reshape(n0, 36), c0[:n0]), s
v = C.c_double(10.0 + rank); capi.check(lib.vk_dist_allreduce_max(C.byref(v)), "max"); assert v.value == 10.0 + world - 1
capi.check(lib.vk_dist_barrier(), "barrier")
lib.vk_dist_finalize()
print("RANK_OK", rank)
"""


def test_capi_two_ranks_on_one_gpu_through_the_file_backed_stand_in(tmp_path):
    """The N > 1 paths of vk_dist.hip on a one-GPU box: two processes share GPU 0 and bind tests/cxx/fake_rccl.cpp (a file-backed stand-in
    for the seven RCCL entry points, loaded through VOLDOR_HIP_RCCL; RCCL itself refuses two ranks on one device).  Five sequences over
    two ranks through vk_dist_init_file + vk_voldor_sharded (uneven shards: one rank sends an empty record in the last step): every rank
    ends up with every sequence's poses and covariances, equal to the one-at-a-time window; max-over-ranks and barrier work.  What this
    cannot show is RCCL with N > 1 -- that is the driver's 8-GPU run."""
    from voldor_amd import build
    build.build_test_lib()
    fake = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    assert os.path.exists(fake)
    env = dict(os.environ, VOLDOR_HIP_RCCL=fake)
    path = str(tmp_path / "id")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANK.format(root=ROOT), str(r), "2", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in so, (r, so[-1000:], se[-3000:])


_EIGHT_RANK = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r})
rank, world, path, fail_rank = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
from voldor_amd import capi, synth
from voldor_amd import dist as vd
lib = capi.lib()
capi.check(lib.vk_set_device(0), "vk_set_device")            # all ranks on GPU 0: the stand-in stages through the host
capi.check(lib.vk_dist_init_file(rank, world, path.encode(), 120), "vk_dist_init_file")
assert lib.vk_dist_world() == world and lib.vk_dist_rank() == rank
N, h, w = 3, 96, 128
good = b"--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2"
bad = good + b" --resize_factor 0.5"                          # rejected by the window call: a LOCAL error
seeds = list(range(233, 233 + 11))                             # eleven sequences over eight ranks: three ranks run two windows, five run one and send an empty record
scenes = {{s: synth.make_scene(w=w, h=h, n_flows=N, fx=64, fy=64, cx=64, cy=48, seed=s) for s in seeds}}
rcs = []
def window(fn, seed, cfg, *tail):
    poses = np.zeros((N, 6), np.float32); covar = np.zeros((N, 36), np.float32); n = C.c_int(0)
    lib.vk_set_rand_epoch(0)
    fl = scenes[seed]["flows"] if seed is not None else None
    rc = fn(capi.fp(fl), None, None, None, None, None, C.c_float(64), C.c_float(64), C.c_float(64), C.c_float(48), C.c_float(0), N, 0, w, h, cfg,
            C.byref(n), capi.fp(poses), capi.fp(covar), None, None, *tail)
    return rc, n.value, poses, covar
def step(seed):
    blocks = np.zeros((world, 1 + 42 * N), np.float32)
    rc, _, _, _ = window(lib.vk_voldor_sharded, seed, bad if (rank == fail_rank and seed is not None) else good, capi.fp(blocks))
    rcs.append(rc)
    return blocks
try:
    res = vd.capi_run_sharded(seeds, step, N)
    assert fail_rank < 0, "a failed window must surface on every rank"
    assert len(res) == len(seeds) and all(r is not None for r in res)
    for s, r in zip(seeds[rank::world] + seeds[:2], [res[seeds.index(q)] for q in seeds[rank::world] + seeds[:2]]):   # own sequences + two of rank 0 / 1: equal to the one-at-a-time window
        rc, n0, p0, c0 = window(lib.vk_voldor_device, s, good)
        assert rc == 0 and r["n_registered"] == n0 == N and np.array_equal(r["poses"], p0[:n0]) and np.array_equal(r["poses_covar"].reshape(n0, 36), c0[:n0]), s
    assert all(rc == 0 for rc in rcs)
except RuntimeError as e:                                        # dist.py: a peer's VK_DIST_FAILED record is an error on EVERY rank, raised after the last step
    assert fail_rank >= 0 and f"failed on rank {{fail_rank}}" in str(e), str(e)
    assert all((rc != 0) == (rank == fail_rank) for rc in rcs[:1]), rcs   # the failing rank's own call returned its error, the others 0; nobody hung
v = C.c_double(10.0 + rank); capi.check(lib.vk_dist_allreduce_max(C.byref(v)), "max"); assert v.value == 10.0 + world - 1
capi.check(lib.vk_dist_barrier(), "barrier")
lib.vk_dist_finalize()
print("RANK_OK", rank)
"""


@pytest.mark.parametrize("fail_rank", [-1, 5])
def test_capi_eight_ranks_on_one_gpu_through_the_stand_in(tmp_path, fail_rank):
    """the world size the driver's scaling run uses.  Eight processes share GPU 0 through the file-backed stand-in: eleven
    sequences over eight ranks (uneven shards, empty records in the second step), file rendezvous, max-over-ranks, barrier; and the same job
    with rank 5's windows failing locally -- it still joins every all-gather, every rank learns which sequence failed where, nobody hangs.
    (RCCL itself with N > 1 has still not run this code: that is the driver's 8-GPU run.)"""
    from voldor_amd import build
    build.build_test_lib()
    fake = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    env = dict(os.environ, VOLDOR_HIP_RCCL=fake, VOLDOR_HIP_JOB_ID=f"eight-{fail_rank}")
    path = str(tmp_path / "id")
    procs = [subprocess.Popen([sys.executable, "-c", _EIGHT_RANK.format(root=ROOT), str(r), "8", path, str(fail_rank)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              cwd=ROOT, env=env) for r in range(8)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in so, (r, so[-1000:], se[-3000:])


def test_file_rendezvous_times_out_instead_of_waiting_forever(tmp_path):
    """A rank whose rank 0 never shows up gets an error from vk_dist_init_file after the timeout it asked for (a launcher can then fail the job)."""
    code = r"""
import sys, time
sys.path.insert(0, {root!r})
from voldor_amd import capi
lib = capi.lib()
capi.check(lib.vk_set_device(0), "vk_set_device")
t0 = time.time()
rc = lib.vk_dist_init_file(1, 2, sys.argv[1].encode(), 3)
assert rc != 0 and 2.0 < time.time() - t0 < 30.0, (rc, time.time() - t0)
assert lib.vk_dist_world() == 0
print("TIMEOUT_OK")
"""
    r = subprocess.run([sys.executable, "-c", code.format(root=ROOT), str(tmp_path / "never")], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "TIMEOUT_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_bench_with_eight_ranks_on_one_gpu_through_the_stand_in():
    """bench.py's whole N = 8 flow as the driver launches it (`--gpus 8`, RANK 0..7, env:// rendezvous), all ranks on GPU 0 through the stand-in:
    the line carries n_gpus 8, the exchange object with 8-rank all-gathers, the per-GPU rate; ranks 1..7 print nothing."""
    from voldor_amd import build
    build.build_test_lib()
    fake = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    procs = []
    for r in range(8):
        env = dict(os.environ, VOLDOR_HIP_RCCL=fake, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--in-flight", "0", "--no-workloads"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=1500) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, so[-500:], se[-3000:])
    j = _line(outs[0][0])
    assert j["n_gpus"] == 8 and j["n_registered"] == 5 and j["value"] > 5 and "x8" in j["config"]["parallelism"] and "fell back" not in j["config"]["parallelism"]
    ex = j["exchange"]
    assert ex["samples"] == 2 and "8 ranks" in ex["collective"] and abs(ex["per_gpu_frames_per_s"] * 8 - j["value"]) < 0.01 * j["value"]
    for r in range(1, 8):
        assert not [ln for ln in outs[r][0].splitlines() if ln.startswith("{")]


def test_bench_under_torchrun_with_eight_ranks_on_one_gpu():
    """The driver's own launch line -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8 -- with the eight ranks pinned to
    GPU 0 (VOLDOR_HIP_FORCE_DEVICE) and the stand-in for RCCL: torchrun's agent store hands rank 0's id to the others (no second bind of MASTER_PORT), every
    rank agrees on the C-ABI front end, the timed steps run, rank 0 prints the line."""
    from voldor_amd import build
    build.build_test_lib()
    fake = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    env = dict(os.environ, VOLDOR_HIP_RCCL=fake, VOLDOR_HIP_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29551",
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--in-flight", "0", "--no-workloads"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=1800)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert j["n_gpus"] == 8 and j["n_registered"] == 5 and "x8" in j["config"]["parallelism"] and "fell back" not in j["config"]["parallelism"], j["config"]
    assert j["exchange"]["samples"] == 2


_FAILING_RANK = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r})
rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
from voldor_amd import capi, synth
lib = capi.lib()
capi.check(lib.vk_set_device(0), "vk_set_device")
if rank == 1:                                                   # a leftover of ANOTHER job under the same path: rank 1 must not join it
    assert os.path.exists(path)
capi.check(lib.vk_dist_init_file(rank, world, path.encode(), 60), "vk_dist_init_file")
N, h, w = 3, 120, 160
sc = synth.make_scene(w=w, h=h, n_flows=N, fx=80, fy=80, cx=80, cy=60, seed=233)
good = b"--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2"
bad = good + b" --resize_factor 0.5"                            # rejected by the window call (vk_voldor.hip): a LOCAL error on rank 1 only
blocks = np.zeros((world, 1 + 42 * N), np.float32)
poses = np.zeros((N, 6), np.float32); covar = np.zeros((N, 36), np.float32); n = C.c_int(0)
rc = lib.vk_voldor_sharded(capi.fp(sc["flows"]), None, None, None, None, None, C.c_float(80), C.c_float(80), C.c_float(80), C.c_float(60), C.c_float(0), N, 0, w, h,
                           bad if rank == 1 else good, C.byref(n), capi.fp(poses), capi.fp(covar), None, None, capi.fp(blocks))
# nobody hangs: the failing rank took part in the all-gather with a marked record and returns its own error; the others return 0
if rank == 1:
    assert rc != 0 and n.value == -2, (rc, n.value)
else:
    assert rc == 0 and n.value == N, (rc, n.value)
assert blocks[0, 0] == N and blocks[1, 0] == -2 and blocks[1, 1] != 0 and not blocks[1, 2:].any(), blocks[:, :3]
capi.check(lib.vk_dist_barrier(), "barrier")                     # the communicator is still usable
dev = np.zeros(8, np.float32); host = np.zeros(8, np.float32); k = C.c_int(0)
capi.check(lib.vk_dist_allgather_stats(capi.fp(dev), capi.fp(host), 8, C.byref(k), 1), "stats")
assert k.value == 1 and dev[0] >= 0 and host[0] > 0, (k.value, dev[:2], host[:2])
lib.vk_dist_finalize()
print("RANK_OK", rank)
"""


def test_a_rank_whose_window_fails_still_joins_the_all_gather(tmp_path):
    """one rank's LOCAL error (here a rejected config key) must not hang the job -- the other ranks are already inside
    ncclAllGather + hipStreamSynchronize with no timeout.  vk_voldor_sharded on the failing rank sends { VK_DIST_FAILED, its error code }
    and returns the error; rank 0 returns 0 and sees which peer failed.  Also: the rendezvous file of ANOTHER job (other job tag) left under
    the same path is ignored by the waiting rank (VOLDOR_HIP_JOB_ID), and vk_dist_allgather_stats reports the collective's latency."""
    from voldor_amd import build
    build.build_test_lib()
    fake = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    path = str(tmp_path / "id")
    with open(path, "wb") as f:  # a stale file of "another job": 64-byte tag + 128 bytes of a dead id
        f.write(b"some-other-job".ljust(64, b"\0") + bytes(128))
    env = dict(os.environ, VOLDOR_HIP_RCCL=fake, VOLDOR_HIP_JOB_ID="this-job")
    p1 = subprocess.Popen([sys.executable, "-c", _FAILING_RANK.format(root=ROOT), "1", "2", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    import time
    time.sleep(3.0)  # rank 1 is now polling the stale file; rank 0 starts late and replaces it
    p0 = subprocess.Popen([sys.executable, "-c", _FAILING_RANK.format(root=ROOT), "0", "2", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    for r, p in ((0, p0), (1, p1)):
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0 and f"RANK_OK {r}" in so, (r, so[-1000:], se[-3000:])


def test_bench_with_two_ranks_on_one_gpu_through_the_stand_in():
    """bench.py's whole N = 2 flow (what the driver launches for its scaling run, minus the second GPU): two processes with RANK 0 / 1,
    both on GPU 0 (LOCAL_RANK 0), the C-ABI front end bound to the file-backed stand-in.  Rendezvous through env://, agreement on the front
    end, timed steps with the exchange inside vk_voldor_sharded, max-over-ranks time, and the rank-0-only measurement legs AFTER the
    timed region, which must not contain a collective (a rank-0-only all-gather would wait for a rank that has already left)."""
    from voldor_amd import build
    build.build_test_lib()
    fake = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    procs = []
    for r in range(2):
        env = dict(os.environ, VOLDOR_HIP_RCCL=fake, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--in-flight", "0"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, so[-500:], se[-3000:])
    j = _line(outs[0][0])
    assert j["n_gpus"] == 2 and j["n_registered"] == 5 and j["value"] > 20 and "below the C-ABI" in j["config"]["parallelism"] and "fell back" not in j["config"]["parallelism"]
    assert j["latency"] is not None and j["roofline"] is not None  # the rank-0-only legs ran (without a collective)
    ex = j["exchange"]  # the collective's own latency and the per-GPU rate are in the line
    assert ex["samples"] == 3 and ex["allgather_us"]["p50"] >= 0 and ex["allgather_host_us"]["p99"] > 0 and abs(ex["per_gpu_frames_per_s"] * 2 - j["value"]) < 0.01 * j["value"]
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]  # rank 1 prints nothing


@pytest.mark.parametrize("world,rccl", [(1, "system"), (2, "stand-in")])
def test_cxx_launcher_of_the_sharded_job(tmp_path, world, rccl):
    """Host code above the C-ABI in C++ (north_star): tests/cxx/dist_client.cpp, built with plain g++ against include/voldor_hip.h and
    linked with -lvoldor_hip, is the whole launcher -- vk_set_device, vk_dist_init_file, vk_voldor_sharded per step, the records of all
    ranks checked against one-at-a-time windows.  One rank on the real (system) RCCL; two ranks on GPU 0 through the file-backed stand-in."""
    import shutil
    from voldor_amd import build, capi
    if shutil.which("g++") is None:
        pytest.skip("no g++ on this box")
    exe = str(tmp_path / "dist_client")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "dist_client.cpp"),
                           "-L", libdir, "-lvoldor_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ)
    env.pop("VOLDOR_HIP_RCCL", None)
    if rccl == "stand-in":
        build.build_test_lib()
        env["VOLDOR_HIP_RCCL"] = os.path.join(ROOT, "voldor_amd", "lib", "libfake_rccl_test.so")
    path = str(tmp_path / "id")
    procs = [subprocess.Popen([exe, str(r), str(world), path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0 and f"DIST CLIENT OK {r}" in out, (r, p.returncode, out[-2000:])
