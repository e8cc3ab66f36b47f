"""bench.py's cpu_baseline leg runs in a child process of its own (round 6: one OpenMP thread per physical core, pinned -- pinning the bench process itself bound
every thread it started later to one core).  The child must work without a GPU and without torch: `python bench.py --cpu-leg <workload> <windows>` prints one JSON
line with the per-window times and the oracle's poses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_leg_child_prints_times_and_poses_without_torch():
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_NUM_THREADS=str(max(1, (os.cpu_count() or 2) // 2)))
    code = ("import sys, runpy; sys.modules['torch'] = None; sys.argv = ['bench.py', '--cpu-leg', 'cfg2', '1']; "  # (importing torch in the child would raise)
            "runpy.run_path('bench.py', run_name='__main__')")
    pr = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    doc = json.loads(pr.stdout.decode().strip().splitlines()[-1])
    assert len(doc["tws"]) == 1 and doc["tws"][0] > 0 and doc["cores"] >= 1
    assert doc["n_registered"] == 5 and len(doc["poses"]) == 5 and len(doc["poses"][0]) == 6
    assert doc["bind"] == "close" and doc["places"] == "cores"
