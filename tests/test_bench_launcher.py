"""bench.py as its own launcher: `python bench.py --gpus N` without torchrun starts N rank
processes before touching the GPU.  Runs on CPU: the rank environment, and the whole spawn -> rendezvous -> barrier ->
MAX-over-ranks -> one JSON line path with the gloo stub step."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_env_is_one_rank_per_gpu_on_loopback():
    sys.path.insert(0, REPO)
    import bench
    envs = [bench.rank_env(r, 8, 29999, base={"PATH": "/usr/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "1"}) for r in range(8)]
    for r, e in enumerate(envs):
        assert e["RANK"] == e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "8"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" and e["PATH"] == "/usr/bin"      # a value the caller set is kept
    assert bench.rank_env(0, 2, 29999, base={"PATH": "/usr/bin"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"   # unset: dmabuf IPC for RCCL
    assert bench.free_port() > 0


def test_gpus_n_without_a_launcher_spawns_n_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--stub",
                        "--no-roofline", "--no-cpu-baseline", "--height", "64", "--width", "96"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["metric"] == "training images/sec at 64x96 bf16"
    # MAX over ranks: rank 1 sleeps 2 ms per step, so 5 steps of global batch 16 cannot beat 16 * 5 / 0.010 images/s
    assert 0 < out["value"] <= 16 * 5 / 0.010


def test_launcher_takes_the_other_ranks_down_when_one_dies():
    """A rank that exits early must not leave the others (and the launcher) in the rendezvous forever: the launcher polls every
    child, terminates the rest on the first failure and returns that exit code."""
    import subprocess
    import time
    env = dict(os.environ, CRD_STUB_FAIL_RANK="1", CRD_BENCH_TIMEOUT="120")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--stub"],
                       env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr[-500:])
    assert time.time() - t0 < 60
