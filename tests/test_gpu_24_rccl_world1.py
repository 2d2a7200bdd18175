"""-m gpu: RCCL really executes.  The test box has ONE GPU, so every multi-rank test of the exchange talks through gloo; here a process
group of ONE rank is created over backend "nccl" (= RCCL on ROCm) and ``dist.force_collectives`` drops the world-size guards, so that
every collective of ``dist.FrameExchange`` / ``graph_view.FrameGraph(exchange=...)`` is issued for real: dtype support (uint8 MAX),
``device_id=`` initialisation, async work handles next to the two-stream graph replay, captures next to RCCL's proxy thread.  A
one-rank all-reduce is the identity: gradients must equal the no-exchange frame's (the reference has no counterpart,
/root/reference/project/models/trainers/base.py:411).  Child processes under a timeout: a hung collective fails the test, not pytest."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_every_collective_of_the_exchange_runs_over_rccl_at_world_1():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], env=_env(), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")][-1]
    out = json.loads(line[len("RCCL_WORLD1 "):])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["issued"]["all_reduce"] > 100 and out["issued"]["broadcast"] >= 1
    fixed = [v for k, v in out["modes"].items() if k.startswith("fixed")]
    assert len(fixed) == 4 and all(m["captures"] == 2 for m in fixed)
    print("[rccl world 1] " + json.dumps(out))


@pytest.mark.parametrize("exchange", ["view", "frame", "auto"])
def test_bench_self_spawn_path_with_forced_collectives(exchange):
    """The driver's launch line at N = 1 (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`) with
    BDS_FORCE_COLLECTIVES=1: bench.py creates the nccl group at world 1 and times the frame WITH the exchange."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "2", "--gaussians", "60000",
           "--width", "480", "--height", "270", "--exchange", exchange, "--no-cpu-baseline", "--no-pair-stats", "--no-api-path",
           "--no-random-views", "--no-exchange-probe"]
    env = _env(BDS_FORCE_COLLECTIVES="1")
    env.pop("MASTER_PORT")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = res["config"]
    assert res["n_gpus"] == 1 and cfg["collectives_forced_at_world1"] is True and res["value"] > 0
    assert cfg["allreduce_bytes_per_step"] > 0 and cfg["exchanges_per_step"] >= 1
    assert cfg["exchange"]["mode"] in ("view", "frame")
    if exchange != "auto":
        assert cfg["exchange"]["mode"] == exchange
    assert "WARNING" not in r.stderr or "fell back" not in r.stderr


def test_bench_exchange_probe_child():
    """The default bench line's `config.exchange_world1` block: the child process that times the frame without an exchange, with the
    per-view exchange and with the per-frame all-reduce over RCCL at world size 1."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--exchange-probe-only", "--steps", "3", "--gaussians", "60000", "--width", "480",
           "--height", "270"]
    env = _env()
    env.pop("MASTER_PORT")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["plain_ms_per_frame"] > 0
    for mode, n in (("per_view", 2 * 6 + 2), ("per_frame", 2)):
        assert out[mode]["all_reduces_per_frame"] == n, out
        assert out[mode]["grad_rel_vs_plain"] < 1e-5, out
    print("[exchange probe] " + json.dumps(out))
