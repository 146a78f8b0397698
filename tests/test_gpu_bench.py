"""bench.py as the driver launches it for N > 1 (BASELINE configs[3]'s plumbing): `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2 ...`, here with both ranks sharing the one GPU of the test box over gloo (RSIS_SHARE_GPU=1, RSIS_DIST_BACKEND=gloo;
RCCL needs one GPU per rank).  Checks the contract of the output -- exactly one JSON line on stdout, n_gpus 2, weak scaling, a finite
loss -- and that the replayed launch mode used the three-graph overlapped gradient exchange (train.GraphedStep)."""
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


def _run(extra, env_extra):
    env = dict(os.environ, RSIS_SHARE_GPU="1", RSIS_DIST_BACKEND="gloo", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--batch", "4",
           "--T", "3", "--skip-cpu", "--skip-roofline", "--skip-secondary", "--no-settle"] + (extra if "--imsize" in extra else ["--imsize", "128"] + extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, "bench.py --gpus 2 failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    return lines, r.stderr


def test_bench_two_ranks_one_json_line_graph_replay():
    # (224x224 / bf16: the geometry and arithmetic of BASELINE configs[2..3], at a batch the shared GPU holds twice)
    lines, err = _run(["--imsize", "224", "--dtype", "bf16"], {})
    assert len(lines) == 1, "stdout must hold exactly one line, got %d: %s" % (len(lines), lines[:3])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3 and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert "hipGraph replay" in out["config"]["launch"], out["config"]["launch"]
    ex = out["config"]["exchange_ms_per_step"]
    assert ex["replays"] == 3 and ex["cuts"] == 2
    assert all(ex[k] >= 0 for k in ("graph_A_fwd_bptt", "graph_B1_trunk_bwd", "graph_B2_trunk_bwd", "exposed_allreduce", "graph_C_adam_repack"))
    assert "communicator size 2" in err and "split-graph schedule" in err
    # the exchange explains itself in the JSON line (not only on stderr): schedule, why, bytes per range, the collectives alone, the
    # same iteration without exchange, per-rank step times
    xc = out["exchange"]
    assert xc["mode"] == "cut-graphs" and xc["world"] == 2 and xc["cuts"] == 2 and xc["backend"] == "gloo", xc
    assert xc["rccl_direct"]["mode"] == "cuts" and "gloo" in xc["rccl_direct"]["why"], xc["rccl_direct"]
    rb = xc["range_bytes"]
    assert set(rb) == {"dec", "trunk_hi", "rest"} and 150e6 < sum(rb.values()) < 200e6 and rb["trunk_hi"] > rb["dec"] > rb["rest"] > 0, rb
    assert isinstance(xc["allreduce_alone_ms"], float) and isinstance(xc["step_ms_without_exchange"], float), xc
    assert len(xc["rank_ms_per_step"]["all"]) == 2 and xc["rank_ms_per_step"]["max"] >= xc["rank_ms_per_step"]["min"] > 0


def test_bench_two_ranks_eager_bucketed_exchange():
    lines, err = _run(["--no-graph"], {})
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert "eager" in out["config"]["launch"]


def _run_plain(gpus, extra, env_extra, timeout=600):
    """plain `python bench.py --gpus N ...` -- the shape of the command the driver runs for N = 1 -- with no launcher around it"""
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "3", "--batch", "4", "--T", "3",
           "--imsize", "128", "--skip-cpu", "--skip-roofline", "--skip-secondary", "--no-settle"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_plain_bench_gpus_2_launches_two_ranks_by_itself():
    r = _run_plain(2, [], {"RSIS_SHARE_GPU": "1", "RSIS_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, "plain bench.py --gpus 2 failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must hold exactly one line, got %d: %s" % (len(lines), lines[:3])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert "self-launch" in r.stderr and "communicator size 2" in r.stderr


def test_plain_bench_refuses_more_ranks_than_devices():
    """one rank per GPU: on a box with fewer than 8 devices `bench.py --gpus 8` exits non-zero and prints no JSON line (it used to run
    one rank and label the line n_gpus 1)"""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 devices visible: the refusal does not apply")
    r = _run_plain(8, [], {}, timeout=300)
    assert r.returncode != 0
    assert not r.stdout.strip(), r.stdout[-500:]
    assert "device(s) visible" in r.stderr
    # ... and a launcher that provides another world size than --gpus is refused as well
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--skip-cpu", "--skip-roofline",
           "--skip-secondary", "--no-settle"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, **env))
    assert r.returncode != 0 and not r.stdout.strip() and "WORLD_SIZE=1" in r.stderr
