"""bench.py's launch contract that can be checked without a GPU: `--gpus N` never runs (or labels) anything but N ranks on N devices."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RSIS_SHARE_GPU"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)


def test_gpus_n_without_n_devices_exits_nonzero_and_prints_no_json():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "1"], {})
    assert r.returncode != 0 and not r.stdout.strip()
    assert "--gpus 8 but only" in r.stderr


def test_world_size_that_contradicts_gpus_is_refused():
    import torch
    if not torch.cuda.is_available():
        # (without a GPU the process stops one line earlier, at "bench.py needs the GPU": still non-zero, still no JSON)
        r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
        assert r.returncode != 0 and not r.stdout.strip()
        return
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not r.stdout.strip() and "WORLD_SIZE=1" in r.stderr
