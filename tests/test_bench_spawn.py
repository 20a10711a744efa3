"""bench.py's own N-rank launch path (`python bench.py --gpus N` with no torchrun environment), driven on the CPU
with gloo: the parent starts the ranks, rank 0 prints one line with n_gpus == N and every rank in ranks_seen; a
label that does not match the job fails loudly instead of running as N = 1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=env, timeout=300)


def test_gpus2_spawns_two_ranks_gloo():
    r = _run(["--gpus", "2", "--selftest-cpu", "--bursts", "1001"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout                       # rank 0 prints exactly one line
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["tmax"] == 2.0


def test_gpus2_without_two_gpus_fails_loudly():
    # no GPU in the CPU container: the real (non-selftest) path must refuse, not run one rank as "n_gpus: 1"
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--selftest-cpu"], {"WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


import pytest


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu():
    """The whole N-rank bench flow with real GPU work: `bench.py --gpus 2 --rehearse-one-gpu` (both ranks on cuda:0, gloo
    collectives -- RCCL refuses two ranks on one device): spawn, table broadcast, context from the broadcast blob, per-rank
    batches, barriers, MAX over ranks, one line with n_gpus == 2.  A launch-path check, not a scaling number."""
    r = _run(["--gpus", "2", "--rehearse-one-gpu", "--steps", "20", "--bursts", "8192", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["value"] > 0 and "rehearsal" in out
    assert out["detected_frac"] > 0.99 and out["clean_hard_bits_ok"]
