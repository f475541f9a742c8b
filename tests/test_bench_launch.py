"""CPU: bench.py's multi-GPU plumbing without a device.  `python bench.py --gpus 2` with WORLD_SIZE unset must itself start two
ranks (fresh child processes) and report n_gpus 2; the ranks go through shard.init_distributed / timed_steps / aggregate_fps -
the same functions the GPU run uses - over gloo.  ORB_BENCH_SELFTEST=1 replaces the device work by a sleep (there is no CPU
fallback of the hot path to run), so only the launcher, the barrier / MAX-over-ranks timing and the JSON contract are tested."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, extra_env):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(ORB_BENCH_SELFTEST="1", ORB_BENCH_BACKEND="gloo")
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks():
    p = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")      # ONE line on stdout, nothing else (library chatter such as Gloo's goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["scaling"] == "weak"
    # MAX over ranks: rank 1 sleeps 4 ms per step, rank 0 2 ms
    assert out["ms_per_step"] >= 4.0
    # whole-job value: both ranks' frames over the max time
    assert abs(out["value"] - 2 * 256 * 224 * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-3   # default step: 224 batches of 256 frames
    # the host-fed leg runs on every rank as well: MAX over ranks (rank 1 sleeps 6 ms, rank 0 3 ms), whole-job aggregate
    hf = out["host_fed"]
    assert hf["n_gpus"] == 2 and hf["seconds_max_over_ranks"] >= 0.006
    assert abs(hf["value"] - 2 * 48 * 256 / hf["seconds_max_over_ranks"]) / hf["value"] < 1e-2


def test_no_host_fed_flag():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--no-host-fed"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["host_fed"] is None


def test_gpus_1_is_a_single_process():
    p = _run(["--steps", "3", "--warmup", "0"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr
