"""bench.py --gpus N started WITHOUT a launcher must start its own ranks (torch.distributed.run as a child process, before this
process touches the GPU) and, where the node has fewer than N devices, fail quickly with a clear message and a non-zero exit code --
never hang, never SystemExit in the parent before trying (round-2 verdict, missing 1)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--lift-only",
                           "--no-cpu-baseline"], capture_output=True, text=True, timeout=240, env=env, cwd=REPO)


def test_bench_self_launch_without_devices_fails_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has two devices: the launch would run the benchmark")
    res = _run(2)
    out = res.stdout + res.stderr
    assert res.returncode != 0
    assert ("needs 2 devices" in out) or ("needs an MI355X" in out), out[-2000:]
    assert "must be launched with" not in out


@pytest.mark.gpu
def test_bench_self_launch_on_a_one_gpu_box_reports_the_missing_devices():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one device")
    res = _run(2)
    out = res.stdout + res.stderr
    assert res.returncode != 0 and "needs 2 devices" in out, out[-2000:]


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_prints_the_n2_line_with_its_distributed_fields():
    """`EGOTAP_DIST_BACKEND=gloo python bench.py --gpus 2`: two ranks sharing the box's one GPU (RCCL needs a device per rank; the collective
    path, bucket layout and JSON fields are backend independent) -- the N = 2 line the driver's scaling run would parse: world size as
    torch.distributed reports it, one per-rank time per rank, the sharded headline checked against the oracle, and the data-parallel
    training legs with their in-backward all-reduce: 5 buckets (pose head / PU / FC encoders / final LayerNorm, one per ViT layer, patch
    embedding), 383.7 MB of fp32 gradients per step (96.98 M - 1.05 M parameters without gradient)."""
    import json
    import torch
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    env = dict(os.environ, EGOTAP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--no-fast-mode",
                          "--train-steps", "1", "--train-batch", "8", "--train-batch-bf16", "8"], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 64 and d["config"]["parallelism"].startswith("dp2")
    dd = d["distributed"]
    assert dd["world_size_seen"] == 2 and dd["backend"] == "gloo" and len(dd["per_rank_ms_per_step"]) == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 32 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-3 * d["value"]      # whole-job frames / max-over-ranks time
    assert d["cpu_baseline"] is None and d["max_abs_diff_vs_oracle"] is not None and d["max_abs_diff_vs_oracle"] < 1e-4, d["max_abs_diff_vs_oracle"]
    assert "rank 0" in d["parity_checked_on"]
    tr = d["train_step_lifting_head"]
    for leg in (tr, tr["bf16x3"], tr["config3_bf16_b1024"]):
        assert "error" not in leg, leg
        assert leg["allreduce_buckets"] == 5, leg["allreduce_buckets"]
        assert abs(leg["allreduce_bytes_per_step"] - 383.7e6) < 0.5e6, leg["allreduce_bytes_per_step"]
        assert len(leg["per_rank_ms_per_step"]) == 2 and leg["allreduce_exposed_ms_last_step"] is not None
        assert leg["allreduce_op"].startswith("sum")                # gloo: sum + scaling pass (RCCL: ReduceOp.AVG)
    assert tr["config3_bf16_b1024"]["config4_global_batch"] == 16
