"""`python bench.py --gpus N` as the driver types it (no torchrun environment): the parent has to start the N ranks itself, as a
child process, and hand rank 0's JSON line through.  Driven here through the `--backend gloo` dry mode (CPU, stand-in nets, no
kernels): launcher, rendezvous on 127.0.0.1, argument plumbing, the two bucketed gradient exchanges per step and the N>1 report."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, text=True, cwd=ROOT)
    return p


@pytest.mark.timeout(300)
def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--backend", "gloo", "--ddp-bucket-mb", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert "torch.distributed.run" in p.stderr            # the parent said what it started
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["dry_run"] is True and "DRY RUN" in out["config"]["workload"]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["value"] > 0
    assert out["weights_identical_on_all_ranks_after_the_steps"] is True
    ddp = out["ddp"]
    assert ddp["world_size_seen"] == 2 and ddp["bucket_mb"] == 1
    assert [r["rank"] for r in ddp["ranks"]] == [0, 1]
    assert len({r["pid"] for r in ddp["ranks"]}) == 2      # two processes, not two threads of one
    for r in ddp["ranks"]:
        assert r["ms_per_step"] > 0
        assert r["exposed_allreduce_ms_D"] is not None and r["exposed_allreduce_ms_G"] is not None
    assert ddp["exposed_allreduce_ms_per_step"]["G"]["max_over_ranks"] >= 0
    assert ddp["allreduce_bytes_per_step"]["G"] > 0 and ddp["buckets"]["G"] >= 2
    assert "OMP_NUM_THREADS" in ddp["env"]


@pytest.mark.timeout(300)
def test_child_failure_is_the_parents_exit_status():
    # two ranks started for a command line that says three: every rank refuses; the launcher must report that, not swallow it
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    code = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.self_launch(2, ['--gpus', '3', '--backend', 'gloo', '--steps', '1']))" % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, text=True, cwd=ROOT)
    assert p.returncode != 0
    assert "--gpus 3 but WORLD_SIZE=2" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(120)
def test_single_rank_dry_mode_needs_no_launcher():
    p = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--backend", "gloo"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["ddp"] is None
