"""bench.py's multi-rank launch path (`--gpus N` must be real).

`python bench.py --gpus N` started plainly has to start N ranks itself -- before it imports torch or touches HIP -- and
any mismatch between --gpus, WORLD_SIZE, the GPUs of the node and the ranks the process group really has must end the
run with a non-zero exit code instead of a line that says n_gpus 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    return p.returncode, p.stdout, p.stderr


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_launcher_starts_two_ranks_and_reports_them():
    rc, out, err = _run(["--gpus", "2", "--launcher-selftest", "--steps", "3", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    lines = _json_lines(out)
    assert len(lines) == 1, out                      # ONE line, from rank 0
    j = lines[0]
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["units_all_ranks"] == 1000 + 1001


def test_world_size_mismatch_is_refused():
    rc, out, err = _run(["--gpus", "2", "--launcher-selftest"], {"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert rc != 0 and "refusing to run" in err and not _json_lines(out)
    rc, out, err = _run(["--gpus", "1", "--launcher-selftest"], {"WORLD_SIZE": "2", "RANK": "0"}, drop=())
    assert rc != 0 and "refusing to run" in err and not _json_lines(out)


def _gpus_visible():
    import torch
    return torch.cuda.device_count()          # does not initialise the GPU on this image


def test_more_ranks_than_gpus_fails_loudly_cpu():
    if _gpus_visible() >= 2:
        pytest.skip("two or more GPUs visible")
    rc, out, err = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-h2d"])
    assert rc != 0, out
    assert not _json_lines(out), "a line was printed although only %d GPU(s) are visible: %s" % (_gpus_visible(), out)
    assert "refusing to run" in err


@pytest.mark.gpu
def test_more_ranks_than_gpus_fails_loudly_on_the_gpu_box():
    """The same on the 1-GPU box: `python bench.py --gpus 2` without torchrun must fail, not report n_gpus 1.  The two
    ranks only count devices (which does not initialise the GPU) and leave."""
    test_more_ranks_than_gpus_fails_loudly_cpu()


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_on_one_gpu():
    """The whole N = 2 path of bench.py on the one-GPU box: `python bench.py --gpus 2 --rehearse-gloo` starts its two
    ranks itself, both on GPU 0, talking over gloo -- launch, per-rank workloads, finalize_distributed in every step, the
    config-3 job cut contiguously over the ranks and merged (its totals and N50 must be the whole job's), MAX / SUM over
    ranks, one JSON line from rank 0.  Not a measurement, and the line says so."""
    rc, out, err = _run(["--gpus", "2", "--rehearse-gloo", "--steps", "2", "--warmup", "1", "--min-seconds", "0",
                         "--reads", "60000", "--config3-reads", "150000", "--no-cpu-baseline", "--no-h2d"])
    assert rc == 0, err[-3000:]
    lines = _json_lines(out)
    assert len(lines) == 1, out
    j = lines[0]
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["backend"].startswith("gloo") and "REHEARSAL" in j["metric"]
    assert j["config"]["reads_per_gpu"] == 60000 and j["value"] > 0
    c3 = j["config3_sharded"]
    assert "error" not in c3, c3
    # the same job on one rank (the 1-rank RCCL group of --dist-path): same tables whatever the sharding
    rc, out, err = _run(["--gpus", "1", "--dist-path", "--config3-leg", "--steps", "2", "--warmup", "1", "--min-seconds", "0",
                         "--reads", "60000", "--config3-reads", "150000", "--no-cpu-baseline", "--no-h2d"])
    assert rc == 0, err[-3000:]
    one = _json_lines(out)[0]["config3_sharded"]
    assert "error" not in one, one
    for k in ("n50", "reads", "bases", "n_fail"):
        assert c3[k] == one[k], (k, c3[k], one[k])
    assert c3["reads"] == 150000 and c3["n_gpus"] == 2 and one["n_gpus"] == 1
