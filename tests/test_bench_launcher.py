"""`python bench.py --gpus N` with N > 1 must start its own ranks (reference
tools/scripts/dist_train.sh:18).  The launcher is exercised here without a GPU: TODA_BENCH_DRYRUN=1 makes the ranks
rendezvous over gloo, all-reduce a one and print the JSON line instead of training."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, TODA_BENCH_DRYRUN="1", OMP_NUM_THREADS="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=600)


@pytest.mark.timeout(900)
def test_plain_invocation_spawns_two_ranks_and_relays_one_line():
    p = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["comm"]["ranks"] == 2        # both ranks took part in the all-reduce


@pytest.mark.timeout(900)
def test_failing_rank_gives_nonzero_exit():
    p = _run({"TODA_BENCH_DRYRUN_FAIL_RANK": "1"}, "--gpus", "2")
    assert p.returncode != 0


def test_world_size_mismatch_is_an_error_not_an_assert():
    env = dict(os.environ, TODA_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("reducer", ["own", "torch"])
def test_two_rank_rehearsal_of_the_n_gpu_bench_path(reducer):
    """What the driver's N = 8 run executes around the kernels, on two gloo ranks: wrap_ddp (both reducers), warm-up, barriers,
    timed steps, max over ranks and comm_report - no_sync() steps, the stand-alone all-reduce, the exposed / hidden split."""
    p = _run({"TODA_BENCH_DRYRUN_REHEARSAL": "tests.bench_rehearsal:make", "TODA_DDP": reducer, "PYTHONPATH": ROOT},
             "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    comm = line["comm"]
    assert comm["ranks"] == 2 and comm["backend"] == "gloo" and comm["grad_bytes"] > 4_000_000
    assert comm["reducer"] == ("DataParallel" if reducer == "own" else "DistributedDataParallel")
    for key in ("allreduce_ms_standalone", "ms_per_step_without_allreduce", "allreduce_exposed_ms", "allreduce_hidden_ms"):
        assert comm[key] >= 0.0
    assert comm["ms_per_step_without_allreduce"] > 0 and line["ms_per_step"] > 0 and line["value"] > 0


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("reducer", ["own", "torch"])
def test_world_8_rehearsal_places_every_rank_and_all_eight_answer(reducer):
    """The shape of the driver's N = 8 run (reference tools/scripts/dist_train.sh:18, tools/train.py:65-74,143) on eight
    gloo ranks of this container: one JSON line, comm.ranks == 8, both reducers, and every rank pinned to its own slice of the allowed
    CPUs before it does anything else (8 cores here: one each; on the GPU box 16: two each - main thread + prefetch worker)."""
    p = _run({"TODA_BENCH_DRYRUN_REHEARSAL": "tests.bench_rehearsal:make", "TODA_DDP": reducer, "PYTHONPATH": ROOT},
             "--gpus", "8", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["comm"]["ranks"] == 8
    assert line["comm"]["reducer"] == ("DataParallel" if reducer == "own" else "DistributedDataParallel")
    cpus = line["rank_cpus"]
    allowed = sorted(os.sched_getaffinity(0))
    per = max(1, len(allowed) // 8)
    assert len(cpus) == 8 and all(len(c) == per for c in cpus), cpus
    if len(allowed) >= 8:
        flat = [c for rank in cpus for c in rank]
        assert len(set(flat)) == len(flat), f"ranks share CPUs: {cpus}"


@pytest.mark.timeout(900)
def test_missing_rank_line_is_a_failure_at_world_8():
    """A rank that dies takes the launcher's exit status with it at N = 8 too (no JSON line is accepted from a partial job)."""
    p = _run({"TODA_BENCH_DRYRUN_FAIL_RANK": "5"}, "--gpus", "8")
    assert p.returncode != 0
