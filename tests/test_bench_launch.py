"""`python bench.py --gpus N` with NO launcher (the driver's command line): bench.py starts its own N ranks -- one child
process per GPU, as `cli(main, cfg)` does at /root/reference/utils/gsplat_utils/gsplat_trainer.py:998 -- from a parent
that makes no HIP call, forwards rank 0's JSON line and turns a failed rank into a non-zero exit.

CPU tests use `--launch-check` (process group over gloo, `config.rccl` report, no kernels: the product has no CPU path);
the GPU test runs the real benchmark with two ranks on the one device of the box (gloo rehearsal)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SPLAT_ONE_AMD_BACKEND")}
    return env


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    rep = out["config"]["rccl"]
    assert out["n_gpus"] == 2 and rep["world_size"] == 2 and rep["self_launched"] is True
    assert [x["rank"] for x in rep["ranks"]] == [0, 1]
    assert len({x["pid"] for x in rep["ranks"]}) == 2          # two processes, neither of them the parent
    import torch
    if torch.cuda.device_count() < 2:
        assert rep["backend"] == "gloo"                          # fewer devices than ranks: flagged rehearsal


def test_failed_rank_gives_nonzero_exit():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--fail-rank", "1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]     # no JSON line from a failed job
    assert "rank 1 exited with code 3" in r.stderr


def test_a_rank_that_dies_late_ends_all_eight_ranks_and_leaves_no_orphans():
    """`bench.py --gpus 8` (what the driver's scaling run starts): the process group comes up, works, and THEN one rank dies
    while the seven others wait in a collective it never joins.  The parent must notice, stop the others -- by PID -- and exit
    non-zero; none of the eight children may outlive it."""
    import re
    import time
    import psutil
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launch-check", "--fail-rank", "5", "--fail-late"], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "rank 5 exited with code 3" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    m = re.search(r"started ranks: pids ([0-9 ]+)", r.stderr)
    assert m, r.stderr[-2000:]
    pids = [int(x) for x in m.group(1).split()]
    assert len(pids) == 8
    time.sleep(0.5)
    alive = [p for p in pids if psutil.pid_exists(p) and psutil.Process(p).status() != psutil.STATUS_ZOMBIE]
    assert not alive, alive
    assert time.time() - t0 < 300            # stopped by the parent, not by a collective's timeout


def test_under_a_launcher_no_second_spawn():
    """RANK in the environment (torch.distributed.run's contract): bench.py adopts the rank it was given."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    rep = _json_line(outs[0][0])["config"]["rccl"]
    assert rep["self_launched"] is False and rep["world_size"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # only rank 0 prints the line


def test_one_gpu_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launch-check"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout) == {"launch_check": True, "n_gpus": 1, "config": {"rccl": None}}


@pytest.mark.gpu
def test_bench_two_ranks_without_launcher(dev):
    """The real benchmark as the driver starts it, `python bench.py --gpus 2 ...`, on a one-GPU box: both ranks share the
    device over gloo (RCCL refuses that), small sizes; the line must be complete and say what it ran on."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--gaussians", "8000",
                        "--width", "320", "--height", "192", "--no-cpu-baseline"], env=_clean_env(), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak"
    assert out["unit"] == "views/s" and abs(out["value"] - 2 * out["optimizer_steps_per_s"]) < 1e-6 * out["value"]
    rep = out["config"]["rccl"]
    assert rep["world_size"] == 2 and rep["self_launched"] is True and len(rep["ranks"]) == 2
    assert "comm_ms" in out["config"]


def test_comm_model_counts_the_bytes_the_row_pieces_move():
    """bench.py's predicted reduce-scatter / all-gather time (config.comm_model) uses the layout of distributed.RowShardedAdam:
    rows rounded up to chunks x world x 64, 236 bytes per row at SH degree 3, 1 / world of them per link and direction."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from splat_one_amd.distributed import RowShardedAdam
    m = bench.comm_model(100_000, 16, 1)
    for w in (2, 4, 8):
        e = m["by_world"][str(w)]
        ra = RowShardedAdam.__new__(RowShardedAdam)
        ra.world, ra.n_chunks = w, 1
        assert e["rows_exchanged"] == ra.span(100_000)
        assert abs(e["bytes_per_link_and_direction_per_phase"] - ra.span(100_000) * 236.0 / w) < 1e-6
        # what every rank sends per phase over its w - 1 links == RowShardedAdam's own count (both phases: x 2)
        assert abs(2 * (w - 1) * e["bytes_per_link_and_direction_per_phase"] - ra.bytes_per_link_and_step(100_000)) < 1e-3
        assert e["reduce_scatter_ms"] > 0 and e["reduce_scatter_ms"] == e["all_gather_ms"]
    m4 = bench.comm_model(500_000, 16, 4)
    assert m4["assumptions"]["collectives_per_phase"] == 8 and m4["by_world"]["8"]["rows_exchanged"] % (4 * 8 * 64) == 0
