"""bench.py's host logic that needs no GPU: which workload an (N, flags) pair names, and the wall-budget planner."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    d = dict(log2_entities=20, log2_entities_total=None, weak=False)
    d.update(kw)
    return argparse.Namespace(**d)


def test_default_workload_is_the_metrics_configuration_at_every_n():
    """BASELINE.json: "2^20 leaves, 1/2/4/8 GPU" -- the total stays 2^20 and N divides it (strong scaling); configs[3] is
    --log2-entities-total 22 on 8 GPUs; --weak keeps the per-GPU size instead."""
    assert bench.plan_workload(_args(), 1) == (1 << 20, 1 << 20, 20, "strong")       # the label of the series: N = 1, 2, 4, 8 agree
    for n in (2, 4, 8):
        assert bench.plan_workload(_args(), n) == (1 << 20, (1 << 20) // n, 20, "strong")
    assert bench.plan_workload(_args(log2_entities_total=22), 8) == (1 << 22, 1 << 19, 22, "strong")
    assert bench.plan_workload(_args(weak=True), 8) == (1 << 23, 1 << 20, 23, "weak")
    assert bench.plan_workload(_args(log2_entities=9, weak=True), 2) == (1024, 512, 10, "weak")
    with pytest.raises(SystemExit):
        bench.plan_workload(_args(), 3)
    with pytest.raises(SystemExit):
        bench.plan_workload(_args(log2_entities=1), 4)


def test_wall_budget_planner():
    # 19 s steps, 450 s left: the driver's 20 + 5 does not fit; timed steps come first, further warm-ups only out of slack
    extra, steps = bench.plan_steps(20, 5, 19.0, 450.0)
    assert steps == 20 and extra == 2
    extra, steps = bench.plan_steps(20, 5, 19.0, 300.0)
    assert steps == 15 and extra == 0
    assert bench.plan_steps(20, 5, 19.0, 10.0) == (0, 3)          # never fewer than three timed steps
    assert bench.plan_steps(3, 1, None, 100.0) == (0, 3)


def test_no_launcher_means_spawn_not_exit(monkeypatch):
    """`--gpus 2` with WORLD_SIZE unset goes to spawn_ranks (which starts torch.distributed.run as a CHILD); under a launcher the
    process is a rank."""
    called = {}
    monkeypatch.setattr(bench, "spawn_ranks", lambda a: called.setdefault("spawn", a.gpus))
    monkeypatch.setattr(bench, "mode_prove", lambda a: called.setdefault("prove", a.gpus))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    bench.main()
    assert called == {"spawn": 2}
    called.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")
    bench.main()
    assert called == {"prove": 2}
    called.clear()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench.main()
    assert called == {"prove": 1}
