"""bench.py as the driver runs it: `python bench.py --gpus N ...` with no launcher around it.  N = 1 prints the contract's JSON
line; N = 2 starts its own ranks (a child `torch.distributed.run`, never a re-exec) -- rehearsed here with the gloo backend so
that both ranks can share the one GPU of the box -- and relays rank 0's line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--scale", "1", "--steps", "3", "--warmup", "1", "--no-strong-x1000", "--no-cpu-baseline"]
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config")


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, text=True, env=e,
                       timeout=900, cwd=REPO)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]          # stdout carries exactly ONE line
    return json.loads(lines[0])


def test_bench_prints_the_contract_line_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = _run(SMALL + ["--no-extras"])
    for k in KEYS:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "weak"
    assert "workload" in out["config"] and out["config"]["ranks"] == 1
    assert 0.0 < out["roofline"]["frac"] <= 1.0 and out["roofline"]["bound"] in ("hbm", "mfma")
    assert "supervised pairs" in out["metric"]


def test_bench_gpus_2_starts_its_own_ranks():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = _run(["--gpus", "2", "--strong-scale", "2", "--scale", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                "--no-kernels"], env={"MMG_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["collective_backend"] == "gloo"
    assert out["value"] > 0 and "segments" in out["config"]["launch"]           # gloo cannot be recorded: the segment chain
    # north_star's curve -- ONE graph sharded over the ranks -- sits inside `config`, where a parser of the contract keeps it
    for k in ("strong_x2_ms_per_step", "strong_x2_edges_per_s", "strong_x2_d128_ms_per_step", "strong_x2_d128_edges_per_s"):
        assert out["config"][k] > 0, k
    assert out["strong_x1000"]["n_gpus"] == 2 and out["strong_x1000"]["scaling"] == "strong"
