"""bench.py on the GPU box: the 1-rank line is well-formed, and a 2-rank run (both ranks on the one
card, gloo for the exchange — the box has a single GPU, RCCL needs one device per rank) gathers exactly
the edge set of the 1-rank run: tile-range sharding + gather on real engine output."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_and_two_rank_equivalence():
    common = ["--steps", "2", "--warmup", "1", "--n-sources", "1500", "--cpu-sample", "0"]
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, {"KSP_BENCH_CHECKSUM": "1"})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in one
    assert one["n_gpus"] == 1 and one["config"]["n_sources"] == 1500 and one["value"] > 0
    assert one["roofline"]["bound"] == "hbm" and one["roofline"]["peak"] == 8000.0
    # a roofline FRACTION: recomputable from the line itself, never above 1
    r = one["roofline"]
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["bytes_per_step"] / (one["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["groups"] and {"partition", "join"} <= {g["group"] for g in r["groups"]}
    assert all(g["ms"] > 0 for g in r["groups"])
    assert one["naive_pairwise"]["bytes_per_step"] > r["bytes_per_step"]
    assert one["config"]["buffers_regrown_in_timed_region"] == 0
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29713", "bench.py", "--gpus", "2"] + common,
               {"KSP_BENCH_CHECKSUM": "1", "KSP_BENCH_SHARE_GPU": "1", "KSP_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2
    assert two["config"]["nonzero_pairs"] == one["config"]["nonzero_pairs"] > 1000
    assert two["config"]["checksum"] == one["config"]["checksum"] != 0
