"""`python bench.py --gpus N` must start its own ranks: the parent spawns a torch.distributed.run
child before touching any GPU and relays rank 0's JSON line.  Rehearsed on CPU with --dry-run (gloo, world_size 2): the
launcher, the rendezvous on 127.0.0.1, the barrier / max-over-ranks timing contract and both scaling modes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout           # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_starts_its_own_ranks_world2(scaling):
    r = _run(["--gpus", "2", "--scaling", scaling])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["dry_run"] is True
    assert r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == scaling and r["value"] > 0
    assert r["config"]["B_local"] == (14 if scaling == "strong" else 28)      # 28 -> 14,14 (jatsr_amd.dist.shard_range)


def test_bench_dry_run_single_rank():
    r = _run(["--gpus", "1"])
    assert r["n_gpus"] == 1 and r["ranks_seen"] == 1


@pytest.mark.parametrize("mode", ["train", "long"])
def test_bench_launcher_other_modes_world2(mode):
    """`--mode train` (configs[3]: the gradient exchange of jatsr_amd.dist.exchange_sum_, slice by slice, summed right on both
    ranks) and `--mode long` (configs[4]: the chunk plan sharded round-robin, one object gather per file) through the same
    self-launch + 127.0.0.1 rendezvous + barrier / max-over-ranks contract, world_size 2 on gloo."""
    r = _run(["--gpus", "2", "--mode", mode])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["dry_run"] is True and r["config"]["mode"] == mode
    assert r["steps"] == 3 and r["value"] > 0 and ("training" in r["metric"] if mode == "train" else "long-sequence" in r["metric"])
