"""bench.py's launcher on the CPU (gloo): `--gpus N` without WORLD_SIZE must start N ranks itself
(round 1 silently ran one), a rank count that differs from --gpus must fail, and the per-rank
block offsets / counter all_reduce / MAX-over-ranks plumbing must see every rank.  `--dry-run`
does no compute (the hot path has no CPU fallback to run here)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=300)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out                      # rank 0 prints ONE line
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--blocks", "5", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["dry_run"] is True and j["value"] is None
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    # rank r shards blocks [r*G, (r+1)*G): first-block offsets 0 and 5 -> (0+1) + (5+1); 2 x 5 blocks
    assert j["first_block_sum"] == 7 and j["blocks_total"] == 10


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr and "{" not in r.stdout
