"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it) and the newest-profile lookup, on CPU.
`--dry` runs the N-rank plumbing -- rendezvous on 127.0.0.1, the one gather, max-over-ranks timing, ONE JSON line from
rank 0 -- on gloo with no GPU work; without `--dry` every rank must fail loudly on a box without a GPU and the launcher must
return non-zero (no silent single-rank run, no CPU fallback)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_n_self_launches_n_ranks_dry():
    r = _run("--gpus", "3", "--dry", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["steps"] == 2 and line["data"] == "dry-run" and line["scaling"] == "weak"
    assert "not a valid result" in line["config"]["workload"]


def test_gpus_n_without_gpu_fails_loudly():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert "no HIP device" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_refused():
    r = _run("--gpus", "2", "--dry", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_newest_traffic_profile_orders_by_round_then_version(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    for name in ("r1_pmc_traffic_v7.json", "r1_pmc_traffic_v13.json", "r1_pmc_traffic.json"):
        (prof / name).write_text("{}")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert os.path.basename(bench.newest_traffic_profile()) == "r1_pmc_traffic_v13.json"      # not _v7 (plain sort)
    (prof / "r2_pmc_traffic_v2.json").write_text("{}")
    assert os.path.basename(bench.newest_traffic_profile()) == "r2_pmc_traffic_v2.json"
