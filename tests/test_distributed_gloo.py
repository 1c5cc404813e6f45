"""N > 1 path on CPU: two processes (torch.distributed, gloo, 127.0.0.1) run the generation
loop of saspa_aug_amd.run_aug with an injected stand-in for the device batch generator, and
must produce exactly the files / pixels / JSON of the single-process run: the plan, the
shard split, the per-item noise slices and the status gather are what is under test (the
device arithmetic is covered by the -m gpu tests)."""
import hashlib
import json
import os
import socket
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from PIL import Image

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import dataset_utils
from saspa_aug_amd import run_aug as R


def fake_generator(batch, noises, sources):
    """Deterministic function of (prompt, noise, source): what a pure sampler is."""
    imgs = []
    for it, n, src in zip(batch, noises, sources):
        h = hashlib.sha256(it.prompt.encode() + n.numpy().tobytes() + src.tobytes()).digest()
        img = np.frombuffer(h * ((src.size // 32) + 1), np.uint8)[: src.size].reshape(src.shape).copy()
        imgs.append(img)
    return np.stack(imgs), sources.copy()


def _settings(prompts_file, root):
    return R.Settings(DATASET="synthetic", NUM_PER_IMAGE=3, SEED=1, RESOLUTION=64, BATCH_SIZE=4, PROMPTS_FILE=prompts_file,
                      DATASET_KWARGS=dict(root_path=root, n_images=7, sizes=((64, 64), (64, 128), (128, 64)), seed=2),
                      SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0)


def _worker(rank, world, port, prompts_file, root, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = _settings(prompts_file, root)
        res = R.main(s, batch_generator=fake_generator, dist=dist)
        if rank == 0:
            q.put((res["json_path"], res["status"].tolist()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _snapshot(folder):
    out = {}
    for f in sorted(os.listdir(folder)):
        out[f] = np.asarray(Image.open(Path(folder) / f)).tobytes()
    return out


@pytest.mark.timeout(300)
def test_two_rank_run_equals_single_process(tmp_path):
    prompts = tmp_path / "prompts.txt"
    prompts.write_text("\n".join(f"An airplane number {k} over a field." for k in range(20)) + "\n")
    # single process
    root1 = str(tmp_path / "one" / "data")
    r1 = R.main(_settings(str(prompts), root1), batch_generator=fake_generator)
    assert (r1["status"] == 1).all() and len(r1["items"]) == 21
    # two ranks, gloo
    root2 = str(tmp_path / "two" / "data")
    dataset_utils.SyntheticUtils(root_path=root2, n_images=7, sizes=((64, 64), (64, 128), (128, 64)), seed=2)  # materialise once
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(prompts), root2, q)) for r in range(2)]
    for p in procs:
        p.start()
    json2, status2 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status2 == [1] * 21
    f1, f2 = _snapshot(r1["output_folder"]), _snapshot(str(Path(json2).parent / "images"))
    assert list(f1) == list(f2) and len([k for k in f1 if "_prompt_" in k]) == 21
    assert f1 == f2                                             # same pixels in every file
    b1 = json.load(open(r1["json_path"]))
    b2 = json.load(open(json2))
    strip = lambda body, root: {k: sorted(Path(p).name for p in v) for k, v in body.items()}  # noqa: E731
    assert strip(b1, root1) == strip(b2, root2) and all(len(v) == 3 for v in b1.values())
    # resume: a second run skips everything (skip-if-exists) and leaves the folder unchanged
    r3 = R.main(_settings(str(prompts), root1), batch_generator=fake_generator)
    assert (r3["status"] == 0).all() and all(it.skip for it in r3["items"])
    assert _snapshot(r1["output_folder"]) == f1


def test_error_isolation_marks_items_failed(tmp_path):
    prompts = tmp_path / "prompts.txt"
    prompts.write_text("An airplane.\nAnother airplane.\n")
    calls = {"n": 0}

    def flaky(batch, noises, sources):
        calls["n"] += 1
        if calls["n"] == 2:
            raise RuntimeError("HIP out of memory (simulated)")
        return fake_generator(batch, noises, sources)
    r = R.main(_settings(str(prompts), str(tmp_path / "d" / "data")), batch_generator=flaky)
    st = r["status"].tolist()
    assert st.count(-1) > 0 and st.count(1) > 0 and st.count(-1) <= 4        # one batch failed, the run went on
    body = json.load(open(r["json_path"]))
    assert sum(len(v) for v in body.values()) == st.count(1)


def _synth_worker(rank, world, port, q, small_shm=False):
    import torch.distributed as dist
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import weights as W
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if small_shm:                                     # a container's default 64 MB /dev/shm
            real_statvfs = os.statvfs
            os.statvfs = lambda p: type("S", (), dict(f_bavail=16, f_frsize=4096))() if p == "/dev/shm" else real_statvfs(p)
        calls = []
        real = W.synth_family
        W.synth_family = lambda cfgs, seed=0: (calls.append(seed), real(cfgs, seed))[1]
        fam = W.synth_family_shared(CFG.tiny(), 3, dist, "tiny")
        digest = hashlib.sha256(b"".join(fam[k][n].numpy().tobytes() for k in sorted(fam) for n in sorted(fam[k]))).hexdigest()
        dist.barrier()                                   # the first rank unlinks after the loader's last barrier
        left = [f for f in os.listdir("/dev/shm") if f.startswith(f"saspa_synth_{os.getuid()}_{port}_")] if os.path.isdir("/dev/shm") else []
        q.put((rank, len(calls), digest, left))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_shared_synthetic_family_is_drawn_once_per_node():
    """bench.py --gpus N: the node's first rank draws the random weights, the others map its /dev/shm file (VERDICT r2 item 9:
    8 CPU synthesisers in one 16-thread cgroup at start-up); same tensors on every rank, nothing left behind."""
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import weights as W
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_synth_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    fam = W.synth_family(CFG.tiny(), 3)
    ref = hashlib.sha256(b"".join(fam[k][n].numpy().tobytes() for k in sorted(fam) for n in sorted(fam[k]))).hexdigest()
    assert [g[1] for g in got] == [1, 0], "rank 1 ran its own synthesiser"
    assert got[0][2] == got[1][2] == ref
    assert got[0][3] == [] and got[1][3] == []


@pytest.mark.timeout(300)
def test_shared_synthetic_family_falls_back_when_dev_shm_is_too_small():
    """Every rank then draws its own copy (same tensors), nothing is left behind and nobody hangs at a barrier."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_synth_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [1, 1] and got[0][2] == got[1][2]
    assert got[0][3] == [] and got[1][3] == []
