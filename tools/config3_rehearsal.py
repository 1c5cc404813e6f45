"""BASELINE.json configs[3] at its REAL size on one box: "SD-v1.5 + Canny ControlNet, full FGVC-Aircraft train set x 4
variants, sharded across 8 x MI355X" = 3 334 source images x NUM_PER_IMAGE 4 = 13 336 work items
(reference loop: run_aug/run_aug.py:282-504; JSON stage: all_utils/utils.py:343-354).

  python tools/config3_rehearsal.py plan  [out.json]          host only (runs without a GPU):
      synthetic Aircraft-shaped dataset (3 334 PNG sources, size histogram below), plan_work, shard_items(world = 8):
      per-rank sum(H*W) imbalance, per-rank buckets and tail batches (< 8 items), noise replay cost of the last rank;
      then create_json_of_image_name_to_augmented_images_paths on a folder holding all 13 336 output names
      (+ 3 334 _source, 10 _control side files).
  python tools/config3_rehearsal.py gpu [n_batches] [out.json]     needs the MI355X:
      rank 0's shard of the same plan (world = 8 through a stand-in for torch.distributed: same shard, same noise slice,
      the gather is local) for n_batches (default 44) through run_aug.main with the PNG writer processes on:
      images/s, deepest writer backlog, host CPU seconds per image.

Size histogram: FGVC-Aircraft photographs are landscape, 1.33 <= W/H <= 1.6 for almost all of them (the 20-pixel
copyright banner included, as the reference feeds it); utils.resize_image (smaller side 512, sides rounded to multiples of
64) sends them to 512x704 / 512x768 mostly.  The synthetic sources are drawn at those ORIGINAL sizes so the device-side
cv2-exact resize runs as in production."""
import json
import os
import resource
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import run_aug as R  # noqa: E402
from saspa_aug_amd import utils  # noqa: E402
from saspa_aug_amd.dataset_utils import SyntheticUtils  # noqa: E402
from saspa_aug_amd.synthetic import synthetic_image  # noqa: E402

N_IMAGES, NUM_PER_IMAGE, WORLD, BATCH = 3334, 4, 8, 8
# (original H, W), share of the train set -> bucket after resize_image
SIZES = [((683, 1024), 0.30), ((695, 1024), 0.12), ((812, 1200), 0.06),      # -> 512 x 768
         ((768, 1024), 0.22), ((731, 1024), 0.10), ((900, 1200), 0.05),      # -> 512 x 704
         ((640, 1024), 0.06),                                                # -> 512 x 832
         ((1024, 1024), 0.04),                                               # -> 512 x 512
         ((576, 1024), 0.03), ((1000, 1500), 0.02)]                          # -> 512 x 896, 512 x 768


def materialise(root, n_images=N_IMAGES, seed=7, templates=6):
    """FGVC-Aircraft layout with PNG sources: `templates` distinct pictures per original size, hard-linked under the image ids
    (the plan reads sizes from the headers; the generation loop decodes whichever picture an id points at)."""
    root = Path(root)
    (root / "images").mkdir(parents=True, exist_ok=True)
    from PIL import Image
    rng = np.random.RandomState(seed)
    shares = np.array([s for _, s in SIZES])
    pick = rng.choice(len(SIZES), n_images, p=shares / shares.sum())
    tdir = root / "_templates"
    tdir.mkdir(exist_ok=True)
    tmpl = {}
    for k, ((h, w), _) in enumerate(SIZES):
        for t in range(templates):
            f = tdir / f"{k}_{t}.png"
            if not f.exists():
                Image.fromarray(synthetic_image(h, w, 100 * k + t)).save(f, compress_level=1)
            tmpl[(k, t)] = f
    ids = [f"{1000000 + i:07d}" for i in range(n_images)]
    for i, image_id in enumerate(ids):
        dst = root / "images" / f"{image_id}.png"
        if not dst.exists():
            os.link(tmpl[(int(pick[i]), i % templates)], dst)
    m, v = SyntheticUtils.MANUFACTURERS, SyntheticUtils.VARIANTS
    (root / "images_train.txt").write_text("\n".join(ids) + "\n")
    (root / "images_manufacturer_train.txt").write_text("".join(f"{a} {m[i % len(m)]}\n" for i, a in enumerate(ids)))
    (root / "images_variant_train.txt").write_text("".join(f"{a} {v[i % len(v)]}\n" for i, a in enumerate(ids)))
    return [SIZES[int(k)][0] for k in pick]


def settings(root, prompts, steps=50, **kw):
    return R.Settings(DATASET="synthetic", BASE_MODEL="sd_v1.5", RESOLUTION=512, NUM_INFERENCE_STEPS=steps, NUM_PER_IMAGE=NUM_PER_IMAGE,
                      SEED=1, SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0, PROMPTS_FILE=prompts, BATCH_SIZE=BATCH,
                      DATASET_KWARGS=dict(root_path=str(root), n_images=N_IMAGES), **kw)


def prompts_file(tmp):
    f = os.path.join(tmp, "prompts.txt")
    # 100 prompts, like prompts_engineering/gpt_prompts/planes-100-gpt_v1.txt
    open(f, "w").write("".join(f"an airplane flying over landscape number {k} at dusk.\n" for k in range(100)))
    return f


def the_plan(s):
    ds = SyntheticUtils(**{**s.DATASET_KWARGS, "print_func": lambda *a: None})
    utils.set_seed(s.SEED)
    out = R.output_folder_for(s, ds.root_path)
    t0 = time.time()
    items = R.plan_work(s, ds.original_images_paths, R.read_prompts(s.PROMPTS_FILE), out, ds.get_image_stem_to_class_str_dict(), ds_utils=ds)
    return ds, out, items, time.time() - t0


def mode_plan(out_json):
    import torch
    tmp = tempfile.mkdtemp(prefix="saspa_c3_")
    try:
        root = os.path.join(tmp, "ds", "data")
        t0 = time.time()
        materialise(root)
        t_mat = time.time() - t0
        s = settings(root, prompts_file(tmp))
        ds, out_folder, items, t_plan = the_plan(s)
        assert len(items) == N_IMAGES * NUM_PER_IMAGE
        t0 = time.time()
        shards = R.shard_items(items, WORLD)
        t_shard = time.time() - t0
        area = [sum(it.height * it.width for it in sh) for sh in shards]
        mean = sum(area) / WORLD
        per_rank = []
        for r, sh in enumerate(shards):
            batches = R.make_batches(sh, BATCH)
            buckets = {}
            for it in sh:
                buckets[f"{it.height}x{it.width}"] = buckets.get(f"{it.height}x{it.width}", 0) + 1
            per_rank.append(dict(rank=r, items=len(sh), first_order=sh[0].order, last_order=sh[-1].order,
                                 sum_hw=area[r], vs_mean=round(area[r] / mean, 5), buckets=dict(sorted(buckets.items())),
                                 batches=len(batches), tail_batches=sum(len(b) < BATCH for b in batches),
                                 tail_items=sum(len(b) for b in batches if len(b) < BATCH),
                                 slots_wasted_frac=round(1 - len(sh) / (BATCH * len(batches)), 4)))
        # the last rank replays the whole CPU noise stream up to its last item: the worst case of noise_for_items
        t0 = time.time()
        noises = R.noise_for_items(items, shards[-1], s.SEED, torch.float16)
        t_noise = time.time() - t0
        assert len(noises) == len(shards[-1])
        # ---- the JSON stage on a folder with every output name of the plan (valid small PNGs: the name matching is what
        # scales as N_orig x N_files; the integrity sweep opens each file) ----
        from PIL import Image
        Path(out_folder).mkdir(parents=True, exist_ok=True)
        proto = os.path.join(tmp, "proto.png")
        Image.fromarray(synthetic_image(16, 16, 0)).save(proto)
        first = {}
        for it in items:
            first.setdefault(it.index, it)
        t0 = time.time()
        for it in items:
            os.link(proto, it.output_path)
        for idx, it in first.items():
            os.link(proto, os.path.join(out_folder, f"{it.image_stem[:R.MAX_FILENAME_LENGTH]}_source.png"))
            if idx < 10:
                os.link(proto, os.path.join(out_folder, f"{it.image_stem[:R.MAX_FILENAME_LENGTH]}_control.png"))
        t_links = time.time() - t0
        n_files = len(os.listdir(out_folder))
        t0 = time.time()
        names = os.listdir(out_folder)
        mapping = utils.match_augmented_images(ds.original_images_paths, names, out_folder)
        t_match = time.time() - t0
        t0 = time.time()
        jp = utils.create_json_of_image_name_to_augmented_images_paths(ds, out_folder, init_log=False,
                                                                       original_images_paths=ds.original_images_paths)
        t_json = time.time() - t0
        body = json.load(open(jp))
        assert len(body) == N_IMAGES and all(len(v) == NUM_PER_IMAGE for v in body.values()), "every image must list its 4 variants"
        assert body == mapping
        # integrity sweep at the real file size: 200 PNGs of a 512 x 704 picture (what PIL verify() reads per generated file)
        real = os.path.join(tmp, "real")
        os.makedirs(real)
        big = synthetic_image(512, 704, 3)
        Image.fromarray(big).save(os.path.join(real, "p.png"))
        for k in range(199):
            shutil.copy(os.path.join(real, "p.png"), os.path.join(real, f"p{k}.png"))
        t0 = time.time()
        utils.check_folder_of_images_with_pil(real, max_delete=50, substrings_to_exclude=utils.SUBSTRINGS_TO_EXCLUDE)
        t_verify200 = time.time() - t0
        res = dict(
            workload=f"BASELINE configs[3] plan: {N_IMAGES} synthetic Aircraft-shaped sources x {NUM_PER_IMAGE} variants = {len(items)} work items, "
                     f"world = {WORLD}, batch {BATCH}; host side only",
            size_histogram={f"{h}x{w}": sh for (h, w), sh in SIZES},
            buckets_all=dict(sorted({k: sum(p["buckets"].get(k, 0) for p in per_rank) for p0 in per_rank for k in p0["buckets"]}.items())),
            seconds=dict(materialise_dataset=round(t_mat, 2), plan_work=round(t_plan, 2), shard_items=round(t_shard, 4),
                         noise_replay_last_rank=round(t_noise, 2), link_output_names=round(t_links, 2),
                         match_augmented_images=round(t_match, 2), create_json_total=round(t_json, 2),
                         pil_verify_200_real_512x704_pngs=round(t_verify200, 3),
                         pil_verify_extrapolated_13336_real_pngs=round(t_verify200 / 200 * len(items), 1)),
            files_in_output_folder=n_files, json_entries=len(body),
            shard_imbalance=dict(max_over_mean=round(max(area) / mean, 5), min_over_mean=round(min(area) / mean, 5),
                                 max_dev_pct=round(100 * max(abs(a / mean - 1) for a in area), 3)),
            per_rank=per_rank,
            json_stage_under_60s=bool(t_json + t_verify200 / 200 * len(items) < 60.0))
        print(json.dumps(res, indent=1))
        if out_json:
            json.dump(res, open(out_json, "w"), indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class LocalWorld:
    """Stand-in for torch.distributed at (rank 0, world 8) on ONE process: main() takes rank 0's shard of the 8-way plan and
    its slice of the noise stream exactly as an 8-rank job would; the gather returns this rank's vector and zeros for the
    ranks that are not running.  (The real collective: tests/test_distributed_gloo.py, bench.py SASPA_FORCE_DIST=1.)"""

    def __init__(self, world):
        self.world = world

    def get_rank(self):
        return 0

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "gloo"

    def barrier(self):
        pass

    def gather(self, t, gathered, dst=0):
        import torch
        gathered[0].copy_(t)
        for g in gathered[1:]:
            g.zero_()


def thread_cpu():
    """{tid: (thread name, CPU seconds)} of this process from /proc (utime + stime per task)."""
    tick = os.sysconf("SC_CLK_TCK")
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
            name = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tick)
        except (OSError, ValueError):
            pass
    return out


def mode_gpu(n_batches, out_json):
    import torch
    tmp = tempfile.mkdtemp(prefix="saspa_c3_")
    try:
        root = os.path.join(tmp, "ds", "data")
        materialise(root)
        prompts = prompts_file(tmp)
        pipe = R.init_pipeline("sd_v1.5", "canny", 0).to("cuda:0", torch.float16)
        # warm: kernels, allocator and the 50-step graphs of the sizes rank 0's first batches have (a production run pays each
        # capture once per size in ~1 600 batches; here it would be a tenth of the run)
        s = settings(root, prompts, MAX_BATCHES=n_batches)
        _, _, items, _ = the_plan(s)
        shard0 = R.shard_items(items, WORLD)[0]
        sizes = []
        for b in R.make_batches(shard0, BATCH)[:n_batches]:
            if (b[0].height, b[0].width) not in sizes:
                sizes.append((b[0].height, b[0].width))
        wroot = os.path.join(tmp, "warm", "data")
        SyntheticUtils(root_path=wroot, n_images=8 * len(sizes), sizes=tuple(sizes), print_func=lambda *a: None)
        sw = R.Settings(**{**s.__dict__, "NUM_PER_IMAGE": 1, "MAX_BATCHES": 0, "DATASET_KWARGS": dict(root_path=wroot, n_images=8 * len(sizes), sizes=tuple(sizes))})
        t0 = time.time()
        R.main(sw, pipe=pipe)
        t_warm = time.time() - t0
        ru0 = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
        th0 = thread_cpu()
        if os.environ.get("SASPA_STACKS_AT"):
            # diagnostics: N seconds into the run, native stacks of every thread (rocgdb attached from a child for a moment) next to
            # the per-thread CPU times so far -> which thread spins, and where
            import ctypes
            import subprocess
            import threading
            ctypes.CDLL(None).prctl(0x59616d61, ctypes.c_ulong(-1 & (2 ** 64 - 1)), 0, 0, 0)          # PR_SET_PTRACER, ANY

            def stacks():
                time.sleep(float(os.environ["SASPA_STACKS_AT"]))
                busy = sorted(((c - th0.get(t, (n, 0.0))[1], t, n) for t, (n, c) in thread_cpu().items()), reverse=True)[:4]
                with open("gpurun_out/r6_host_thread_stacks.txt", "w") as f:
                    f.write("busiest threads (cpu s since start, tid, name): " + repr(busy) + "\n")
                    f.flush()
                    subprocess.run(["rocgdb", "-p", str(os.getpid()), "-batch", "-ex", "info threads", "-ex", "thread apply all bt 14"],
                                   stdout=f, stderr=subprocess.STDOUT, timeout=240)
            threading.Thread(target=stacks, daemon=True).start()
        torch.cuda.synchronize()
        t0 = time.time()
        res = R.main(s, pipe=pipe, dist=LocalWorld(WORLD))
        torch.cuda.synchronize()
        dt = time.time() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
        th1 = thread_cpu()
        done = [it for it in res["mine"] if it.status == 1]
        n = len(done)
        # CPU seconds per thread of THIS process over the run (threads that ended in between are not listed)
        per_thread = sorted(((name, th1[t][1] - th0.get(t, (name, 0.0))[1], t == os.getpid()) for t, (name, _) in th1.items()),
                            key=lambda r: -r[1])
        threads = [dict(thread=("MAIN " if is_main else "") + name, cpu_s=round(c, 2), cpu_s_per_image=round(c / max(n, 1), 4))
                   for name, c, is_main in per_thread if c > 0.005 * dt]
        # images/s per (H, W) bucket: batches of one bucket are consecutive (make_batches sorts the buckets); a bucket's time runs
        # from the drain of the previous bucket's last batch to the drain of its own last batch
        per_bucket, prev_t, cur = [], t0, None
        for (bh, bw, cnt, t) in res["batch_log"]:
            if cur is None or cur["size"] != f"{bh}x{bw}":
                if cur is not None:
                    prev_t = cur["_t"]
                cur = dict(size=f"{bh}x{bw}", batches=0, images=0, _t0=prev_t)
                per_bucket.append(cur)
            cur["batches"] += 1
            cur["images"] += cnt
            cur["_t"] = t
        for c in per_bucket:
            span = c.pop("_t") - c.pop("_t0")
            c["seconds"] = round(span, 2)
            c["images_per_s"] = round(c["images"] / span, 3) if span > 0 else None
        cpu_self = (ru1[0].ru_utime + ru1[0].ru_stime) - (ru0[0].ru_utime + ru0[0].ru_stime)
        cpu_child = (ru1[1].ru_utime + ru1[1].ru_stime) - (ru0[1].ru_utime + ru0[1].ru_stime)
        by_size = {}
        for it in done:
            by_size[f"{it.height}x{it.width}"] = by_size.get(f"{it.height}x{it.width}", 0) + 1
        flop_equiv = sum(it.height * it.width for it in done) / (512 * 512)
        body = json.load(open(res["json_path"]))
        listed = sum(len(v) for v in body.values())
        out = dict(
            workload=f"BASELINE configs[3] rehearsal: rank 0's shard of the {len(items)}-item plan (world {WORLD}) for {res['n_batches']} batches of {BATCH} "
                     "through run_aug.main: PNG decode, device resize, Canny, 50 DDIM steps, safety checker, D2H, PNG writer "
                     "processes (outputs + _source + _control), status vector, JSON over the whole train list; synthetic weights",
            shard0_items=len(res["mine"]), batches_run=res["n_batches"], images_generated=n, sizes_generated=by_size,
            seconds=round(dt, 2), images_per_s=round(n / dt, 3),
            images_per_s_512x512_equivalent=round(flop_equiv / dt, 3),
            equivalent_note="pixel-count-scaled (conv / linear FLOPs scale with H*W; self-attention grows faster, so this understates)",
            warmup_seconds_incl_graph_captures=round(t_warm, 1),
            png_files_submitted=res["png_submitted"], png_writer_max_backlog=res["png_max_queue"], png_writer_processes=4,
            host_cpu_s_per_image=dict(main_process=round(cpu_self / n, 4), png_writer_children=round(cpu_child / n, 4)),
            main_process_threads=threads, per_bucket=per_bucket, failed_items=sum(1 for it in res["mine"] if it.status == -1),
            env={k: os.environ[k] for k in ("SASPA_FORK", "SASPA_SIDE_STREAM", "OMP_NUM_THREADS") if k in os.environ},
            json_entries=len(body), json_paths_listed=listed, json_lists_only_generated=bool(listed == n),
            dtype="bf16", data="synthetic")
        print(json.dumps(out, indent=1))
        if out_json:
            json.dump(out, open(out_json, "w"), indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ----------------------------------------------------------------------------------------------------------------------
# `hosts`: do N co-resident ranks keep their GPUs fed on the host cores they get?  (VERDICT r5 "next" #4; CPU only.)
# Every rank is a real process running the real run_aug.main on its own shard of the 13 336-item plan -- planner, noise replay,
# loader thread (PNG decode of the sources), software pipeline, PNG writer processes, status vector -- with the GPU replaced by
# a stand-in that costs the host what the MI355X path was MEASURED to cost it (profiles/r6_host_budget.json, gpu arm):
#   * the launch thread is busy for `enqueue_cpu_s` per batch (graph replays 50 x 1.4 ms + the eager pre / post launches) and
#     then sleeps until the "GPU" is done (ops.sleep_wait in the product);
#   * one extra thread per rank burns a full core for the whole run: ROCr's AsyncEventsLoop, which the rocgdb stacks show spinning
#     in hsaKmtWaitOnMultipleEvents while kernels complete at ~70 k / s;
#   * a batch of bucket (H, W) takes the measured GPU time of that bucket; images come back as photo-like pixels (smooth content +
#     sensor-like noise) so that the PNG writers work as hard as on generated pictures.
# Reported per rank: images/s against the GPU-limited rate, writer backlog, CPU seconds; and the node's total cores in use.
# ----------------------------------------------------------------------------------------------------------------------
GPU_BATCH_S = {(512, 512): 1.44, (512, 704): 1.55, (512, 768): 1.70, (512, 832): 1.90, (512, 896): 1.89}   # r6 full shard / budget runs


def _photo_like(h, w, k, cache={}):
    if (h, w, k) not in cache:
        base = synthetic_image(h, w, 900 + k).astype(np.float32)
        rs = np.random.RandomState(k)
        # low-pass structure + fine noise: PNG (zlib level Pillow's default) lands near the cost of generated photographs
        noise = rs.normal(0, 6.0, (h, w, 3)).astype(np.float32)
        cache[(h, w, k)] = np.clip(base + noise, 0, 255).astype(np.uint8)
    return cache[(h, w, k)]


def host_rank(rank, world, root, prompts, n_batches, enqueue_cpu_s, spin_thread, out_file):
    import threading

    import torch
    torch.set_num_threads(1)
    s = settings(root, prompts, MAX_BATCHES=n_batches)
    state = dict(gpu_free_at=0.0, busy=True)

    if spin_thread:
        import hashlib

        def burn():                                   # a C loop that releases the GIL, like a runtime thread would
            while state["busy"]:
                hashlib.pbkdf2_hmac("sha256", b"x", b"y", 20000)
        threading.Thread(target=burn, daemon=True).start()

    def enqueue(batch, noises, sources, subjects=None, category=None):
        t_end = time.process_time() + enqueue_cpu_s
        t0 = time.thread_time()
        while time.thread_time() - t0 < enqueue_cpu_s:   # the launch thread's own CPU work for this batch
            pass
        h, w = batch[0].height, batch[0].width
        start = max(time.time(), state["gpu_free_at"])
        state["gpu_free_at"] = start + GPU_BATCH_S.get((h, w), 1.7) * len(batch) / BATCH
        return (batch, state["gpu_free_at"])

    def finish(handle):
        batch, t_done = handle
        while time.time() < t_done:
            time.sleep(0.002)
        h, w = batch[0].height, batch[0].width
        imgs = np.stack([_photo_like(h, w, (it.order + 1) % 24) for it in batch])
        srcs = np.stack([_photo_like(h, w, it.index % 24) for it in batch])
        ctrl = (imgs > 128).astype(np.uint8) * 255
        return imgs, ctrl, srcs, None

    def gen(*a):
        return finish(enqueue(*a))[:2]
    gen.enqueue, gen.finish = enqueue, finish

    class World(LocalWorld):
        def get_rank(self):
            return rank

        def gather(self, t, gathered, dst=0):          # every rank is its own process here: rank 0 sees its own vector only
            if gathered is not None:
                super().gather(t, gathered, dst)
    ru0 = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    res = R.main(s, batch_generator=gen, dist=World(world))
    dt = time.time() - t0
    state["busy"] = False
    ru1 = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
    done = [it for it in res["mine"] if it.status == 1]
    log = res["batch_log"]
    steady = (sum(c for _, _, c, _ in log[2:]) / (log[-1][3] - log[1][3])) if len(log) > 3 else None
    gpu_s = sum(GPU_BATCH_S.get((h, w), 1.7) * c / BATCH for h, w, c, _ in log[2:])
    out = dict(rank=rank, images=len(done), seconds=round(dt, 1), images_per_s_steady=round(steady, 3) if steady else None,
               gpu_limited_images_per_s=round(sum(c for _, _, c, _ in log[2:]) / gpu_s, 3) if len(log) > 3 else None,
               png_writer_max_backlog=res["png_max_queue"],
               cpu_s_main_process=round((ru1[0].ru_utime + ru1[0].ru_stime) - (ru0[0].ru_utime + ru0[0].ru_stime), 1),
               cpu_s_png_writers=round((ru1[1].ru_utime + ru1[1].ru_stime) - (ru0[1].ru_utime + ru0[1].ru_stime), 1),
               failed=sum(1 for it in res["mine"] if it.status == -1))
    json.dump(out, open(out_file, "w"))


def mode_hosts(world, n_batches, out_json, cores=None, enqueue_cpu_s=0.15, spin_thread=True):
    import subprocess
    tmp = tempfile.mkdtemp(prefix="saspa_c3_")
    try:
        root = os.path.join(tmp, "ds", "data")
        materialise(root)
        prompts = prompts_file(tmp)
        ncpu = len(os.sched_getaffinity(0))
        cores = min(cores or ncpu, ncpu)
        cpus = sorted(os.sched_getaffinity(0))[:cores]
        procs = []
        t0 = time.time()
        for r in range(world):
            # separate output trees per rank would hide contention on one directory: all ranks write into ONE tree, as in production
            cmd = ["taskset", "-c", ",".join(map(str, cpus)), sys.executable, os.path.abspath(__file__), "_host_rank", str(r), str(world), root,
                   prompts, str(n_batches), str(enqueue_cpu_s), str(int(spin_thread)), os.path.join(tmp, f"rank{r}.json")]
            procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL if r else None))
        codes = [p.wait() for p in procs]
        wall = time.time() - t0
        ranks = [json.load(open(os.path.join(tmp, f"rank{r}.json"))) for r in range(world) if os.path.exists(os.path.join(tmp, f"rank{r}.json"))]
        tot_cpu = sum(r["cpu_s_main_process"] + r["cpu_s_png_writers"] for r in ranks)
        out = dict(
            what=f"{world} co-resident generation ranks on {cores} host cores (taskset), run_aug.main with a GPU stand-in that costs the host what the "
                 f"MI355X path was measured to cost (launch thread {enqueue_cpu_s} CPU-s per batch then asleep, "
                 + ("one runtime thread spinning a full core per rank, " if spin_thread else "") +
                 f"measured GPU seconds per batch and bucket), {n_batches} batches of {BATCH} per rank on the configs[3] plan; CPU only",
            exit_codes=codes, wall_s=round(wall, 1), cores=cores, ranks=ranks,
            cores_in_use_avg=round(tot_cpu / wall, 2),
            every_rank_keeps_its_gpu_fed=bool(ranks and all(r["images_per_s_steady"] and r["images_per_s_steady"] >= 0.97 * r["gpu_limited_images_per_s"] for r in ranks)),
            min_rate_vs_gpu_limited=round(min(r["images_per_s_steady"] / r["gpu_limited_images_per_s"] for r in ranks), 4) if ranks else None,
            max_writer_backlog=max(r["png_writer_max_backlog"] for r in ranks) if ranks else None)
        print(json.dumps(out, indent=1))
        if out_json:
            json.dump(out, open(out_json, "w"), indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "plan"
    if mode == "_host_rank":
        a = sys.argv[2:]
        host_rank(int(a[0]), int(a[1]), a[2], a[3], int(a[4]), float(a[5]), bool(int(a[6])), a[7])
    elif mode == "hosts":
        # hosts <world> <batches per rank> [out.json] [cores] [enqueue cpu s per batch] [spin thread 0|1]
        a = sys.argv[2:]
        mode_hosts(int(a[0]) if a else 8, int(a[1]) if len(a) > 1 else 24, a[2] if len(a) > 2 else None,
                   int(a[3]) if len(a) > 3 else None, float(a[4]) if len(a) > 4 else 0.15, bool(int(a[5])) if len(a) > 5 else True)
    elif mode == "plan":
        mode_plan(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        mode_gpu(int(sys.argv[2]) if len(sys.argv) > 2 else 44, sys.argv[3] if len(sys.argv) > 3 else None)
