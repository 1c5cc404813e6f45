"""One process per GPU without an external launcher: `python bench.py --gpus N` / `SASPA_GPUS=N python run_aug/run_aug.py`.

The reference runs one process on one GPU (`DEVICE = "cuda:0"`, run_aug/run_aug.py:509-516); this build shards the work
items over the GPUs of a node, one rank per GPU.  The parent that calls `launch_ranks` must NOT have touched the GPU (no
torch.cuda call, no HIP library call): it only starts N fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays rank 0's stdout and propagates failure.  Nothing here imports torch."""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_supervised(n, script, argv, fallback_env=None, max_restarts=1):
    """`launch_ranks`, plus ONE supervised restart for a generation run (round 6, advisor finding on the two-branch step
    graphs): when a rank is killed by a SIGNAL (negative return code -- e.g. the segmentation fault inside hipGraphLaunch that
    long multi-pipeline sessions showed on ROCm 7.2, profiles/r5_graph_replay_segv_backtrace.txt), every rank is started again
    as a FRESH child process with `fallback_env` (default SASPA_FORK=0: single-branch step graphs, -4.7 % at 512x512) on top
    of the environment.  A generation run is resumable by construction -- `plan_work` skips every output file that exists
    (run_aug/run_aug.py:430-432) -- so the restart continues where the dead run stopped.  Never an exec of this or of a
    GPU-holding process; an ordinary failure (positive exit code) is not retried and is returned as it is."""
    fallback_env = {"SASPA_FORK": "0"} if fallback_env is None else dict(fallback_env)
    rc = launch_ranks(n, script, argv)
    restarts = 0
    while rc < 0 and restarts < max_restarts:
        restarts += 1
        print(f"launcher: a rank died on signal {-rc}; restarting all {n} rank(s) in fresh processes with "
              f"{' '.join(f'{k}={v}' for k, v in fallback_env.items())} (restart {restarts} of {max_restarts}; finished files are skipped)",
              file=sys.stderr, flush=True)
        rc = launch_ranks(n, script, argv, extra_env=fallback_env)
    return rc


def launch_ranks(n, script, argv, extra_env=None):
    """Run `python script argv...` as ranks 0..n-1; returns the exit code (0 only if every rank exited 0).  As soon as
    one rank fails the others are terminated by PID (they would otherwise wait at the rendezvous forever)."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: required for RCCL on this driver
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, live = 0, set(range(n))
    chunks = []
    os.set_blocking(procs[0].stdout.fileno(), False)
    while live and rc == 0:
        time.sleep(0.2)
        try:
            data = procs[0].stdout.read()                        # drain so rank 0 never blocks on a full pipe
            if data:
                chunks.append(data)
        except (BlockingIOError, ValueError):
            pass
        for r in list(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    print(f"launcher: rank {r} exited with {code}", file=sys.stderr)
    if rc != 0:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    try:
        os.set_blocking(procs[0].stdout.fileno(), True)
        chunks.append(procs[0].stdout.read() or b"")
    except (OSError, ValueError):
        pass
    sys.stdout.write(b"".join(chunks).decode(errors="replace"))
    sys.stdout.flush()
    return rc
