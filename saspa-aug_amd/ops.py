"""Tensor-level wrappers over the C ABI (include/saspa_hip.h).

PyTorch is used for device memory and the current HIP stream only; every function here
enqueues hand-written gfx950 kernels through ctypes and does no arithmetic of its own.
Activations are channels-last ``[B, H, W, C]`` (or ``[M, C]`` token matrices) whose last
dim is contiguous; the pixel pitch (``stride(-2)``) may exceed C (views into wider
buffers, e.g. the q/k slices of a fused projection)."""
import os
import ctypes as C

import torch

from . import _lib

ACT_NONE, ACT_SILU = 0, 1
ACT_QUICK_GELU = 2  # saspa_activation only
ACT_GELU = 4        # saspa_activation only: erf GELU (BERT / Q-Former)
ACT_GEGLU = 3       # saspa_gemm only: fused GEGLU epilogue (bf16, weights packed by weights.pack_geglu)
ACT_RELU = 5        # saspa_gemm / saspa_activation: ReLU (before the residual add)
ACT_ADD_RELU = 6    # saspa_gemm only: ReLU AFTER the residual add (ResNet bottleneck)
GEMM_AUTO, GEMM_TILED, GEMM_WIDE, GEMM_WS, GEMM_AS = 0, 1, 2, 3, 4   # SaspaGemmParams.variant


# Optional launch recorder (bench.py / profiling only): called as recorder(kind, flops, call)
# and must return call()'s result.  `flops` is the ALGORITHMIC work of the launch (2*M*N*K
# for the implicit GEMM, 4*nq*nk*D per head for attention), used for the roofline figures; `meta` of a GEMM launch is
# (M, N, K, kh, stride, upsample, concat, has_residual, output columns): kh = 0 linear, kh < 0 = -(batch count) of a raw
# batched GEMM.
_RECORDER = None


def set_recorder(rec):
    global _RECORDER
    _RECORDER = rec


_IN_LAUNCH = [False]


def _launch(kind, flops, call, meta=None):
    if _RECORDER is None:
        return call()
    _IN_LAUNCH[0] = True
    try:
        return _RECORDER(kind, flops, call, meta)
    finally:
        _IN_LAUNCH[0] = False


# library entry points that launch nothing (host-side queries), and the clock probe (a sleeping wave used as a gate / clock
# sample, not work of the path): never recorded
_HOST_ONLY = ("eligible", "suggest", "ksplit", "which", "as_auto", "abi_version", "build_arch", "clock_probe")


class _RecordingLib:
    """What `_L()` hands out while a launch recorder is installed: every kernel-launching entry point that is NOT already wrapped
    by `_launch` (norms, element-wise passes, scheduler updates ...) is timed too, under its own name with zero FLOPs -- so that the
    recorder's table covers the whole step (bench.py `timed_path_check`), not only the MFMA kernels."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("saspa_") or any(t in name for t in _HOST_ONLY):
            return fn

        def wrapped(*a):
            rec = _RECORDER
            if rec is None or _IN_LAUNCH[0]:
                return fn(*a)
            _IN_LAUNCH[0] = True
            try:
                return rec(name[len("saspa_"):], 0.0, lambda: fn(*a), None)
            finally:
                _IN_LAUNCH[0] = False
        return wrapped


def _L():
    lib = _lib.load()
    return lib if _RECORDER is None else _RecordingLib(lib)


GEMM_FAMILY_FP8 = 8                  # recorder meta only: launches of saspa_gemm_fp8 (not a SaspaGemmParams.variant)
GEMM_FAMILY_FF_BLOCK = 16                 # saspa_ff_block (not a saspa_gemm dispatch: the launch names its family itself)
GEMM_FAMILY_NAMES = {1: "tiled_4wave", 2: "wide_8wave", 3: "wave_specialised", 4: "a_stationary", 8: "fp8_e4m3", 16: "ff_block"}


def _meta_kernel(p, meta):
    """Recorder runs only: append (kernel family, K slices) of the launch `p` describes -- the library's own dispatch, executed dry
    (saspa_gemm_which) -- to the launch's meta tuple, so that per-launch timings can be grouped by the kernel that really ran."""
    if _RECORDER is None or meta is None:
        return meta
    w = int(_L().saspa_gemm_which(C.byref(p)))
    return tuple(meta) + ((w & 0xff, w >> 8) if w > 0 else (0, 1))


def _probe_launch(kind, flops, call, meta=None):
    """A launch that the library may REFUSE without launching anything (SASPA_ERANGE on a `defer_reduce` contract it cannot
    honour): goes to the recorder only when it really ran (rc == 0) -- a refused probe used to be logged as a 2*M*N*K-FLOP
    launch of ~zero duration and the fallback's real launch was logged again."""
    if _RECORDER is None:
        return call()
    box = []

    def run():
        box.append(call())
        return box[0]
    rec = _RECORDER
    _IN_LAUNCH[0] = True
    try:
        if hasattr(rec, "conditional"):
            return rec.conditional(kind, flops, run, meta, lambda rc: rc == 0)
        return call()                # a recorder without the hook: record nothing for the probe (under-counts one launch)
    finally:
        _IN_LAUNCH[0] = False


def _dt(t):
    if t.dtype == torch.bfloat16:
        return _lib.SASPA_BF16
    if t.dtype == torch.float32:
        return _lib.SASPA_F32
    raise TypeError(f"unsupported activation dtype {t.dtype}")


_F32_GEMM_MODE = ["exact"]


class f32_gemm_mode:
    """Context: how fp32 GEMMs / convs issued inside run -- "exact" (fp32 MFMA, the parity path) or "x3" (SASPA_F32X3: three
    bf16 MFMAs per product, ~2^-17 relative; the upcast SDXL VAE).  Storage and every other kernel stay fp32."""

    def __init__(self, mode):
        if mode not in ("exact", "x3"):
            raise ValueError(mode)
        self.mode = mode

    def __enter__(self):
        self.prev = _F32_GEMM_MODE[0]
        _F32_GEMM_MODE[0] = self.mode

    def __exit__(self, *a):
        _F32_GEMM_MODE[0] = self.prev


def _gemm_dt(t):
    d = _dt(t)
    return _lib.SASPA_F32X3 if (d == _lib.SASPA_F32 and _F32_GEMM_MODE[0] == "x3") else d


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("saspa_aug_amd ops run on the GPU only (tensor is on %s)" % t.device)


_SIDE_STREAMS = {}


def side_stream(device):
    """THE second stream of a device: one per process and device, shared by every step graph's ControlNet branch (round 6) and
    by a twin launch recorder's paired region.  Each _StepGraph used to take its own stream out of torch's 32-stream
    pool; a long session that captured two-branch graphs for dozens of pipelines ended in a segmentation fault inside
    hipGraphLaunch (hip::Graph::UpdateStreams, profiles/r5_graph_replay_segv_backtrace.txt); with one shared stream the full GPU
    suite runs two-branch graphs everywhere (profiles/r6_forkall_tests.log).  SASPA_SIDE_STREAM=per_graph = the old behaviour."""
    if os.environ.get("SASPA_SIDE_STREAM", "shared") == "per_graph":
        return torch.cuda.Stream(device=device)
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def sleep_wait(event, poll_s=0.001):
    """Wait for a recorded torch.cuda.Event WITHOUT spinning a host core: hipEventSynchronize busy-waits on this stack even for
    hipEventBlockingSync events (rocr::core::BusyWaitSignal::WaitRelaxed under hip::Event::synchronize -- rocgdb stack of the
    generation loop's launch thread, profiles/r6_host_thread_stacks.txt), so the host polls hipEventQuery and sleeps in between.
    1 ms granularity against 20-35 ms sampling steps / 1-2 s batches."""
    import time
    while not event.query():
        time.sleep(poll_s)


def h2d(t, device, dtype=None):
    """Host -> device copy that does NOT block the host: through pinned memory, asynchronous in the current stream (a
    pageable-memory copy would wait for everything queued before it -- e.g. the previous batch's whole launch sequence --
    and defeat the batch pipeline of run_aug).  Device tensors pass through."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if t.is_cuda:
        return t if dtype is None or t.dtype == dtype else t.to(dtype)
    d = t.contiguous().pin_memory().to(device, non_blocking=True)
    return d if dtype is None or d.dtype == dtype else d.to(dtype)


def _pitch4(x):
    """[B,H,W,C] channels-last view -> pixel pitch; validates regular pixel rows."""
    b, h, w, c = x.shape
    if c > 1 and x.stride(3) != 1:
        raise ValueError("last dim must be contiguous")
    ld = x.stride(2)
    if ld < c:
        raise ValueError("pixel pitch smaller than the channel count")
    if h > 1 and x.stride(1) != w * ld:
        raise ValueError("rows must be densely packed at the pixel pitch")
    if b > 1 and x.stride(0) != h * w * ld:
        raise ValueError("batch images must be densely packed at the pixel pitch")
    return ld


def round8(n):
    return (n + 7) // 8 * 8


_TWIN = [False]


class twin_branch:
    """Context: the launches issued inside run on ONE of two concurrent graph branches that walk the same layer shapes at the
    same time (UNet encoder || ControlNet encoder, pipeline._StepGraph).  The library sizes a launch for the whole chip; with
    a twin beside it half the chip is what it gets, so the launches carry SaspaGemmParams.sharing = 1 (ABI 15): tile choice
    and split-K are made for 128 CUs -- the convs / projections of the 32x32 level (128 wide tiles) run un-split next to
    their twin (no fp32 slabs, no reduce launch), the 16x16 level takes two K slices instead of four.  SASPA_TWIN_KS=0 turns
    the hint off (A/B knob).  Results differ from the single-branch dispatch only by the summation order of K."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _TWIN[0]
        _TWIN[0] = self.on and os.environ.get("SASPA_TWIN_KS", "1") != "0"

    def __exit__(self, *a):
        _TWIN[0] = self.prev


def _set_splitk(p, m, n, k, t, force=None):
    """Split-K factor from the library's own heuristic (saspa_gemm_suggest_ksplit: every other field of p is already
    filled in) or the caller's override; allocates the fp32 slab workspace."""
    p.ksplit, p.workspace = 1, None
    p.sharing = 1 if _TWIN[0] else 0
    ks = _L().saspa_gemm_suggest_ksplit(C.byref(p)) if force is None else int(force)
    if ks > 1:
        ws = torch.empty((ks * m * n,), device=t.device, dtype=torch.float32)
        p.ksplit, p.workspace = ks, C.c_void_p(ws.data_ptr())
        return ws           # keep alive until the launch is enqueued
    p.ksplit, p.workspace = 1, None
    return None


def _as_over_stats():
    return os.environ.get("SASPA_GEMM_AS_OVER_STATS", "1") != "0"


def splitk_gn_enabled():
    """SASPA_SPLITK_GN=0: conv1 -> norm2 of the small levels runs as conv (+ reduce) and GroupNorm launches again (A/B knob)."""
    return os.environ.get("SASPA_SPLITK_GN", "1") != "0"


def gn_onepass_enabled():
    """SASPA_GN_ONEPASS=0: small-image GroupNorms run as statistics pass + apply pass again (A/B knob)."""
    return os.environ.get("SASPA_GN_ONEPASS", "1") != "0"


def gn_fusion_enabled():
    """SASPA_GN_FUSE=0: every GroupNorm runs its own statistics pass (A/B knob for the epilogue statistics)."""
    return os.environ.get("SASPA_GN_FUSE", "1") != "0"


def _gn_stats_for(p, out, gn_unit, b, hw, n):
    """Epilogue GroupNorm statistics (SaspaGemmParams.gn_stats): allocate the [rows / 128, N / unit, 2] fp32 buffer, hang it
    on the output tensor (`saspa_gn` = (stats, unit): `groupnorm` picks it up and skips its statistics pass) -- when the
    shape allows it: bf16, whole 128-row blocks per image, whole 160-column tiles, dense output rows."""
    if hasattr(out, "saspa_gn"):
        del out.saspa_gn                  # a reused `out=`: statistics of an earlier launch must not outlive this one
    if not gn_unit or not gn_fusion_enabled() or out.dtype != torch.bfloat16:
        return None
    if hw % 128 or n % 160 or 80 % gn_unit or gn_unit % 2 or gn_unit > 16 or out.shape[-1] != n or p.ldo % 8 or (p.residual and p.ldr % 8):
        return None
    stats = torch.empty((b * hw // 128, n // gn_unit, 2), device=out.device, dtype=torch.float32)
    p.gn_stats, p.gn_unit = _ptr(stats), int(gn_unit)
    # (statistics, unit, the buffer and tensor version they describe): `groupnorm` re-checks the last two, so a tensor that
    # was re-pointed or written in place through torch since the launch falls back to its own statistics pass
    out.saspa_gn = (stats, int(gn_unit), out.data_ptr(), out._version)
    return stats


def conv(x, w, bias=None, *, kh=1, kw=1, stride=1, pad=0, upsample=False, x2=None, rowvec=None,
         residual=None, alpha=1.0, act=ACT_NONE, out=None, n_out=None, variant=0, korder=None, ksplit=None, out_hw=None,
         gn_unit=None, fuse_gn=None):
    """Implicit-GEMM conv of channels-last ``x`` (optionally channel-concatenated with
    ``x2``) with packed weights ``w`` [N, kh*kw*(C0+C1)].  Returns [B, Ho, Wo, round8(N)]
    (pad channels zero).  ``korder``: K order the weights were packed in (default: the tensor's
    ``saspa_korder`` attribute set by the packer, else tap-major).  ``gn_unit``: the output feeds a GroupNorm whose
    groups are multiples of this many channels -> the epilogue leaves its statistics (`_gn_stats_for`).
    ``fuse_gn`` = (gamma, beta, groups, eps, act): the conv's ONLY consumer is this GroupNorm (ResnetBlock2D.conv1 -> norm2 ->
    SiLU), so the call returns the NORMALISED tensor: where the conv runs on K slices and an image's group fits one workgroup
    (the 8x8 / 16x16 levels) the reduce launch, the statistics and the apply pass are one launch (saspa_splitk_groupnorm, ABI
    18: the un-normalised output is never written); otherwise the conv runs as usual and `groupnorm` follows."""
    _check_dev(x, w, bias, x2, rowvec, residual, out)
    lib = _L()
    b, h, wd, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[3]
    hv, wv = (2 * h, 2 * wd) if upsample else (h, wd)
    ho = (hv + 2 * pad - kh) // stride + 1
    wo = (wv + 2 * pad - kw) // stride + 1
    if out_hw is not None:
        # asymmetric zero padding (Downsample2D(padding=0) of the VAE encoder pads right / bottom only): windows may hang
        # over the far edges, where the loaders' bounds masks read zeros; the C ABI checks the window origins
        ho, wo = out_hw
    n = w.shape[0] if n_out is None else n_out
    # geometry is validated HERE: the kernel trusts it (a mismatched skip / residual would be read out of bounds)
    if x2 is not None and (tuple(x2.shape[:3]) != (b, h, wd) or x2.dtype != x.dtype):
        raise ValueError(f"concat source {tuple(x2.shape)} does not match {tuple(x.shape)} (batch / height / width / dtype)")
    if w.shape[1] < kh * kw * (c0 + c1) or w.dtype != x.dtype:
        raise ValueError(f"weights {tuple(w.shape)} {w.dtype} do not fit a {kh}x{kw} window over {c0}+{c1} channels of {x.dtype}")
    if ho <= 0 or wo <= 0:
        raise ValueError("empty convolution output")
    for name, t in (("residual", residual), ("out", out)):
        if t is not None and (t.dim() != 4 or tuple(t.shape[:3]) != (b, ho, wo) or t.shape[3] < n or t.dtype != x.dtype):
            raise ValueError(f"{name} {tuple(t.shape)} {t.dtype} does not match the output [{b},{ho},{wo},>={n}] {x.dtype}")
    if bias is not None and bias.numel() < n:
        raise ValueError("bias shorter than N")
    if rowvec is not None and (rowvec.shape[-1] < n or (rowvec.dim() == 2 and rowvec.shape[0] not in (1, b))):
        raise ValueError(f"rowvec {tuple(rowvec.shape)} must be [N] or [batch, N]")
    if out is None:
        nc = round8(n)
        out = (torch.zeros if nc != n else torch.empty)((b, ho, wo, nc), device=x.device, dtype=x.dtype)
    p = _lib.GemmParams()
    p.dtype = _gemm_dt(x)
    p.a0, p.a1 = _ptr(x), _ptr(x2)
    p.c0, p.c1 = c0, c1
    p.lda0 = _pitch4(x)
    p.lda1 = 0 if x2 is None else _pitch4(x2)
    p.batch, p.hin, p.win, p.hout, p.wout = b, h, wd, ho, wo
    p.kh, p.kw, p.stride, p.pad, p.upsample = kh, kw, stride, pad, int(upsample)
    p.w, p.ldw = _ptr(w), w.stride(0)
    p.M, p.N, p.K = b * ho * wo, n, kh * kw * (c0 + c1)
    p.bias = _ptr(bias)
    p.rowvec = _ptr(rowvec)
    p.ldrv = 0 if (rowvec is None or rowvec.dim() == 1 or rowvec.shape[0] == 1) else rowvec.stride(0)
    p.residual = _ptr(residual)
    p.ldr = 0 if residual is None else _pitch4(residual)
    p.alpha, p.act = float(alpha), int(act)
    p.out, p.ldo = _ptr(out), _pitch4(out)
    p.nb1 = p.nb2 = 1
    p.variant = int(variant)
    p.korder = int(getattr(w, "saspa_korder", 0)) if korder is None else int(korder)
    if getattr(w, "saspa_wsplit", 0):
        # weights.presplit_x3 (ABI 20): [hi | lo] bf16 pairs per K-tile -- only the SASPA_F32X3 loop can read them
        if p.dtype != _lib.SASPA_F32X3:
            raise RuntimeError("pre-split (SASPA_F32X3) weights used outside ops.f32_gemm_mode('x3')")
        p.w_split = 1
    # a level-0 pointwise layer the A-stationary kernel takes on whole rounds (Transformer2DModel.proj_out): that kernel has no
    # statistics epilogue, and A-stationary + the consumer's own statistics pass (27 + 12 us) beats the tiled kernel with the
    # statistics in its epilogue (53 us); SASPA_GEMM_AS_OVER_STATS=0 keeps the statistics
    if gn_unit and kh == 1 and kw == 1 and c1 == 0 and x.dtype == torch.bfloat16 and _as_over_stats() and \
            lib.saspa_gemm_as_auto(C.byref(p)) == 1:
        gn_unit = None
    meta = (p.M, p.N, p.K, kh, stride, int(upsample), c1 > 0, residual is not None, n // 2 if act == ACT_GEGLU else n)
    if fuse_gn is not None:
        gamma, beta, groups, eps, gact = fuse_gn
        if splitk_gn_enabled() and residual is None and act == ACT_NONE and out.shape[-1] == n:
            _ws = _set_splitk(p, p.M, p.N, p.K, x, ksplit)          # the heuristic WITHOUT epilogue statistics: nobody reads them
            q = _lib.GroupNormParams()
            q.dtype = _dt(out)
            q.x0, q.x1, q.c0, q.c1 = _ptr(out), None, n, 0
            q.ldx0, q.ldx1 = _pitch4(out), 0
            q.batch, q.hw, q.groups, q.eps = b, ho * wo, int(groups), float(eps)
            q.gamma, q.beta = _ptr(gamma), _ptr(beta)
            q.partial, q.nsplit, q.scale_shift = None, 0, None
            q.act, q.y, q.ldy = int(gact), _ptr(out), _pitch4(out)
            if p.ksplit > 1 and lib.saspa_splitk_groupnorm_eligible(C.byref(p), C.byref(q)):
                p.defer_reduce = 1
                # SASPA_ERANGE: the dispatch would run this problem on ONE slice / a kernel without slabs (a variant pin, an A/B
                # knob): nothing was launched, fall back to conv + groupnorm below
                rc = _probe_launch("gemm", 2.0 * p.M * p.N * p.K, lambda: lib.saspa_gemm(C.byref(p), _stream()), _meta_kernel(p, meta))
                if rc == 0:
                    _launch("splitk_gn", 0.0, lambda: _lib.check(lib.saspa_splitk_groupnorm(C.byref(p), C.byref(q), _stream()), "saspa_splitk_groupnorm"))
                    return out
                if rc != _lib.SASPA_ERANGE:
                    _lib.check(rc, "saspa_gemm(conv, deferred reduce)")
            p.ksplit, p.workspace, p.defer_reduce = 1, None, 0
        h = conv(x, w, bias, kh=kh, kw=kw, stride=stride, pad=pad, upsample=upsample, x2=x2, rowvec=rowvec, residual=residual, alpha=alpha,
                 act=act, out=out, n_out=n_out, variant=variant, korder=korder, ksplit=ksplit, out_hw=out_hw, gn_unit=gn_unit)
        return groupnorm(h, gamma, beta, groups, eps, gact)
    _gs = _gn_stats_for(p, out, gn_unit, b, ho * wo, n)  # noqa: F841   (set BEFORE the split-K heuristic looks at p)
    _ws = _set_splitk(p, p.M, p.N, p.K, x, ksplit)  # noqa: F841   (ksplit: tuning override of the heuristic)
    _launch("gemm", 2.0 * p.M * p.N * p.K, lambda: _lib.check(lib.saspa_gemm(C.byref(p), _stream()), "saspa_gemm(conv)"), _meta_kernel(p, meta))
    return out


def _gn_source_stats(x, x2, groups):
    """Epilogue statistics of the producers of x (| x2), as `groupnorm` would use them: (stats0, stats1, unit) or None."""
    def _fresh(t):
        g = getattr(t, "saspa_gn", None) if t is not None else None
        if g is None or g[2] != t.data_ptr() or g[3] != t._version or g[0].shape[0] * 128 != t.shape[0] * t.shape[1] * t.shape[2]:
            return None
        return g
    b, h, w, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[3]
    g0, g1 = _fresh(x), _fresh(x2)
    ok = (g0 is not None and (x2 is None or g1 is not None) and gn_fusion_enabled() and (h * w) % 128 == 0
          and (x2 is None or g1[1] == g0[1]) and c0 % g0[1] == 0 and c1 % g0[1] == 0 and ((c0 + c1) // groups) % g0[1] == 0
          and g0[0].shape[1] * g0[1] == c0 and (x2 is None or g1[0].shape[1] * g1[1] == c1))
    return (g0[0], g1[0] if g1 is not None else None, g0[1]) if ok else None


def conv_gn(x, gn, w32, bias=None, *, x2=None, rowvec=None, residual=None, alpha=1.0, out=None, ksplit=None, gn_unit=None,
            defer_to=None):
    """out = conv3x3(act(GroupNorm(cat(x, x2)))) in ONE launch (saspa_conv3x3_halo, ABI 19): the GroupNorm (+ SiLU) is applied
    to the conv's input tile in LDS -- ResnetBlock2D's norm -> SiLU -> conv.  gn = (gamma_beta32, groups, eps, act) or None
    (plain halo conv of an already normalised x); w32: weights packed SASPA_KORDER_CHUNK32 (weights.to_chunk32_major).
    Returns None when the launch is not eligible (the caller falls back to groupnorm + conv).  Statistics of x come from its
    producer's epilogue when it left them (`saspa_gn`), else from a statistics pass launched here.  defer_to = (gamma, beta,
    groups, eps, act): the conv's only consumer is that GroupNorm and the conv runs on K slices -> saspa_splitk_groupnorm sums the
    slabs and normalises (as ops.conv(fuse_gn=...))."""
    _check_dev(x, w32, bias, x2, rowvec, residual, out)
    if x.dtype != torch.bfloat16:
        return None
    lib = _L()
    b, h, wd, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[3]
    n = w32.shape[0]
    if x2 is not None and (tuple(x2.shape[:3]) != (b, h, wd) or x2.dtype != x.dtype):
        raise ValueError(f"concat source {tuple(x2.shape)} does not match {tuple(x.shape)} (batch / height / width / dtype)")
    if w32.shape[1] < 9 * (c0 + c1) or w32.dtype != x.dtype:
        raise ValueError(f"weights {tuple(w32.shape)} {w32.dtype} do not fit a 3x3 window over {c0}+{c1} channels of {x.dtype}")
    for name, t in (("residual", residual), ("out", out)):
        if t is not None and (t.dim() != 4 or tuple(t.shape[:3]) != (b, h, wd) or t.shape[3] < n or t.dtype != x.dtype):
            raise ValueError(f"{name} {tuple(t.shape)} {t.dtype} does not match the output [{b},{h},{wd},>={n}] {x.dtype}")
    if bias is not None and bias.numel() < n:
        raise ValueError("bias shorter than N")
    if rowvec is not None and (rowvec.shape[-1] < n or (rowvec.dim() == 2 and rowvec.shape[0] not in (1, b))):
        raise ValueError(f"rowvec {tuple(rowvec.shape)} must be [N] or [batch, N]")
    p = _lib.GemmParams()
    p.dtype = _lib.SASPA_BF16
    p.a0, p.a1, p.c0, p.c1 = _ptr(x), _ptr(x2), c0, c1
    p.lda0 = _pitch4(x)
    p.lda1 = 0 if x2 is None else _pitch4(x2)
    p.batch, p.hin, p.win, p.hout, p.wout = b, h, wd, h, wd
    p.kh, p.kw, p.stride, p.pad, p.upsample = 3, 3, 1, 1, 0
    p.w, p.ldw = _ptr(w32), w32.stride(0)
    p.M, p.N, p.K = b * h * wd, n, 9 * (c0 + c1)
    p.bias, p.rowvec = _ptr(bias), _ptr(rowvec)
    p.ldrv = 0 if (rowvec is None or rowvec.dim() == 1 or rowvec.shape[0] == 1) else rowvec.stride(0)
    p.residual = _ptr(residual)
    p.ldr = 0 if residual is None else _pitch4(residual)
    p.alpha, p.act = float(alpha), ACT_NONE
    p.nb1 = p.nb2 = 1
    p.korder = 2
    q = None
    keep = []
    if gn is not None:
        gb32, groups, eps, gact = gn
        q = _lib.ConvGnParams()
        q.gamma_beta32, q.groups, q.eps, q.act = _ptr(gb32), int(groups), float(eps), int(gact)
        st = _gn_source_stats(x, x2, groups)
        if st is not None:
            q.stats0, q.stats1, q.unit = _ptr(st[0]), _ptr(st[1]), int(st[2])
            keep.append(st)
    # eligibility is decided BEFORE any allocation / statistics launch (out / ldo of a would-be output: dense rows of round8(N))
    p.out, p.ldo = _ptr(x), round8(n) if out is None else _pitch4(out)
    if q is not None and not q.stats0:
        q.partial, q.nsplit = _ptr(x), 1            # placeholders for the eligibility check
    if not lib.saspa_conv3x3_halo_eligible(C.byref(p), C.byref(q) if q is not None else None):
        return None
    if q is not None and not q.stats0:
        # no epilogue statistics on the input: the statistics pass of `groupnorm`, then the fused apply + conv
        ctot = c0 + c1
        nsplit = _gn_nsplit(b, h * wd, ctot // 8)
        partial = torch.empty((b * nsplit * ctot * 2,), device=x.device, dtype=torch.float32)
        g = _lib.GroupNormParams()
        g.dtype = _dt(x)
        g.x0, g.x1, g.c0, g.c1 = _ptr(x), _ptr(x2), c0, c1
        g.ldx0, g.ldx1 = _pitch4(x), (0 if x2 is None else _pitch4(x2))
        g.batch, g.hw, g.groups, g.eps = b, h * wd, int(groups), float(eps)
        g.gamma, g.beta = _ptr(gb32), _ptr(gb32)       # not read by the statistics pass
        g.partial, g.nsplit, g.scale_shift = _ptr(partial), nsplit, None
        g.act, g.y, g.ldy = 0, None, 0
        _launch("gn_stats", 0.0, lambda: _lib.check(lib.saspa_groupnorm_stats(C.byref(g), _stream()), "saspa_groupnorm_stats"))
        q.partial, q.nsplit = _ptr(partial), nsplit
        keep.append(partial)
    if out is None:
        nc = round8(n)
        out = (torch.zeros if nc != n else torch.empty)((b, h, wd, nc), device=x.device, dtype=x.dtype)
    p.out, p.ldo = _ptr(out), _pitch4(out)
    meta = (p.M, p.N, p.K, 3, 1, 0, c1 > 0, residual is not None, n)
    flops = 2.0 * p.M * p.N * p.K

    def _split(force):
        p.ksplit, p.workspace = 1, None
        p.sharing = 1 if _TWIN[0] else 0
        p.korder = 1                                  # the heuristic knows the im2col kernels' orders only; same K
        ks = lib.saspa_gemm_suggest_ksplit(C.byref(p)) if force is None else int(force)
        p.korder = 2
        ks = lib.saspa_conv3x3_halo_ksplit(C.byref(p), ks)
        if ks > 1:
            ws = torch.empty((ks * p.M * p.N,), device=x.device, dtype=torch.float32)
            p.ksplit, p.workspace = ks, C.c_void_p(ws.data_ptr())
            return ws
        return None

    qq = C.byref(q) if q is not None else None
    if defer_to is not None:
        gamma, beta, groups2, eps2, gact2 = defer_to
        if splitk_gn_enabled() and residual is None and out.shape[-1] == n:
            _ws = _split(ksplit)
            g2 = _lib.GroupNormParams()
            g2.dtype = _dt(out)
            g2.x0, g2.x1, g2.c0, g2.c1 = _ptr(out), None, n, 0
            g2.ldx0, g2.ldx1 = _pitch4(out), 0
            g2.batch, g2.hw, g2.groups, g2.eps = b, h * wd, int(groups2), float(eps2)
            g2.gamma, g2.beta = _ptr(gamma), _ptr(beta)
            g2.partial, g2.nsplit, g2.scale_shift = None, 0, None
            g2.act, g2.y, g2.ldy = int(gact2), _ptr(out), _pitch4(out)
            if p.ksplit > 1 and lib.saspa_splitk_groupnorm_eligible(C.byref(p), C.byref(g2)):
                p.defer_reduce = 1
                rc = _probe_launch("gemm", flops, lambda: lib.saspa_conv3x3_halo(C.byref(p), qq, _stream()), meta)
                if rc == 0:
                    _launch("splitk_gn", 0.0, lambda: _lib.check(lib.saspa_splitk_groupnorm(C.byref(p), C.byref(g2), _stream()), "saspa_splitk_groupnorm"))
                    return out
                if rc != _lib.SASPA_ERANGE:
                    _lib.check(rc, "saspa_conv3x3_halo(deferred reduce)")
            p.ksplit, p.workspace, p.defer_reduce = 1, None, 0
        _gs = _gn_stats_for(p, out, gn_unit, b, h * wd, n)  # noqa: F841
        _ws = _split(ksplit)  # noqa: F841
        _launch("gemm", flops, lambda: _lib.check(lib.saspa_conv3x3_halo(C.byref(p), qq, _stream()), "saspa_conv3x3_halo"), meta)
        return groupnorm(out, gamma, beta, groups2, eps2, gact2)
    _gs = _gn_stats_for(p, out, gn_unit, b, h * wd, n)  # noqa: F841
    _ws = _split(ksplit)  # noqa: F841
    _launch("gemm", flops, lambda: _lib.check(lib.saspa_conv3x3_halo(C.byref(p), qq, _stream()), "saspa_conv3x3_halo"), meta)
    return out


def _linear_params(x2, w, bias, r2, o2, alpha, act, rowvec, variant, m, n, k):
    p = _lib.GemmParams()
    p.dtype = _gemm_dt(x2)
    p.a0, p.a1, p.c0, p.c1 = _ptr(x2), None, k, 0
    p.lda0, p.lda1 = (x2.stride(0) if m > 1 else max(k, x2.stride(0))), 0
    p.batch, p.hin, p.win, p.hout, p.wout = 1, m, 1, m, 1
    p.kh = p.kw = p.stride = 1
    p.pad = p.upsample = 0
    p.w, p.ldw = _ptr(w), w.stride(0)
    p.M, p.N, p.K = m, n, k
    p.bias, p.rowvec, p.ldrv = _ptr(bias), _ptr(rowvec), 0
    p.residual = _ptr(r2)
    p.ldr = 0 if r2 is None else (r2.stride(0) if m > 1 else max(n, r2.stride(0)))
    p.alpha, p.act = float(alpha), int(act)
    p.out = _ptr(o2)
    p.ldo = o2.stride(0) if m > 1 else max(o2.shape[-1], o2.stride(0))
    p.nb1 = p.nb2 = 1
    p.variant = int(variant)
    return p


def linear_ln_fusable(x, w, *, act=ACT_NONE, n_out=None):
    """Non-zero if `linear(x, w, ..., ln=...)` / `out_t=` can run: the A-stationary kernel takes the problem
    (saspa_gemm_as_eligible: bf16, K = 320, N % 64 == 0, >= 192 blocks of 256 rows; SASPA_GEMM_AS=0 turns it off).
    2 = whole rounds of row blocks (the kernel wins on every layer shape), 1 = a ragged last round."""
    if not x.is_cuda or x.dtype != torch.bfloat16:
        return 0
    k = x.shape[-1]
    x2 = x.reshape(-1, k) if x.dim() != 2 else x
    m, n = x2.shape[0], w.shape[0]
    nc = (n // 2 if act == ACT_GEGLU else n) if n_out is None else n_out
    p = _linear_params(x2, w, None, None, x2, 1.0, act, None, 0, m, n, k)
    p.ldo = round8(nc)                    # the output a call would allocate (only its pitch / alignment are looked at)
    return int(_L().saspa_gemm_as_eligible(C.byref(p)))


def linear(x, w, bias=None, *, residual=None, alpha=1.0, act=ACT_NONE, out=None, rowvec=None, variant=0, ksplit=None,
           ln=None, out_t=None, n_split=None, rows_per_batch=None):
    """x: [..., K] (last dim contiguous, uniform row pitch) @ w[N, K]^T -> [..., round8(N)].
    ln = (gamma, beta, eps): LayerNorm over K applied to x inside the launch (SaspaGemmParams.ln_gamma: the A-stationary
    kernel only -- check `linear_ln_fusable` first; the library returns ERANGE otherwise).
    out_t [B, N - n_split, ld] with n_split, rows_per_batch: output columns >= n_split are written transposed per batch of
    rows_per_batch rows (the V^T operand of flash_attn next to Q | K); the returned tensor then has n_split columns."""
    _check_dev(x, w, bias, residual, out, out_t)
    lib = _L()
    k = x.shape[-1]
    x2 = x.reshape(-1, k) if x.dim() != 2 else x
    m = x2.shape[0]
    n = w.shape[0]
    if out is None:
        nc = round8(n // 2 if act == ACT_GEGLU else (n if out_t is None else n_split))
        out = (torch.zeros if (nc != n and act != ACT_GEGLU and out_t is None) else torch.empty)((m, nc), device=x.device, dtype=x.dtype)
    o2 = out.view(-1, out.shape[-1]) if out.dim() != 2 else out
    r2 = None if residual is None else (residual.reshape(-1, residual.shape[-1]) if residual.dim() != 2 else residual)
    p = _linear_params(x2, w, bias, r2, o2, alpha, act, rowvec, variant, m, n, k)
    if ln is not None:
        g, b_, eps = ln
        _check_dev(g, b_)
        if g.dtype != torch.float32 or b_.dtype != torch.float32 or g.numel() < k or b_.numel() < k:
            raise ValueError("LayerNorm gamma / beta must be fp32 vectors of K elements")
        p.ln_gamma, p.ln_beta, p.ln_eps = _ptr(g), _ptr(b_), float(eps)
    if out_t is not None:
        if out_t.dim() != 3 or out_t.stride(2) != 1 or out_t.dtype != x.dtype or n_split is None or not rows_per_batch:
            raise ValueError("out_t must be [batch, N - n_split, ld] in the input dtype, with n_split and rows_per_batch given")
        if out_t.shape[0] * rows_per_batch != m or out_t.shape[1] != n - n_split or out_t.shape[2] < rows_per_batch:
            raise ValueError(f"out_t {tuple(out_t.shape)} does not hold {n - n_split} transposed columns of {m} rows in batches of {rows_per_batch}")
        p.out_t, p.ldt, p.st = _ptr(out_t), out_t.stride(1), out_t.stride(0)
        p.n_split, p.rows_per_batch = int(n_split), int(rows_per_batch)
    _ws = _set_splitk(p, m, n, k, x, ksplit) if (act != ACT_GEGLU and ln is None and out_t is None) else None  # noqa: F841
    if act == ACT_GEGLU:
        p.ksplit, p.workspace = 1, None
    _launch("gemm", 2.0 * m * n * k, lambda: _lib.check(lib.saspa_gemm(C.byref(p), _stream()), "saspa_gemm(linear)"),
            _meta_kernel(p, (m, n, k, 0, 1, 0, False, residual is not None, n // 2 if act == ACT_GEGLU else n)))
    if x.dim() != 2 and out.dim() == 2:
        return out.reshape(*x.shape[:-1], out.shape[-1])
    return out


def xattn_block(x, ln, w, bias, kf, vf, nk, rows_per_sample, residual=None, out=None):
    """The cross-attention half of a level-0 transformer block in one launch (saspa_xattn_block, include/saspa_hip.h):
    out = residual + to_out(softmax(to_q(LayerNorm(x)) K^T) V) + bias.  x [..., 320] bf16 (uniform row pitch), ln = (gamma, beta,
    eps), w / bias from weights.pack_xattn_w, kf / vf [B, ...] from weights.xattn_kv_fragments, residual defaults to x."""
    _check_dev(x, w, bias, kf, vf, residual, out)
    lib = _L()
    c = x.shape[-1]
    x2 = x.reshape(-1, c) if x.dim() != 2 else x
    m = x2.shape[0]
    if x.dtype != torch.bfloat16 or c != 320:
        raise ValueError("xattn_block: bf16 rows of 320 channels")
    r2 = x2 if residual is None else (residual.reshape(-1, c) if residual.dim() != 2 else residual)
    if out is None:
        out = torch.empty((m, c), device=x.device, dtype=x.dtype)
    o2 = out.view(-1, c) if out.dim() != 2 else out
    g, b_, eps = ln
    p = _lib.XattnBlockParams()
    p.x, p.ldx = _ptr(x2), x2.stride(0)
    p.residual, p.ldr = _ptr(r2), r2.stride(0)
    p.M, p.rows_per_sample = m, int(rows_per_sample)
    p.ln_gamma, p.ln_beta, p.ln_eps = _ptr(g), _ptr(b_), float(eps)
    p.w, p.ldw, p.bias = _ptr(w), w.stride(0), _ptr(bias)
    p.kf, p.kf_stride = _ptr(kf), kf.stride(0) * kf.element_size()
    p.vf, p.vf_stride = _ptr(vf), vf.stride(0) * vf.element_size()
    p.nk = int(nk)
    p.out, p.ldo = _ptr(o2), o2.stride(0)
    if kf.shape[0] * rows_per_sample != m or vf.shape[0] != kf.shape[0]:
        raise ValueError(f"xattn_block: {kf.shape[0]} samples of {rows_per_sample} rows do not make {m} rows")
    # algorithmic work: to_q + to_out (2 x 2 M C^2) and the two attention products over the nk keys (2 x 2 M nk C)
    flops = 4.0 * m * c * c + 4.0 * m * nk * c
    _launch("gemm", flops, lambda: _lib.check(lib.saspa_xattn_block(C.byref(p), _stream()), "saspa_xattn_block"),
            (m, 2 * c + 2 * nk, c, 0, 1, 0, False, True, c))
    if x.dim() != 2 and out.dim() == 2:
        return out.reshape(*x.shape[:-1], c)
    return out


def gemm_batched(a, lda, sa, w, ldw, sw, out, ldo, so, m, n, k, nb1, nb2, alpha=1.0, residual=None, ldr=0, act=ACT_NONE):
    """Raw batched GEMM out[z] = act(alpha * a[z] @ w[z]^T) (+ residual[z], same batch strides as out); s* = (stride1,
    stride2) in elements.  ``a``/``w``/``out``/``residual`` are tensors whose data_ptr is the z=0 origin."""
    _check_dev(a, w, out, residual)
    lib = _L()
    p = _lib.GemmParams()
    p.dtype = _gemm_dt(a)
    p.a0, p.a1, p.c0, p.c1, p.lda0, p.lda1 = _ptr(a), None, k, 0, lda, 0
    p.batch, p.hin, p.win, p.hout, p.wout = 1, m, 1, m, 1
    p.kh = p.kw = p.stride = 1
    p.pad = p.upsample = 0
    p.w, p.ldw = _ptr(w), ldw
    p.M, p.N, p.K = m, n, k
    p.bias = p.rowvec = None
    p.residual, p.ldr = _ptr(residual), int(ldr)
    p.ldrv = 0
    p.alpha, p.act = float(alpha), int(act)
    p.out, p.ldo = _ptr(out), ldo
    p.nb1, p.nb2 = nb1, nb2
    p.sa1, p.sa2 = sa
    p.sw1, p.sw2 = sw
    p.so1, p.so2 = so
    p.ksplit, p.workspace = 1, None
    _launch("gemm", 2.0 * m * n * k * nb1 * nb2,
            lambda: _lib.check(lib.saspa_gemm(C.byref(p), _stream()), "saspa_gemm(batched)"),
            _meta_kernel(p, (m, n, k, -nb1 * nb2, 1, 0, False, residual is not None, n)))
    return out


def _ff_params(x, ln, w1p, b1p, w2f, b2, residual, out):
    c = x.shape[-1]
    x2 = x.reshape(-1, c) if x.dim() != 2 else x
    m = x2.shape[0]
    r2 = x2 if residual is None else (residual.reshape(-1, c) if residual.dim() != 2 else residual)
    p = _lib.FfBlockParams()
    p.x, p.ldx, p.residual, p.ldr, p.M, p.F = _ptr(x2), x2.stride(0), _ptr(r2), r2.stride(0), m, w2f.shape[0] * 32
    if ln is not None:
        p.ln_gamma, p.ln_beta, p.ln_eps = _ptr(ln[0]), _ptr(ln[1]), float(ln[2])
    p.w1, p.ldw1, p.b1, p.w2f, p.b2 = _ptr(w1p), w1p.stride(0), _ptr(b1p), _ptr(w2f), _ptr(b2)
    if out is not None:
        o2 = out.reshape(-1, c) if out.dim() != 2 else out
        p.out, p.ldo = _ptr(o2), o2.stride(0)
    return p, x2, r2, m, c


def ff_block_eligible(x, w1p, w2f):
    """Can `ff_block` take these tokens (bf16, 320 channels, rows % 128 == 0 ...)?  Host-side, launches nothing."""
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.shape[-1] != 320 or w1p.dtype != torch.bfloat16 or w2f.dtype != torch.bfloat16:
        return False
    p, x2, _, m, _ = _ff_params(x, None, w1p, w1p, w2f, w1p, None, None)
    p.out, p.ldo = p.x, p.ldx
    return bool(_lib.load().saspa_ff_block_eligible(C.byref(p)))


def ff_block(x, ln, w1p, b1p, w2f, b2, residual=None, out=None):
    """The feed-forward half of a level-0 transformer block in one launch (saspa_ff_block, include/saspa_hip.h):
    out = residual + W2 (v * gelu(g)) + b2, [v ; g] = W1 LayerNorm(x) + b1.  x [..., 320] bf16 (uniform row pitch), ln = (gamma, beta,
    eps) or None, w1p / b1p / w2f / b2 from weights.pack_ff_block (bf16 / fp32 / bf16 / fp32), residual defaults to x."""
    _check_dev(x, w1p, b1p, w2f, b2, residual, out)
    if x.dtype != torch.bfloat16 or w1p.dtype != torch.bfloat16 or w2f.dtype != torch.bfloat16:
        raise TypeError("ff_block: bf16 activations / weights")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    p, x2, r2, m, c = _ff_params(x, ln, w1p, b1p, w2f, b2, residual, out)
    f = p.F
    # algorithmic work: the two projections (2 M C 2F + 2 M F C)
    flops = 2.0 * m * c * 2 * f + 2.0 * m * f * c
    _launch("gemm", flops, lambda: _lib.check(_L().saspa_ff_block(C.byref(p), _stream()), "saspa_ff_block"),
            (m, 3 * f, c, 0, 1, 0, False, True, c) + ((GEMM_FAMILY_FF_BLOCK, 1) if _RECORDER is not None else ()))
    return out


def flash_attn(q, k, vt, out, heads, d, nq, nk, scale, causal=False, prescaled=False, v_rowmajor=False):
    """q: [B, nq, >=heads*d] view, k: [B, nk, >=heads*d] view, vt: [B, heads*d, ldvt] (keys
    contiguous), out: [B, nq, >=heads*d] view.  bf16 only.  prescaled: q already carries scale * log2(e)
    (folded into the to_q weights, weights.ATTN_LOG2E) -> SASPA_ATTN_QPRESCALED, `scale` is ignored.
    v_rowmajor: `vt` is V itself, [B, nk, >=heads*d] like k (a view into a fused Q | K | V projection) ->
    SASPA_ATTN_V_ROWMAJOR: the kernel transposes between LDS and the MFMA, no V^T projection is needed."""
    _check_dev(q, k, vt, out)
    lib = _L()
    if q.dtype != torch.bfloat16:
        raise TypeError("flash_attn is the bf16 path; fp32 uses the unfused GEMM+softmax path")
    p = _lib.AttnParams()
    p.q, p.ldq, p.sqb = _ptr(q), q.stride(1), q.stride(0)
    p.k, p.ldk, p.skb = _ptr(k), k.stride(1), k.stride(0)
    p.vt, p.ldvt, p.svb = _ptr(vt), vt.stride(1), vt.stride(0)
    p.o, p.ldo, p.sob = _ptr(out), out.stride(1), out.stride(0)
    p.batch, p.heads, p.D, p.nq, p.nk = q.shape[0], heads, d, nq, nk
    p.scale, p.causal, p.flags = float(scale), int(causal), (1 if prescaled else 0) | (2 if v_rowmajor else 0)
    _launch("flash_attn", 4.0 * q.shape[0] * heads * nq * nk * d,
            lambda: _lib.check(lib.saspa_flash_attn_bf16(C.byref(p), _stream()), "saspa_flash_attn_bf16"),
            (q.shape[0], heads, nq, nk, d))
    return out


def softmax_rows(x, n, scale, causal=False, rows_per_mat=1):
    """In-place softmax(scale*x) over the first n columns of [rows, ld]; pad columns zeroed."""
    _check_dev(x)
    lib = _L()
    x2 = x.view(-1, x.shape[-1])
    _lib.check(lib.saspa_softmax_rows(_dt(x), _ptr(x2), x2.shape[0], n, x2.stride(0), float(scale), int(causal),
                                      int(rows_per_mat), _stream()), "saspa_softmax_rows")
    return x


def _gn_nsplit(batch, hw, c8):
    """Pixel splits of the statistics pass: about 1024 workgroups per channel slab (the apply pass re-reads
    nsplit * slabs * groups sums per workgroup, so no more than needed to fill the chip)."""
    want = max(1, int(os.environ.get("SASPA_GN_BLOCKS", "1024")) // max(1, batch))
    return max(1, min(want, 64, hw // 16 if hw >= 16 else 1))


def groupnorm(x, gamma, beta, groups, eps, act=ACT_NONE, x2=None, out=None):
    """GroupNorm(+SiLU) over channels-last x (optionally concatenated with x2) -> [B,H,W,C]."""
    _check_dev(x, gamma, beta, x2, out)
    lib = _L()
    b, h, w, c0 = x.shape
    c1 = 0 if x2 is None else x2.shape[3]
    ctot = c0 + c1
    if x2 is not None and (tuple(x2.shape[:3]) != (b, h, w) or x2.dtype != x.dtype):
        raise ValueError(f"concat source {tuple(x2.shape)} does not match {tuple(x.shape)}")
    if gamma.numel() < ctot or beta.numel() < ctot or ctot % groups:
        raise ValueError("GroupNorm parameters / groups do not match the channel count")
    if out is not None and (tuple(out.shape[:3]) != (b, h, w) or out.shape[3] < ctot or out.dtype != x.dtype):
        raise ValueError("GroupNorm output does not match the input geometry")
    if out is None:
        out = torch.empty((b, h, w, ctot), device=x.device, dtype=x.dtype)
    # statistics left by the producers' epilogues (conv(..., gn_unit=...)): no statistics pass
    def _fresh(t):
        g = getattr(t, "saspa_gn", None) if t is not None else None
        if g is None or g[2] != t.data_ptr() or g[3] != t._version or g[0].shape[0] * 128 != t.shape[0] * t.shape[1] * t.shape[2]:
            return None
        return g
    g0, g1 = _fresh(x), _fresh(x2)
    fused = (g0 is not None and (x2 is None or g1 is not None) and gn_fusion_enabled() and (h * w) % 128 == 0
             and (x2 is None or g1[1] == g0[1]) and c0 % g0[1] == 0 and c1 % g0[1] == 0 and (ctot // groups) % g0[1] == 0
             and g0[0].shape[1] * g0[1] == c0 and (x2 is None or g1[0].shape[1] * g1[1] == c1))
    if fused:
        p = _lib.GroupNormParams()
        p.dtype = _dt(x)
        p.x0, p.x1, p.c0, p.c1 = _ptr(x), _ptr(x2), c0, c1
        p.ldx0 = _pitch4(x)
        p.ldx1 = 0 if x2 is None else _pitch4(x2)
        p.batch, p.hw, p.groups, p.eps = b, h * w, groups, float(eps)
        p.gamma, p.beta = _ptr(gamma), _ptr(beta)
        p.partial, p.nsplit, p.scale_shift = None, 0, None
        p.act, p.y, p.ldy = int(act), _ptr(out), _pitch4(out)
        p.stats0, p.stats1, p.unit = _ptr(g0[0]), (_ptr(g1[0]) if g1 is not None else None), g0[1]
        _lib.check(lib.saspa_groupnorm_apply(C.byref(p), _stream()), "saspa_groupnorm_apply(epilogue statistics)")
        return out
    if gn_onepass_enabled():
        # small images (the 8x8 level: no 128-row statistics blocks): statistics + apply in one launch
        p = _lib.GroupNormParams()
        p.dtype = _dt(x)
        p.x0, p.x1, p.c0, p.c1 = _ptr(x), _ptr(x2), c0, c1
        p.ldx0 = _pitch4(x)
        p.ldx1 = 0 if x2 is None else _pitch4(x2)
        p.batch, p.hw, p.groups, p.eps = b, h * w, groups, float(eps)
        p.gamma, p.beta = _ptr(gamma), _ptr(beta)
        p.partial, p.nsplit, p.scale_shift = None, 0, None
        p.act, p.y, p.ldy = int(act), _ptr(out), _pitch4(out)
        if lib.saspa_groupnorm_onepass_eligible(C.byref(p)):
            _lib.check(lib.saspa_groupnorm_onepass(C.byref(p), _stream()), "saspa_groupnorm_onepass")
            return out
    nsplit = _gn_nsplit(b, h * w, ctot // 8)
    partial = torch.empty((b * nsplit * ctot * 2,), device=x.device, dtype=torch.float32)
    p = _lib.GroupNormParams()
    p.dtype = _dt(x)
    p.x0, p.x1, p.c0, p.c1 = _ptr(x), _ptr(x2), c0, c1
    p.ldx0 = _pitch4(x)
    p.ldx1 = 0 if x2 is None else _pitch4(x2)
    p.batch, p.hw, p.groups, p.eps = b, h * w, groups, float(eps)
    p.gamma, p.beta = _ptr(gamma), _ptr(beta)
    p.partial, p.nsplit, p.scale_shift = _ptr(partial), nsplit, None
    p.act, p.y, p.ldy = int(act), _ptr(out), _pitch4(out)
    s = _stream()
    _lib.check(lib.saspa_groupnorm_stats(C.byref(p), s), "saspa_groupnorm_stats")
    _lib.check(lib.saspa_groupnorm_apply(C.byref(p), s), "saspa_groupnorm_apply")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _check_dev(x, gamma, beta, out)
    lib = _L()
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    o2 = out.view(-1, c)
    rows = x2.shape[0]
    _lib.check(lib.saspa_layernorm(_dt(x), _ptr(x2), x2.stride(0) if rows > 1 else c, _ptr(o2),
                                   o2.stride(0) if rows > 1 else c, rows, c, _ptr(gamma), _ptr(beta), float(eps),
                                   _stream()), "saspa_layernorm")
    return out


def layernorm_quant_fp8(x, gamma, beta, eps=1e-5):
    """LayerNorm over the last dim of bf16 x, quantised per row to e4m3 -> (q uint8 [..., C], scale fp32 [rows])."""
    _check_dev(x, gamma, beta)
    if x.dtype != torch.bfloat16:
        raise TypeError("the fp8 path takes bf16 activations")
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    rows = x2.shape[0]
    q = torch.empty((rows, c), device=x.device, dtype=torch.uint8)
    scale = torch.empty((rows,), device=x.device, dtype=torch.float32)
    _lib.check(_L().saspa_layernorm_quant_fp8(_ptr(x2), x2.stride(0) if rows > 1 else c, _ptr(q), c, _ptr(scale), rows, c,
                                                     _ptr(gamma), _ptr(beta), float(eps), _stream()), "saspa_layernorm_quant_fp8")
    return q.view(*x.shape[:-1], c), scale


def linear_fp8(xq, xscale, wq, wscale, bias=None, *, residual=None, act=ACT_NONE, out_fp8_scale=None, amax=None):
    """e4m3 activations (uint8 [..., K] + per-row scale, or ONE scale for the whole tensor: a 1-element `xscale`) @ e4m3 weights
    (uint8 [N, K] + per-channel scale)^T -> bf16 [..., N] (N / 2 with the fused GEGLU): `saspa_gemm_fp8`.
    GEGLU only (ABI 20): `out_fp8_scale` (device fp32 scalar) makes the result e4m3 bytes under that tensor-wide scale -- uint8
    [..., N / 2], what the feed-forward output projection reads --, `amax` (device fp32 scalar) receives the atomic maximum of the
    gated values' magnitudes (calibration of that scale)."""
    _check_dev(xq, xscale, wq, wscale, bias, residual, out_fp8_scale, amax)
    k = xq.shape[-1]
    x2 = xq.reshape(-1, k)
    m, n = x2.shape[0], wq.shape[0]
    nout = n // 2 if act == ACT_GEGLU else n
    if (out_fp8_scale is not None or amax is not None) and act != ACT_GEGLU:
        raise ValueError("fp8 emission / amax exist in the GEGLU epilogue only")
    out = torch.empty((m, nout), device=xq.device, dtype=torch.uint8 if out_fp8_scale is not None else torch.bfloat16)
    r2 = None if residual is None else residual.reshape(-1, residual.shape[-1])
    p = _lib.GemmF8Params()
    p.a, p.lda, p.w, p.ldw = _ptr(x2), x2.stride(0) if m > 1 else k, _ptr(wq), wq.stride(0)
    p.M, p.N, p.K = m, n, k
    p.sa, p.sw, p.bias = _ptr(xscale), _ptr(wscale), _ptr(bias)
    if xscale.numel() == 1 and m > 1:
        p.sa_broadcast = 1
    elif xscale.numel() < m:
        raise ValueError(f"{xscale.numel()} activation scales for {m} rows")
    p.residual, p.ldr = _ptr(r2), 0 if r2 is None else r2.stride(0)
    p.act, p.out, p.ldo = int(act), _ptr(out), nout
    if out_fp8_scale is not None:
        if out_fp8_scale.dtype != torch.float32 or out_fp8_scale.numel() != 1:
            raise ValueError("out_fp8_scale is one fp32 value on the device")
        p.out_fp8, p.out_scale = 1, _ptr(out_fp8_scale)
    if amax is not None:
        if amax.dtype != torch.float32 or amax.numel() != 1:
            raise ValueError("amax is one fp32 value on the device")
        p.amax = _ptr(amax)
    _launch("gemm", 2.0 * m * n * k, lambda: _lib.check(_L().saspa_gemm_fp8(C.byref(p), _stream()), "saspa_gemm_fp8"),
            (m, n, k, 0, 1, 0, False, residual is not None, nout, GEMM_FAMILY_FP8, 1))
    return out.reshape(*xq.shape[:-1], nout)


def fp8_pow2_scale(amax, margin=16.0):
    """Tensor-wide e4m3 scale from a calibration maximum (device fp32 scalar -> device fp32 scalar, no host sync): the power of two
    >= margin * amax / 448.  e4m3 is a floating format: a power-of-two scale shifts exponents only, so the quantised values do not
    depend on the calibration batch as long as nothing overflows (the margin: 4 binades; the conversion saturates beyond) or
    underflows (values below scale * 2^-9 ~ 1e-4 of the calibration maximum flush to zero)."""
    a = amax.float().reshape(1)
    s = torch.exp2(torch.ceil(torch.log2((a * (margin / 448.0)).clamp_min(2.0 ** -60))))
    return torch.where(a > 0, s, torch.ones_like(s))


def geglu(x, out=None):
    """x: [..., 2F] -> [..., F] = x[..., :F] * gelu_erf(x[..., F:])"""
    _check_dev(x, out)
    lib = _L()
    f = x.shape[-1] // 2
    x2 = x.reshape(-1, 2 * f)
    if out is None:
        out = torch.empty((*x.shape[:-1], f), device=x.device, dtype=x.dtype)
    o2 = out.view(-1, f)
    _lib.check(lib.saspa_geglu(_dt(x), _ptr(x2), x2.stride(0), _ptr(o2), o2.stride(0), x2.shape[0], f, _stream()),
               "saspa_geglu")
    return out


def activation(x, act, out=None):
    _check_dev(x, out)
    lib = _L()
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    o2 = out.view(-1, c)
    rows = x2.shape[0]
    _lib.check(lib.saspa_activation(_dt(x), int(act), _ptr(x2), x2.stride(0) if rows > 1 else c, _ptr(o2),
                                    o2.stride(0) if rows > 1 else c, rows, c, _stream()), "saspa_activation")
    return out


def pool2d(x, k, stride=None, pad=0, mode="avg"):
    """nn.MaxPool2d(k, stride, pad) / nn.AvgPool2d(k) over channels-last x [B,H,W,C] -> [B,Ho,Wo,C]."""
    _check_dev(x)
    stride = k if stride is None else stride
    b, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((b, ho, wo, c), device=x.device, dtype=x.dtype)
    _lib.check(_L().saspa_pool2d(_dt(x), 0 if mode == "max" else 1, _ptr(x), _pitch4(x), _ptr(out), c, b, h, w, c, int(k),
                                        int(stride), int(pad), _stream()), "saspa_pool2d")
    return out


def signsqrt_l2norm(x, eps, scale=1.0):
    """rows of fp32 [R, C]: scale * normalize(sign(x) * sqrt(|x| + eps)) (WSDAN bilinear attention pooling tail)."""
    _check_dev(x)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("signsqrt_l2norm expects an fp32 [rows, C] matrix")
    out = torch.empty((x.shape[0], x.shape[1]), device=x.device, dtype=torch.float32)
    _lib.check(_L().saspa_signsqrt_l2norm(_ptr(x), x.stride(0), _ptr(out), out.stride(0), x.shape[0], x.shape[1], float(eps),
                                                 float(scale), _stream()), "saspa_signsqrt_l2norm")
    return out


def embed_tokens(ids, tok, pos, npos):
    _check_dev(ids, tok, pos)
    lib = _L()
    n = ids.numel()
    c = tok.shape[1]
    out = torch.empty((n, c), device=tok.device, dtype=tok.dtype)
    ids32 = ids.reshape(-1).to(torch.int32)
    _lib.check(lib.saspa_embed_tokens(_dt(tok), _ptr(ids32), n, npos, _ptr(tok), _ptr(pos), c, _ptr(out), _stream()),
               "saspa_embed_tokens")
    return out


def embed_tokens_ctx(ids, ctx, ctx_begin, tok, pos):
    """ids [B, ntok] prompt tokens, ctx [B, nctx, C] (or None) spliced in at `ctx_begin`
    -> [B, ntok + nctx, C] = token embeddings + position embeddings (ContextCLIPTextEmbeddings)."""
    _check_dev(ids, ctx, tok, pos)
    lib = _L()
    b, ntok = ids.shape
    c = tok.shape[1]
    nctx = 0 if ctx is None else ctx.shape[1]
    if ctx is not None:
        if ctx.shape[0] != b or ctx.shape[2] != c or ctx.dtype != tok.dtype:
            raise ValueError("ctx must be [B, nctx, C] in the embedding dtype")
        ctx = ctx.contiguous()
    if ntok + nctx > pos.shape[0]:
        raise ValueError("sequence longer than the position table")
    out = torch.empty((b, ntok + nctx, c), device=tok.device, dtype=tok.dtype)
    ids32 = ids.reshape(-1).to(torch.int32)
    _lib.check(lib.saspa_embed_tokens_ctx(_dt(tok), _ptr(ids32), b, ntok, _ptr(ctx), nctx, int(ctx_begin), _ptr(tok), _ptr(pos), c,
                                          _ptr(out), _stream()), "saspa_embed_tokens_ctx")
    return out


def cfg_plms_step(eps, x, hist, sample, nimg, hw, c, guidance, store_slot, w_cur, w_hist, coef_sample, coef_model):
    """eps, x: [2*nimg, hw, 8]; hist: [4, nimg, hw, 8] history of CFG-combined outputs; sample: [nimg, hw, 8] or None.
    One PNDM/PLMS update (see saspa_cfg_plms_step); updates x (both CFG halves) and hist[store_slot] in place."""
    _check_dev(eps, x, hist, sample)
    lib = _L()
    wh = (C.c_float * 4)(*[float(v) for v in w_hist])
    _lib.check(lib.saspa_cfg_plms_step(_dt(x), _ptr(eps), _ptr(x), _ptr(hist), _ptr(sample), nimg, hw, c, 8, float(guidance),
                                       int(store_slot), float(w_cur), wh, float(coef_sample), float(coef_model), _stream()),
               "saspa_cfg_plms_step")
    return x


def cfg_ddim_step(eps, x, nimg, hw, c, guidance, sa_t, s1m_t, sa_p, s1m_p):
    """eps, x: [2*nimg, hw, 8]; updates x (both CFG halves) in place."""
    _check_dev(eps, x)
    lib = _L()
    _lib.check(lib.saspa_cfg_ddim_step(_dt(x), _ptr(eps), _ptr(x), nimg, hw, c, 8, float(guidance), float(sa_t),
                                       float(s1m_t), float(sa_p), float(s1m_p), _stream()), "saspa_cfg_ddim_step")
    return x


def ddim_step(eps, x, nimg, hw, c, sa_t, s1m_t, sa_p, s1m_p):
    """eps, x: [nimg, hw, 8]; DDIM update without CFG (SDXL-Turbo, guidance off), x in place."""
    _check_dev(eps, x)
    lib = _L()
    _lib.check(lib.saspa_ddim_step(_dt(x), _ptr(eps), _ptr(x), nimg, hw, c, 8, float(sa_t), float(s1m_t), float(sa_p),
                                   float(s1m_p), _stream()), "saspa_ddim_step")
    return x


def gather_row(table, index, dst):
    """dst[:] = table[index[0]] for a [rows, ...] fp32 table and a device int32 index (hipGraph step state)."""
    _check_dev(table, index, dst)
    lib = _L()
    row = table[0].numel()
    if table.dtype != torch.float32 or dst.dtype != torch.float32 or index.dtype != torch.int32 or not table.is_contiguous() \
            or not dst.is_contiguous() or dst.numel() != row:
        raise ValueError("gather_row: fp32 contiguous table [rows, ...], dst of one row, int32 index")
    _lib.check(lib.saspa_gather_row_f32(_ptr(table), row, _ptr(index), _ptr(dst), row, _stream()), "saspa_gather_row_f32")
    return dst


def ddim_step_dev(eps, x, nimg, hw, c, guidance, coefs, index, cfg=True):
    """(CFG +) DDIM update with the coefficients of row index[0] of the device table coefs [steps, 4]."""
    _check_dev(eps, x, coefs, index)
    lib = _L()
    if coefs.dtype != torch.float32 or coefs.dim() != 2 or coefs.shape[1] != 4 or not coefs.is_contiguous() or index.dtype != torch.int32:
        raise ValueError("ddim_step_dev: coefs fp32 [steps, 4], int32 index")
    _lib.check(lib.saspa_ddim_step_dev(_dt(x), _ptr(eps), _ptr(x), nimg, hw, c, 8, int(bool(cfg)), float(guidance), _ptr(coefs),
                                       _ptr(index), _stream()), "saspa_ddim_step_dev")
    return x


def cfg_plms_step_dev(eps, x, hist, saved, nimg, hw, c, guidance, table, index):
    """PLMS update with the evaluation's parameters read from row index[0] of the device table [evaluations, 10]."""
    _check_dev(eps, x, hist, saved, table, index)
    if table.dtype != torch.float32 or table.dim() != 2 or table.shape[1] != 10 or not table.is_contiguous() or index.dtype != torch.int32:
        raise ValueError("cfg_plms_step_dev: table fp32 [evaluations, 10], int32 index")
    _lib.check(_L().saspa_cfg_plms_step_dev(_dt(x), _ptr(eps), _ptr(x), _ptr(hist), _ptr(saved), nimg, hw, c, 8, float(guidance),
                                                   _ptr(table), _ptr(index), _stream()), "saspa_cfg_plms_step_dev")
    return x


def cfg_unipc_step(eps, x, state, nimg, hw, c, guidance, row=None, table=None, index=None):
    """CFG + one UniPC step; state [3, nimg, hw, 8] (last | m0 | m1).  `row`: 12 host floats, or (`table` fp32 [steps, 12],
    `index` int32 [1]) on the device."""
    _check_dev(eps, x, state, table, index)
    if table is not None and (table.dtype != torch.float32 or table.dim() != 2 or table.shape[1] != 12 or not table.is_contiguous()
                              or index is None or index.dtype != torch.int32):
        raise ValueError("cfg_unipc_step: table fp32 [steps, 12], int32 index")
    r = None if row is None else (C.c_float * 12)(*[float(v) for v in row])
    _lib.check(_L().saspa_cfg_unipc_step(_dt(x), _ptr(eps), _ptr(x), _ptr(state), nimg, hw, c, 8, float(guidance), r,
                                                _ptr(table), _ptr(index), _stream()), "saspa_cfg_unipc_step")
    return x


def unipc_step(eps, x, state, nimg, hw, c, row=None, table=None, index=None):
    """One UniPC step WITHOUT classifier-free guidance (sd_xl-turbo): eps / x [nimg, hw, 8]; state [3, nimg, hw, 8]."""
    _check_dev(eps, x, state, table, index)
    if table is not None and (table.dtype != torch.float32 or table.dim() != 2 or table.shape[1] != 12 or not table.is_contiguous()
                              or index is None or index.dtype != torch.int32):
        raise ValueError("unipc_step: table fp32 [steps, 12], int32 index")
    r = None if row is None else (C.c_float * 12)(*[float(v) for v in row])
    _lib.check(_L().saspa_unipc_step(_dt(x), _ptr(eps), _ptr(x), _ptr(state), nimg, hw, c, 8, r, _ptr(table), _ptr(index),
                                            _stream()), "saspa_unipc_step")
    return x


def index_add(index, delta=1):
    _check_dev(index)
    _lib.check(_L().saspa_index_add(_ptr(index), int(delta), _stream()), "saspa_index_add")
    return index


def clock_probe(out2, iters=250):
    """out2: int64 device tensor [2] <- (shader clocks, 100 MHz ticks) of a window of iters x s_sleep 127 (~1 ms at 250),
    measured by one sleeping wave on the current stream (bench.py: a side stream, while the hot path runs)."""
    _check_dev(out2)
    if out2.dtype != torch.int64 or out2.numel() < 2 or not out2.is_contiguous():
        raise ValueError("clock_probe wants a contiguous int64 tensor of >= 2 elements")
    _lib.check(_L().saspa_clock_probe(_ptr(out2), int(iters), _stream()), "saspa_clock_probe")
    return out2


def vae_sample_noise(moments, e1, e2, scaling, sa, s1m):
    """moments [B,h,w,8] (mean | logvar), e1 / e2 [B,h,w,8] noise draws -> noised start latents [B,h,w,8] (img2img)."""
    _check_dev(moments, e1, e2)
    out = torch.empty_like(moments)
    npix = moments.numel() // 8
    _lib.check(_L().saspa_vae_sample_noise(_dt(moments), _ptr(moments.contiguous()), _ptr(e1.contiguous()), _ptr(e2.contiguous()),
                                                  _ptr(out), npix, float(scaling), float(sa), float(s1m), _stream()),
               "saspa_vae_sample_noise")
    return out


def scale(x, s, out=None):
    _check_dev(x, out)
    lib = _L()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.saspa_scale(_dt(x), _ptr(x), _ptr(out), x.numel(), float(s), _stream()), "saspa_scale")
    return out


def u8_to_act(img_u8, dtype):
    """u8 [n,H,W,3] -> [n,H,W,8] in [0,1]"""
    _check_dev(img_u8)
    lib = _L()
    n, h, w, _ = img_u8.shape
    out = torch.empty((n, h, w, 8), device=img_u8.device, dtype=dtype)
    _lib.check(lib.saspa_u8_to_act(_dt(out), _ptr(img_u8), _ptr(out), n * h * w, _stream()), "saspa_u8_to_act")
    return out


def act_to_u8(x):
    """[n,H,W,>=4] (3 live channels) -> u8 [n,H,W,3]"""
    _check_dev(x)
    lib = _L()
    n, h, w, _ = x.shape
    out = torch.empty((n, h, w, 3), device=x.device, dtype=torch.uint8)
    _lib.check(lib.saspa_act_to_u8(_dt(x), _ptr(x), _pitch4(x), _ptr(out), n * h * w, _stream()), "saspa_act_to_u8")
    return out


def canny(img_u8, low, high):
    """u8 [n,H,W,3] (device) -> u8 [n,H,W,3] in {0,255}"""
    _check_dev(img_u8)
    lib = _L()
    n, h, w, c = img_u8.shape
    if c != 3 or img_u8.dtype != torch.uint8 or not img_u8.is_contiguous():
        raise ValueError("canny expects a contiguous u8 [n,H,W,3] tensor")
    out = torch.empty_like(img_u8)
    work = torch.empty((8 * n * h * w,), device=img_u8.device, dtype=torch.uint8)
    _lib.check(lib.saspa_canny(_ptr(img_u8), _ptr(out), _ptr(work), n, h, w, int(low), int(high), _stream()),
               "saspa_canny")
    return out
