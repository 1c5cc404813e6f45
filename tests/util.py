"""Shared helpers for the parity tests (tests only)."""
import torch


def to_nhwc(x_nchw, dtype, dev, cpad=None):
    """NCHW fp32 (cpu) -> channels-last device tensor, channels zero-padded to cpad."""
    x = x_nchw.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad != x.shape[-1]:
        x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
    return x.to(dev, dtype).contiguous()


def from_nhwc(y, c=None):
    """device channels-last -> NCHW fp32 cpu (first c channels)."""
    y = y.float().cpu()
    if c is not None:
        y = y[..., :c]
    return y.permute(0, 3, 1, 2).contiguous()


def q(x, dtype):
    """Round a reference input to the storage dtype (so bf16 tests compare like with like)."""
    return x.to(dtype).float()


def tol(dtype):
    # fp32: accumulation-order differences only; bf16: output rounding 2^-8 relative + input
    # products accumulated in fp32
    return dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)


def assert_close(got, ref, dtype, scale=1.0, what=""):
    t = tol(dtype)
    err = (got - ref).abs()
    bound = t["atol"] * scale + t["rtol"] * ref.abs()
    bad = err > bound
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err {err.max().item():.4g} "
            f"(ref scale {ref.abs().max().item():.4g}); first bad at {idx}: got {got[tuple(idx)].item():.6g} "
            f"ref {ref[tuple(idx)].item():.6g}")
