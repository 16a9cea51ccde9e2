"""Device-memory plumbing: PyTorch-ROCm is used for HBM allocations and streams only."""
import numpy as np
import torch


def device(index=None):
    if not torch.cuda.is_available():
        raise RuntimeError("hilo_mpc_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device('cuda', torch.cuda.current_device() if index is None else index)


def to_dev(a, dev, shape=None):
    """float64 contiguous device tensor from numpy / list / scalar / torch input (copies host data over PCIe)."""
    if isinstance(a, torch.Tensor):
        if shape is None and a.dtype is torch.float64 and a.device == dev and a.is_contiguous():
            return a                       # (the common case of a control loop: nothing to do - a call's host time is counted in us)
        t = a.to(device=dev, dtype=torch.float64)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=dev)
    if shape is not None:
        t = t.reshape(shape)
    return t.contiguous()


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(dev):
    """The current stream of `dev` as a raw hipStream_t (what torch's own generated code calls; the Stream object costs ~4 us)."""
    if _raw_stream is not None and dev.index is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


def like_input(t, ref):
    """Return numpy when the caller passed host data, the device tensor otherwise (mirrors gp.py:714-716)."""
    if isinstance(ref, torch.Tensor):
        return t
    return t.cpu().numpy()
