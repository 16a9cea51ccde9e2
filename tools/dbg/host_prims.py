"""Host cost of the primitives a filter / controller call is made of (us per operation, device idle)."""
import time
import torch
dev = torch.device('cuda:0')
B, K = 4096, 16
u = torch.rand(B, 2, dtype=torch.float64, device=dev)
p = torch.rand(B, 4, dtype=torch.float64, device=dev)
buf = torch.empty(1, B, 6, dtype=torch.float64, device=dev)
y = torch.rand(K, B, 2, dtype=torch.float64, device=dev)
out = torch.empty(K, B, 4, 5, dtype=torch.float64, device=dev)


def t(name, f, n=20000):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        f()
    b = time.perf_counter()
    torch.cuda.synchronize()
    print(f'{name:40s} {(b - a) / n * 1e6:7.2f} us')


t('torch.empty(K,B,4,5)', lambda: torch.empty(K, B, 4, 5, dtype=torch.float64, device=dev))
t('cat(out=buf)', lambda: torch.cat([u.reshape(-1, 2).expand(B, -1)[None], p.expand(B, -1)[None].expand(1, -1, -1)], dim=2, out=buf), 5000)
t('two sliced assigns', lambda: (buf.__setitem__((slice(None), slice(None), slice(None, 2)), u[None]), buf.__setitem__((slice(None), slice(None), slice(2, None)), p[None])), 5000)
t('reshape+contiguous', lambda: y.reshape(K, -1, 2).contiguous())
t('out[:, :, :, 0]', lambda: out[:, :, :, 0])
t('out[-1]', lambda: out[-1])
t('data_ptr', lambda: out.data_ptr())
t('key tuple', lambda: tuple((id(q), q._version, tuple(q.shape)) for q in (u, p)))
t('raw stream', lambda: torch._C._cuda_getCurrentRawStream(0))
t('current_stream.cuda_stream', lambda: torch.cuda.current_stream(dev).cuda_stream)
t('isinstance+device eq', lambda: isinstance(u, torch.Tensor) and u.dtype is torch.float64 and u.device == dev and u.is_contiguous())
t('numel % n', lambda: y.numel() % 32)
