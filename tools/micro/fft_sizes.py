#!/usr/bin/env python3
"""rocFFT timing (through torch.fft) of real 3D transforms of PME-sized grids: whole 3D plans vs. per-axis batched 1D plans."""
import torch, time
dev = torch.device("cuda")
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for shape in [(56, 80, 80), (64, 80, 80), (60, 80, 80), (64, 64, 64), (64, 96, 96), (72, 72, 72), (56, 84, 84), (64, 81, 81), (60, 90, 90), (64, 100, 100), (96, 96, 96), (80, 112, 112), (100, 144, 144), (128, 128, 128)]:
    x = torch.randn(shape, device=dev)
    t3 = timeit(lambda: torch.fft.irfftn(torch.fft.rfftn(x), s=shape))
    def axes():
        y = torch.fft.rfft(x, dim=2); y = torch.fft.fft(y, dim=1); y = torch.fft.fft(y, dim=0)
        y = torch.fft.ifft(y, dim=0); y = torch.fft.ifft(y, dim=1); return torch.fft.irfft(y, n=shape[2], dim=2)
    t1 = timeit(axes)
    def two_one():
        y = torch.fft.rfft2(x, dim=(1, 2)); y = torch.fft.fft(y, dim=0)
        y = torch.fft.ifft(y, dim=0); return torch.fft.irfft2(y, s=shape[1:], dim=(1, 2))
    t2 = timeit(two_one)
    print("%-16s 3D fwd+inv %7.1f us   per-axis 1D x6 %7.1f us   2D+1D %7.1f us" % (shape, t3, t1, t2), flush=True)
