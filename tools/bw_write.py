"""Dev tool: plain device write / copy bandwidth on this GPU (context for the backward GEMM's 2 GB of plane stores)."""
import torch, time
x = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device="cuda")  # 2 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill 2 GiB", lambda: x.fill_(1.0), x.numel() * 4), ("copy 2 GiB", lambda: y.copy_(x), 2 * x.numel() * 4),
                         ("mul 2 GiB (r+w)", lambda: torch.mul(x, 2.0, out=y), 2 * x.numel() * 4)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.3f} ms  {nbytes/dt/1e12:.2f} TB/s", flush=True)
