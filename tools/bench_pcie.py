import torch, time
for mb in (8, 47, 234):
    h = torch.empty(mb * 1024 * 1024, dtype=torch.uint8).pin_memory()
    d = torch.empty_like(h, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"H2D {mb} MB pinned: {dt*1e3:.3f} ms = {mb/1024/dt:.1f} GB/s")
    for _ in range(3): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"D2H {mb} MB pinned: {dt*1e3:.3f} ms = {mb/1024/dt:.1f} GB/s")
