import torch, time
mb = 200
h = [torch.empty(mb * 1024 * 1024 // 2, dtype=torch.uint8).pin_memory() for _ in range(2)]
d = [torch.empty_like(x, device="cuda") for x in h]
s = [torch.cuda.Stream() for _ in range(2)]
def run(two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        if two:
            for k in range(2):
                with torch.cuda.stream(s[k]): d[k].copy_(h[k], non_blocking=True)
        else:
            for k in range(2): d[k].copy_(h[k], non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10
for _ in range(2): run(False); run(True)
a, b = run(False), run(True)
print(f"H2D {mb} MB pinned: one stream {a*1e3:.3f} ms = {mb/1024/a:.1f} GB/s, two streams {b*1e3:.3f} ms = {mb/1024/b:.1f} GB/s")
# four chunks on four streams
s4 = [torch.cuda.Stream() for _ in range(4)]
h4 = [torch.empty(mb * 1024 * 1024 // 4, dtype=torch.uint8).pin_memory() for _ in range(4)]
d4 = [torch.empty_like(x, device="cuda") for x in h4]
def run4():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        for k in range(4):
            with torch.cuda.stream(s4[k]): d4[k].copy_(h4[k], non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10
run4(); c = run4()
print(f"four streams {c*1e3:.3f} ms = {mb/1024/c:.1f} GB/s")
