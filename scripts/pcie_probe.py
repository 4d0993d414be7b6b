import torch, time
n = 3840*2160*3//2
def run(nstreams, direction, iters=60):
    hs = [torch.empty(n // nstreams, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
    ds = [torch.empty(n // nstreams, dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
    ss = [torch.cuda.Stream() for _ in range(nstreams)]
    for _ in range(5):
        for h, d, s in zip(hs, ds, ss):
            with torch.cuda.stream(s):
                (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for h, d, s in zip(hs, ds, ss):
            with torch.cuda.stream(s):
                (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return n / dt / 1e9, dt * 1e3
for direction in ("h2d", "d2h"):
    for ns in (1, 2, 4, 8):
        gbs, ms = run(ns, direction)
        print(f"{direction} {ns} streams: {gbs:.1f} GB/s, {ms:.3f} ms per 4K I420 frame")
# both directions at once
def both(ns, iters=60):
    hs = [torch.empty(n // ns, dtype=torch.uint8).pin_memory() for _ in range(2 * ns)]
    ds = [torch.empty(n // ns, dtype=torch.uint8, device="cuda") for _ in range(2 * ns)]
    ss = [torch.cuda.Stream() for _ in range(2 * ns)]
    def go():
        for i, (h, d, s) in enumerate(zip(hs, ds, ss)):
            with torch.cuda.stream(s):
                (d.copy_(h, non_blocking=True) if i < ns else h.copy_(d, non_blocking=True))
    for _ in range(5): go()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): go()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    return n / dt / 1e9, dt * 1e3
for ns in (1, 2, 4):
    gbs, ms = both(ns)
    print(f"both directions, {ns} streams each: {gbs:.1f} GB/s each way, {ms:.3f} ms per frame pair")
