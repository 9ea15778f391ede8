import torch, time
dev = torch.device("cuda:0")
S = 64
x = torch.zeros(1024, 1024, device=dev)
def work():
    # ~0.5 ms of GPU work per call, like a forward
    y = x
    for _ in range(6): y = y @ x
    return y
def loop(draw, n=200):
    for _ in range(10): draw(); work()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        j = draw(); work()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def d_pin(): return torch.rand(S, pin_memory=True).to(dev, non_blocking=True)
def d_page(): return torch.rand(S).to(dev, non_blocking=True)
ring = [torch.empty(S).pin_memory() for _ in range(64)]; ev = [None] * 64; k = [0]
def d_ring():
    i = k[0] % 64; k[0] += 1
    if ev[i] is not None: ev[i].synchronize()
    torch.rand(S, out=ring[i])
    t = ring[i].to(dev, non_blocking=True)
    e = torch.cuda.Event(); e.record(); ev[i] = e
    return t
def d_none(): return None
for name, f in (("none", d_none), ("pin_memory per call", d_pin), ("pageable", d_page), ("pinned ring", d_ring), ("pin_memory per call", d_pin)):
    print(f"{name:22s} {loop(f):.4f} ms per iteration")
