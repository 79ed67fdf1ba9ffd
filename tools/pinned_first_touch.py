"""First-use cost of a pinned host block in hipMemcpyAsync (MI355X box): is it per allocation, per page, per direction?"""
import time, torch
dev = torch.device("cuda:0")
d = torch.zeros(32 << 20, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return "call %.3f ms, done %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3)
for trial in range(2):
    p = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
    print("fresh block A: D2H 16 MB #1:", t(lambda: p.copy_(d[:16 << 20], non_blocking=True)))
    print("               D2H 16 MB #2:", t(lambda: p.copy_(d[:16 << 20], non_blocking=True)))
    print("               H2D 16 MB #1:", t(lambda: d[:16 << 20].copy_(p, non_blocking=True)))
    print("               H2D 16 MB #2:", t(lambda: d[:16 << 20].copy_(p, non_blocking=True)))
    q = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
    print("fresh block B: D2H 256 B   :", t(lambda: q[:256].copy_(d[:256], non_blocking=True)))
    print("               D2H 8 MB #1 :", t(lambda: q[:8 << 20].copy_(d[:8 << 20], non_blocking=True)))
    print("               D2H 16 MB #1:", t(lambda: q.copy_(d[:16 << 20], non_blocking=True)))
    print("               D2H 16 MB #2:", t(lambda: q.copy_(d[:16 << 20], non_blocking=True)))
    r = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
    r.zero_()
    print("fresh block C (zeroed by the CPU first): D2H 16 MB #1:", t(lambda: r.copy_(d[:16 << 20], non_blocking=True)))
    print("               H2D 16 MB #1:", t(lambda: d[:16 << 20].copy_(r, non_blocking=True)))
    del p, q, r
