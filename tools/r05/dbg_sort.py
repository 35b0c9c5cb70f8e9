"""finalize_pairs' ordering chain in isolation: int64 vs int32 keys, argsort vs sort, indexing vs index_select (208k pairs)."""
import time, torch
dev = "cuda:0"
n, B, N = 208000, 4096, 1536
g = torch.Generator(device=dev); g.manual_seed(0)
node = torch.randint(0, N, (n,), device=dev, generator=g, dtype=torch.int32); ray = torch.randint(0, B, (n,), device=dev, generator=g, dtype=torch.int32)
pid = torch.stack([ray, node], 1).contiguous(); px = torch.randn(n, 4, device=dev); pg = torch.randn(n, 4, device=dev)
pon = torch.randint(-1, n, (N, B), device=dev, generator=g, dtype=torch.int32)
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t) / reps
def cur():
    key = pid[:, 1].to(torch.int64) * B + pid[:, 0].to(torch.int64)
    perm = torch.argsort(key)
    inv = torch.empty(n, dtype=torch.int32, device=dev); inv[perm] = torch.arange(n, dtype=torch.int32, device=dev)
    a, b, c = pid[perm], px[perm], pg[perm]
    flat = pon.view(-1)
    return torch.where(flat >= 0, inv[flat.clamp(min=0).to(torch.int64)], flat).view(N, B)
def new():
    key = pid[:, 1] * B + pid[:, 0]                      # < N * B < 2^31
    _, perm = torch.sort(key)
    inv = torch.empty(n, dtype=torch.int32, device=dev); inv.index_copy_(0, perm, torch.arange(n, dtype=torch.int32, device=dev))
    a, b, c = pid.index_select(0, perm), px.index_select(0, perm), pg.index_select(0, perm)
    flat = pon.view(-1)
    return torch.where(flat >= 0, inv.index_select(0, flat.clamp(min=0)), flat).view(N, B)
assert torch.equal(cur(), new())
print(f"current chain {timed(cur):.0f} us, int32 keys + sort + index_select {timed(new):.0f} us")
for name, f in (("argsort int64", lambda: torch.argsort(pid[:, 1].to(torch.int64) * B + pid[:, 0].to(torch.int64))), ("sort int32", lambda: torch.sort(pid[:, 1] * B + pid[:, 0])),
                ("px[perm] int64", lambda: px[torch.arange(n, device=dev)]), ("where/inv over N x B", lambda: torch.where(pon.view(-1) >= 0, pon.view(-1), pon.view(-1)))):
    print(f"  {name}: {timed(f):.0f} us")
