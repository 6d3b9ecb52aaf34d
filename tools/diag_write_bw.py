"""Write-only HBM bandwidth of this GPU (development aid): what a pure store stream reaches."""
import time
import torch
n = 4_080_501 * 401
x = torch.empty(n, dtype=torch.float64, device="cuda")
def tm(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for name, f in [("fill_(1.0)", lambda: x.fill_(1.0)), ("zero_()", lambda: x.zero_())]:
    s = tm(f)
    print(f"{name}: {s*1e3:.2f} ms for {n*8/1e9:.2f} GB = {n*8/s/1e12:.2f} TB/s written")
y = torch.empty_like(x)
s = tm(lambda: y.copy_(x))
print(f"copy_: {s*1e3:.2f} ms = {2*n*8/s/1e12:.2f} TB/s read+write ({n*8/s/1e12:.2f} TB/s written)")
s = tm(lambda: x.sum())
print(f"sum (read only): {s*1e3:.2f} ms = {n*8/s/1e12:.2f} TB/s read")

# Round 6: the volume's OWN write pattern.  A stacking launch writes, per (node, time tile), 2 KB of a node's row
# (256 samples); rows are n_samples * 8 bytes apart (48 KB for the 6000-sample volume, 3.2 KB for the 401-sample
# window).  What does a pure store stream reach on that pattern -- every tile of every node written, tile by tile
# (one strided fill per tile, as the tiles' workgroups do side by side), nothing else running?
for ns, n_nodes in ((6000, 4_080_501 // 2), (401, 4_080_501)):
    v = torch.empty((n_nodes, ns), dtype=torch.float64, device="cuda")
    tiles = [(t, min(t + 256, ns)) for t in range(0, ns, 256)]
    def by_tiles():
        for a, b in tiles:
            v[:, a:b].fill_(1.0)
    s = tm(by_tiles, reps=3)
    c = tm(lambda: v.fill_(1.0), reps=3)
    print(f"[{n_nodes}][{ns}] volume: by 256-sample tiles {s*1e3:.1f} ms = {v.numel()*8/s/1e12:.2f} TB/s written; "
          f"contiguous {c*1e3:.1f} ms = {v.numel()*8/c/1e12:.2f} TB/s")
    del v
