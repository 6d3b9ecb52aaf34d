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
