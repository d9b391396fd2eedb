"""Read-only / copy streaming ceilings of this device with torch ops (development aid)."""
import torch, time
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for gb in (1.3, 4.0, 16.0):
    N=int(gb*1e9/8)
    x=torch.empty(N,dtype=torch.float64,device="cuda").uniform_(); y=torch.empty_like(x)
    xi=x.view(torch.int64)
    print("size %.1f GB: sum(f64) %.0f GB/s | sum(int64 view) %.0f | max %.0f | copy %.0f (r+w) | mul_ inplace %.0f (r+w) | dot(x,y) %.0f | fill %.0f (w)" % (
        gb, 8*N/t(lambda: x.sum())/1e9, 8*N/t(lambda: xi.sum())/1e9, 8*N/t(lambda: x.max())/1e9, 16*N/t(lambda: y.copy_(x))/1e9,
        16*N/t(lambda: x.mul_(1.0000001))/1e9, 16*N/t(lambda: torch.dot(x,y))/1e9, 8*N/t(lambda: y.fill_(1.0))/1e9), flush=True)
    del x,y,xi
