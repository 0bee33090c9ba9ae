import torch, time
def t(f, n=20):
    f(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for mb in (128, 512):
    n=mb*1024*1024//4
    a=torch.empty(n, device='cuda'); b=torch.empty(n, device='cuda')
    us=t(lambda: a.fill_(1.0)); print(f"fill {mb}MB: {us:.1f} us  {mb*1.048576/us*1e3:.0f} GB/s write")
    us=t(lambda: b.copy_(a)); print(f"copy {mb}MB: {us:.1f} us  {2*mb*1.048576/us*1e3:.0f} GB/s r+w")
    us=t(lambda: a.sum()); print(f"sum  {mb}MB: {us:.1f} us  {mb*1.048576/us*1e3:.0f} GB/s read")
