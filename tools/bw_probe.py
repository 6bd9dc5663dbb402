import torch, time
dev="cuda:0"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for mb in (168, 512, 2048):
    n = mb*1024*1024//4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    us = t(lambda: x.fill_(1.0)); print(f"fill  {mb:5d} MB: {us:8.1f} us  {mb*1.048576/us:6.2f} TB/s")
    us = t(lambda: y.copy_(x));  print(f"copy  {mb:5d} MB: {us:8.1f} us  r+w {2*mb*1.048576/us:6.2f} TB/s")
    us = t(lambda: x.sum());     print(f"read  {mb:5d} MB: {us:8.1f} us  {mb*1.048576/us:6.2f} TB/s")
