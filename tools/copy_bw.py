import torch, time
torch.cuda.init()
for mb in (17, 34, 68, 136, 512):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"copy {mb} MB -> {mb} MB: {us:.1f} us  {2*mb*1.048576/us:.2f} TB/s (read + write)")
