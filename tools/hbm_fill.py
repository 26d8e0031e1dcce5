import torch, time
x = torch.empty(417_000_000 // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in (("fill_", lambda: x.fill_(1.0), x.numel()*4), ("zero_", lambda: x.zero_(), x.numel()*4), ("copy_", lambda: y.copy_(x), 2*x.numel()*4), ("read-reduce (sum)", lambda: x.sum(), x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:20s} {ms*1e3:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s")
