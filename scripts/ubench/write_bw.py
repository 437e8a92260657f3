import torch, time
dev="cuda"
vol=torch.empty(8,192,512,640,device=dev)
src=torch.empty_like(vol)
def t(fn,n=10):
    for _ in range(3): fn()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
ms=t(lambda: vol.fill_(1.0)); print(f"fill 2GB {ms:.3f} ms {vol.numel()*4/ms/1e9:.2f} TB/s")
ms=t(lambda: vol.copy_(src)); print(f"copy 2GB {ms:.3f} ms {2*vol.numel()*4/ms/1e9:.2f} TB/s (r+w)")
ms=t(lambda: vol.mul_(1.5)); print(f"mul_ 2GB {ms:.3f} ms {2*vol.numel()*4/ms/1e9:.2f} TB/s (r+w)")
