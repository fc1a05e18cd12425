import torch, torch.nn.functional as F
dev=torch.device("cuda",0)
x=torch.randn(256,2048,7,7,device=dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
ones=torch.full((49,),1/49,device=dev)
print("mean(2,3)      %.1f us"%t(lambda: x.mean(dim=(2,3))))
print("view mean(-1)  %.1f us"%t(lambda: x.view(256,2048,49).mean(-1)))
print("adaptive pool  %.1f us"%t(lambda: F.adaptive_avg_pool2d(x,1)))
print("avg_pool2d 7   %.1f us"%t(lambda: F.avg_pool2d(x,7)))
print("matmul ones    %.1f us"%t(lambda: x.view(256*2048,49) @ ones))
print("sum(-1)        %.1f us"%t(lambda: x.view(256*2048,49).sum(-1)))
print("amax check     %.1f us"%t(lambda: x.view(256*2048,49).amax(-1)))
