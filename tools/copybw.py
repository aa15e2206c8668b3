import torch, time
a=torch.empty(100_000_000, dtype=torch.int64, device="cuda"); b=torch.empty_like(a)
for fn,name,bytes_ in ((lambda: b.copy_(a),"copy (r+w 1.6 GB)",1.6e9),(lambda: a.sum(),"read-only sum (0.8 GB)",0.8e9),(lambda: b.fill_(1),"write-only fill (0.8 GB)",0.8e9)):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print("%-28s %.3f ms  %.2f TB/s"%(name,ms,bytes_/ms/1e9))
