"""How long do small torch ops take while a gather kernel occupies the GPU? (host-timed, one op at a time)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kwage_amd as ka
from kwage_amd import synth

torch.cuda.set_device(0)
ctx = ka.Context(0)
s = synth.build(ctx, synth.WORKLOADS["c2"])
thr = synth.WORKLOADS["c2"].threshold
x = torch.zeros(4096, device="cuda"); pin = torch.empty(4096, pin_memory=True)
hi = torch.cuda.Stream(priority=-1)
def ops():
    return {
        "add_+sync": lambda: (x.add_(1), torch.cuda.current_stream().synchronize()),
        "item()": lambda: x[0].item(),
        "d2h pinned+sync": lambda: (pin.copy_(x, non_blocking=True), torch.cuda.current_stream().synchronize()),
        "sort 64k": lambda: (torch.sort(torch.arange(65536, device="cuda").flip(0)), torch.cuda.current_stream().synchronize()),
    }
for name, f in ops().items():
    f(); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); idle = time.perf_counter() - t0
    res = []
    for stream in (None, hi):
        p = s.group.submit(s.batch, thr, 0)
        time.sleep(0.0003)                       # let the gather kernel start
        t0 = time.perf_counter()
        if stream is None: f()
        else:
            with torch.cuda.stream(stream): f()
        res.append(time.perf_counter() - t0)
        p.collect()
    print("%-18s idle %.3f ms | during gather: default stream %.3f ms, high-priority stream %.3f ms" % (name, idle*1e3, res[0]*1e3, res[1]*1e3))
