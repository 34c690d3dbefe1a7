"""Host-side phase timing of the sharded pipelined step (nccl world=1): submit / collect / exchange+merge."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np
import torch, torch.distributed as dist
import kwage_amd as ka
from kwage_amd import synth
from kwage_amd.distributed import ShardedSearch, PipelinedDeviceSearcher

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = ka.Context(0)
w = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
s = synth.build(ctx, w)
ss = ShardedSearch(dist, 0, 1, s.group.num_columns, None, device="cuda:0")
pipe = PipelinedDeviceSearcher(s.group, int(os.environ.get("PROBE_FLAGS", "0")), "cuda:0")
thr = w.threshold
def sync():
    ctx.sync(); torch.cuda.synchronize()
for _ in range(3):
    ss.exchange_counted(*pipe.collect_counted(pipe.submit(s.batch, thr)))
sync()
# unloaded exchange
t = pipe.collect_counted(pipe.submit(s.batch, thr)); sync()
t0 = time.perf_counter()
for _ in range(20): m = ss.exchange_counted(*t)
print("hits", len(m), "exchange+merge unloaded %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
# loaded
ph = np.zeros(3); N = 50; kms = []
tk = pipe.submit(s.batch, thr)
T0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); nxt = pipe.submit(s.batch, thr)
    b = time.perf_counter(); t = pipe.collect_counted(tk); kms.append(pipe.last_kernel_ms)
    c = time.perf_counter(); m = ss.exchange_counted(*t)
    d = time.perf_counter(); ph += (b - a, c - b, d - c); tk = nxt
pipe.collect_counted(tk); sync()
print("per step %.3f ms: submit %.3f collect %.3f exchange %.3f; gather kernel %.3f ms (HIP events)" % ((time.perf_counter() - T0) / N * 1e3, *(ph / N * 1e3), float(np.mean(kms))))
# the same loop without the exchange (device search only)
tk = pipe.submit(s.batch, thr); kms = []
T0 = time.perf_counter()
for _ in range(N):
    nxt = pipe.submit(s.batch, thr); pipe.collect_counted(tk); kms.append(pipe.last_kernel_ms); tk = nxt
pipe.collect_counted(tk); sync()
print("no exchange: per step %.3f ms; gather kernel %.3f ms" % ((time.perf_counter() - T0) / N * 1e3, float(np.mean(kms))))
# exchange only the collective
tk = pipe.submit(s.batch, thr); kms = []
T0 = time.perf_counter()
for _ in range(N):
    nxt = pipe.submit(s.batch, thr); buf, n = pipe.collect_counted(tk); kms.append(pipe.last_kernel_ms)
    ss.dist.all_gather_into_tensor(ss._crecv, buf[:ss.capacity + 1]); tk = nxt
pipe.collect_counted(tk); sync()
print("all_gather only: per step %.3f ms; gather kernel %.3f ms" % ((time.perf_counter() - T0) / N * 1e3, float(np.mean(kms))))
dist.destroy_process_group()
