import sys, os, time
sys.path.insert(0, os.getcwd())
import kwage_amd as ka
ctx = ka.Context(0)
for L, cols in ((23, 100_000), (23, 100_000), (21, 100_000), (20, 1_000_000)):
    t = time.time()
    g = ka.Group(ctx, 31, 1, L, cols)
    print("group create %.2f s" % (time.time() - t), g.placement, flush=True)
    t = time.time(); g.close(); print("close %.2f s" % (time.time() - t), flush=True)
