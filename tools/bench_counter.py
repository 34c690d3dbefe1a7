"""Throughput of the device counting pass (kwage_bloom_counter_*) next to the sequential CPU restatement.
Read set: 150-base reads drawn from a random genome at the given coverage, 1 % substitution errors."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import kwage_amd as ka
from kwage_amd.pipeline import BloomCounter

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=5_000_000)
ap.add_argument("--coverage", type=float, default=40.0)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--min-count", type=int, default=5)
ap.add_argument("--cpu-bases", type=int, default=20_000_000)
args = ap.parse_args()

rng = np.random.default_rng(1)
g = rng.integers(0, 4, size=args.genome, dtype=np.uint8)
n_reads = int(args.genome * args.coverage / args.read_len)
starts = rng.integers(0, args.genome - args.read_len, size=n_reads)
idx = starts[:, None] + np.arange(args.read_len)[None, :]
reads = g[idx]
err = rng.random(reads.shape) < 0.01
reads[err] = (reads[err] + rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)) & 3
flat = np.frombuffer(b"ACGT", dtype=np.uint8)[reads].reshape(-1)
buf = flat.tobytes()
offs = (np.arange(n_reads + 1, dtype=np.uint64) * args.read_len)
num_bp = n_reads * args.read_len
L = ka.native.lib()
logc = L.kwage_counting_filter_log2(num_bp)
print("reads %d x %d = %.1f M bases, counting filters 2^%d, min_kmer_count %d" % (n_reads, args.read_len, num_bp / 1e6, logc, args.min_count))

with ka.Context(0) as ctx:
    for rep in range(2):
        bc = BloomCounter(ctx, 31, args.min_count, logc, 32)
        ctx.sync()
        t0 = time.perf_counter()
        ka.native.check(L.kwage_bloom_counter_add(bc._h, buf, offs.ctypes.data, n_reads))
        st = bc.stats()
        dt = time.perf_counter() - t0
        status, prm = bc.finish(0.25, 18)
        t1 = time.perf_counter() - t0
        print("device: %.3f s counting (%.1f M bases/s), %.3f s with fold; num_kmer %d, chunks %d, commit rounds %d (max %d in a chunk), committed %d of %d positions; status %d L=%d h=%d"
              % (dt, num_bp / dt / 1e6, t1, st.num_valid_kmer, st.chunks, st.rounds, st.max_rounds, st.occurrences_committed, st.positions, status, prm.log_2_filter_len, prm.num_hash))
        if rep == 0:
            head_counts = bc.counts(0, 1 << 20)
        bc.close()

import kwage_oracle as oracle
oracle.build()
n_cpu = min(n_reads, args.cpu_bases // args.read_len)
ref = oracle.CountingPass(31, args.min_count, logc, 32)
t0 = time.perf_counter()
lib = oracle.lib()
for i in range(n_cpu):
    lib.kwo_counter_add(ref._c, buf[i * args.read_len:(i + 1) * args.read_len], args.read_len)
dt = time.perf_counter() - t0
print("cpu restatement (1 thread): %.1f M bases in %.2f s = %.1f M bases/s" % (n_cpu * args.read_len / 1e6, dt, n_cpu * args.read_len / dt / 1e6))
if n_cpu == n_reads:
    print("counters equal on the first 2^20 elements:", bool(np.array_equal(ref.counts()[:1 << 20], head_counts)), " num_kmer", ref.num_valid_kmer)
