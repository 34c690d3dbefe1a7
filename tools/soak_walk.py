#!/usr/bin/env python3
"""Soak of the cut-pair protocols of the persistent gather kernels.

and_walk_kernel, and_band_walk_kernel and count_walk_kernel finish the (query, tile) pairs that their wave shares cut
through device-scope atomics ordered by `s_waitcnt vmcnt(0)` instead of release fences (kernels.hpp).
A rare mis-ordering would be a silent wrong hit list, so this hammers them: for every kernel and every wave count in
{5, 3001, 16384, 30000} -- shares from a fifth of the batch down to two or three positions, i.e. nearly every pair cut,
many of them across dozens of waves -- `--launches` searches of a ragged batch (tests/test_gpu_parity.py::
test_walk_rows_many_queries: query lengths 0 ... 300, planted windows so that parts carry surviving columns, N-runs,
too-short queries), software-pipelined through the context's two slots so that the NEXT search's k-mer stage runs beside
the gather kernel, every hit list compared with the first one (which is compared with the tiled kernel's, itself checked
against the CPU oracle once), and at the end of every setting the kernels' exchange buffers read back: all zero, or a
pair was left unfinished (kwage_ctx_scratch_nonzero).  (The C++ release / acquire form that round 4 soaked for contrast was a
knob, `walk_fences`, and went with it in round 5.)

    python tools/soak_walk.py [--launches 5000] [--out gpurun_out/soak/r05_soak.txt]

Not part of pytest (suite time).  Exit status 1 on any mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

WAVES = (5, 3001, 16384, 30000)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)].tobytes().decode()


def main():
    args = sys.argv[1:]
    launches, out_path = 5000, os.path.join(ROOT, "gpurun_out", "soak", "r05_soak.txt")
    while args:
        if args[0] == "--launches":
            launches, args = int(args[1]), args[2:]
        elif args[0] == "--out":
            out_path, args = args[1], args[2:]
        else:
            raise SystemExit("unknown option " + args[0])
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    log = open(out_path, "w")

    def say(msg):
        print(msg, flush=True)
        log.write(msg + "\n")
        log.flush()

    import kwage_amd as ka
    import kwage_oracle as oracle
    oracle.build()
    os.environ.setdefault("KWAGE_GROUP_PLACEMENT_PROBE", "0")
    rng = np.random.default_rng(20260404)
    k, nh, L, n_cols = 31, 2, 10, 40000                                   # 5 KB rows: five KiB-steps per row, one column tile
    nb = (n_cols + 7) // 8
    # density 0.75: a 48-base query (36 rows) keeps ~1 column, a 300-base one none; the planted windows keep theirs
    image = rng.integers(0, 256, size=(1 << L, nb), dtype=np.uint8) | rng.integers(0, 256, size=(1 << L, nb), dtype=np.uint8)
    image[:, -1] &= np.uint8((1 << (n_cols % 8)) - 1) if n_cols % 8 else np.uint8(255)
    genome = rand_seq(rng, 1200)
    cols = sorted({0, 1, n_cols // 2, n_cols - 130, n_cols - 1})
    grows = oracle.row_indices(oracle.unique_kmers(genome, k), k, nh, L).reshape(-1)
    for c in cols:
        image[grows, c // 8] |= np.uint8(1 << (c % 8))
    seqs = []
    for i in range(960):
        n = 40 if i % 200 == 7 else int(rng.choice([0, 30, 48, 64, 100, 150, 300]))      # (a 40-base query reports ~100 columns)
        if i % 2 == 0 and n >= 31:
            a = int(rng.integers(0, len(genome) - n + 1))
            seqs.append(genome[a:a + n])
        elif i % 11 == 0:
            seqs.append("N" * n)
        else:
            seqs.append(rand_seq(rng, n))
    say("# soak of the cut-pair protocols: %d launches per (kernel, wave count); batch of %d ragged queries (%d k-mer positions), %d columns x 2^%d rows, %d hash functions"
        % (launches, len(seqs), sum(max(len(s) - k + 1, 0) for s in seqs), n_cols, L, nh))
    bad = 0
    with ka.Context(0) as ctx:
        import bench
        fp = ctx.fingerprint()
        say("# code hash %s (bench.kernel_code_hash: kernels.hpp + kmer_device.hpp + engine_state.hpp + engine.hip); box %s (%s)" % (bench.kernel_code_hash(), fp.get("uuid"), fp.get("pci")))
        g = ka.Group(ctx, k, nh, L, n_cols)
        g.add_columns(image, n_cols)
        g.finalize()
        b = ka.Batch(ctx, seqs)
        # a second batch of other queries: its k-mer stage (and gather kernel) is what runs beside / between the soaked searches
        other = ka.Batch(ctx, [rand_seq(rng, int(rng.choice([64, 150, 300]))) for _ in range(700)])

        # references: the tiled kernels, the AND one against the oracle
        with ctx.tuning(walk=0, count_walk=0):
            ref_and = g.search(b, 1.0)
            ref_cnt = g.search(b, 0.95)
            ref_other = {1.0: g.search(other, 1.0), 0.95: g.search(other, 0.95)}
        assert ref_and.search_kernel.startswith("and_kernel<") and ref_cnt.search_kernel.startswith("count_kernel<")
        exp = [oracle.search_image(image, image.shape[1], k, nh, L, n_cols, oracle.unique_kmers(s, k), 1.0)[0] for s in seqs]
        assert ref_and.per_query() == exp
        say("# reference lists: t=1.0 %d records (== CPU oracle), t=0.95 %d records" % (len(ref_and.hits), len(ref_cnt.hits)))

        def soak(label, thr, ref, knobs, want_kernel, n):
            nonlocal bad
            t0 = time.perf_counter()
            mism = 0
            with ctx.tuning(**knobs):
                pend = []
                done = 0
                for i in range(n + 1):
                    if i < n:
                        # every fourth search is the OTHER batch: its k-mer stage overlaps the soaked gather kernel, and the
                        # soaked kernel follows a different one through the same exchange buffers
                        which = other if (i % 4 == 3) else b
                        pend.append((which, g.submit(which, thr)))
                    if len(pend) == 2 or (i == n and pend):
                        which, p = pend.pop(0)
                        r = p.collect()
                        want = ref if which is b else ref_other[thr]
                        if which is b and not r.search_kernel.startswith(want_kernel):
                            raise SystemExit("%s: expected %s, the engine chose %s" % (label, want_kernel, r.search_kernel))
                        if not (np.array_equal(r.hits, want.hits) and np.array_equal(r.num_query_kmer, want.num_query_kmer)):
                            mism += 1
                        done += 1
                while pend:
                    which, p = pend.pop(0)
                    r = p.collect()
                    want = ref if which is b else ref_other[thr]
                    if not (np.array_equal(r.hits, want.hits) and np.array_equal(r.num_query_kmer, want.num_query_kmer)):
                        mism += 1
                    done += 1
                left = ctx.scratch_nonzero()
            dirty = {kk: v for kk, v in left.items() if v}
            bad += mism + (1 if dirty else 0)
            say("%-58s launches %6d  mismatches %d  exchange buffers non-zero words %s  %.1f s"
                % (label, done, mism, dirty or "none", time.perf_counter() - t0))

        base = dict(walk=4, walk_min_rows=1, walk_max_kib=64, walk_bands=0, count_walk=1, count_walk_min_rows=1)
        for waves in WAVES:
            soak("and_walk_kernel       walk_waves=%-6d" % waves, 1.0, ref_and, dict(base, walk_waves=waves), "and_walk_kernel<", launches)
        for waves in WAVES:
            for bands in (3, 16):
                soak("and_band_walk_kernel  walk_waves=%-6d bands=%-2d" % (waves, bands), 1.0, ref_and,
                     dict(base, walk_waves=waves, walk_bands=bands, walk_bands_min_gib=0), "and_band_walk_kernel<", launches // 2)
        for waves in WAVES:
            soak("count_walk_kernel     count_walk_waves=%-6d" % waves, 0.95, ref_cnt, dict(base, count_walk_waves=waves), "count_walk_kernel<", launches)
        b.close()
        other.close()
        g.close()
    say("# %s" % ("ZERO mismatches, every exchange buffer left all zero" if bad == 0 else "%d FAILURES" % bad))
    log.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
