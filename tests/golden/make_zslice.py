#!/usr/bin/env python3
"""Generate tests/golden/zslice/ from the REFERENCE's own slice codec (run in the build container only).

oracle/_ref/ref_tool includes the reference's slice_z.h in place; `ref_tool zslice` runs its CompressSlice
(deflateInit2 level 9, windowBits -9, memLevel 9, default strategy; "store raw unless smaller",
slice_z.h:153-267) over the slices written here and records, per slice, what it would store.  The fixtures
are DATA: the input slices and the reference codec's output.  Deterministic (fixed seed).

    make -C oracle ref && python tests/golden/make_zslice.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_tool")


def slices(rng, n, nbytes, ncol):
    """n slices of nbytes (ncol columns, pad bits of the last byte zero), a mixture of what a database holds:
    sparse rows (compressible), dense random rows (not), all-zero / all-one rows, runs and periodic patterns."""
    out = np.zeros((n, nbytes), dtype=np.uint8)
    for i in range(n):
        kind = i % 8
        if kind == 0:
            bits = rng.random(nbytes * 8) < 0.01
        elif kind == 1:
            bits = rng.random(nbytes * 8) < 0.05
        elif kind == 2:
            bits = rng.random(nbytes * 8) < 0.25
        elif kind == 3:
            bits = rng.random(nbytes * 8) < 0.5
        elif kind == 4:
            bits = np.zeros(nbytes * 8, dtype=bool) if (i // 8) % 2 == 0 else np.ones(nbytes * 8, dtype=bool)
        elif kind == 5:
            bits = np.zeros(nbytes * 8, dtype=bool)
            a = int(rng.integers(0, nbytes * 8)); bits[a:a + int(rng.integers(1, 400))] = True
        elif kind == 6:
            period = int(rng.integers(2, 40))
            bits = (np.arange(nbytes * 8) % period) < int(rng.integers(1, period))
        else:
            bits = rng.random(nbytes * 8) < 0.9
        bits = bits.copy()
        bits[ncol:] = False
        out[i] = np.packbits(bits, bitorder="little")
    return out


def main():
    if not os.path.exists(REF_TOOL):
        sys.exit("build the reference first: make -C oracle ref")
    d = os.path.join(HERE, "zslice")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(20191025)
    for n, ncol in ((256, 2048), (256, 1000), (128, 100), (64, 8)):      # slice = 256, 125, 13, 1 bytes
        nbytes = (ncol + 7) // 8
        s = slices(rng, n, nbytes, ncol)
        base = os.path.join(d, "n%d_cols%d" % (n, ncol))
        s.tofile(base + ".slices")
        subprocess.check_call([REF_TOOL, "zslice", str(nbytes), base + ".slices", base + ".z"])
        subprocess.check_call([REF_TOOL, "zinflate", str(nbytes), base + ".z", base + ".back"])
        assert open(base + ".back", "rb").read() == s.tobytes()
        os.remove(base + ".back")
        print("%s: %d slices x %d bytes -> %d bytes of records" % (os.path.basename(base), n, nbytes, os.path.getsize(base + ".z")))


if __name__ == "__main__":
    main()
