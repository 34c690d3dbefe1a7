"""Compressed `.dbz` container (north-star extension; the reference has a codec, slice_z.h, but no
container, so parity with it is UNPINNED -- see DESIGN.md section 7).  Validated by round trip and by an
independent Python reading of the layout; the GPU tests check that search results do not change."""
import ctypes as C
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN

from kwage_amd import native


def _sparse_db(oracle, path, n=300, L=11, seed=5):
    rng = np.random.default_rng(seed)
    rows = (rng.random((1 << L, ((n + 7) // 8) * 8)) < 0.02)
    rows[:, n:] = False
    img = np.packbits(rows, axis=1, bitorder="little")
    infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("ERR%06d" % (j + 1)),
                               experiment_title="title %d" % j if j % 7 == 0 else "") for j in range(n)]
    oracle.write_db(path, 31, 2, L, img, n, infos)
    return img


def _read_container(path):
    """Independent reader of the layout documented in include/kwage_amd.h."""
    raw = open(path, "rb").read()
    (magic, ver, crc, k, nh, lg, n, hf, comp, info_start) = struct.unpack_from("<IIIIIIIiIQ", raw, 0)
    assert magic == 0x20191025 and comp == 2
    nrows, ss = 1 << lg, (n + 7) // 8
    offs = struct.unpack_from("<%dQ" % (nrows + 1), raw, 44)
    assert offs[0] == 44 + 8 * (nrows + 1) and offs[-1] == info_start
    out = bytearray()
    n_deflated = 0
    for i in range(nrows):
        payload = raw[offs[i]:offs[i + 1]]
        if len(payload) == ss:
            out += payload
        else:
            n_deflated += 1
            d = zlib.decompressobj(-9)              # ZLIB_WINDOW_BITS, slice_z.h:9
            out += d.decompress(payload) + d.flush()
    assert zlib.crc32(bytes(out)) & 0xFFFFFFFF == crc
    return bytes(out), n_deflated, raw[info_start:]


def test_round_trip_and_layout(oracle, tmp_path):
    L = native.lib()
    sparse = str(tmp_path / "sparse.db")
    _sparse_db(oracle, sparse)
    for src in (sparse, os.path.join(GOLDEN, "basic", "db", "basic.db"), os.path.join(GOLDEN, "k32", "k32.db")):
        z = str(tmp_path / "x.dbz")
        back = str(tmp_path / "back.db")
        native.check(L.kwage_db_compress(src.encode(), z.encode(), 4))
        native.check(L.kwage_db_decompress(z.encode(), back.encode()))
        orig = open(src, "rb").read()
        assert open(back, "rb").read() == orig                      # byte-exact round trip
        db = oracle.read_db(src)
        slices, n_deflated, tail = _read_container(z)
        assert slices == db.rows.tobytes()
        if src == sparse:
            assert n_deflated > 1000 and os.path.getsize(z) < 0.85 * len(orig)
        # metadata reachable through the same reader as for raw files
        d = C.c_void_p()
        native.check(L.kwage_dbinfo_open(z.encode(), C.byref(d)))
        buf = C.create_string_buffer(64)
        native.check(L.kwage_dbinfo_csv_string(d, db.header.num_filter - 1, buf, 64))
        assert buf.value.decode() == db.info(db.header.num_filter - 1).csv_string()
        L.kwage_dbinfo_close(d)
    assert L.kwage_db_compress(z.encode(), str(tmp_path / "zz.dbz").encode(), 1) != 0      # already compressed
    bad = bytearray(open(z, "rb").read())
    bad[60] ^= 0xFF                                                                         # corrupt the offset table
    open(z, "wb").write(bad)
    assert L.kwage_db_decompress(z.encode(), back.encode()) != 0


@pytest.mark.gpu
def test_search_is_unchanged_by_compression(oracle, tmp_path):
    import kwage_amd as ka
    L = native.lib()
    sparse = str(tmp_path / "sparse.db")
    img = _sparse_db(oracle, sparse)
    z = str(tmp_path / "sparse.dbz")
    native.check(L.kwage_db_compress(sparse.encode(), z.encode(), 0))
    rng = np.random.default_rng(1)
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in (40, 80, 200)]
    with ka.Context(0) as ctx:
        res = []
        for path in (sparse, z):
            g = ka.Group(ctx, 31, 2, 11, 300)
            assert g.add_db_file(path) == (0, 300)
            g.finalize()
            assert np.array_equal(g.read_rows(np.arange(1 << 11))[:, :img.shape[1]], img)   # inflated correctly into HBM
            b = ka.Batch(ctx, seqs)
            res.append(g.search(b, 0.01))
            b.close(); g.close()
        assert np.array_equal(res[0].hits, res[1].hits) and len(res[0].hits) > 0
    # CLI: a directory holding only the .dbz gives the same report as the .db
    d1, d2 = tmp_path / "raw", tmp_path / "packed"
    d1.mkdir(); d2.mkdir()
    shutil.copy(os.path.join(GOLDEN, "basic", "db", "basic.db"), d1 / "basic.db")
    native.check(L.kwage_db_compress(str(d1 / "basic.db").encode(), str(d2 / "basic.dbz").encode(), 0))
    q = os.path.join(GOLDEN, "basic", "q.fa")
    outs = [subprocess.run([native.KWAGE_BIN, "-d", str(d), "-i", q, "-t", "0.8", "--o.json"], capture_output=True) for d in (d1, d2)]
    assert outs[0].returncode == 0 and outs[1].returncode == 0, outs[1].stderr
    assert outs[0].stdout == outs[1].stdout
