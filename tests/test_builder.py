"""Database construction: the device bit-transpose builder against the
reference's own build_db() output (tests/golden/*/bloom/*.bloom -> *.db, both written by the
reference through oracle/_ref/ref_tool) and against the numpy restatement on synthetic inputs."""
import ctypes as C
import glob
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _blooms(case):
    return sorted(glob.glob(os.path.join(GOLDEN, case, "bloom", "*.bloom")))


def test_oracle_build_db_slices_match_reference(oracle):
    """numpy restatement of the transpose == the reference's file (slice block, header, index)."""
    for case, db in (("basic", "basic/db/basic.db"), ("k32", "k32/k32.db")):
        ref = open(os.path.join(GOLDEN, db), "rb").read()
        got = oracle.build_db_bytes(_blooms(case))
        hdr = oracle.DBHeader.unpack(ref)
        assert got[: hdr.info_start + 8 * hdr.num_filter] == ref[: hdr.info_start + 8 * hdr.num_filter]
        assert len(got) == len(ref)


def _device_build(ka, ctx, paths, params, out):
    from kwage_amd import native
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    st = native.BuildStats()
    p = native.Params(*params)
    native.check(native.lib().kwage_build_db(ctx._h, out.encode(), C.byref(p), arr, len(paths), C.byref(st)))
    return st


@pytest.mark.gpu
def test_device_builder_is_byte_identical_to_reference(tmp_path):
    import kwage_amd as ka
    with ka.Context(0) as ctx:
        for case, db, params in (("basic", "basic/db/basic.db", (31, 3, 12, 0)), ("k32", "k32/k32.db", (32, 5, 10, 0))):
            out = str(tmp_path / (case + ".db"))
            st = _device_build(ka, ctx, _blooms(case), params, out)
            ref = open(os.path.join(GOLDEN, db), "rb").read()
            assert open(out, "rb").read() == ref, case
            assert st.db_bytes == len(ref) and st.bits_transposed == (1 << params[2]) * len(_blooms(case))


@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(1, 3), (3, 5), (64, 7), (65, 10), (1000, 12), (1024, 11), (1025, 11), (2048, 14), (2500, 13)])
def test_device_builder_vs_oracle_random(oracle, tmp_path, n, L):
    import kwage_amd as ka
    rng = np.random.default_rng(n * 31 + L)
    paths = []
    for j in range(n):
        bits = rng.integers(0, 256, size=((1 << L) + 7) // 8, dtype=np.uint8)
        if L < 3:
            bits &= (1 << (1 << L)) - 1
        fi = oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%d" % (j + 1)),
                               sample_attributes=[("k", "v%d" % j)] if j % 3 == 0 else [],
                               experiment_title="t%d" % j if j % 5 == 0 else "")
        p = str(tmp_path / ("f%05d.bloom" % j))
        oracle.write_bloom(p, 21, L, 2, fi, bits)
        paths.append(p)
    out = str(tmp_path / "out.db")
    with ka.Context(0) as ctx:
        _device_build(ka, ctx, paths, (21, 2, L, 0), out)
        assert open(out, "rb").read() == oracle.build_db_bytes(paths)
        # and the engine can search what it built
        g = ka.Group(ctx, 21, 2, L, n)
        assert g.add_db_file(out) == (0, n)
        g.close()


@pytest.mark.gpu
def test_device_builder_rejects_bad_input(oracle, tmp_path):
    import kwage_amd as ka
    paths = _blooms("k32")
    bad = str(tmp_path / "bad.bloom")
    raw = bytearray(open(paths[0], "rb").read())
    raw[-1] ^= 0x10                                   # flip a filter bit: CRC32 mismatch (build_db.cpp:343-362)
    open(bad, "wb").write(raw)
    with ka.Context(0) as ctx:
        with pytest.raises(ka.KwageError) as e:
            _device_build(ka, ctx, [bad] + paths[1:], (32, 5, 10, 0), str(tmp_path / "o.db"))
        assert "CRC32" in str(e.value)
        assert not os.path.exists(tmp_path / "o.db")  # nothing is written before validation passes
        with pytest.raises(ka.KwageError):
            _device_build(ka, ctx, paths, (31, 5, 10, 0), str(tmp_path / "o.db"))    # inconsistent params
        raw = bytearray(open(paths[0], "rb").read())
        raw[0] = 0x00                                 # BLOOM_MAGIC_IN_PROGRESS
        open(bad, "wb").write(raw)
        with pytest.raises(ka.KwageError):
            _device_build(ka, ctx, [bad], (32, 5, 10, 0), str(tmp_path / "o.db"))


# ---------------------------------------------------------------------------------------------
# column-wise re-pack (the bit-level work of merge_db.cpp): several files -> one wide file
# ---------------------------------------------------------------------------------------------
def _repack(ctx, paths, out):
    from kwage_amd import native
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    native.check(native.lib().kwage_repack_db(ctx._h, out.encode(), arr, len(paths)))


@pytest.mark.gpu
def test_repack_is_the_column_concatenation(oracle, tmp_path):
    import subprocess
    import zlib
    import kwage_amd as ka
    from kwage_amd import native
    rng = np.random.default_rng(12)
    paths, mats, names = [], [], []
    for f, n in enumerate((13, 21, 1, 64, 7, 130)):       # ragged widths: every bit offset gets exercised
        bits = rng.random((1 << 10, ((n + 7) // 8) * 8)) < 0.3
        bits[:, n:] = False
        rows = np.packbits(bits, axis=1, bitorder="little")
        infos = [oracle.FilterInfo(run_accession=oracle.str_to_accession("ERR%d" % (100 * f + j + 1)),
                                   experiment_title="file %d col %d" % (f, j) if j % 4 == 0 else "",
                                   sample_attributes=[("a", "b"), ("c", "d")] if j % 5 == 0 else []) for j in range(n)]
        p = str(tmp_path / ("part%d.db" % f))
        oracle.write_db(p, 25, 2, 10, rows, n, infos)
        paths.append(p); mats.append(bits[:, :n]); names += [i.csv_string() for i in infos]
    z = str(tmp_path / "part1.dbz")                       # one input in the compressed container
    native.check(native.lib().kwage_db_compress(paths[1].encode(), z.encode(), 2))
    inputs = [paths[0], z] + paths[2:]
    out = str(tmp_path / "wide.db")
    with ka.Context(0) as ctx:
        _repack(ctx, inputs, out)
        with pytest.raises(ka.KwageError):
            _repack(ctx, inputs + [os.path.join(GOLDEN, "k32", "k32.db")], str(tmp_path / "bad.db"))   # other parameters
        # a source whose slices do not match the CRC32 in its header is refused (merge_db.cpp:608-614), raw or compressed,
        # and no output is left behind
        for victim in (paths[3], z):
            raw = bytearray(open(victim, "rb").read())
            raw[-1 - len(raw) // 3 if victim == z else 44 + 500] ^= 0x10
            flipped = str(tmp_path / ("flipped" + os.path.splitext(victim)[1]))
            open(flipped, "wb").write(raw)
            bad_out = str(tmp_path / "bad_crc.db")
            with pytest.raises(ka.KwageError) as ei:
                _repack(ctx, [paths[0], flipped], bad_out)
            if victim != z:
                assert "Invalid CRC32 value for source database file" in str(ei.value)
            assert not os.path.exists(bad_out)
        # ... and so does the loader when asked to check (KWAGE_VERIFY_CRC=1; the knob is read once per process)
        code = ("import sys; sys.path.insert(0, %r)\nimport kwage_amd as ka\n"
                "with ka.Context(0) as ctx:\n    g = ka.Group(ctx, 25, 2, 10, 4096)\n    print(g.add_db_file(sys.argv[1]))\n" % ROOT)
        flipped_raw = str(tmp_path / "flipped.db")
        for env, path, ok in (({"KWAGE_VERIFY_CRC": "1"}, paths[3], True), ({"KWAGE_VERIFY_CRC": "1"}, flipped_raw, False),
                              ({}, flipped_raw, True)):
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=dict(os.environ, **env))
            assert (r.returncode == 0) == ok, r.stderr[-800:]
            if not ok:
                assert "Invalid CRC32 value" in r.stderr
    wide = oracle.read_db(out)
    total = sum(m.shape[1] for m in mats)
    assert wide.header.num_filter == total and wide.header.compression == 0
    exp = np.concatenate(mats, axis=1)
    got = np.unpackbits(wide.rows, axis=1, bitorder="little")[:, :total].astype(bool)
    assert np.array_equal(got, exp)
    assert np.all(np.unpackbits(wide.rows, axis=1, bitorder="little")[:, total:] == 0)       # pad bits stay zero
    assert zlib.crc32(wide.rows.tobytes()) & 0xFFFFFFFF == wide.header.crc32
    assert [wide.info(j).csv_string() for j in range(total)] == names
    assert wide.info(0).experiment_title == "file 0 col 0" and len(wide.info(13).sample_attributes) == 2
    # the REFERENCE binary reads the wide file and reports what it reports on the parts
    if os.access(oracle.REF_KWAGE, os.X_OK):
        q = str(tmp_path / "q.fa")
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        with open(q, "w") as fh:
            for i in range(6):
                fh.write(">q%d\n%s\n" % (i, acgt[rng.integers(0, 4, size=60)].tobytes().decode()))
        parts_dir = tmp_path / "parts"; parts_dir.mkdir()
        for p in paths:
            os.link(p, parts_dir / os.path.basename(p))
        for thr in ("0.2", "0.05"):
            run = lambda d: subprocess.run([oracle.REF_KWAGE, "-d", d, "-i", q, "-t", thr, "--o.csv"], capture_output=True, text=True,
                                           env=dict(os.environ, OMP_NUM_THREADS="1"))
            a, b = run(str(parts_dir)), run(out)
            assert a.returncode == 0 and b.returncode == 0
            assert sorted(a.stdout.splitlines()) == sorted(b.stdout.splitlines()) and len(a.stdout.splitlines()) > 1


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [1, 2, 3])
def test_every_transpose_tile_shape_builds_the_same_file(oracle, tmp_path, shape):
    """KWAGE_BUILD_TILE (read once per process) selects the transpose kernel's tile: 1300 filters x 2^12 bits cover partial
    filter tiles, row tiles and an output row that is not a multiple of 16 bytes, for the three shapes that are not the default."""
    import subprocess
    from kwage_amd import native
    tool = os.path.join(os.path.dirname(native.KWAGE_BIN), "kwage_dbtool")
    rng = np.random.default_rng(77)
    paths = []
    for j in range(1300):
        p = str(tmp_path / ("f%04d.bloom" % j))
        oracle.write_bloom(p, 21, 12, 1, oracle.FilterInfo(run_accession=oracle.str_to_accession("SRR%d" % (j + 1))),
                           rng.integers(0, 256, size=512, dtype=np.uint8))
        paths.append(p)
    out = str(tmp_path / "out.db")
    r = subprocess.run([tool, "build", out, "21", "12", "1"] + paths, capture_output=True, text=True, env=dict(os.environ, KWAGE_BUILD_TILE=str(shape)))
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == oracle.build_db_bytes(paths)


@pytest.mark.gpu
def test_dbtool_front_end(tmp_path):
    """kwage_dbtool drives the same C-ABI entry points from the shell."""
    import subprocess
    from kwage_amd import native
    tool = os.path.join(os.path.dirname(native.KWAGE_BIN), "kwage_dbtool")
    run = lambda *a: subprocess.run([tool] + list(a), capture_output=True, text=True)
    out = str(tmp_path / "k32.db")
    r = run("build", out, "32", "10", "5", *_blooms("k32"))
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "k32", "k32.db"), "rb").read()
    r = run("info", out)
    assert "kmer_len\t32" in r.stdout and "num_filter\t8" in r.stdout
    assert run("accessions", out).stdout.splitlines()[2] == "2\tSRR3"
    z = str(tmp_path / "k32.dbz")
    assert run("compress", out, z, "2").returncode == 0 and run("decompress", z, str(tmp_path / "b.db")).returncode == 0
    assert open(tmp_path / "b.db", "rb").read() == open(out, "rb").read()
    a, b = os.path.join(GOLDEN, "multi/dbs/a/k31_L10_h1.db"), os.path.join(GOLDEN, "multi/dbs/a/deeper/k31_L10_h1_b.db")
    wide = str(tmp_path / "wide.db")
    assert run("repack", wide, a, b).returncode == 0
    assert "num_filter\t34" in run("info", wide).stdout
    fa = tmp_path / "g.fasta"
    fa.write_text(">c\n" + "ACGTTGCAAGGCTTAACCGGATATCGCGAT" * 20 + "\n")
    bl = str(tmp_path / "g.bloom")
    r = run("mkbloom", bl, "SRR424242", "21", "0", "0", str(fa))
    assert r.returncode == 0 and "distinct k-mers -> log_2_filter_len 18" in r.stderr, r.stderr
    assert run("build", str(tmp_path / "g.db"), "21", "18", r.stderr.split("num_hash ")[1].split()[0], bl).returncode == 0
    assert run("repack", wide, a, os.path.join(GOLDEN, "k32", "k32.db")).returncode == 1      # different parameters
    assert run("nonsense", "x").returncode == 2
