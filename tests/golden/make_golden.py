#!/usr/bin/env python3
"""Generate tests/golden/ from the REFERENCE ITSELF (run in the build container only).

Uses oracle/_ref/ref_tool (reference BloomFilter + build_db linked by a small driver) to
write real `.db` files, and oracle/_ref/kwage (the reference search binary) to produce the
expected CSV / JSON for every case.  Build both first:  make -C oracle ref

Everything written here is DATA (inputs + the reference's outputs); no reference source is
copied.  Re-running is deterministic (fixed seeds), except that `readdir` order decides the
order of equal-score hits, which the parity tests ignore.

    python tests/golden/make_golden.py
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_KWAGE = os.path.join(ROOT, "oracle", "_ref", "kwage")
REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_tool")


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def mkdb(out_db, k, L, nhash, filters, keep_bloom_dir=None):
    """filters: list of dicts {acc, seed, noise, seqs:[...], meta:{}, attrs:[(k,v)], n:(spots,bases), date}
    keep_bloom_dir: also keep the reference-written .bloom files (inputs of the builder parity test)."""
    os.makedirs(os.path.dirname(out_db), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="kwgold_")
    spec = os.path.join(tmp, "spec.tsv")
    with open(spec, "w") as f:
        f.write("DB\t%s\t%d\t%d\t%d\t%s\n" % (out_db, k, L, nhash, tmp))
        if keep_bloom_dir:
            f.write("KEEP\n")
        for fl in filters:
            f.write("F\t%s\t%d\t%d\n" % (fl["acc"], fl.get("seed", 1), fl.get("noise", 0)))
            for s in fl.get("seqs", []):
                f.write("S\t%s\n" % s)
            for key, v in fl.get("meta", {}).items():
                f.write("M\t%s\t%s\n" % (key, v))
            for key, v in fl.get("attrs", []):
                f.write("A\t%s\t%s\n" % (key, v))
            if "n" in fl:
                f.write("N\t%d\t%d\n" % fl["n"])
            if "date" in fl:
                f.write("D\t%s\n" % fl["date"])
    subprocess.check_call([REF_TOOL, "mkdb", spec])
    if keep_bloom_dir:
        shutil.rmtree(keep_bloom_dir, ignore_errors=True)
        os.makedirs(keep_bloom_dir)
        for fn in sorted(os.listdir(tmp)):
            if fn.endswith(".bloom"):
                shutil.move(os.path.join(tmp, fn), os.path.join(keep_bloom_dir, fn))
    shutil.rmtree(tmp)


def run_ref(args, cwd):
    """Run the reference kwage; returns stdout text."""
    r = subprocess.run([REF_KWAGE] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    if r.returncode != 0:
        raise RuntimeError("reference kwage failed: %s\n%s" % (args, r.stderr.decode()))
    return r.stdout.decode("latin-1")


def main():
    if not (os.path.exists(REF_KWAGE) and os.path.exists(REF_TOOL)):
        sys.exit("build the reference first: make -C oracle ref")
    rng = random.Random(20191025)
    manifest = {"cases": []}

    def add_case(name, db_args, query_files, cmdline_seqs, thresholds, formats=("csv", "json"), tag=""):
        cdir = os.path.join(HERE, name)
        for t in thresholds:
            for fmt in formats:
                args = []
                for d in db_args:
                    args += ["-d", d]
                for q in query_files:
                    args += ["-i", q]
                args += ["-t", t, "--o." + fmt] + list(cmdline_seqs)
                out = run_ref(args, cdir)
                fn = "expected%s_t%s.%s" % (tag, t, fmt)
                with open(os.path.join(cdir, fn), "w", encoding="latin-1") as f:
                    f.write(out)
                manifest["cases"].append({"name": name, "db": db_args, "queries": query_files,
                                          "cmdline": list(cmdline_seqs), "threshold": t,
                                          "format": fmt, "expected": fn})

    # ------------------------------------------------------------------------------
    # case "basic": k=31, L=12, 3 hashes, 100 filters (N % 8 != 0), rich metadata
    # ------------------------------------------------------------------------------
    name = "basic"
    cdir = os.path.join(HERE, name)
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    genomes = [rand_seq(rng, 400) for _ in range(6)]
    filters = []
    for j in range(100):
        fl = {"acc": "SRR%07d" % (1000 + j), "seed": 100 + j, "noise": 250 + 13 * (j % 7), "seqs": []}
        for g, gen in enumerate(genomes):
            if (j * 7 + g * 3) % 11 == 0:
                fl["seqs"].append(gen)
            elif (j + g) % 17 == 0:
                fl["seqs"].append(gen[:200])  # half the genome: partial matches for t<1
        if j % 9 == 0:
            fl["meta"] = {"experiment_accession": "SRX%06d" % j, "sample_accession": "SRS%05d" % j,
                          "study_accession": "SRP%04d" % j, "experiment_title": "title %d" % j,
                          "experiment_design_description": "design, with comma",
                          "experiment_library_name": "lib%d" % j,
                          "experiment_library_strategy": "WGS", "experiment_library_source": "GENOMIC",
                          "experiment_library_selection": "RANDOM",
                          "experiment_instrument_model": "Illumina HiSeq 2000",
                          "sample_taxa": "Escherichia coli", "study_title": "study %d" % j,
                          "study_abstract": "abstract text %d" % j}
            fl["attrs"] = [("strain", "K-12"), ("host", "Homo sapiens"), ("collection_date", "2019"),
                           ("geo_loc_name", "USA"), ("isolation_source", "gut")][: 1 + j % 5]
            fl["n"] = (1000 + j, 150000 + j)
            fl["date"] = "2019-%02d-%02dT03:10:22Z" % (1 + j % 12, 1 + j % 28)
        elif j % 9 == 1:
            fl["date"] = "2010-03-24"
        filters.append(fl)
    mkdb(os.path.join(cdir, "db", "basic.db"), 31, 12, 3, filters, keep_bloom_dir=os.path.join(cdir, "bloom"))

    with open(os.path.join(cdir, "q.fa"), "w") as f:
        f.write(">g0 full genome 0\n")
        for i in range(0, 400, 70):
            f.write(genomes[0][i:i + 70] + "\n")
        f.write(">g1_rc reverse complement of genome 1 window\n" + revcomp(genomes[1][50:350]) + "\n")
        f.write(">g2_lower lower case\n" + genomes[2][:250].lower() + "\n")
        f.write(">g3_withN genome 3 with N\n" + genomes[3][:120] + "N" + genomes[3][121:300] + "\n")
        f.write(">short too short for k\n" + genomes[0][:30] + "\n")
        f.write(">random no match\n" + rand_seq(rng, 300) + "\n")
        f.write(">dup repeated kmers\n" + genomes[4][:100] + genomes[4][:100] + genomes[4][:100] + "\n")
        f.write(">chimera half genome 5 half random\n" + genomes[5][:150] + rand_seq(rng, 150) + "\n")
        f.write(">polyA\n" + "A" * 80 + "\n")
        f.write(">iupac RYKM\n" + genomes[2][:40] + "RYKM" + genomes[2][44:120] + "\n")
    add_case(name, ["db"], ["q.fa"], [genomes[1][:100], "ACGTNACGT"],
             ["1.0", "0.8", "0.5", "0.05", "0.0001"])

    # ------------------------------------------------------------------------------
    # case "multi": several files, different (k, L, num_hash), nested dirs, FASTQ + gz
    # ------------------------------------------------------------------------------
    name = "multi"
    cdir = os.path.join(HERE, name)
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    g = [rand_seq(rng, 300) for _ in range(4)]
    mkdb(os.path.join(cdir, "dbs", "a", "k31_L10_h1.db"), 31, 10, 1,
         [{"acc": "ERR%06d" % (1 + j), "seed": j, "noise": 60,
           "seqs": [g[j % 4]] if j % 3 == 0 else []} for j in range(13)])
    mkdb(os.path.join(cdir, "dbs", "a", "deeper", "k31_L10_h1_b.db"), 31, 10, 1,
         [{"acc": "DRR%06d" % (100 + j), "seed": 50 + j, "noise": 40,
           "seqs": [g[(j + 1) % 4]] if j % 2 == 0 else []} for j in range(21)])
    mkdb(os.path.join(cdir, "dbs", "b", "k15_L11_h2.DB"), 15, 11, 2,
         [{"acc": "SRR%09d" % (5 + j), "seed": 200 + j, "noise": 100,
           "seqs": [g[j % 4][:100]] if j % 5 == 0 else []} for j in range(64)])
    mkdb(os.path.join(cdir, "dbs", "k31_L12_h3.db"), 31, 12, 3,
         [{"acc": "SRR%05d" % (7 + j), "seed": 300 + j, "noise": 500,
           "seqs": [g[3]] if j in (0, 8) else []} for j in range(9)])
    with open(os.path.join(cdir, "dbs", "not_a_db.txt"), "w") as f:
        f.write("ignored by the .db filter\n")
    with open(os.path.join(cdir, "reads.fastq"), "w") as f:
        for i in range(6):
            s = g[i % 4][10 * i: 10 * i + 150] if i != 4 else rand_seq(rng, 150)
            f.write("@read%d some description\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    with open(os.path.join(cdir, "contigs.fa.gz"), "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:     # mtime=0: reproducible bytes
            f.write((">c0\n" + g[0] + "\n>c1\n" + g[3][:200] + "\n").encode())
    add_case(name, ["dbs"], ["reads.fastq", "contigs.fa.gz"], [], ["1.0", "0.7"])
    add_case(name, ["dbs/a", "dbs/k31_L12_h3.db"], ["contigs.fa.gz"], [g[3][20:90]], ["1.0"],
             formats=("csv",), tag="_subset")

    # ------------------------------------------------------------------------------
    # case "k32": k=32, 5 hashes, N=8 exactly (one full byte)
    # ------------------------------------------------------------------------------
    name = "k32"
    cdir = os.path.join(HERE, name)
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    g32 = rand_seq(rng, 200)
    mkdb(os.path.join(cdir, "k32.db"), 32, 10, 5,
         [{"acc": "SRR%d" % (1 + j), "seed": j, "noise": 150,
           "seqs": [g32] if j in (2, 5) else ([g32[:120]] if j == 7 else [])} for j in range(8)],
         keep_bloom_dir=os.path.join(cdir, "bloom"))
    with open(os.path.join(cdir, "q.fna"), "w") as f:
        f.write(">one\n" + g32[:150] + "\n")
    add_case(name, ["k32.db"], ["q.fna"], [], ["1.0", "0.6"])

    # ------------------------------------------------------------------------------
    # known-answer vectors for k-mer packing + hashing, straight from the reference code
    # ------------------------------------------------------------------------------
    kat = []
    kat_inputs = [(31, 5, "ACGTACGTACGTACGTACGTACGTACGTACG"), (31, 5, "T" * 31),
                  (31, 5, "GATTACAGATTACAGATTACAGATTACAGAT"), (32, 5, "ACGTTGCAACGTTGCAACGTTGCAACGTTGCA"),
                  (15, 5, "ACGTTGCAACGTTGC"), (31, 1, "ACGTN" + "ACGT" * 8 + "AC"),
                  (1, 2, "ACGT"), (2, 2, "ACGTA"), (3, 3, "acgtnACGT"), (4, 1, "ACGTACGT"),
                  (5, 4, rand_seq(rng, 12)), (7, 5, rand_seq(rng, 20)), (16, 5, rand_seq(rng, 40)),
                  (21, 5, rand_seq(rng, 50)), (29, 5, rand_seq(rng, 60)), (30, 5, rand_seq(rng, 60)),
                  (31, 5, rand_seq(rng, 100)), (32, 5, rand_seq(rng, 100)),
                  (31, 2, rand_seq(rng, 40) + "x" + rand_seq(rng, 40) + "-" + rand_seq(rng, 30))]
    for k, nh, s in kat_inputs:
        out = subprocess.check_output([REF_TOOL, "kmers", str(k), str(nh), s]).decode()
        rows = [ln.split("\t") for ln in out.splitlines()]
        kat.append({"k": k, "num_hash": nh, "seq": s,
                    "kmers": [{"pos": int(r[0]), "canon": r[1], "hash": r[2:]} for r in rows]})
    with open(os.path.join(HERE, "kat_kmers.json"), "w") as f:
        json.dump(kat, f, indent=0)

    acc = []
    for s in ["SRR1234567", "ERR000001", "DRR9999999999", "srr42", "SRX0000010", "ZZZ1"]:
        out = subprocess.check_output([REF_TOOL, "accession", s]).decode().split()
        acc.append({"str": s, "packed": out[0], "round_trip": out[1]})
    with open(os.path.join(HERE, "kat_accession.json"), "w") as f:
        json.dump(acc, f, indent=0)

    # optimal_bloom_param (bloom.cpp:10-68) over a sweep of k-mer counts / bounds
    opt = []
    for p in ("0.25", "0.05", "0.5"):
        for nk in [1, 2, 10, 1000, 75000, 75366, 100000, 150733, 301467, 1000000, 4800000, 5_000_000, 123456789,
                   3_000_000_000, 10_000_000_000]:
            for lo, hi in ((18, 32), (8, 20)):
                out = subprocess.check_output([REF_TOOL, "param", "31", str(nk), p, str(lo), str(hi)]).decode().split()
                opt.append({"num_kmer": nk, "p": p, "min": lo, "max": hi, "result": out})
    with open(os.path.join(HERE, "kat_optimal_bloom_param.json"), "w") as f:
        json.dump(opt, f, indent=0)

    # ------------------------------------------------------------------------------
    # case "bloomgen": filters that hold ONLY the k-mers of given sequences (no noise), kept as
    # .bloom files: what make_bloom_filter yields at min_kmer_count == 1 -- inputs for the device
    # Bloom-construction parity test
    # ------------------------------------------------------------------------------
    name = "bloomgen"
    cdir = os.path.join(HERE, name)
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    samples = []
    for j in range(6):
        seqs = [rand_seq(rng, int(n)) for n in ([40, 700, 31][: 1 + j % 3] + ([70000] if j == 4 else []))]
        if j == 2:
            seqs.append(seqs[0][:20] + "NNN" + seqs[0][20:].lower())
        fl = {"acc": "ERR%07d" % (j + 1), "seed": 0, "noise": 0, "seqs": seqs}
        if j % 2 == 0:
            fl["meta"] = {"experiment_accession": "ERX%05d" % j, "sample_accession": "ERS%05d" % j,
                          "study_accession": "ERP%04d" % j, "experiment_title": "bloomgen %d" % j,
                          "sample_taxa": "Bacillus subtilis", "study_title": "study"}
            fl["attrs"] = [("strain", "168"), ("host", "soil"), ("note", "x y z")][: 1 + j % 3]
            fl["n"] = (10 + j, 1000 + j)
            fl["date"] = "2021-0%d-1%d" % (1 + j, j)
        samples.append(fl)
    mkdb(os.path.join(cdir, "bloomgen.db"), 27, 16, 4, samples, keep_bloom_dir=os.path.join(cdir, "bloom"))
    with open(os.path.join(cdir, "samples.json"), "w") as f:
        json.dump({"kmer_len": 27, "log_2_filter_len": 16, "num_hash": 4, "samples": samples}, f, indent=0)

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", len(manifest["cases"]), "expected outputs")


if __name__ == "__main__":
    main()
