"""Randomised differential check on the GPU box: random k, genome counts and sizes, assembly shapes, abundance filters and engine knobs;
every matrix and every counted set against the CPU oracle.  python scripts/fuzz_paths.py [seconds] [seed]"""
import os
import sys
import time
from importlib import import_module

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grm_amd                                          # noqa: E402
from oracle import oracle_ctypes as orc                 # noqa: E402

synth = import_module("genomic-resistance-mapping-grm-_amd.synth")


class Mismatch(AssertionError):
    pass


def run(ctx, budget, seed, say=print):
    """cases until `budget` seconds have passed; -> number of cases.  Raises Mismatch with the case's description."""
    rng = np.random.RandomState(seed)
    t0 = time.time()
    n_cases = 0
    while time.time() - t0 < budget:
        k = int(rng.choice([19, 21, 25, 31, 32, 33, 34, 47, 63, 64, 65, 96, 128]))
        n = int(rng.choice([1, 2, 3, 7, 40, 70, 129, 150]))
        L = int(rng.choice([2_000, 20_000, 150_000])) if n <= 7 else int(rng.choice([1_500, 8_000, 25_000]))
        pg = synth.realistic(genome_len=L, seed=int(rng.randint(1 << 30)), contigs=(1, 6), indel_sites=max(1, L // 3000), n_snps=max(1, L // 80),
                             n_accessory=2, accessory_len=min(600, L // 3))
        genomes = []
        for i in range(n):
            g = pg.genome(i).tobytes()
            if rng.rand() < 0.3:
                g = g + g[: int(rng.randint(0, len(g)))]                # a repeated stretch
            if rng.rand() < 0.1:
                g = b""
            genomes.append([g])
        amin = int(rng.choice([1, 1, 2]))
        filt = bool(rng.rand() < 0.5)
        opts = {}
        if rng.rand() < 0.3:
            opts["rec_part_bits"] = int(rng.randint(0, 7))
        if rng.rand() < 0.3:              # (forced bucket counts: not so few that a bucket's k-mers fit no LDS table -- that is a loud error by design)
            lo_bits = max(3, int(np.ceil(np.log2(max(2.0, 2.2 * L / 1500.0)))))
            opts["bucket_bits"] = int(rng.randint(lo_bits, max(lo_bits + 1, 12)))
        if rng.rand() < 0.15:
            opts["records"] = 0
        if rng.rand() < 0.15:
            opts["cap_log2"] = int(rng.randint(8, 12))
        desc = "k=%d n=%d L=%d amin=%d filt=%d opts=%s" % (k, n, L, amin, filt, opts)
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            want = orc.build_matrix(genomes, k, amin, filt)
            b = ctx.batch(n)
            for gi, files in enumerate(genomes):
                for f in files:
                    b.add(gi, f)
            b.upload()
            m = b.run(k, amin, filt)
            ok = (m.kmers().shape == want["kmers"].shape and (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
                  and b.n_occurrences == want["n_occurrences"])
            if ok and rng.rand() < 0.25:
                # the matrix through the HDF5 writer (chunks deflated on the device, odd chunk widths) and back through libhdf5's inflate
                import tempfile
                kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
                cw = int(rng.choice([1, 7, 64, 1000, 100000]))
                gz = int(rng.choice([1, 4, 9]))
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, "f.kover")
                    kd.write_header(path, "contigs", "fuzz", None, None, gz, ["g%d" % i for i in range(n)], None, None, None, "nothing")
                    m.write_kover_h5(path, gz, cw)
                    rd = kd.KoverDatasetReader(path)
                    ok = bool((rd.kmer_matrix == want["matrix"]).all()) and rd.kmer_sequences == orc.decode_kmers(want["kmers"], k)
                    if not ok:
                        desc += " [through the .kover writer, chunk_cols=%d gzip=%d]" % (cw, gz)
            m.free()
            if ok and n >= 2 and rng.rand() < 0.35:
                # the staged calls, as the chunked and multi-GPU routes use them: the genomes in two batches (the second starts a new
                # word-row block only if the cut is a multiple of 64: the rows are compared genome by genome instead), both local
                # dictionaries in an accumulator, each batch filled against the merged dictionary
                cut = int(rng.randint(1, n))
                acc = ctx.dict_accum()
                parts = []
                for lo, hi in ((0, cut), (cut, n)):
                    bb = ctx.batch(hi - lo)
                    for gi in range(lo, hi):
                        for f in genomes[gi]:
                            bb.add(gi - lo, f)
                    bb.upload()
                    bb.partition(k, amin)
                    bb.local_dict()
                    acc.add(bb)
                    parts.append((lo, hi, bb))
                for lo, hi, bb in parts:
                    u = bb.set_global_dict_accum(acc, filt)
                    mm = bb.fill()
                    ok = ok and u == want["kmers"].shape[0] and (mm.kmers() == want["kmers"]).all()
                    d = mm.data()
                    for gi in range(lo, hi):
                        mine = (d[(gi - lo) // 64] >> np.uint64(63 - (gi - lo) % 64)) & np.uint64(1)
                        theirs = (want["matrix"][gi // 64] >> np.uint64(63 - gi % 64)) & np.uint64(1)
                        ok = ok and bool((mine == theirs).all())
                    mm.free()
                    bb.free()
                acc.free()
                if not ok:
                    desc += " [staged: two batches cut at %d, through the accumulator]" % cut
            if ok and k <= 32:
                b.partition_counts(k, amin)
                for gi in sorted(set([0, n // 2, n - 1])):
                    km, ct, nocc = orc.count_genome(genomes[gi], k, amin)
                    s = b.genome_set(gi)
                    ok = ok and s.occurrences == nocc and s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
                    s.free()
            b.free()
            if not ok:
                raise Mismatch("result differs from the oracle: " + desc)
        except Mismatch:
            raise
        except Exception:
            say("ERROR in case " + desc)
            raise
        finally:
            for name in opts:
                ctx.set_option(name, -1)
        n_cases += 1
        if n_cases % 10 == 0:
            say("%d cases ok (%.0f s) last: %s" % (n_cases, time.time() - t0, desc))
    return n_cases


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    with grm_amd.Context(0) as ctx:
        try:
            n_cases = run(ctx, budget, seed, say=lambda m: print(m, flush=True))
        except Mismatch as e:
            print("MISMATCH", e, flush=True)
            sys.exit(1)
    print("fuzz ok: %d cases in %.0f s (seed %d)" % (n_cases, time.time() - t0, seed))
