#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It never copies
reference source text into the repo: it (a) harvests k-mer *data* from the
reference's shipped result files and (b) executes three pure functions of
bin/kover/core/kover/utils.py (AST-extracted in memory, Python-2 `xrange`
mapped to `range`) on seeded inputs and stores only inputs + outputs.

Outputs
  canonical_31mers.txt   196 distinct 31-mers found in page/results/** (SCM rules
                         + equivalent-rule FASTA files).  These were emitted by the
                         real DSK/dsk2kover pipeline, so every one of them must be
                         the canonical form under GATB's A<C<T<G order.
  pack_vectors.json      _pack_binary_bytes_to_ints / _unpack_... / _minimum_uint_size
                         input/output pairs (utils.py:117-187).
"""
import ast, glob, json, os, re, sys
import numpy as np

REF = os.environ.get("GRM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def harvest_kmers():
    kmers = set()
    for mj in glob.glob(os.path.join(REF, "page/results/datasets/*/model.json")):
        d = json.load(open(mj))
        for r in d.get("rules", []):
            m = re.match(r"(?:Presence|Absence)\(([ACGT]+)\)", r)
            if m:
                kmers.add(m.group(1))
    for fa in glob.glob(os.path.join(REF, "page/results/datasets/*/*.fasta")):
        for line in open(fa):
            line = line.strip()
            if line and not line.startswith(">") and re.fullmatch(r"[ACGT]+", line):
                kmers.add(line)
    kmers = sorted(k for k in kmers if len(k) == 31)
    with open(os.path.join(HERE, "canonical_31mers.txt"), "w") as f:
        f.write("# 31-mers shipped in the reference's page/results/** (model.json rules, *.fasta)\n")
        f.write("# each is a canonical k-mer as emitted by DSK (GATB order A<C<T<G)\n")
        for k in kmers:
            f.write(k + "\n")
    return len(kmers)


def load_utils_functions():
    src = open(os.path.join(REF, "bin/kover/core/kover/utils.py")).read()
    tree = ast.parse(src)
    want = {"_minimum_uint_size", "_pack_binary_bytes_to_ints", "_unpack_binary_bytes_from_ints"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    mod = ast.Module(body=body, type_ignores=[])
    from math import ceil
    ns = {"np": np, "ceil": ceil, "xrange": range}
    exec(compile(mod, "<reference utils.py (in-memory)>", "exec"), ns)
    return ns


def pack_vectors():
    ns = load_utils_functions()
    rng = np.random.RandomState(20240607)
    cases = []
    for (n_rows, n_cols) in [(1, 3), (63, 4), (64, 4), (65, 4), (70, 5), (128, 2), (130, 7), (200, 3)]:
        a = (rng.rand(n_rows, n_cols) < 0.4).astype(np.uint8)
        for pack_size in (64, 32):
            b = ns["_pack_binary_bytes_to_ints"](a, pack_size)
            u = ns["_unpack_binary_bytes_from_ints"](b)
            assert (u[:n_rows] == a).all() and not u[n_rows:].any()
            cases.append({
                "n_rows": n_rows, "n_cols": n_cols, "pack_size": pack_size,
                "bits": a.tolist(),
                "packed": [[str(int(x)) for x in row] for row in b],   # decimal strings: JSON has no u64
                "unpacked_rows": int(u.shape[0]),
            })
    uint_sizes = []
    for v in [0, 1, 255, 256, 65535, 65536, 2**32 - 1, 2**32, 2**40]:
        uint_sizes.append({"max_value": str(v), "dtype": np.dtype(ns["_minimum_uint_size"](v)).name})
    json.dump({"source": "bin/kover/core/kover/utils.py:117-187 executed under Python 3",
               "pack_cases": cases, "minimum_uint_size": uint_sizes},
              open(os.path.join(HERE, "pack_vectors.json"), "w"), indent=0)
    return len(cases)


if __name__ == "__main__":
    print("kmers:", harvest_kmers())
    print("pack cases:", pack_vectors())
