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
  popcount_vectors.json  masked popcount + row masks of the learner side: the reference's only native file,
                         learning/common/popcount.pyx, is compiled with this container's Cython into a
                         temporary directory (nothing of it enters the repo) and inplace_popcount_64 is run
                         on seeded blocks; build_row_mask (a nested function of
                         KmerRuleClassifications.sum_rows, learning/common/rules.py:210-222) is AST-extracted
                         and executed with Python-2 integer division; the column sums are what
                         rules.py:243-262 accumulates.  Inputs + outputs only.
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


def load_build_row_mask():
    """build_row_mask is defined inside sum_rows (rules.py:210-222).  The reference is Python 2: `/` between
    two ints is floor division there (true division as soon as one side is a float), so every `a / b` of the
    extracted function is evaluated by a helper with exactly that rule."""
    src = open(os.path.join(REF, "bin/kover/core/kover/learning/common/rules.py")).read()
    tree = ast.parse(src)
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "build_row_mask":
            fn = node
    assert fn is not None

    class Py2Div(ast.NodeTransformer):
        def visit_BinOp(self, node):
            self.generic_visit(node)
            if isinstance(node.op, ast.Div):
                return ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right], keywords=[])
            return node

    def _py2div(a, b):
        ints = (int, np.integer)
        return a // b if isinstance(a, ints) and isinstance(b, ints) else a / b
    fn = Py2Div().visit(fn)
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    from math import ceil
    ns = {"np": np, "ceil": ceil, "xrange": range, "_py2div": _py2div}
    exec(compile(mod, "<reference rules.py build_row_mask (in-memory)>", "exec"), ns)
    return ns["build_row_mask"]


def load_reference_popcount():
    """compile learning/common/popcount.pyx (Cython -> C -> shared object) in a temporary directory and import it"""
    import importlib.util, shutil, subprocess, sysconfig, tempfile
    d = tempfile.mkdtemp(prefix="ref_popcount_")
    pyx = os.path.join(d, "popcount.pyx")
    shutil.copy(os.path.join(REF, "bin/kover/core/kover/learning/common/popcount.pyx"), pyx)
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-o", os.path.join(d, "popcount.c"), pyx])
    so = os.path.join(d, "popcount" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
                           "-DNPY_NO_DEPRECATED_API=0", os.path.join(d, "popcount.c"), "-o", so])
    spec = importlib.util.spec_from_file_location("popcount", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, d


def popcount_vectors():
    import shutil
    build_row_mask = load_build_row_mask()
    pop, tmp = load_reference_popcount()
    try:
        rng = np.random.RandomState(20261004)
        cases = []
        for (n_genomes, n_cols, n_sel) in [(1, 4, 1), (64, 5, 20), (65, 5, 33), (130, 9, 57), (200, 6, 200), (1000, 3, 400)]:
            n_rows = (n_genomes + 63) // 64
            sel = sorted(rng.choice(n_genomes, size=n_sel, replace=False).tolist())
            mask = build_row_mask(sel, n_genomes, 64)
            assert mask.dtype == np.uint64 and mask.shape == (n_rows,)
            block = rng.randint(0, 2**63, size=(n_rows, n_cols), dtype=np.int64).astype(np.uint64) * np.uint64(2) + \
                rng.randint(0, 2, size=(n_rows, n_cols)).astype(np.uint64)
            block[0, 0] = 0                                       # the `!= 0` shortcut of the reference loop
            # bits beyond the last genome are never set in a real dataset (padding of the last word-row)
            pad = n_rows * 64 - n_genomes
            if pad:
                block[-1] &= ~np.uint64((1 << pad) - 1)
            work = block.copy()
            pop.inplace_popcount_64(work, mask)                   # popcount.pyx:76-95
            sums = work.sum(axis=0)                               # rules.py:262
            cases.append({"n_genomes": n_genomes, "selected": sel,
                          "row_mask": [str(int(x)) for x in mask],
                          "block": [[str(int(x)) for x in row] for row in block],
                          "popcounts": [[int(x) for x in row] for row in work],
                          "column_sums": [int(x) for x in sums]})
        json.dump({"source": "learning/common/popcount.pyx:76-95 compiled with Cython %s and executed; build_row_mask of "
                             "learning/common/rules.py:210-222 executed with Python-2 division" % __import__("Cython").__version__,
                   "cases": cases}, open(os.path.join(HERE, "popcount_vectors.json"), "w"), indent=0)
        return len(cases)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print("kmers:", harvest_kmers())
    print("pack cases:", pack_vectors())
    print("popcount cases:", popcount_vectors())
