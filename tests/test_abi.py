"""The C-ABI library loads without a GPU and exports every symbol include/grm_kmer.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "grm_kmer.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import grm_amd
    if not os.path.exists(grm_amd._lib.LIB_PATH):
        grm_amd._lib.build_library()
    raw = C.CDLL(grm_amd._lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(raw, n), "missing export: " + n
    bound = {p[0] for p in grm_amd._lib.PROTOTYPES}
    assert set(names) == bound, (set(names) ^ bound)
    grm_amd._lib.load()


def test_no_cpu_fallback_without_device():
    import grm_amd
    L = grm_amd._lib.load()
    assert L.grm_create(-1, 1) is None          # "CPU mode" does not exist
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        with pytest.raises(grm_amd.GrmError):
            grm_amd.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "genomic-resistance-mapping-grm-_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("import oracle", "from oracle", "grm_oracle", "oracle_ctypes", "pyoracle", "oracle/"):
                    assert needle not in src, (os.path.join(dirpath, f), needle)


def test_header_is_c99_and_a_c_client_links(tmp_path):
    """the boundary is a C ABI: include/grm_kmer.h must compile as plain C, and a C program must be able
    to link libgrmkmer.so and use it (here the GPU-free part: host-only matrix -> TSV writer)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "grm_kmer.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    import grm_amd
    so = grm_amd._lib.LIB_PATH
    assert os.path.exists(so)
    exe = str(tmp_path / "c_client")
    libdir = os.path.dirname(so)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-o", exe, os.path.join(root, "tests", "abi", "c_client.c"),
                           "-L" + libdir, "-lgrmkmer", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe, str(tmp_path / "m.tsv")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c client ok" in r.stdout

