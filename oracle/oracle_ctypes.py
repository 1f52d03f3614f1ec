"""ctypes binding of oracle/libgrm_oracle.so.  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcSet(C.Structure):
    _fields_ = [("kmers", C.POINTER(C.c_uint64)), ("counts", C.POINTER(C.c_uint32)),
                ("n", C.c_size_t), ("words", C.c_int), ("k", C.c_int),
                ("n_occurrences", C.c_uint64)]


class OrcMatrix(C.Structure):
    _fields_ = [("kmers", C.POINTER(C.c_uint64)), ("matrix", C.POINTER(C.c_uint64)),
                ("n_genomes_with", C.POINTER(C.c_uint32)), ("n_kmers", C.c_size_t),
                ("n_rows", C.c_size_t), ("n_genomes", C.c_int), ("words", C.c_int),
                ("k", C.c_int)]


def build(force=False):
    if os.environ.get("GRM_ORACLE_SANITIZED") == "1":        # tests/test_sanitizers.py: the AddressSanitizer / UBSan build (oracle/Makefile: asan)
        subprocess.check_call(["make", "-C", _HERE, "-s", "asan"])
        return os.path.join(_HERE, "libgrm_oracle_asan.so")
    so = os.path.join(_HERE, "libgrm_oracle.so")
    src = os.path.join(_HERE, "grm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgrm_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_count_buffers.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int,
                                        C.c_int, C.c_uint32, C.POINTER(OrcSet)]
        L.orc_count_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_uint32,
                                      C.POINTER(OrcSet)]
        L.orc_build_matrix.argtypes = [C.POINTER(OrcSet), C.c_int, C.c_int, C.POINTER(OrcMatrix)]
        L.orc_canonical_ascii.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_decode.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_char_p]
        L.orc_pack_bits.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_minimum_uint_bytes.argtypes = [C.c_uint64]
        L.orc_write_tsv.argtypes = [C.POINTER(OrcMatrix), C.POINTER(C.c_char_p), C.c_char_p]
        L.orc_pipeline_buffers.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int,
                                           C.c_int, C.c_uint32, C.c_int, C.c_int,
                                           C.POINTER(OrcMatrix), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.orc_set_free.argtypes = [C.POINTER(OrcSet)]
        L.orc_matrix_free.argtypes = [C.POINTER(OrcMatrix)]
        _LIB = L
    return _LIB


def _set_to_np(s):
    n, w = s.n, s.words
    km = np.ctypeslib.as_array(s.kmers, shape=(n * w,)).copy().reshape(n, w) if n else np.zeros((0, w), np.uint64)
    ct = np.ctypeslib.as_array(s.counts, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
    return km, ct


def count_genome(buffers, k, abundance_min=1):
    """buffers: list[bytes] (file images of one genome) -> (kmers[n,words] u64, counts u32, n_occ)"""
    L = lib()
    arr = (C.c_char_p * len(buffers))(*buffers)
    lens = (C.c_size_t * len(buffers))(*[len(b) for b in buffers])
    s = OrcSet()
    rc = L.orc_count_buffers(arr, lens, len(buffers), k, abundance_min, C.byref(s))
    if rc:
        raise RuntimeError("orc_count_buffers rc=%d" % rc)
    km, ct = _set_to_np(s)
    nocc = int(s.n_occurrences)
    L.orc_set_free(C.byref(s))
    return km, ct, nocc


def build_matrix(genomes, k, abundance_min=1, filter_singleton=False):
    """genomes: list of list[bytes].  -> dict(kmers[U,words], matrix[rows,U], n_genomes_with[U])"""
    L = lib()
    sets = (OrcSet * max(1, len(genomes)))()
    keep = []
    for g, bufs in enumerate(genomes):
        arr = (C.c_char_p * len(bufs))(*bufs)
        lens = (C.c_size_t * len(bufs))(*[len(b) for b in bufs])
        keep.append((arr, lens))
        rc = L.orc_count_buffers(arr, lens, len(bufs), k, abundance_min, C.byref(sets[g]))
        if rc:
            raise RuntimeError("orc_count_buffers rc=%d" % rc)
    m = OrcMatrix()
    rc = L.orc_build_matrix(sets, len(genomes), 1 if filter_singleton else 0, C.byref(m))
    if rc:
        raise RuntimeError("orc_build_matrix rc=%d" % rc)
    out = matrix_to_np(m)
    out["per_genome"] = [_set_to_np(sets[g]) for g in range(len(genomes))]
    out["n_occurrences"] = sum(int(sets[g].n_occurrences) for g in range(len(genomes)))
    for g in range(len(genomes)):
        L.orc_set_free(C.byref(sets[g]))
    L.orc_matrix_free(C.byref(m))
    return out


def matrix_to_np(m):
    U, w, r = m.n_kmers, m.words, m.n_rows
    return {
        "kmers": np.ctypeslib.as_array(m.kmers, shape=(U * w,)).copy().reshape(U, w) if U else np.zeros((0, w), np.uint64),
        "matrix": np.ctypeslib.as_array(m.matrix, shape=(r * U,)).copy().reshape(r, U) if U and r else np.zeros((r, U), np.uint64),
        "n_genomes_with": np.ctypeslib.as_array(m.n_genomes_with, shape=(U,)).copy() if U else np.zeros(0, np.uint32),
        "n_genomes": m.n_genomes, "k": m.k, "words": w,
    }


def pipeline(buffers, k, abundance_min, filter_singleton, n_threads):
    """one buffer per genome; returns (matrix dict, count_s, merge_s, n_occurrences)"""
    L = lib()
    n = len(buffers)
    ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(b), C.c_void_p) for b in buffers])
    lens = (C.c_size_t * n)(*[len(b) for b in buffers])
    m = OrcMatrix()
    cs, ms, occ = C.c_double(), C.c_double(), C.c_uint64()
    rc = L.orc_pipeline_buffers(ptrs, lens, n, k, abundance_min, 1 if filter_singleton else 0,
                                n_threads, C.byref(m), C.byref(cs), C.byref(ms), C.byref(occ))
    if rc:
        raise RuntimeError("orc_pipeline_buffers rc=%d" % rc)
    out = matrix_to_np(m)
    L.orc_matrix_free(C.byref(m))
    return out, cs.value, ms.value, int(occ.value)


def canonical_ascii(s, k=None):
    L = lib()
    k = k or len(s)
    out = (C.c_uint64 * 4)()
    rc = L.orc_canonical_ascii(s.encode(), k, out)
    if rc:
        return None
    buf = C.create_string_buffer(k)
    L.orc_decode(out, k, buf)
    return buf.raw[:k].decode(), [int(out[i]) for i in range((k + 31) // 32)]


def pack_bits(bits, pack_size):
    L = lib()
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    n_rows, n_cols = bits.shape
    out = np.zeros(((n_rows + pack_size - 1) // pack_size, n_cols), dtype=np.uint64)
    L.orc_pack_bits(bits.ctypes.data, n_rows, n_cols, pack_size, out.ctypes.data)
    return out


def minimum_uint_bytes(v):
    return lib().orc_minimum_uint_bytes(v)


def decode_kmers(kmers, k):
    """kmers [n,words] u64 -> list[str]"""
    L = lib()
    out = []
    buf = C.create_string_buffer(k)
    for row in np.ascontiguousarray(kmers, dtype=np.uint64):
        arr = (C.c_uint64 * len(row))(*[int(x) for x in row])
        L.orc_decode(arr, k, buf)
        out.append(buf.raw[:k].decode())
    return out
