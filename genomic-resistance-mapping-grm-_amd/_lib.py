"""ctypes binding of libgrmkmer.so (C ABI: include/grm_kmer.h).

The shared library is the product; this module only declares prototypes.  There is no
CPU fallback: if the library is missing it must be built (build_library), and every
compute call needs a HIP device.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgrmkmer.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/grm_kmer.h declares: (name, restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_U64P = C.POINTER(C.c_uint64)
PROTOTYPES = [
    ("grm_create", _P, [C.c_int, C.c_int]),
    ("grm_destroy", None, [_P]),
    ("grm_ctx_live_handles", C.c_int, [_P]),
    ("grm_last_error", C.c_char_p, [_P]),
    ("grm_version", C.c_char_p, []),
    ("grm_set_option", C.c_int, [_P, C.c_char_p, C.c_int]),
    ("grm_timing_enable", C.c_int, [_P, C.c_int]),
    ("grm_timing_reset", C.c_int, [_P]),
    ("grm_timing_count", C.c_int, [_P]),
    ("grm_timing_get", C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), _U64P]),
    ("grm_count_genome", C.c_int, [_P, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_uint32, _PP]),
    ("grm_count_genome_buffers", C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_uint32, _PP]),
    ("grm_kmer_set_from_host", C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, _PP]),
    ("grm_merge_counted_sets", C.c_int, [_P, _PP, C.c_int, C.c_uint32, _PP]),
    ("grm_kmer_set_size", C.c_size_t, [_P]),
    ("grm_kmer_set_k", C.c_int, [_P]),
    ("grm_kmer_set_words", C.c_int, [_P]),
    ("grm_kmer_set_occurrences", C.c_uint64, [_P]),
    ("grm_kmer_set_kmers", _U64P, [_P]),
    ("grm_kmer_set_counts", C.POINTER(C.c_uint32), [_P]),
    ("grm_kmer_set_free", None, [_P]),
    ("grm_build_matrix", C.c_int, [_P, _PP, C.c_int, C.c_int, _PP]),
    ("grm_matrix_n_kmers", C.c_size_t, [_P]),
    ("grm_matrix_n_rows", C.c_size_t, [_P]),
    ("grm_matrix_n_genomes", C.c_int, [_P]),
    ("grm_matrix_k", C.c_int, [_P]),
    ("grm_matrix_words", C.c_int, [_P]),
    ("grm_matrix_kmers", _U64P, [_P]),
    ("grm_matrix_data", _U64P, [_P]),
    ("grm_matrix_dev_kmers", _P, [_P]),
    ("grm_matrix_dev_data", _P, [_P]),
    ("grm_matrix_column_counts", C.c_int, [_P, _P]),
    ("grm_matrix_sum_rows", C.c_int, [_P, _P, _P]),
    ("grm_matrix_from_host", C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _PP]),
    ("grm_matrix_to_device", C.c_int, [_P, _P]),
    ("grm_matrix_risk_errors", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P]),
    ("grm_matrix_risk_index", C.c_int, [_P, _P, _P, C.c_uint32, _P, _P]),
    ("grm_matrix_last_error", C.c_char_p, [_P]),
    ("grm_matrix_free", None, [_P]),
    ("grm_write_tsv", C.c_int, [_P, C.POINTER(C.c_char_p), C.c_char_p]),
    ("grm_write_tsv_slice", C.c_int, [_P, C.POINTER(C.c_char_p), C.c_char_p, C.c_uint64, C.c_uint64]),
    ("grm_write_kover_h5", C.c_int, [_P, C.c_char_p, C.c_int, C.c_int]),
    ("grm_matrix_deflate_rows", C.c_int, [_P, C.c_int, _PP, _PP, _PP, _U64P]),
    ("grm_matrix_deflate_kmer_strings", C.c_int, [_P, C.c_int, _PP, _PP, _PP, _U64P]),
    ("grm_host_free", None, [_P]),
    ("grm_write_kover_h5_parts", C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_int, _PP, _PP, _PP, _U64P, _U64P]),
    ("grm_batch_create", C.c_int, [_P, C.c_int, _PP]),
    ("grm_batch_add", C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    ("grm_batch_add_file", C.c_int, [_P, C.c_int, C.c_char_p]),
    ("grm_batch_upload", C.c_int, [_P]),
    ("grm_batch_run", C.c_int, [_P, C.c_int, C.c_uint32, C.c_int, _PP]),
    ("grm_batch_partition", C.c_int, [_P, C.c_int, C.c_uint32]),
    ("grm_batch_partition_counts", C.c_int, [_P, C.c_int, C.c_uint32]),
    ("grm_batch_local_dict", C.c_int, [_P, _U64P]),
    ("grm_batch_export_dict", C.c_int, [_P, _P, _P]),
    ("grm_batch_set_global_dict", C.c_int, [_P, _P, _P, C.c_uint64, C.c_int, _U64P]),
    ("grm_batch_fill", C.c_int, [_P, _PP]),
    ("grm_exchange_layout", None, [C.c_uint64, C.c_int, C.c_int, _U64P, _U64P, _U64P]),
    ("grm_batch_bucket_bits", C.c_int, [_P]),
    ("grm_batch_export_dict_ordered", C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    ("grm_batch_export_dict_record", C.c_int, [_P, _P, C.c_uint64, C.c_int, C.POINTER(C.c_int)]),
    ("grm_batch_set_global_dict_gathered", C.c_int, [_P, _P, C.c_int, C.c_uint64, _U64P, C.POINTER(C.c_int), C.c_int, _U64P]),
    ("grm_batch_set_global_dict_gathered_from", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint64, _U64P, C.POINTER(C.c_int), C.c_int, _U64P]),
    ("grm_dict_accum_create", C.c_int, [_P, _PP]),
    ("grm_dict_accum_add", C.c_int, [_P, _P]),
    ("grm_dict_accum_size", C.c_uint64, [_P]),
    ("grm_dict_accum_free", None, [_P]),
    ("grm_batch_set_global_dict_accum", C.c_int, [_P, _P, C.c_int, _U64P]),
    ("grm_matrix_stack_rows", C.c_int, [_PP, C.c_int, _PP]),
    ("grm_batch_n_symbols", C.c_uint64, [_P]),
    ("grm_batch_n_occurrences", C.c_uint64, [_P]),
    ("grm_batch_input_bytes", C.c_uint64, [_P]),
    ("grm_batch_n_local", C.c_uint64, [_P]),
    ("grm_batch_memo_stats", C.c_int, [_P, _U64P]),
    ("grm_batch_genome_set", C.c_int, [_P, C.c_int, _PP]),
    ("grm_batch_free", None, [_P]),
]

_lib = None


def build_library(force=False, quiet=True):
    """compile csrc/*.hip + *.cpp for gfx950 with hipcc (cross-compiles without a GPU)"""
    if force:
        subprocess.call(["make", "-C", CSRC, "-s", "clean"])
    cmd = ["make", "-C", CSRC, "-j4"] + (["-s"] if quiet else [])
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce %s" % LIB_PATH)
    return LIB_PATH


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (soname libamdhip64.so.7) and ask for it by the un-versioned name, so if the system copy
    were mapped first a later `import torch` would map a SECOND runtime and find no GPUs.
    Mapping torch's copy first (when torch is installed) makes both users share it; without
    torch the loader falls through to /opt/rocm's copy via libgrmkmer.so's RUNPATH."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the library and attach prototypes.  Raises (never falls back) if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with __graft_entry__.build() / `make -C %s` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % (LIB_PATH, CSRC))
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in PROTOTYPES:
            fn = getattr(L, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
