"""grm-kmer-mi355x: MI355X-native k-mer-matrix engine (drop-in for DSK / multidsk /
dsk2kover / Ray Surveyor as GRM uses them).  See DESIGN.md."""
from . import _lib, engine
from .engine import Batch, Context, GrmError, HostMatrix, KmerSet, Matrix, decode_kmers

__all__ = ["Batch", "Context", "GrmError", "HostMatrix", "KmerSet", "Matrix", "decode_kmers", "engine", "_lib"]
