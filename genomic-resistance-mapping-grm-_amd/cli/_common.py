"""shared helpers of the drop-in executables (dsk, multidsk, dsk2kover, Ray)"""
import os
import struct
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

MAGIC = b"GRMKSET1"


def engine():
    import grm_amd
    return grm_amd


def gatb_args(argv, known):
    """GATB-style single-dash `-flag value` pairs (kmer_count.py:28-37, kmer_pack.py:28-36, src/app.py:1372).
    known: {flag: default}.  Unknown flags with a value are accepted and ignored (DSK has many)."""
    out = dict(known)
    i = 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("-"):
            raise SystemExit("unexpected argument %r" % a)
        key = a.lstrip("-")
        if i + 1 < len(argv) and not (argv[i + 1].startswith("-") and not argv[i + 1].lstrip("-").isdigit()):
            out[key] = argv[i + 1]
            i += 2
        else:
            out[key] = "1"
            i += 1
    return out


def truthy(v):
    """Kover passes Python values through str(): '-progress True', '-verbose 0' (kmer_count.py:36-37, kmer_pack.py:36)"""
    return str(v).strip().lower() not in ("", "0", "false", "none", "no")


def die(msg, code=1):
    """return codes are ignored by Kover (kmer_count.py:28, kmer_pack.py:28): be loud on stderr too"""
    sys.stderr.write("ERROR: %s\n" % msg)
    sys.stderr.flush()
    sys.exit(code)


def write_kset(path, k, abundance_min, kmers, counts, n_occ):
    """per-genome solid k-mer set, the artefact multidsk leaves under the name Kover expects
    (<out-dir>/<stem>.h5, dataset/create.py:375,488).  The pair multidsk/dsk2kover is opaque
    to Kover, so the container is our own: magic, header, uint64 k-mers (two words each, most
    significant first, when k > 32), uint32 counts."""
    tmp = path + ".tmp"
    words = (k + 31) // 32
    kmers = np.ascontiguousarray(kmers, dtype="<u8").reshape(-1)
    if len(kmers) != words * len(counts):
        raise ValueError("k-mer array does not match k=%d (%d words per k-mer)" % (k, words))
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<IIQQ", k, abundance_min, len(counts), n_occ))
        f.write(kmers.tobytes())
        f.write(np.ascontiguousarray(counts, dtype="<u4").tobytes())
    os.replace(tmp, path)


def read_kset(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s is not a k-mer set written by this multidsk" % path)
        k, amin, n, n_occ = struct.unpack("<IIQQ", f.read(24))
        words = (k + 31) // 32
        kmers = np.frombuffer(f.read(8 * n * words), dtype="<u8")
        counts = np.frombuffer(f.read(4 * n), dtype="<u4")
    if len(kmers) != n * words or len(counts) != n:
        raise ValueError("%s is truncated" % path)
    return k, amin, kmers, counts, n_occ


# ---- multidsk's combined artefact --------------------------------------------------------------
# The pair multidsk / dsk2kover is opaque to Kover (only the file NAMES of dataset/create.py:375,488
# matter).  By default multidsk therefore does the device work for the whole list at once and leaves
# ONE artefact -- dictionary, unfiltered presence matrix, carrier counts -- plus a small reference file
# under every name Kover expects; dsk2kover then only selects rows / filters columns / writes HDF5.
# (GRM_MULTIDSK_SETS=1 restores one counted k-mer set per genome: 12 bytes per distinct k-mer and
# genome on disk, 60 GB for 1000 bacterial genomes.)
MATRIX_MAGIC = b"GRMKMAT1"
REF_MAGIC = b"GRMKREF1"


def write_matrix_artifact(path, k, abundance_min, kmers, data, counts, n_genomes):
    kmers = np.ascontiguousarray(kmers, dtype="<u8")
    data = np.ascontiguousarray(data, dtype="<u8")
    counts = np.ascontiguousarray(counts, dtype="<u4")
    words = (k + 31) // 32
    U = len(counts)
    assert kmers.size == U * words and data.shape == ((n_genomes + 63) // 64, U)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MATRIX_MAGIC)
        f.write(struct.pack("<IIQQ", k, abundance_min, n_genomes, U))
        f.write(kmers.tobytes())
        f.write(counts.tobytes())
        f.write(data.tobytes())
    os.replace(tmp, path)


def read_matrix_artifact(path):
    with open(path, "rb") as f:
        if f.read(8) != MATRIX_MAGIC:
            raise ValueError("%s is not a matrix written by this multidsk" % path)
        k, amin, n_genomes, U = struct.unpack("<IIQQ", f.read(24))
        words = (k + 31) // 32
        rows = (n_genomes + 63) // 64
        kmers = np.fromfile(f, dtype="<u8", count=U * words)
        counts = np.fromfile(f, dtype="<u4", count=U)
        data = np.fromfile(f, dtype="<u8", count=rows * U)
    if kmers.size != U * words or counts.size != U or data.size != rows * U:
        raise ValueError("%s is truncated" % path)
    return k, amin, int(n_genomes), kmers.reshape(U, words), counts, data.reshape(rows, U)


def write_ref(path, artifact, index, k, abundance_min, n_solid):
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(REF_MAGIC)
        f.write(struct.pack("<IIQQ", k, abundance_min, index, n_solid))
        f.write(os.path.abspath(artifact).encode())
    os.replace(tmp, path)


def read_ref(path):
    """-> (artifact path, genome index, k, abundance_min, n_solid) or None when `path` is not a reference file"""
    with open(path, "rb") as f:
        if f.read(8) != REF_MAGIC:
            return None
        k, amin, index, n_solid = struct.unpack("<IIQQ", f.read(24))
        return f.read().decode(), int(index), k, amin, int(n_solid)


def select_rows(data, n_genomes, order):
    """word-rows of the genomes `order` (indices into a [rows][U] bit matrix, genome i at word-row i//64,
    bit 63 - i%64) -> the same layout for len(order) genomes.  Identity orders are returned as is."""
    if list(order) == list(range(n_genomes)):
        return data
    U = data.shape[1]
    out = np.zeros(((len(order) + 63) // 64, U), dtype=np.uint64)
    for new, old in enumerate(order):
        bit = (data[old // 64] >> np.uint64(63 - old % 64)) & np.uint64(1)
        out[new // 64] |= bit << np.uint64(63 - new % 64)
    return out


def list_stem(line):
    """output name of one multidsk list line: basename without extension of the LAST file of the
    line (create.py:375 for contigs, :488 for reads; '.fastq.gz' keeps '.fastq')"""
    last = line.strip().split(",")[-1]
    return os.path.basename(os.path.splitext(last)[0])
