"""GPU: the device-side zlib encoder (csrc/grm_deflate.hip) behind the Kover HDF5 writer.  Every stream must (1) inflate with
stock zlib to the chunk's bytes and (2) be bit-identical to what the host emulation of the same format functions writes
(tests/host/deflate_emul.cpp) -- the kernel's lockstep is deterministic; and the file written from a device-resident matrix must
read back, through libhdf5's own inflate, as the matrix."""
import os
import zlib
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle_ctypes as orc
from tests import test_deflate_emul as emu

PKG = "genomic-resistance-mapping-grm-_amd"


@pytest.fixture(scope="module")
def ctx():
    import grm_amd
    with grm_amd.Context(0) as c:
        yield c


@pytest.fixture(scope="module")
def emul():
    return emu.load_emul()


def device_matrix(ctx, kmers, data, n_genomes, k):
    import grm_amd
    return grm_amd.HostMatrix(kmers, data, n_genomes, k).to_device(ctx)


def some_kmers(rng, U, k):
    km = emu.sorted_kmers(rng, U + U // 8 + 16, k)
    assert km.shape[0] >= U
    return km[:U]


@pytest.mark.parametrize("n_genomes,U,cw,kind", [
    (64, 1, 100000, "pan"), (64, 63, 100000, "pan"), (130, 1000, 100, "pan"), (200, 100000, 100000, "pan"), (64, 250001, 100000, "pan"),
    (1000, 120000, 50000, "pan"), (64, 100000, 100000, "random"), (64, 100000, 100000, "zeros"), (128, 70000, 4096, "pan"),
    (64, 300000, 300000, "pan"), (64, 9000, 9000, "far")])
def test_matrix_row_streams(ctx, emul, n_genomes, U, cw, kind):
    rng = np.random.default_rng(U * 31 + n_genomes)
    R = (n_genomes + 63) // 64
    if kind == "pan":
        data = np.stack([emu.pan_rows(rng, U) for _ in range(R)])
    elif kind == "random":
        data = rng.integers(0, 2 ** 63, (R, U), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (R, U), dtype=np.uint64)
    elif kind == "zeros":
        data = np.zeros((R, U), np.uint64)
    else:
        # distinct words with repeats at the window's edge (4032 words back: found; one further: not)
        data = rng.integers(1, 2 ** 63, (R, U), dtype=np.uint64)
        data[0, 4032 + 300] = data[0, 300]
        data[0, 4033 + 400 + 64] = data[0, 400]
    k = 31
    m = device_matrix(ctx, some_kmers(rng, U, k), data, n_genomes, k)
    st = m.deflate_rows(cw)
    w = min(U, cw)
    per_row = (U + w - 1) // w
    assert len(st) == per_row * R
    for i in range(len(st)):
        r, c0 = i // per_row, (i % per_row) * w
        raw = np.zeros(w, np.uint64)
        part = data[r, c0:c0 + w]
        raw[: part.shape[0]] = part
        s = st.chunk(i)
        assert zlib.decompress(s) == raw.tobytes(), (i, kind)
        want, _ = emu.row_stream(emul, part, w)
        assert s == want, "chunk %d: the kernel's stream differs from the host emulation's (first byte %d)" % (
            i, next((j for j in range(min(len(s), len(want))) if s[j] != want[j]), -1))
    if kind == "pan" and U >= 100000:
        total = sum(int(n) for n in st.lens)
        assert total < 0.55 * data.nbytes
    m.free()


@pytest.mark.parametrize("k,U,ce", [(31, 70000, 65536), (31, 100, 65536), (21, 5000, 4096), (1, 4, 65536), (3, 60, 16), (32, 4000, 65536),
                                     (33, 3000, 1000), (63, 50000, 65536), (64, 2000, 65536), (65, 2000, 512), (101, 1500, 65536), (128, 3000, 2048)])
def test_kmer_string_streams(ctx, emul, k, U, ce):
    rng = np.random.default_rng(k * 977 + U)
    km = emu.sorted_kmers(rng, U, k)
    U = km.shape[0]
    m = device_matrix(ctx, km, np.zeros((1, U), np.uint64), 1, k)
    st = m.deflate_kmer_strings(ce)
    e = min(U, ce)
    assert len(st) == (U + e - 1) // e
    letters = emu.kmer_strings(km, k)
    for i in range(len(st)):
        raw = np.zeros((e, k), np.uint8)
        part = letters[i * e:(i + 1) * e]
        raw[: part.shape[0]] = part
        s = st.chunk(i)
        assert zlib.decompress(s) == raw.tobytes(), i
        want, _ = emu.kmer_stream(emul, km[i * e:(i + 1) * e], k, e)
        assert s == want, i
    m.free()


@pytest.mark.parametrize("k,n_genomes,U,chunk_cols,gzip", [(31, 70, 5000, 100000, 4), (31, 200, 250000, 100000, 5), (63, 64, 30000, 7000, 1),
                                                            (101, 130, 2000, 100000, 9)])
def test_kover_file_from_a_device_matrix(ctx, tmp_path, monkeypatch, k, n_genomes, U, chunk_cols, gzip):
    kd = import_module(PKG + ".kover_dataset")
    rng = np.random.default_rng(U + k)
    km = emu.sorted_kmers(rng, U, k)
    U = km.shape[0]
    R = (n_genomes + 63) // 64
    data = np.stack([emu.pan_rows(rng, U) for _ in range(R)])
    ids = ["g%d" % i for i in range(n_genomes)]
    files = {}
    for mode in ("device", "host"):
        path = str(tmp_path / (mode + ".kover"))
        kd.write_header(path, "contigs", "l", None, None, gzip, ids, None, None, None, "nothing")
        m = device_matrix(ctx, km, data, n_genomes, k)
        if mode == "host":
            monkeypatch.setenv("GRM_DEFLATE", "host")
        else:
            monkeypatch.delenv("GRM_DEFLATE", raising=False)
        m.write_kover_h5(path, gzip, chunk_cols)
        m.free()
        r = kd.KoverDatasetReader(path)
        assert (r.kmer_matrix == data).all() and r.kmer_matrix.dtype == np.uint64
        assert r.kmer_sequences == orc.decode_kmers(km, k)
        assert (r.kmer_by_matrix_column == np.arange(U)).all()
        lay = r.layout("kmer_matrix")
        assert lay["chunks"] == (1, min(U, chunk_cols)) and lay["n_filters"] == 1
        files[mode] = os.path.getsize(path)
    # the device's streams are about as small as the host library's (a different encoder: not byte-identical)
    assert files["device"] < 1.15 * files["host"] + 4096


def test_parts_written_by_another_rank(ctx, tmp_path):
    """two 'ranks' deflate the word-rows they hold, the first one appends everything: the file equals the one-rank file's content"""
    kd = import_module(PKG + ".kover_dataset")
    rng = np.random.default_rng(5)
    k, U, n = 31, 150000, 300
    km = emu.sorted_kmers(rng, U, k)
    U = km.shape[0]
    R = (n + 63) // 64
    data = np.stack([emu.pan_rows(rng, U) for _ in range(R)])
    ids = ["g%d" % i for i in range(n)]
    cut = 3
    a = device_matrix(ctx, km, data[:cut], cut * 64, k)
    b = device_matrix(ctx, km, data[cut:], n - cut * 64, k)
    sa, sb = a.deflate_rows(100000), b.deflate_rows(100000)
    spool = str(tmp_path / "rank1.chunks")
    sb.tofile(spool)
    grm = import_module(PKG + ".engine")
    sb2 = grm.ChunkStreams.fromfile(spool)
    path = str(tmp_path / "parts.kover")
    kd.write_header(path, "contigs", "l", None, None, 4, ids, None, None, None, "nothing")
    a.write_kover_h5_parts(path, [(0, cut, sa), (cut, R - cut, sb2)], R, 4, 100000)
    r = kd.KoverDatasetReader(path)
    assert (r.kmer_matrix == data).all() and r.kmer_sequences == orc.decode_kmers(km, k)
    with pytest.raises(grm.GrmError):
        a.write_kover_h5_parts(path, [(0, cut, sa)], R, 4, 100000)          # rows not covered
    a.free()
    b.free()


def test_failed_append_from_the_device_path(ctx, tmp_path, monkeypatch):
    kd = import_module(PKG + ".kover_dataset")
    h5lite = import_module(PKG + ".h5lite")
    grm = import_module(PKG + ".engine")
    rng = np.random.default_rng(9)
    k, U, n = 31, 30000, 100
    km = emu.sorted_kmers(rng, U, k)
    U = km.shape[0]
    data = np.stack([emu.pan_rows(rng, U) for _ in range(2)])
    m = device_matrix(ctx, km, data, n, k)
    path = str(tmp_path / "f.kover")
    kd.write_header(path, "contigs", "l", None, None, 4, ["g%d" % i for i in range(n)], None, None, None, "nothing")
    monkeypatch.setenv("GRM_FAULT_H5_CHUNK", "4")
    with pytest.raises(grm.GrmError):
        m.write_kover_h5(path, 4, 10000)
    with h5lite.File(path) as f:
        assert not any(f.exists(x) for x in ("kmer_sequences", "kmer_matrix", "kmer_by_matrix_column"))
    monkeypatch.delenv("GRM_FAULT_H5_CHUNK")
    m.write_kover_h5(path, 4, 10000)
    assert (kd.KoverDatasetReader(path).kmer_matrix == data).all()
    m.free()
