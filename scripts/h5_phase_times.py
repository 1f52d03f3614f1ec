import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import numpy as np
import grm_amd
kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
U, G, k = 7_600_000, 1000, 31
rng = np.random.default_rng(1)
kmers = np.sort(rng.integers(0, 1 << 62, size=U, dtype=np.uint64))
R = (G + 63) // 64
# pan-genome-like: most columns nearly full or nearly empty
p = rng.beta(0.3, 0.3, size=U)
data = np.zeros((R, U), dtype=np.uint64)
for r in range(R):
    for bit in range(0, 64, 8):
        bits = (rng.random((8, U)) < p[None, :])
        for j in range(8):
            data[r] |= bits[j].astype(np.uint64) << np.uint64(63 - bit - j)
data[-1] &= np.uint64(~((1 << (64 * R - G)) - 1) & (2**64 - 1))
m = grm_amd.HostMatrix(kmers, data, G, k)
d = tempfile.mkdtemp()
path = os.path.join(d, "t.kover")
kd.write_header(path, "contigs", "l", None, None, 4, ["g%d" % i for i in range(G)], None, None, None, "singleton")
t0 = time.time(); m.write_kover_h5(path, 4, 100000); print("h5 %.2fs size %d MB" % (time.time() - t0, os.path.getsize(path) >> 20))
