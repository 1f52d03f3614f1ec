"""Which kernels wait for their loads one at a time?  Compiles every .hip of the engine to gfx950 assembly and lists the kernels whose
ISA holds chains of (global load, `s_waitcnt vmcnt(0)`) pairs -- a load that is used (stored to LDS, compared, selected) before the next
one is asked for is a round trip of its own.  Binary searches show up too (their chains are real dependencies).
    python scripts/isa_wait_scan.py [min chain length, default 3]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genomic-resistance-mapping-grm-_amd", "csrc")
least = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with tempfile.TemporaryDirectory() as td:
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        out = os.path.join(td, os.path.basename(src) + ".s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-S", "--cuda-device-only",
                        "-I", CSRC, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
        for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", txt, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if "rocprim" in name:
                continue
            seq = []
            for line in body.split("\n"):
                line = line.strip()
                if line.startswith(("global_load", "buffer_load", "flat_load")):
                    seq.append("L")
                elif line.startswith("s_waitcnt") and "vmcnt(0)" in line:
                    seq.append("W")
                elif line.startswith("s_waitcnt") and "vmcnt(" in line:
                    seq.append("w")
            runs = re.findall(r"(?:LW){%d,}" % least, "".join(seq))
            if runs:
                plain = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                print("%-22s %-100s longest chain %d" % (os.path.basename(src), plain[:100], max(len(r) // 2 for r in runs)))
