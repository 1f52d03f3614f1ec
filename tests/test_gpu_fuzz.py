"""GPU: the randomised differential check (scripts/fuzz_paths.py) as part of the suite -- fixed seeds, a time budget: random k, genome
counts and sizes, assembly shapes (own contigs, indels, strands, repeated stretches, empty genomes), abundance filters and forced
engine geometries; every matrix and every counted set against the CPU oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 20261005])
def test_random_paths_against_the_oracle(seed):
    import grm_amd
    spec = importlib.util.spec_from_file_location("fuzz_paths", os.path.join(ROOT, "scripts", "fuzz_paths.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    with grm_amd.Context(0) as ctx:
        n = fz.run(ctx, 30.0, seed, say=lambda m: None)
    assert n >= 20, "only %d cases in 30 s" % n
