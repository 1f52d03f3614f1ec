"""Handle lifetime across the C ABI: a context may be closed (grm_destroy) while batches, matrices, k-mer sets
and accumulators made from it are still alive -- a pytest traceback or a CLI that leaves `with Context` on an
exception does exactly that.  Round 2 ended such processes with SIGABRT (a freed context's stream handed to
hipStreamSynchronize); the library now counts references.  Every case runs in a child process so that an abort
shows as an exit code."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = """
import sys
sys.path.insert(0, %r)
import numpy as np
import grm_amd
rng = np.random.RandomState(5)
def genome(n):
    return b">g\\n" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)) + b"\\n"
def filled(ctx, n=3):
    b = ctx.batch(n)
    for i in range(n):
        b.add(i, genome(5000))
    b.upload()
    return b
""" % ROOT


def _child(body, timeout=300):
    p = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def test_handles_freed_after_grm_destroy():
    rc, out, err = _child("""
        ctx = grm_amd.Context(0)
        b = filled(ctx)
        m = b.run(31, 1, False)
        s = ctx.count_genome([genome(3000)], 21, 1)
        acc = ctx.dict_accum()
        assert ctx.L.grm_ctx_live_handles(ctx.h) == 4
        ctx.L.grm_destroy(ctx.h)                                # NOT Context.close(): the handles stay alive
        assert ctx.L.grm_ctx_live_handles(ctx.h) == 4
        n = m.n_kmers
        assert n > 0 and m.kmers().shape[0] == n                # a download after the owner let go still works
        m.free(); s.free(); acc.free(); b.free()                # the last free takes the context down
        print("ok", n)
    """)
    assert rc == 0, (rc, err[-2000:])
    assert out.startswith("ok")


def test_close_frees_live_children_first():
    rc, out, err = _child("""
        with grm_amd.Context(0) as ctx:
            b = filled(ctx)
            m = b.run(31, 1, True)
        assert b._h is None and m._h is None                    # Context.close() freed them
        for dead in (b, m):                                     # ... and a freed handle says so instead of handing NULL to the C ABI
            try:
                dead.h
            except grm_amd.GrmError as e:
                assert e.code == -7
            else:
                raise SystemExit("a freed handle gave out a pointer")
        try:
            m.kmers()
        except grm_amd.GrmError:
            pass
        else:
            raise SystemExit("an accessor of a freed Matrix returned")
        del b, m
        print("ok")
    """)
    assert rc == 0, (rc, err[-2000:])


def test_exception_inside_with_context_is_a_clean_exit():
    # the failing frame keeps `b` and `m` alive in the traceback while __exit__ closes the context
    rc, out, err = _child("""
        def work():
            with grm_amd.Context(0) as ctx:
                b = filled(ctx)
                m = b.run(31, 1, False)
                raise RuntimeError("boom after %d columns" % m.n_kmers)
        work()
    """)
    assert rc == 1, (rc, err[-2000:])
    assert "RuntimeError: boom" in err and "terminate called" not in err and "Aborted" not in err


def test_error_return_keeps_the_process_alive():
    # an engine error (k out of range) raised inside the block: message on stderr, exit code 1, no abort
    rc, out, err = _child("""
        with grm_amd.Context(0) as ctx:
            b = filled(ctx)
            b.run(129, 1, False)
    """)
    assert rc == 1, (rc, err[-2000:])
    assert "GRM_ERR" in err and "terminate called" not in err


def test_double_destroy_and_late_queries_are_harmless():
    """after grm_destroy AND the last handle's free the context no longer exists: a second grm_destroy, grm_ctx_live_handles and
    grm_last_error on the stale pointer recognise that instead of reading freed memory"""
    rc, out, err = _child("""
        ctx = grm_amd.Context(0)
        b = filled(ctx)
        h = ctx.h
        ctx.L.grm_destroy(h)
        ctx.L.grm_destroy(h)                                    # twice while a handle keeps it alive
        assert ctx.L.grm_ctx_live_handles(h) == 1
        b.free()                                                # the context goes with its last handle
        ctx.L.grm_destroy(h)                                    # and once more: no-op
        assert ctx.L.grm_ctx_live_handles(h) == 0
        assert b"no longer exists" in ctx.L.grm_last_error(h)
        ctx.h = None
        print("ok")
    """)
    assert rc == 0, (rc, err[-2000:])
    assert out.startswith("ok")
