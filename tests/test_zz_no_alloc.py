"""mdx_get_forces_noalloc (fixed scratch, no allocation) against mdx_get_forces through the C ABI: two calls in a
row on one GPU, the second on a smaller system, bit for bit; sizes beyond the scratch are refused."""
import ctypes

import numpy as np
import pytest

import mdx  # noqa: F401
from mdx import engine as eng

pytestmark = pytest.mark.gpu


def test_noalloc_forces_equal_allocating_entry_point(small_complex, small_water):
    big = eng.Engine(small_complex, rc=1.0, nstcalcenergy=1)
    small = eng.Engine(small_water, rc=1.0, nstcalcenergy=1)
    assert small.n < big.n
    f_big, _ = big.compute()                 # mdx_compute_forces + mdx_get_forces (allocating)
    f_small, _ = small.compute()
    got_big = big.forces(noalloc=True)       # first call: the larger system fills most of what it needs of the scratch
    got_small = small.forces(noalloc=True)   # second call, smaller input, same scratch on the same GPU
    assert np.array_equal(got_big, f_big)
    assert np.array_equal(got_small, f_small)
    assert np.abs(f_big).max() > 0 and np.abs(f_small).max() > 0
    # and again in the other order, interleaved with the allocating entry point
    assert np.array_equal(big.forces(noalloc=True), big.forces())
    # no forces at the current positions: refused like mdx_get_forces
    small.step(1)
    with pytest.raises(eng.MdxError):
        small.forces(noalloc=True)
    big.close(); small.close()


def test_noalloc_refuses_sizes_beyond_the_scratch():
    L = eng.load_library()
    n = (1 << 20) + 64
    h = L.mdx_create(0, n)
    assert h, L.mdx_last_error(None)
    buf = np.zeros(3, np.float32)
    rc = L.mdx_get_forces_noalloc(h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    msg = L.mdx_last_error(h).decode()
    L.mdx_destroy(h)
    assert rc == -4 and "1048576" in msg
