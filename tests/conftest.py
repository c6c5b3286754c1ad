import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's C part is test infrastructure: build it on demand where gcc exists
    so = os.path.join(ROOT, "oracle", "libnbref.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def small_water():
    import mdx  # noqa: F401
    from mdx import synth
    import numpy as np
    s = synth.water_box(d=3.4)
    s.x = s.x.astype(np.float32).astype(np.float64)
    return s


@pytest.fixture(scope="session")
def small_complex():
    import mdx  # noqa: F401
    from mdx import synth
    import numpy as np
    s = synth.pseudo_complex(d=6.0, protein_radius=1.0, n_active=60, ligand_offset=1.2)
    s.x = s.x.astype(np.float32).astype(np.float64)
    return s
