"""Register allocations of the kernels that run beside each other, read from the code objects inside the built libmdx.so.

What runs beside the nonbonded kernel is decided by vector-register allocations: three nonbonded waves per SIMD at up to
128 registers leave a SIMD 128 of its 512, which hold two 64-register waves of the PME transform kernels or two to three
waves of the spreading and gather kernels. A nonbonded kernel that grows past 128, or a PME kernel past 64, still gives
right results and the same time when run alone, and costs tens of microseconds per step inside the step, so the budget
is checked here, without a GPU."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "molecular-dynamics_amd", "libmdx.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_table():
    """{kernel symbol: (vgpr_count, private_segment_fixed_size)} over every gfx950 code object bundled in the library"""
    data = open(LIB, "rb").read()
    table = {}
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        nent = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(nent):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
                f.write(data[i + off:i + off + size])
                name = f.name
            try:
                out = subprocess.run([READELF, "--notes", name], capture_output=True, text=True, check=True).stdout
            finally:
                os.unlink(name)
            # one metadata block per kernel: fields are sorted by name, .name comes before the counts
            for blk in out.split("  - .agpr_count:")[1:]:
                m = re.search(r"^\s+\.name:\s+(\S+)\s*$", blk, re.M)
                v = re.search(r"^\s+\.vgpr_count:\s+(\d+)\s*$", blk, re.M)
                s = re.search(r"^\s+\.private_segment_fixed_size:\s+(\d+)\s*$", blk, re.M)
                if m and v and s:
                    table[m.group(1)] = (int(v.group(1)), int(s.group(1)))
        pos = i + len(MAGIC)
    return table


@pytest.mark.skipif(not os.path.exists(LIB), reason="libmdx.so is not built")
@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf is not installed")
def test_kernels_that_share_a_simd_keep_their_register_budget():
    t = kernel_table()
    assert len(t) > 40, "kernel metadata not found in the library"

    def find(prefix):
        hits = {k: v for k, v in t.items() if prefix in k}
        assert hits, "no kernel matches %s" % prefix
        return hits

    # nonbonded kernels of the step loop: three waves per SIMD that leave 128 registers, nothing in scratch memory
    for pat in ("k_nonbonded_ft6NbArgs", "k_nonbonded_f6NbArgs", "k_nonbonded_ev6NbArgs", "k_nonbonded_ev_count6NbArgs"):
        for k, (vgpr, scratch) in find(pat).items():
            assert vgpr <= 128, (k, vgpr)
            assert scratch == 0, (k, scratch)
    for k, (vgpr, scratch) in find("k_nonbonded_ev_dd6NbArgs").items():
        assert vgpr <= 128, (k, vgpr)
    # PME transform kernels: two waves per SIMD beside them
    for pat in ("k_pme_fftILi", "k_pme_fft_yzILb"):
        for k, (vgpr, scratch) in find(pat).items():
            assert vgpr <= 64, (k, vgpr)
    # spreading (with the bonded rows) and gather: two waves of at most 48, three of at most 40
    for k, (vgpr, scratch) in find("k_pme_spread_slab_bonded14").items():
        assert vgpr <= 48 and scratch == 0, (k, vgpr, scratch)
    for k, (vgpr, scratch) in find("k_pme_gather").items():
        assert vgpr <= 40 and scratch == 0, (k, vgpr, scratch)
    # fused update of a topology without large constraint components: four waves per SIMD, nothing in scratch memory
    for k, (vgpr, scratch) in find("k_update_fusedILb0ELb0EE").items():
        assert vgpr <= 104 and scratch == 0, (k, vgpr, scratch)
    # constraint launch of the energy steps, same topologies: nothing in scratch memory
    for pat in ("k_constrainILb0ELb0EE", "k_constrainILb1ELb0EE"):
        for k, (vgpr, scratch) in find(pat).items():
            assert vgpr <= 128 and scratch == 0, (k, vgpr, scratch)
