"""The co-residency budget of the built kernels, read from the code objects inside libmi355jpeg.so.

Batched calls run a part's tail kernels under the next part's block-encode kernel.  That only works while, on every CU,
two workgroups of k_screen_encode leave room for k_merge -- one workgroup of its full-window form, two of the half-window
form that batches run: LDS in 1280-byte granules within 160 KiB, and on every SIMD two encode waves plus one k_merge wave
per workgroup within the 512 registers of the unified file (8-register granules).
What counts is what the hardware ALLOCATES -- the kernel descriptor's granulated count -- not what the kernel uses: the
compiler raises a kernel's allocation when it has worked out a lower occupancy from workgroup size and LDS, and did so
for the 192-thread k_merge (32 registers used, 72 allocated) until round 4 gave it launch bounds of 256.  Measured
(gpurun r4w, r4x, r4cs-r4cu): encode 224 + merge 72: 187-194 Gpixel/s; encode 216 + merge 72,
encode 224 + merge 32: no loss; the headline is 256.  One register granule or one LDS granule over the line costs a
quarter of the throughput of batched calls, and nothing but the bench shows it.  This test does, on CPU, from the build."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "jpeg-encoder-opencl_amd", "libmi355jpeg.so")


def kernel_table(tmp_path):
    """{mangled name: {vgpr (allocated: next_free_vgpr of the kernel descriptor, in granules of 8), used, lds, scratch}}
    of every gfx950 kernel in the library."""
    lib = shutil.copy(LIB, tmp_path / "lib.so")  # llvm-objdump writes the bundles next to its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    table = {}
    for f in sorted(os.listdir(tmp_path)):
        if not f.endswith("gfx950"):
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(tmp_path / f)], check=True,
                               capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            def field(k):
                return re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
            table[field("name")] = dict(used=int(field("vgpr_count")), lds=int(field("group_segment_fixed_size")),
                                        scratch=int(field("private_segment_fixed_size")))
        # the kernel descriptors, disassembled back into .amdhsa_ directives
        kd = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--section=.rodata", str(tmp_path / f)], check=True,
                            capture_output=True, text=True).stdout
        for name, body in re.findall(r"<(\S+)\.kd>:(.*?)\.end_amdhsa_kernel", kd, re.S):
            table[name]["vgpr"] = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    return table


def up(x, g):
    return (x + g - 1) // g * g


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no ROCm LLVM tools")
def test_tail_kernels_fit_beside_the_block_encode(tmp_path):
    assert os.path.exists(LIB), "libmi355jpeg.so not built"
    t = kernel_table(tmp_path)
    enc = {k: v for k, v in t.items() if "k_screen_encodeILb0E" in k}   # the shipped (non-probe) forms: strict, 4:4:4, 4:2:0
    merge = {k: v for k, v in t.items() if "k_mergeIL" in k}     # <S420, SMALL>: full window (one per CU), half window (two)
    assert len(enc) == 3 and len(merge) == 4, sorted(t)
    for k, v in {**enc, **merge}.items():
        assert v["scratch"] == 0, (k, v)  # no spills on the hot path
    for ek, e in enc.items():
        s420 = "ILb0ELi2E" in ek
        for mk, m in merge.items():
            if ("k_mergeILb1E" in mk) != s420:
                continue
            per_cu = 2 if "ELb1EEEv" in mk else 1  # workgroups of this form that must fit beside the block encode
            lds = 2 * up(e["lds"], 1280) + per_cu * up(m["lds"], 1280)
            regs = 2 * up(e["vgpr"], 8) + per_cu * up(m["vgpr"], 8)
            assert m["vgpr"] <= up(m["used"], 8), (mk, m)  # no occupancy-driven inflation of the tail kernel's allocation
            assert lds <= 160 * 1024, (ek, mk, e, m, lds)
            assert regs <= 512, (ek, mk, e, m, regs)
    # the small tail kernels of every frame
    heads = [v for k, v in t.items() if "k_dc_heads" in k]
    assert heads and all(v["vgpr"] <= 32 and v["lds"] <= 1280 for v in heads)
