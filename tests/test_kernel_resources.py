"""Static check of the built kernels' scratch (no GPU): guards the by-value-argument cliff described in
genjax_amd/_native.py::build -- above LLVM's cap on the users of a by-value kernel argument the world tables are
copied to scratch at kernel entry (+1.5 KB per lane and a 3-6x slower kernel; profiles/r03_ray_counts.txt).
Reads the code-object notes of genjax_amd/_lib/obj/gjx_loc.hip.o with the ROCm LLVM tools; skipped where the
object or the tools are missing (the library itself is checked by tests/test_abi_and_host.py)."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "..", "genjax_amd", "_lib", "obj", "gjx_loc.hip.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_scratch(tmp_path):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(OBJ) or not all(os.path.exists(t) for t in tools):
        pytest.skip("no built object or no ROCm LLVM tools here")
    objcopy, bundler, readelf = tools
    fat = str(tmp_path / "loc.fatbin")
    subprocess.check_call([objcopy, "-O", "binary", "--only-section=.hip_fatbin", OBJ, fat])
    targets = subprocess.check_output([bundler, "--list", "--type=o", f"--input={fat}"], text=True).split()
    tgt = [t for t in targets if "gfx950" in t]
    assert tgt, targets
    co = str(tmp_path / "loc.co")
    subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={fat}", f"--targets={tgt[0]}", f"--output={co}"])
    notes = subprocess.check_output([readelf, "--notes", co], text=True)
    out, name = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


def test_no_kernel_keeps_the_world_tables_in_scratch(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc: nothing was built here")
    scratch = _kernel_scratch(tmp_path)
    loc = {k: v for k, v in scratch.items() if k.startswith("_Z") and "k_loc_" in k}
    assert len(loc) > 50, len(loc)
    # the reference's setting (8 rays = the 4-slot paired instantiations, `ILi4ELb1E`): spills only
    head = {k: v for k, v in loc.items() if re.search(r"k_loc_\w+ILi4ELb1E", k)}
    assert head and max(head.values()) <= 256, sorted(head.items(), key=lambda kv: -kv[1])[:3]
    # 16 rays (8 slots): under 512 B; nothing anywhere near the 1.5 KB of a copied world plus spills
    mid = {k: v for k, v in loc.items() if re.search(r"k_loc_\w+ILi8E", k)}
    assert mid and max(mid.values()) <= 512, sorted(mid.items(), key=lambda kv: -kv[1])[:3]
    # kernels that touch few rays at a time hold no scratch at all
    for k, v in loc.items():
        if re.search(r"k_loc_(regenerate|hmc_propose|logp_grad)I", k):
            assert v <= 64, (k, v)
    assert max(loc.values()) <= 2048, sorted(loc.items(), key=lambda kv: -kv[1])[:3]
