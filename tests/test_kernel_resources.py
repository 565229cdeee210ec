"""Static checks of the built kernels (no GPU), with the ROCm LLVM tools on the objects under genjax_amd/_lib/obj
(skipped where the objects or the tools are missing; the library itself is checked by tests/test_abi_and_host.py):
 * scratch per lane: guards the by-value-argument cliff described in genjax_amd/_native.py::build -- above LLVM's
   cap on the users of a by-value kernel argument the world tables are copied to scratch at kernel entry (+1.5 KB
   per lane and a 3-6x slower kernel; profiles/r03_ray_counts.txt);
 * spill code in front of an exec restore: hipcc 7.2 can place register spill stores / reloads at the top of a
   divergent join, BEFORE the s_or_b64 that restores exec -- a block the wave reaches with exec = 0 when no lane
   took the branch, so the spill never happens.  k_loc_extend<16, true, true> lost its particle index that way
   (memory-aperture fault at 32 rays).  Every kernel of the library is disassembled and scanned."""
import glob
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "..", "genjax_amd", "_lib", "obj", "gjx_loc.hip.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def _code_object(obj, tmp_path):
    """The gfx950 code object inside a host object (None when it holds no device code)."""
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump")]
    if not os.path.exists(obj) or not all(os.path.exists(t) for t in tools):
        pytest.skip("no built object or no ROCm LLVM tools here")
    objcopy, bundler = tools[:2]
    base = os.path.basename(obj)
    fat = str(tmp_path / (base + ".fatbin"))
    subprocess.check_call([objcopy, "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    targets = subprocess.check_output([bundler, "--list", "--type=o", f"--input={fat}"], text=True).split()
    tgt = [t for t in targets if "gfx950" in t]
    if not tgt:
        return None
    co = str(tmp_path / (base + ".co"))
    subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={fat}", f"--targets={tgt[0]}", f"--output={co}"])
    return co


def _kernel_scratch(tmp_path):
    co = _code_object(OBJ, tmp_path)
    assert co is not None
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
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


_RESTORE = re.compile(r"s_or_b64 exec, exec,|s_andn2_saveexec_b64|s_or_saveexec_b64")


def _spills_before_exec_restore(co):
    """{kernel: [(address, instruction), ...]}: scratch / buffer accesses at the top of a block that an
    s_cbranch_execz jumps to, in front of that block's exec restore (the block's first write of exec)."""
    dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)
    kernels, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+): ", line)
        if m and cur:
            kernels[cur].append((int(m.group(2), 16), m.group(1)))
    out = {}
    for name, ins in kernels.items():
        targets = set()
        for addr, text in ins:
            m = re.match(r"s_cbranch_execz (-?\d+)", text)       # the skip edge: its target is reached with exec = 0
            if m:
                off = int(m.group(1))
                off = off - 65536 if off >= 32768 else off
                targets.add(addr + 4 + 4 * off)
        pre, hits = None, []
        for addr, text in ins:
            if addr in targets:
                pre = []
            if pre is None:
                continue
            if _RESTORE.match(text):
                hits += [x for x in pre if x[1].startswith(("scratch_", "buffer_"))]
                pre = None
            elif re.match(r"s_(cbranch|branch|endpgm)", text) or re.search(r"\bexec\b", text.split(",")[0]) \
                    or re.match(r"s_\w*saveexec", text):
                pre = None      # the block ends, or narrows exec first: what came before ran at its entry mask by design
            else:
                pre.append((hex(addr), text))
        if hits:
            out[name] = hits
    return out


def test_no_spill_code_in_front_of_an_exec_restore(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc: nothing was built here")
    objs = sorted(glob.glob(os.path.join(os.path.dirname(OBJ), "*.o")))
    if not objs:
        pytest.skip("no built objects here")
    n_kernels, bad = 0, {}
    for obj in objs:
        co = _code_object(obj, tmp_path)
        if co is None:
            continue
        found = _spills_before_exec_restore(co)
        bad.update({f"{os.path.basename(obj)}:{k}": v[:2] for k, v in found.items()})
        n_kernels += 1
    assert n_kernels >= 5
    assert not bad, bad
