"""Per-kernel register / scratch / LDS figures of the BUILT library, from the code objects' own metadata.

    python tools/kernel_resources.py [lib.so] [--spills] [--grep PATTERN]

Unbundles the gfx950 code objects of libtinyvc_hip.so into a temporary directory (llvm-objdump --offloading), reads each one's
NT_AMDGPU_METADATA note (llvm-readelf --notes) and prints one line per kernel: VGPRs, AGPRs, SGPRs, VGPR / SGPR spill counts,
scratch bytes per lane, static LDS bytes, workgroup bound, demangled name.  `kernels(lib)` is what tests/test_build_resources.py
asserts on (no kernel of the library may spill).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("TVC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "tinyvc_amd", "libtinyvc_hip.so")

_FIELDS = {
    ".name": "name", ".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr",
    ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill",
    ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds",
    ".max_flat_workgroup_size": "wg", ".uses_dynamic_stack": "dyn_stack",
}


def _demangle(names):
    tool = os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(tool):
        tool = shutil.which("c++filt")
    if not tool:
        return names
    r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.split("\n")
    return out[:len(names)] if r.returncode == 0 and len(out) >= len(names) else names


def kernels(lib=DEFAULT_LIB):
    """List of dicts (one per kernel of every gfx950 code object in `lib`)."""
    tmp = tempfile.mkdtemp(prefix="tvc_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, capture_output=True, check=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   capture_output=True, text=True, check=True).stdout
            cur = None
            for line in notes.split("\n"):
                m = re.match(r"^  - (\.\w+):\s*(.*)$", line)        # first key of a kernel entry (two-space list item)
                if m:
                    cur = {"object": f}
                    out.append(cur)
                else:
                    m = re.match(r"^    (\.\w+):\s*(.*)$", line)
                if m and cur is not None and m.group(1) in _FIELDS:
                    v = m.group(2).strip()
                    cur[_FIELDS[m.group(1)]] = v if m.group(1) in (".name", ".uses_dynamic_stack") else int(v)
        out = [k for k in out if "name" in k]
        for k, d in zip(out, _demangle([k["name"] for k in out])):
            k["demangled"] = d
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(name, n=150):
    name = name.replace("tvc::(anonymous namespace)::", "").replace("tvc::", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main(argv):
    lib = DEFAULT_LIB
    only_spills = False
    pat = None
    it = iter(argv)
    for a in it:
        if a == "--spills":
            only_spills = True
        elif a == "--grep":
            pat = re.compile(next(it))
        else:
            lib = a
    ks = kernels(lib)
    print(f"# {lib}: {len(ks)} kernels in {len({k['object'] for k in ks})} code objects")
    print("# vgpr agpr sgpr | vspill sspill scratchB | ldsB wg | kernel")
    n = 0
    for k in sorted(ks, key=lambda k: (-k.get("vgpr_spill", 0), -k.get("sgpr_spill", 0), k["demangled"])):
        if only_spills and not (k.get("vgpr_spill", 0) or k.get("sgpr_spill", 0) or k.get("scratch", 0)):
            continue
        if pat and not pat.search(k["demangled"]):
            continue
        n += 1
        print(f"{k.get('vgpr', 0):4d} {k.get('agpr', 0):4d} {k.get('sgpr', 0):4d} | {k.get('vgpr_spill', 0):4d} {k.get('sgpr_spill', 0):4d} "
              f"{k.get('scratch', 0):5d} | {k.get('lds', 0):6d} {k.get('wg', 0):4d} | {short(k['demangled'])}")
    print(f"# {n} listed")


if __name__ == "__main__":
    main(sys.argv[1:])
