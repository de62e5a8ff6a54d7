"""Per-kernel register / scratch / LDS usage of the built library, read from the gfx950 code object's metadata notes
(the `.vgpr_spill_count`, `.private_segment_fixed_size` ... fields hipcc records for every kernel).  CPU-only.

    python tools/kernel_resources.py [--spills] [path/to/liblgd_hip.so]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def kernel_resources(lib=None):
    """[{name (demangled), vgpr_count, agpr_count, ..., vgpr_spill_count, private_segment_fixed_size, ...}] of every
    kernel in the gfx950 code object embedded in `lib`."""
    lib = lib or os.path.join(ROOT, "llm-groundeddiffusion_amd", "liblgd_hip.so")
    notes = ""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"                  # one bundle per translation unit, concatenated
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, st in enumerate(starts):
            part, co = os.path.join(tmp, f"b{i}"), os.path.join(tmp, f"co{i}")
            with open(part, "wb") as fh:
                fh.write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                    text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "agpr_count":                      # first field of every kernel record
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        if key == "name":
            cur["mangled"] = val.strip()
        elif key in FIELDS:
            cur[key] = int(val)
    out = [k for k in out if "mangled" in k]
    names = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in out),
                           capture_output=True, text=True, check=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["name"] = n.replace("(anonymous namespace)::", "")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = kernel_resources(args[0] if args else None)
    only = "--spills" in sys.argv
    print(f"{len(ks)} kernels")
    for k in sorted(ks, key=lambda k: -k.get("vgpr_spill_count", 0)):
        if only and not (k.get("vgpr_spill_count") or k.get("private_segment_fixed_size")):
            continue
        print(f"vgpr {k.get('vgpr_count', 0):3d} agpr {k.get('agpr_count', 0):3d} spill v{k.get('vgpr_spill_count', 0):4d} "
              f"s{k.get('sgpr_spill_count', 0):3d} scratch {k.get('private_segment_fixed_size', 0):5d} B  lds "
              f"{k.get('group_segment_fixed_size', 0):6d}  {k['name'][:150]}")
