"""Builds liblgd_hip.so (all HIP kernels, gfx950 only) in-tree with hipcc.

    python llm-groundeddiffusion_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblgd_hip.so")
STAMP = OUT + ".stamp"
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "attn_w4.hip", "attn_bwd.hip", "misc.hip", "energy.hip", "boxdiff.hip", "sam.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-Wno-unused-value"]
# Per-file extras.  The attention kernels run softmax VALU work on MFMA results every key tile: keep
# the accumulators in VGPRs (not AGPRs) so that no v_accvgpr_read/write traffic is generated.
EXTRA_FLAGS = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "attn_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files.append(os.path.join(HERE, "..", "include", "lgd_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP):
        if open(STAMP).read().strip() == dig:
            return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    with open(STAMP, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
