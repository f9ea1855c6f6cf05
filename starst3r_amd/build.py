"""Builds libst3r_hip.so (gfx950 only) in-tree with hipcc.

    python -m starst3r_amd.build [--force] [-v]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
gs_project.hip / gs_isect.hip carry integer outputs that must be bit-exact against the CPU
oracle and are therefore compiled with -ffp-contract=off (see the file headers).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libst3r_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-DNDEBUG"] + os.environ.get("ST3R_DEFS", "").split()
PER_FILE = {
    "gs_project.hip": ["-ffp-contract=off"],
    "gs_isect.hip": ["-ffp-contract=off"],
    # MFMA results straight into VGPRs (gfx950 has one unified register file): the arg-max epilogue would otherwise
    # start with one v_accvgpr_read per score
    "recip_nn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
}
# Measured on MI355X: a v_pk_{mul,fma,add}_f32 costs two issue slots and the register pairing it needs costs extra
# v_mov, so hipcc's packed-fp32 vectorisation slows the VALU-bound loops down (blend forward -12 %, backward -5 %
# with it switched off).  No kernel here wants it.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "st3r.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, verbose):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + COMMON + NO_PACKED_F32 + PER_FILE.get(src, []) + ["-c", srcp, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    # the host pass of hipcc does not know the device feature switched off above and says so on every file
    noise = "is not a recognized feature for this target"
    err = "\n".join(l for l in r.stderr.splitlines() if l.strip() and noise not in l)
    if verbose and err:
        print(err)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=("-v" in sys.argv or True)))
