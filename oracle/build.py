"""Build recipe for the CPU oracle (test infrastructure only).

`python -m oracle.build` compiles oracle/*.c with gcc into oracle/_build/ (git-ignored,
travels to the GPU box with the gpurun snapshot).  Flags matter: -ffp-contract=off and no
fast-math are part of the oracle's floating-point contract (see gs_oracle.c header).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")

SOURCES = {
    "libgs_oracle.so": ["gs_oracle.c"],
}

CFLAGS = ["-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
          "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter"]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    built = []
    for lib, srcs in SOURCES.items():
        srcs = [os.path.join(HERE, s) for s in srcs]
        target = os.path.join(OUT, lib)
        if force or _stale(target, srcs):
            cmd = ["gcc"] + CFLAGS + srcs + ["-o", target, "-lm"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        built.append(target)
    return built


def lib_path(name="libgs_oracle.so"):
    p = os.path.join(OUT, name)
    if not os.path.exists(p):
        build()
    return p


if __name__ == "__main__":
    for p in build(force="--force" in sys.argv, verbose=True):
        print("built", p)
