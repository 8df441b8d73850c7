"""In-tree build of libdrs_hip.so (hipcc, gfx950 only).  `python -m diffusionremotesensing_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libdrs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "drs_hip.h"))
    jobs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        if force or not _newer(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in srcs]
    if force or jobs or not _newer(LIB, objs):
        # the C-ABI is the dynamic symbol table: drs_* (include/drs_hip.h, DRS_API) and nothing else - libstdc++'s inline
        # templates carry default visibility whatever -fvisibility says, so the linker gets the list as well
        vers = os.path.join(OBJ, "exports.map")
        with open(vers, "w") as f:
            f.write("{ global: drs_*; local: *; };\n")
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vers, "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
