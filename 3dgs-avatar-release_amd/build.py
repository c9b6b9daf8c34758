"""Builds lib/libgsplat_mi355.so from csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

Usage:  python 3dgs-avatar-release_amd/build.py [--force] [--verbose]
The library is built IN-TREE (it travels with the source snapshot to the GPU box) and links only
against the HIP runtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# GSPLAT_VARIANT=name: an experimental build next to the product one (variants/name/, with GSPLAT_EXTRA_HIPCC_FLAGS;
# loaded through GSPLAT_LIB_PATH) -- the product library is lib/libgsplat_mi355.so
_VARIANT = os.environ.get("GSPLAT_VARIANT")
OUT_DIR = os.path.join(HERE, "..", "variants", _VARIANT) if _VARIANT else os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj") if _VARIANT else os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libgsplat_mi355.so")

ARCH = "gfx950"
COMMON = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# translation units whose integer outputs must match the CPU oracle bit for bit: no FMA contraction
STRICT = {"preprocess.hip", "knn.hip"}
# packed fp32 (v_pk_*) issues at half rate on gfx950 (tools/valu_rate.hip), so pairing scalars buys
# nothing and the register shuffling the SLP vectoriser adds to form the pairs costs VALU slots
NO_SLP = {"render_fwd.hip", "render_bwd.hip"}
SOURCES = ["capi.hip", "preprocess.hip", "radix_sort.hip", "depth_sort.hip", "binning.hip", "render_fwd.hip", "render_bwd.hip",
           "gaussian_bwd.hip", "knn.hip", "loss.hip", "prepass.hip", "optim.hip", "debug_stats.hip"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "gsplat_mi355.h"),
                                                                 os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_dep():
        return LIB
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        flags = list(COMMON) + (["-ffp-contract=off"] if src in STRICT else ["-ffp-contract=fast"])
        if src in NO_SLP:
            flags.append("-fno-slp-vectorize")
        flags += os.environ.get("GSPLAT_EXTRA_HIPCC_FLAGS", "").split()  # experiments (e.g. -DKNN_BOX=64)
        cmd = [cc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # render_bwd.hip's exec-masked LDS reads land in live registers while other instructions run: verify in the generated code
    # that nothing between their issue and their wait touches those registers (check_inflight.py, next to this file)
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import check_inflight
    check_inflight.check(os.path.join(OBJ_DIR, "render_bwd.o"), cc)
    # code-object extracts that `llvm-objdump --offloading` leaves next to what it inspects do not belong in a directory
    # that travels to the GPU box with every push
    for d in (OUT_DIR, OBJ_DIR):
        for f in os.listdir(d):
            if ".hipv4-amdgcn" in f or ".host-x86_64" in f:
                os.remove(os.path.join(d, f))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
