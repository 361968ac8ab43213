"""In-tree build of the native pieces (no JIT cache: the built .so files travel
with the source tree to the GPU box).

    libfjsp_amd.so      hipcc --offload-arch=gfx950   csrc/*.hip + csrc/*.cpp   (the product)
    liboracle           gcc                            oracle/fjsp_oracle.c      (test infrastructure)

Only `__graft_entry__.build()` and the test suite call this; importing the
package never compiles anything and never falls back to a CPU path.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
# FJSP_AMD_LIB lets a developer point the binding at an alternative build of the same ABI
LIB_PATH = os.environ.get("FJSP_AMD_LIB") or os.path.join(PKG_DIR, "libfjsp_amd.so")
ORACLE_DIR = os.path.join(REPO_DIR, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libfjsp_oracle.so")

HIP_SOURCES = ["fjsp_kernels.hip", "fjsp_group.hip", "fjsp_lp_device.hip", "fjsp_env.hip", "fjsp_rollout_buffer.hip", "fjsp_ppo.hip", "fjsp_mlp_train.hip", "fjsp_policy_mlp.hip"]
CPP_SOURCES = ["fjsp_instance.cpp", "fjsp_lp.cpp"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def build_library(force=False, verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 into libfjsp_amd.so."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + CPP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    deps.append(os.path.join(REPO_DIR, "include", "fjsp_amd.h"))
    if not force and not _newer(LIB_PATH, deps):
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # bit-exact f64 decision chain: never contract a*b+c into an FMA
           "-ffp-contract=off", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(REPO_DIR, "include"), "-I", CSRC] + srcs + ["-o", LIB_PATH, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def build_oracle(force=False, verbose=False):
    """Compile the CPU oracle (plain C).  Test infrastructure, never loaded by the product."""
    src = os.path.join(ORACLE_DIR, "fjsp_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "fjsp_oracle.h")
    if not force and not _newer(ORACLE_LIB, [src, hdr]):
        return ORACLE_LIB
    cmd = ["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-builtin-pow",
           "-Wall", "-Wextra", src, "-o", ORACLE_LIB, "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return ORACLE_LIB
