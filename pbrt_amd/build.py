"""Builds libpbrt_hip.so (the C-ABI library of include/pbrt_hip.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  Flags that matter for parity with the CPU oracle:
  -ffp-contract=off   no fused multiply-add is formed (hipcc's default would contract)
  (no -ffast-math)    IEEE division / sqrt stay correctly rounded, NaN semantics kept
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PBRT_HIP_LIB_DIR: build / load a variant of the library elsewhere in the tree (A-B experiments, tools/ab_variants.sh)
LIB_DIR = os.path.abspath(os.environ.get("PBRT_HIP_LIB_DIR") or os.path.join(HERE, "lib"))
LIB_PATH = os.path.join(LIB_DIR, "libpbrt_hip.so")
CLI_PATH = os.path.join(LIB_DIR, "pbrt")  # the C++ command line (csrc/pbrt_main.cpp)
SOURCES = ["capi.cpp", "multi_gpu.cpp", "bvh_build.cpp", "reinsert_batch.cpp", "imageio.cpp", "scene_parser.cpp", "ply_reader.cpp", "kernels.hip", "bvh_gpu.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


# -fno-slp-vectorize: the SLP vectoriser turns pairs of f32 operations into v_pk_mul / v_pk_add_f32, which issue at half
# rate on gfx950 (tools/ubench) and need v_mov_b64 copies into aligned register pairs: without it render_kernel has the
# same instruction count, 13 VGPRs fewer and runs 3 % faster (the packed forms written by hand in the node step stay)
# -amdgpu-sdwa-peephole=0: SDWA forms are half rate too, and the peephole costs render_kernel 10 VGPRs (106 -> 96: five
# waves per SIMD without a spill)
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sdwa-peephole=0", "-Wall",
                "-Wno-unused-function", f"--offload-arch={ARCH}"]


def compiler_version():
    try:
        return subprocess.run([HIPCC, "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        return "unknown"


def source_id(extra_flags=()):
    """Identity of what a build is made from: a hash of every source under csrc/, the public headers, EVERY flag the kernels are
    compiled with (the fixed list above and the extra ones) and the compiler's version.  It is compiled into the library
    (pbrt_hip_build_id()) and printed in every bench line.  (Until round 4 it also keyed the counter profiles; they are keyed by the
    measured kernel's machine code now, pbrt_amd/isa_id.py: an edit that does not reach that kernel leaves a profile valid.)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "pbrt_hip.h"), os.path.join("..", "..", "include", "pbrt_hip_debug.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(COMMON_FLAGS + list(extra_flags)).encode())
    h.update(compiler_version().encode())
    return h.hexdigest()[:16]


def build_hip(force=False, verbose=False, extra_flags=()):
    """Compile every translation unit with hipcc and link the shared library."""
    os.makedirs(LIB_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "pbrt_hip.h"))
    deps.append(os.path.join(HERE, "..", "include", "pbrt_hip_debug.h"))
    deps.append(os.path.abspath(__file__))
    if not force and _newer(LIB_PATH, deps):
        return LIB_PATH
    objs = []
    extra_flags = list(extra_flags) + os.environ.get("PBRT_HIP_EXTRA_FLAGS", "").split()
    build_id = source_id(extra_flags)
    common = COMMON_FLAGS + [f'-DPBRT_HIP_BUILD_ID="{build_id}"'] + extra_flags
    def compile_one(src):
        obj = os.path.join(LIB_DIR, src.rsplit(".", 1)[0] + ".o")
        cmd = [HIPCC] + common + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as pool:  # the translation units are independent
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    # the command-line binary links the library by relative rpath so that it runs from the snapshot
    cmd = [HIPCC, "-O2", "-std=c++17", os.path.join(CSRC, "pbrt_main.cpp"), "-o", CLI_PATH, "-L", LIB_DIR, "-lpbrt_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
