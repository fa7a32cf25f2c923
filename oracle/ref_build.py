"""Builds oracle/_ref/libref.so: the reference's own __global__ kernel bodies, compiled for the CPU.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (the build container).  The kernel text
is read from the reference tree at build time, written to oracle/_ref/gen/kernels.inc (git-ignored)
and compiled with g++ against oracle/ref_shim.h (a fiber-based CPU execution model for CUDA kernels)
plus oracle/ref_driver.cpp (launch shapes + zero-initialised outputs as in the reference's launchers).
No reference source is copied into the repository; only the resulting .so travels to the GPU box.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/third_party"
SOURCES = [
    "pvcnn/functional/src/voxelization/vox.cu",
    "pvcnn/functional/src/interpolate/trilinear_devox.cu",
    "pvcnn/functional/src/ball_query/ball_query.cu",
    "pvcnn/functional/src/grouping/grouping.cu",
    "pvcnn/functional/src/sampling/sampling.cu",
    "pvcnn/functional/src/interpolate/neighbor_interpolate.cu",
    "ChamferDistancePytorch/chamfer3D/chamfer3D.cu",
    "PyTorchEMD/cuda/emd_kernel.cu",
]


def extract_kernels(text):
    """every `[template<...>] __global__ void name(...) { ... }` definition, verbatim."""
    out = []
    for m in re.finditer(r"(template\s*<[^>]*>\s*)?__global__\s+void\s+(\w+)\s*\(", text):
        i = text.index("{", text.index(")", m.end() - 1) if False else _sig_end(text, m.end() - 1))
        depth, j = 0, i
        while True:
            if text[j] == "{":
                depth += 1
            elif text[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        out.append((m.group(2), text[m.start():j + 1]))
    return out


def _sig_end(text, open_paren):
    depth, j = 0, open_paren
    while True:
        if text[j] == "(":
            depth += 1
        elif text[j] == ")":
            depth -= 1
            if depth == 0:
                return j
        j += 1


def main():
    if not os.path.isdir(REF):
        print("ref_build: /root/reference not present, nothing to do")
        return 0
    gen = os.path.join(HERE, "_ref", "gen")
    os.makedirs(gen, exist_ok=True)
    names = []
    with open(os.path.join(gen, "kernels.inc"), "w") as f:
        f.write("// GENERATED at build time from /root/reference -- do not commit (oracle/_ref is git-ignored)\n")
        for src in SOURCES:
            text = open(os.path.join(REF, src)).read()
            for name, body in extract_kernels(text):
                if name == "furthest_point_sampling_kernel":
                    # The kernel reads dists_i[0] after its last barrier and then starts the next round,
                    # whose first shared-memory write (dists[threadIdx.x] = best) is not separated from
                    # that read by a barrier (sampling.cu:162-165 -> :141): a benign race on a GPU (every
                    # warp reads long before any warp finishes the next distance loop), but fatal for
                    # fibers that run one thread at a time.  One extra barrier after the read makes the
                    # intended order explicit without changing any result.
                    assert body.count("old = dists_i[0];") == 1
                    body = body.replace("old = dists_i[0];", "old = dists_i[0]; __syncthreads();")
                f.write(f"\n// ---- {src} :: {name}\n{body}\n")
                names.append(name)
    so = os.path.join(HERE, "_ref", "libref.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-fvisibility=hidden", "-w", "-I", HERE, os.path.join(HERE, "ref_driver.cpp"), "-o", so]
    subprocess.check_call(cmd)
    print(f"ref_build: {len(names)} reference kernels -> {so}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
