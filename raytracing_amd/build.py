"""Builds librt_core.so (HIP, gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "librt_core.so")
SOURCES = ["rt_core.hip", "host/rt_host.cpp", "host/rt_multi.cpp"]
HEADERS = ["rt_kernels.hip.h", "rt_trace_refill.hip.h", "rt_bvh_sweep.hip.h", "rt_math.h", "rt_types.h", "rt_compact.h", "../../include/rt_core.h",
           "../../include/rt_host.h", "host/rt_host.hpp", "host/bvh_build.h", "host/json.h", "host/png.h", "host/jpeg.h"]
# -ffp-contract=off: the integrator's float sequence is part of the parity contract.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-shared", "-fPIC"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, out=None, extra_flags=()):
    """out / extra_flags: an experimental variant of the library beside the product build (kernel A/B runs load it through
    the RT_CORE_LIB environment variable, raytracing_amd/_lib.py); never what the tests or the bench load by default."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-o", out or LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lz", "-pthread", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:      # python -m raytracing_amd.build <variant.so> -DRT_EXP_...   (experimental variant)
        build(force=True, verbose=True, out=os.path.abspath(sys.argv[1]), extra_flags=sys.argv[2:])
    else:
        build(force=True, verbose=True)
