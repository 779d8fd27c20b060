"""Build recipe for libtinyvc_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("TVC_LIB_PATH") or os.path.join(HERE, "libtinyvc_hip.so")
SOURCES = ["api.hip", "ragged.hip", "frontdoor.hip", "frontend.hip", "fft.hip", "encoder.hip", "knn.hip", "knn_general.hip", "decoder.hip", "filter_up24s.hip", "conv48s.hip", "sola.hip"]
# -ffp-contract=on: a*b+c written as ONE expression may become an fma, but products and sums that the source keeps apart
# (the __fmul_rn / __fadd_rn helpers are plain inline functions in HIP, not barriers) are NOT fused after inlining - the
# default (fast) fused them, which broke the op-for-op restatements of ATen arithmetic (shift_frequency, pitch softmax, OLA).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=on"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = FLAGS + os.environ.get("TVC_EXTRA_FLAGS", "").split()   # e.g. -DUP24_NT=1024 for A/B runs
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "tinyvc_hip.h"))
    objs, jobs = [], []
    objdir = os.environ.get("TVC_OBJ_DIR") or CSRC      # variant builds (TVC_EXTRA_FLAGS + TVC_LIB_PATH) keep their objects apart
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
