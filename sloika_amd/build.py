"""Build recipe for libsloika_amd.so: every csrc/*.hip compiled for gfx950 with hipcc and linked into ONE
C-ABI shared library kept in-tree (sloika_amd/_build/), next to the sources.

    python -m sloika_amd.build [--force]

hipcc cross-compiles without a GPU, so this also serves as the CPU-side "does it build" check.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libsloika_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Werror", "-Wno-unused-function",
         "-Wno-unused-command-line-argument",          # (hipcc's own --hip-link when a test compiles to ISA only)
         "-fno-fast-math", "-ffp-contract=off", "-fvisibility=hidden"]
# -fvisibility=hidden: the library exports what include/sloika_amd.h declares (SLK_API) and nothing else.
# IEEE divide/sqrt, no reassociation, and NO implicit fma contraction (hipcc's default is -ffp-contract=fast, and
# __fmul_rn/__fadd_rn are plain operators in the HIP headers): prepare_post, the normalisation and the DP kernels
# must round exactly like numpy's float32 evaluation.  Kernels that want fused multiply-adds call fmaf / MFMA.


# -amdgpu-mfma-vgpr-form: compiler-generated MFMAs keep their accumulators in ordinary registers.  Without it a kernel that
# names accumulation registers in inline asm (weights as "a" operands) gets ALL its MFMA results in accumulation registers and
# pays four v_accvgpr_read per tile before the gate arithmetic can touch them (gru_bar16_kernel<96,96>: 112 of them, 28 per
# step on the serial chain).  Only for the files written for it: their asm reads of accumulators keep the eight wait states
# behind an MFMA themselves (the compiler pads in front of its own v_accvgpr_read, never inside asm).
VGPR_FORM = ("gru_bar16.hip", "gru_bar16d.hip", "gru_bar16q.hip")


def flags_for(src):
    """Compile flags of one source file of csrc/."""
    return FLAGS + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.path.basename(src) in VGPR_FORM else [])


#: The hipcc the hand-counted wait states of the scan kernels and the static ISA screens (tests/test_isa_hygiene.py) were validated on.
#: A kernel that reads MFMA results or writes MFMA operands from inline asm depends on the instruction schedule the compiler emits around
#: it (design/gru_round4.md 1, 7; -amdgpu-mfma-vgpr-form is an internal flag): another compiler is another schedule, so the build says so
#: loudly and tests/test_isa_hygiene.py::test_hipcc_is_the_validated_one fails until the screens have been re-run and this updated.
VALIDATED_HIPCC = "roc-7.2.0"


def hipcc_version():
    """First line of `hipcc --version` that names the compiler build (e.g. "AMD clang version 22.0.0git (... roc-7.2.0 ...)")."""
    try:
        out = subprocess.run([hipcc(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError, RuntimeError):
        return None
    for ln in out.splitlines():
        if "clang version" in ln:
            return ln.strip()
    return out.strip().splitlines()[0] if out.strip() else None


def hipcc_is_validated():
    v = hipcc_version()
    return bool(v and VALIDATED_HIPCC in v)


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the sloika_amd HIP extension cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "sloika_amd.h")]
    cc = hipcc()
    jobs = []
    objs = []
    for src in srcs:
        obj = os.path.join(OUT, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([cc] + flags_for(src) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r.returncode, r.stdout

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for cmd, rc, out in ex.map(run, jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
                if verbose and out.strip():
                    print(out)
    if jobs or force or _stale(LIB, objs):
        cmd, rc, out = run([cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs)
        if rc != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), out))
        # which compiler built it, next to the library (read by tests/test_isa_hygiene.py and printed by `python -m sloika_amd.build`)
        with open(os.path.join(OUT, "hipcc_version.txt"), "w") as fh:
            fh.write("%s\nvalidated: %s (%s)\n" % (hipcc_version(), VALIDATED_HIPCC, "yes" if hipcc_is_validated() else "NO"))
    if not hipcc_is_validated():
        sys.stderr.write("sloika_amd.build: WARNING: this hipcc (%s) is not the one the hand-scheduled kernels' wait states were validated "
                         "on (%s): run tests/test_isa_hygiene.py and the GPU suite before trusting the library\n"
                         % (hipcc_version(), VALIDATED_HIPCC))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print("hipcc:", hipcc_version(), "| validated:", "yes" if hipcc_is_validated() else "NO (%s)" % VALIDATED_HIPCC)
