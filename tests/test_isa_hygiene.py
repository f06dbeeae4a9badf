"""ISA-level screen of the MFMA kernels: hipcc (ROCm 7.2) may give a freshly defined MFMA destination (C = 0) the registers of an
operand that dies at that instruction or at the MFMA just before it; on MI355X that corrupted rows of v_mfma_f32_32x32x16_f16 under
matrix-pipe contention (csrc/softmax_viterbi.hip, mma_pair).  The 16x16x32 kernels never showed it, but carry the same guards.  The sources keep the operands alive with
empty asm statements; this test compiles them to ISA (hipcc cross-compiles without a GPU) and checks that no such overlap is
left."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

_ISA_CACHE = {}


def _isa(src, flags=()):
    """Path of the gfx950 ISA of csrc/<src> built with the library's flags (+ flags); every file is compiled once per test session."""
    import atexit
    import shutil
    import tempfile
    from sloika_amd import build
    key = (src, tuple(flags))
    if key not in _ISA_CACHE:
        if "dir" not in _ISA_CACHE:
            _ISA_CACHE["dir"] = tempfile.mkdtemp(prefix="slk_isa_")
            atexit.register(shutil.rmtree, _ISA_CACHE["dir"], True)
        out = os.path.join(_ISA_CACHE["dir"], "%s_%d.s" % (src, len(_ISA_CACHE)))
        cmd = [build.hipcc()] + build.flags_for(src) + list(flags) + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        _ISA_CACHE[key] = out
    return _ISA_CACHE[key]


@pytest.mark.parametrize("src,flags", [("softmax_viterbi.hip", ["-DSV_ONLY_KS=4"]), ("softmax_viterbi.hip", ["-DSV_ONLY_KS=6"]),
                                       ("gemm_rows_f16x3.hip", []), ("lstm_scan16.hip", []), ("gru_bar16.hip", []),
                                       ("gru_bar16d.hip", []), ("gru_bar16q.hip", []), ("gru_scan16.hip", []), ("lstm_bwd16.hip", []),
                                       ("lstm_fused16.hip", []), ("gru_bwd16.hip", []),
                                       ("gru_scan1t.hip", []), ("gemm_bf16x6.hip", []), ("train.hip", []),
                                       ("softmax_viterbi.hip", ["-DSV_ONLY_KS=7"]), ("softmax_viterbi.hip", ["-DSV_ONLY_KS=8"])])
def test_no_mfma_destination_over_live_operands(tmp_path, src, flags):
    import mfma_overlap_scan
    out = _isa(src, flags)
    own, war = mfma_overlap_scan.scan(out)
    assert own == 0 and war == 0, "%s: %d MFMA destinations over their own operands, %d over the preceding MFMA's" % (src, own, war)
    # ... and no MFMA reads a register the vector instruction right in front of it wrote: not interlocked either
    # (tools/probes/valu_to_mfma_hazard_probe.hip), and hipcc counts no wait states for the MFMAs inside inline asm
    import mfma_operand_hazard_scan
    bad = mfma_operand_hazard_scan.scan(out)
    assert not bad, "%s: %s" % (src, bad[:3])
    # ... and no vector instruction reads a 16x16x32 MFMA's result before its seventh wait state (tools/probes/mfma_read_hazard_probe.hip:
    # not interlocked; the accumulators are read from inline asm, where hipcc pads nothing)
    import mfma_result_hazard_scan
    early = mfma_result_hazard_scan.scan(out)
    assert not early, "%s: %s" % (src, early[:3])


@pytest.mark.parametrize("src,nloads", [("gru_scan16.hip", 50), ("lstm_scan16.hip", 8), ("lstm_bwd16.hip", 8),
                                        ("lstm_fused16.hip", 4), ("gru_bwd16.hip", 8),
                                        ("gru_scan1t.hip", 6)])
def test_no_instruction_touches_a_register_an_asm_load_is_filling(tmp_path, src, nloads):
    """csrc/gru_scan16.hip and lstm_scan16.hip issue their projection loads as asm, three steps ahead, and count them themselves; the
    compiler must not move such a destination (it once spilled one to an accumulation register right behind the load:
    tools/inflight_load_scan.py)."""
    import inflight_load_scan
    out = _isa(src)
    assert open(out).read().count("global_load_dword") > nloads         # the scan has something to look at
    bad = inflight_load_scan.scan(out)
    assert not bad, bad[:3]


@pytest.mark.parametrize("src", ["gru_bar16.hip", "gru_bar16d.hip", "gru_bar16q.hip", "gru_scan16.hip", "gru_scan1t.hip", "lstm_fused16.hip",
                                 "gru_bwd16.hip", "lstm_bwd16.hip"])
def test_recurrent_kernels_keep_everything_in_registers(tmp_path, src):
    """The persistent scan kernels are sized against the register file by hand (weights in registers for the whole scan); a spill puts
    scratch traffic on the serial chain.  (gru_bar16_kernel<128, 96> once spilled 36 registers unnoticed when its service wave gained
    the x rows it now loads itself.)"""
    out = _isa(src)
    sizes = [int(ln.split()[-1]) for ln in open(out) if ".amdhsa_private_segment_fixed_size" in ln]
    assert sizes and max(sizes) == 0, sizes


#: Scratch (bytes per lane) the compiler is allowed in the MFMA kernels OUTSIDE the persistent scans above: instantiation substring ->
#: bytes.  Everything on the headline path (K = 96: KS 6) is at zero.  The row-GEMM instantiations for K >= 112 run nine waves per
#: workgroup at 168 registers and spill two to seven of them (one, the K = 192 statistics pass, 45); they serve FeedForward / unfused
#: Softmax layers of the wider models and the training step's softmax layer, none of them a latency chain.  The table is a ceiling: a
#: kernel that grows past its entry, or a new instantiation with scratch, fails here instead of slowing down unnoticed.
SCRATCH_BUDGET = {
    "gemm_rows_f16x3.hip": {"ILi7ELb0ELi1E": 8, "ILi7ELb1ELi0ELb1ELi1E": 20, "ILi8ELb1ELi0E": 20, "ILi8ELb0ELi0ELb1ELi0E": 8,
                            "ILi8ELb0ELi1E": 12, "ILi8ELb0ELi2E": 12, "ILi8ELb0ELi0ELb1ELi2E": 28, "ILi9ELb1E": 20, "ILi9ELb0E": 12,
                            "ILi10ELb1E": 20, "ILi10ELb0E": 12, "ILi11ELb0ELi3E": 28, "ILi11ELb0E": 12, "ILi12ELb1E": 180,
                            "ILi12ELb0ELi2E": 16, "ILi12ELb0ELi3E": 68},
    "recurrent.hip": {"gru_mfma_kernelILi144E": 36},
    "lstm_scan16.hip": {"lstm_scan16_kernelILi128E": 16},
    "gru_backward_mfma.hip": {"gru_backward_mfma_kernelILi112E": 68},
    # the log-posterior DUMP instantiations exist for the tests only (they store every log-posterior: 3.4 GB at full size)
    "softmax_viterbi.hip": {"Lb1EEv": 160},
    "gemm_bf16x6.hip": {}, "gemm.hip": {}, "train.hip": {}, "lstm_mfma.hip": {}, "gemm_rows.hip": {}, "decode.hip": {},
}


@pytest.mark.parametrize("src", sorted(SCRATCH_BUDGET))
def test_mfma_kernels_stay_within_their_scratch_budget(tmp_path, src):
    out = _isa(src)
    name, over = None, []
    for ln in open(out):
        if ln.startswith(".amdhsa_kernel "):
            name = ln.split()[1]
        elif ".amdhsa_private_segment_fixed_size" in ln and name:
            size = int(ln.split()[-1])
            allowed = max([v for k, v in SCRATCH_BUDGET[src].items() if k in name] or [0])
            if size > allowed:
                over.append((name, size, allowed))
    assert not over, over


def test_hipcc_is_the_validated_one():
    """The wait states of the hand-scheduled kernels were counted, and the static screens of this file run, on ONE compiler; on another
    the schedule around the inline asm is another.  Fails (not skips) so that a toolchain change cannot go unnoticed: re-run this file
    and the GPU suite on the new compiler, then update build.VALIDATED_HIPCC."""
    from sloika_amd import build
    assert build.hipcc_is_validated(), "hipcc is %r, validated on %r" % (build.hipcc_version(), build.VALIDATED_HIPCC)


def test_backtrace_rows_kernel_owns_m0():
    """viterbi_backtrace_rows_kernel (csrc/decode.hip) sets m0 inside its asm rows (the lane v_writelane writes) without naming it as
    a clobber -- hipcc refuses reserved registers there.  That is sound as long as the compiler itself has no use for m0 in that
    kernel: every mention of m0 in its ISA must be one of the two instructions of the asm."""
    out = _isa("decode.hip")
    inside, seen = False, 0
    for ln in open(out):
        if ln.startswith("_Z29viterbi_backtrace_rows_kernel"):
            inside = True
        elif inside and "s_endpgm" in ln:
            break
        elif inside and "m0" in ln.split(";")[0]:
            ins = ln.split()[0]
            assert ins in ("s_mov_b32", "v_writelane_b32"), ln
            seen += 1
    assert inside and seen >= 2
