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


@pytest.mark.parametrize("src,flags", [("softmax_viterbi.hip", ["-DSV_ONLY_KS=4"]), ("softmax_viterbi.hip", ["-DSV_ONLY_KS=6"]),
                                       ("gemm_rows_f16x3.hip", []), ("lstm_scan16.hip", []), ("gru_bar16.hip", []),
                                       ("gru_bar16d.hip", []), ("gru_bar16q.hip", []), ("gru_scan16.hip", []), ("lstm_bwd16.hip", []),
                                       ("lstm_fused16.hip", []), ("gru_bwd16.hip", []),
                                       ("gru_scan1t.hip", []), ("gemm_bf16x6.hip", []), ("train.hip", []),
                                       ("softmax_viterbi.hip", ["-DSV_ONLY_KS=7"]), ("softmax_viterbi.hip", ["-DSV_ONLY_KS=8"])])
def test_no_mfma_destination_over_live_operands(tmp_path, src, flags):
    import mfma_overlap_scan
    from sloika_amd import build
    out = str(tmp_path / (src + ".s"))
    cmd = [build.hipcc()] + build.flags_for(src) + flags + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
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
    from sloika_amd import build
    out = str(tmp_path / (src + ".s"))
    cmd = [build.hipcc()] + build.flags_for(src) + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert open(out).read().count("global_load_dword") > nloads         # the scan has something to look at
    bad = inflight_load_scan.scan(out)
    assert not bad, bad[:3]


@pytest.mark.parametrize("src", ["gru_bar16.hip", "gru_bar16d.hip", "gru_bar16q.hip", "gru_scan16.hip", "gru_scan1t.hip", "lstm_fused16.hip",
                                 "gru_bwd16.hip", "lstm_bwd16.hip"])
def test_recurrent_kernels_keep_everything_in_registers(tmp_path, src):
    """The persistent scan kernels are sized against the register file by hand (weights in registers for the whole scan); a spill puts
    scratch traffic on the serial chain.  (gru_bar16_kernel<128, 96> once spilled 36 registers unnoticed when its service wave gained
    the x rows it now loads itself.)"""
    from sloika_amd import build
    out = str(tmp_path / (src + ".s"))
    cmd = [build.hipcc()] + build.flags_for(src) + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    sizes = [int(ln.split()[-1]) for ln in open(out) if ".amdhsa_private_segment_fixed_size" in ln]
    assert sizes and max(sizes) == 0, sizes
