"""Scan gfx950 ISA (hipcc --cuda-device-only -S) for MFMA destinations that overlap operand registers:
   (1) of the same instruction, (2) of the MFMA issued just before it (within 12 lines).
hipcc (ROCm 7.2) emits both for v_mfma_f32_32x32x16_f16 when the destination is freshly defined (C = 0); on MI355X they
produced wrong rows under matrix-pipe contention (csrc/softmax_viterbi.hip, mma_pair): a 32x32 tile is sixteen registers that
the instruction writes back in groups while later passes still read A and B.  A 16x16 tile is four registers written once, after
the last pass has read its operands: there the overlap is the compiler's normal register reuse (every kernel built with
-amdgpu-mfma-vgpr-form has it), counted separately (`scan(path, wide_only=False)`) and screened on the device instead
(tests/test_gpu_gru_bar16.py: repeated full-size launches bit for bit, agreement with the float32 kernels under load).
    python tools/mfma_overlap_scan.py file.s ..."""
import re
import sys

PAT = re.compile(r'\s*(v_mfma_\S+)\s+([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*(\S+)')


def scan(path, wide_only=True):
    """(own, war) over the MFMAs with 32x32 tiles (wide_only) or over all of them."""
    own, war, prev = 0, 0, []
    for ln, line in enumerate(open(path)):
        m = PAT.match(line)
        if not m:
            continue
        op, dt, d0, d1, at, a0, a1, bt, b0, b1, c = m.groups()
        d0, d1, a0, a1, b0, b1 = map(int, (d0, d1, a0, a1, b0, b1))
        srcs = [(at, a0, a1), (bt, b0, b1)]
        if wide_only and "32x32" not in op:
            prev.append((ln, [], (d0, d1)))
            continue
        for (t, lo, hi) in srcs:
            if t == dt and not (hi < d0 or lo > d1):
                own += 1
        for (pln, psrcs, pd) in prev[-1:]:
            if ln - pln > 12:
                continue
            for (t, lo, hi) in psrcs:
                if t == dt and not (hi < d0 or lo > d1) and (lo, hi) != pd:
                    war += 1
        prev.append((ln, srcs, (d0, d1)))
    return own, war


if __name__ == "__main__":
    bad = 0
    for f in sys.argv[1:]:
        own, war = scan(f)
        print("%s: 32x32 destinations over their own operands %d times, over operands of the preceding MFMA %d times; all tile "
              "shapes: %s" % (f, own, war, scan(f, wide_only=False)))
        bad += own + war
    sys.exit(1 if bad else 0)
