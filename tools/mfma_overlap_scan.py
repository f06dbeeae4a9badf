"""Scan gfx950 ISA (hipcc --cuda-device-only -S) for MFMA destinations that overlap operand registers:
   (1) of the same instruction, (2) of the MFMA issued just before it (within 12 lines).
hipcc (ROCm 7.2) emits both for v_mfma_f32_32x32x16_f16 when the destination is freshly defined (C = 0); on MI355X they
produced wrong rows under matrix-pipe contention (csrc/softmax_viterbi.hip, mma_pair).
    python tools/mfma_overlap_scan.py file.s ..."""
import re
import sys

PAT = re.compile(r'\s*(v_mfma_\S+)\s+([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*(\S+)')


def scan(path):
    own, war, prev = 0, 0, []
    for ln, line in enumerate(open(path)):
        m = PAT.match(line)
        if not m:
            continue
        op, dt, d0, d1, at, a0, a1, bt, b0, b1, c = m.groups()
        d0, d1, a0, a1, b0, b1 = map(int, (d0, d1, a0, a1, b0, b1))
        srcs = [(at, a0, a1), (bt, b0, b1)]
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
        print("%s: destination overlaps its own operands %d times, operands of a preceding MFMA %d times" % (f, own, war))
        bad += own + war
    sys.exit(1 if bad else 0)
