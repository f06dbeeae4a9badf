"""Scan gfx950 ISA (hipcc --cuda-device-only -S) for an MFMA whose A, B or C operand was written by the vector instruction right in
front of it.  The hardware does not interlock that (tools/probes/valu_to_mfma_hazard_probe.hip: with no wait state in between the MFMA
multiplies what the register held BEFORE, with one it is correct); hipcc separates the two for the MFMAs it generates, but counts
nothing for an MFMA inside inline asm -- and materialises constants (the zero state of a scan's first step) with v_mov right where they
are first used.  csrc/bar16_common.h settle() is the cure; this scan finds the places that need it.
    python tools/mfma_operand_hazard_scan.py file.s ..."""
import re
import sys

MFMA = re.compile(r'\s*(v_mfma_\S+)\s+([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*([av])\[(\d+):(\d+)\],\s*(\S+)')
REG = re.compile(r'([av])(\d+)\b|([av])\[(\d+):(\d+)\]')


def _dest(line):
    """(file, lo, hi) written by a vector ALU instruction, or None (first operand; v_cmp writes no vector register)."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return None
    op, rest = parts
    if not op.startswith("v_") or op.startswith("v_mfma") or op.startswith("v_cmp") or op.startswith("v_nop"):
        return None
    first = rest.split(",")[0].strip()
    m = REG.fullmatch(first)
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), int(m.group(2))
    return m.group(3), int(m.group(4)), int(m.group(5))


def scan(path):
    """[(line number, MFMA line, writer line)] for MFMAs that read what the instruction in front of them wrote."""
    bad, prev = [], None
    for ln, line in enumerate(open(path), 1):
        text = line.split(";")[0].rstrip()
        if not text.strip() or text.lstrip().startswith(".") or text.rstrip().endswith(":"):
            continue                                  # comments, directives, labels
        m = MFMA.match(text)
        if m and prev is not None:
            _, _, _, _, at, a0, a1, bt, b0, b1, c = m.groups()
            srcs = [(at, int(a0), int(a1)), (bt, int(b0), int(b1))]
            mc = REG.fullmatch(c.strip())
            if mc and mc.group(3):
                srcs.append((mc.group(3), int(mc.group(4)), int(mc.group(5))))
            d = _dest(prev)
            if d and any(t == d[0] and not (hi < d[1] or lo > d[2]) for t, lo, hi in srcs):
                bad.append((ln, text.strip(), prev.strip()))
        prev = text
    return bad


if __name__ == "__main__":
    n = 0
    for f in sys.argv[1:]:
        b = scan(f)
        print("%s: %d MFMAs read what the vector instruction in front of them wrote" % (f, len(b)))
        for ln, mf, wr in b[:10]:
            print("   line %d: %s   <-   %s" % (ln, mf, wr))
        n += len(b)
    sys.exit(1 if n else 0)
