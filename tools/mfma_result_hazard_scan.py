"""Scan gfx950 ISA for a vector instruction that reads the destination of a 16x16x32 MFMA too early.  The hardware does not interlock
that read (tools/probes/mfma_read_hazard_probe.hip: the finished product is there from seven wait states on, garbage before); hipcc
pads for instructions it generates but counts an inline-asm statement as one wait state whatever is inside, and csrc/bar16_common.h
reads accumulators from asm (pick_mix).  Rule checked here: between the MFMA and the first vector instruction that reads one of its
destination registers there must be at least `need` wait states, counting one per instruction and N+1 per `s_nop N`, four per MFMA
issued in between (an MFMA holds the issue port of its wave for its first pass).  Dependent MFMAs OF THE SAME SHAPE (the destination
read as C) are the hardware's own business and not counted as readers; an MFMA of another shape (csrc/gru_scan1t.hip mixes 16x16x32
and 16x16x16) is a reader like any vector instruction.
    python tools/mfma_result_hazard_scan.py file.s ..."""
import re
import sys

MFMA = re.compile(r'\s*(v_mfma_f32_16x16x(?:32|16)_\S+)\s+([av])\[(\d+):(\d+)\],')
REGS = re.compile(r'\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]')


def _reads(text):
    """registers read by a vector instruction: every register operand but the first (the destination)"""
    parts = text.split(None, 1)
    if len(parts) < 2:
        return []
    ops = parts[1].split(",")
    if parts[0].startswith(("global_store", "ds_write", "buffer_store", "flat_store")):
        pass                                          # stores read all their operands
    else:
        ops = ops[1:]
    out = []
    for op in ops:
        for m in REGS.finditer(op):
            if m.group(1):
                out.append((m.group(1), int(m.group(2)), int(m.group(2))))
            else:
                out.append((m.group(3), int(m.group(4)), int(m.group(5))))
    return out


def _writes(text):
    parts = text.split(None, 1)
    if len(parts) < 2 or parts[0].startswith(("global_store", "ds_write", "buffer_store", "flat_store", "s_", "v_cmp")):
        return []
    m = REGS.search(parts[1].split(",")[0])
    if not m:
        return []
    return [(m.group(1), int(m.group(2)), int(m.group(2)))] if m.group(1) else [(m.group(3), int(m.group(4)), int(m.group(5)))]


def scan(path, need=7):
    bad, pending = [], []                             # pending: [file, lo, hi, wait states so far, line, text]
    for ln, line in enumerate(open(path), 1):
        text = line.split(";")[0].rstrip()
        st = text.strip()
        if not st or st.startswith("."):
            continue
        if st.endswith(":"):                          # a label: control flow joins, give up on what was pending (the compiler pads per path)
            pending = []
            continue
        op = st.split()[0]
        m = MFMA.match(text)
        if op.startswith("v_"):                       # a vector reader?  (an MFMA is one of MFMAs of another shape)
            rd = _reads(st)
            for p in pending:
                if m and p[5].split()[0] == op:
                    continue
                if p[3] < need and any(t == p[0] and not (hi < p[1] or lo > p[2]) for t, lo, hi in rd):
                    bad.append((ln, st, p[4], p[5], p[3]))
        if op.startswith(("global_store", "ds_write", "buffer_store")):
            rd = _reads(st)
            for p in pending:
                if p[3] < need and any(t == p[0] and not (hi < p[1] or lo > p[2]) for t, lo, hi in rd):
                    bad.append((ln, st, p[4], p[5], p[3]))
        # a write to a pending destination ends its watch (the value is no longer the MFMA's)
        wr = _writes(st) if not m else []
        pending = [p for p in pending if not any(t == p[0] and not (hi < p[1] or lo > p[2]) for t, lo, hi in wr)]
        step = 1
        if op == "s_nop":
            step = int(st.split()[1]) + 1
        elif op.startswith("v_mfma"):
            step = 4
        for p in pending:
            p[3] += step
        pending = [p for p in pending if p[3] < need]
        if m:
            pending.append([m.group(2), int(m.group(3)), int(m.group(4)), 0, ln, st])
    return bad


if __name__ == "__main__":
    n = 0
    for f in sys.argv[1:]:
        b = scan(f)
        print("%s: %d early reads of an MFMA result" % (f, len(b)))
        for ln, rd, mln, mf, ws in b[:12]:
            print("   line %d: %s   reads after %d wait states what line %d wrote: %s" % (ln, rd, ws, mln, mf))
        n += len(b)
    sys.exit(1 if n else 0)
