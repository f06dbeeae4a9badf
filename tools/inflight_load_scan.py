"""Scan gfx950 ISA (hipcc --cuda-device-only -S) for registers that an asm-issued global load is still filling when another
instruction reads them: csrc/gru_scan16.hip requests the projection of later steps with global_load_dword written as asm (the
compiler does not track them; the kernel counts them with s_waitcnt vmcnt(n) itself).  Under register pressure hipcc copied such
a destination to an accumulation register right after the load was ISSUED -- before the data arrived -- and every output became
NaN.  The loads in question use the `vaddr, s[base]` form, which the compiler's own loads in that file do not.
    python tools/inflight_load_scan.py file.s ..."""
import re
import sys

LOAD = re.compile(r'\s*global_load_dword(?:x[234])? (v\d+|v\[\d+:\d+\]), v\d+, s\[')
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    lines = open(path).read().split('\n')
    bad = []
    for i, l in enumerate(lines):
        m = LOAD.match(l)
        if not m:
            continue
        dst = _regs(m.group(1))
        for j in range(i + 1, min(i + 600, len(lines))):
            t = lines[j]
            if 's_waitcnt' in t and 'vmcnt' in t:
                break
            if t.lstrip().startswith(';') or 'global_load' in t:
                continue
            if dst & _regs(t.split(';')[0]):
                bad.append((i + 1, l.strip(), j + 1, t.strip()))
                break
    return bad


if __name__ == "__main__":
    n = 0
    for f in sys.argv[1:]:
        bad = scan(f)
        for b in bad:
            print("%s:%d %s  <- read at line %d: %s" % ((f,) + b))
        print("%s: %d in-flight destinations touched" % (f, len(bad)))
        n += len(bad)
    sys.exit(1 if n else 0)
