"""Shader clock the Gru kernel gets INSIDE the whole basecalling step (SLOIKA_AMD_BAR16_DIAG=11|16 must be set before the
library loads: every launch then records where its workgroups ran and for how long; 16 = the zero-column experiment: the state in column
group 0 of the recurrent MFMAs only).   usage: SLOIKA_AMD_BAR16_DIAG=11 python3 tools/bar16_pipeline_clock.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sloika_amd import _lib, models
from sloika_amd.pipeline import Basecaller
L = _lib.lib()
L.slk_debug_read_bar16_wg.argtypes = [ctypes.c_void_p]
net = models.randomise_zero_layers(models.build_model("raw_0.98_rgrgr", klen=5, sd=0.5, seed=11))
bc = Basecaller(net, kmer_len=5)
B, T = 1024, 800
g = torch.Generator(device="cuda"); g.manual_seed(3)
chunks = torch.randn(B, 4000, device="cuda", generator=g)
for _ in range(5): bc.call_chunks(chunks)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): bc.call_chunks(chunks)
e1.record(); torch.cuda.synchronize()
out = np.zeros((1024, 4), dtype=np.uint64)
assert L.slk_debug_read_bar16_wg(out.ctypes.data) == 0
o = out[:256]
cyc, real = o[:, 0].astype(np.float64), o[:, 1].astype(np.float64)
mhz = cyc / real * 100.0
print("DIAG=%s: %.3f ms per step; last Gru launch of the step: %.3f ms (slowest workgroup), %.0f cycles per scan step, shader clock %.0f..%.0f MHz (median %.0f)"
      % (os.environ.get("SLOIKA_AMD_BAR16_DIAG"), e0.elapsed_time(e1) / 20, real.max() / 100e3, np.median(cyc) / T, mhz.min(), mhz.max(), np.median(mhz)))
