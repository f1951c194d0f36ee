"""Cycle stamps of the producer/consumer Gram kernel (workgroup 0): where a tile's time goes.
usage: PAROPT_AMD_WGRAM_ABLATE=16 python tools/dbg/wgram_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa
from paropt_amd.lib import lib

ctx = pa.Context(0)
n, m = 50_000_000, int(os.environ.get("STAMP_COLS", "43"))
d = pa.PVec(ctx, n).fill_hash(0, 9, 0, 1.0, 0.5)
V = [pa.PVec(ctx, n).fill_hash(0, 20 + j, 0, 2.0, -1.0) for j in range(m)]
for rep in range(3):
    ctx.time_wgram(True)
    pa.wgram(d, V, rhs_last=True)
    ms = ctx.time_wgram_result(0)[0]
    out = (C.c_double * 8)()
    lib.po_debug_wgram_stamps.restype = C.c_int
    lib.po_debug_wgram_stamps(out)
    o = list(out)
    nt = max(o[5], 1.0)
    cyc = (o[2] + o[3] + o[4]) / nt
    print("kernel %.3f ms, %d tiles per workgroup -> %.3f us per tile | shader cycles per tile (s_memtime): consumer wait %.0f "
          "work %.0f | producer stage (incl. wait for its loads) %.0f load-issue %.0f barrier-wait %.0f | implied clock "
          "%.2f GHz, %.1f B/cycle/CU" % (ms, nt, ms * 1e3 / nt, o[0] / nt, o[1] / nt, o[2] / nt, o[3] / nt, o[4] / nt,
                                          cyc / (ms * 1e3 / nt) * 1e-3, (m + 1) * int(os.environ.get("STAMP_ROWS", "128")) * 8 / cyc))
    if o[7] > 0:
        print("   loop: %.0f s_memtime ticks in %.3f ms of the 100 MHz counter -> s_memtime runs at %.0f MHz" % (
            o[6], o[7] / 1e5, o[6] / o[7] * 100.0))
