import os, sys, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
exec(open(os.path.join(ROOT, "tools", "decoder_bench.py")).read())
from gags_amd import _lib
lib = _lib.load()
buf = np.zeros(8192 * 4 * 8, np.uint64)
lib.gags_debug_x3_ts.argtypes = [ctypes.c_void_p]
print("rc", lib.gags_debug_x3_ts(buf.ctypes.data))
t = buf.reshape(8192 * 4, 8).astype(np.float64)
t = t[t[:, 6] > 0]
tot = t[:, 6]
print("waves", len(t), "cycles/wave", tot.mean())
for k, nme in enumerate(["prologue fetch", "commit (split)", "sync1", "fetch+tile_step", "sync2", "epilogue"]):
    print(f"{nme:16s} {t[:, k].mean():10.0f}  {100 * t[:, k].sum() / tot.sum():5.1f} %")
