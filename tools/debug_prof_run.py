import ctypes, sys, os, subprocess, json
sys.path.insert(0, os.getcwd())
import torch
from desco_amd import _lib
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
L = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
import io, contextlib
bench.main()
torch.cuda.synchronize()
L.desco_debug_shmp_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.desco_debug_shmp_prof(buf, 1)
w, mm, epi, top, tiles, waves, sw = [buf[i] for i in range(7)]
print("PROF cycles per tile per wave: wait %.0f  put+mfma %.0f  epilogue %.0f  top %.0f  switch %.0f   tiles/wave %.1f  total/tile %.0f" % (w / tiles, mm / tiles, epi / tiles, top / tiles, sw / tiles, tiles / waves, (w + mm + epi + top + sw) / tiles))
