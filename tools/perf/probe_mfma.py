import ctypes as C, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from playableenvironments_amd import _lib
import torch
torch.zeros(1).cuda()
lib = _lib.load()
for name in ("pr_probe_mfma_f32", "pr_probe_mfma_f16"):
    for rnd in (0, 1):
        for rep in range(2):
            tf, ms = C.c_double(), C.c_double()
            _lib.check(getattr(lib, name)(200000 if "f32" in name else 100000, rnd, C.byref(tf), C.byref(ms), None), name)
            print(name, "random" if rnd else "const", round(tf.value, 1), "TF/s", round(ms.value, 2), "ms")
