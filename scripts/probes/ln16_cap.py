"""Workgroup cap of the tiled 16-bit LayerNorm (Swin): does an oversubscribed grid help there as it does for the int8 kernel?"""
import os; os.environ["IVIT_USE_LAB_LIBRARY"] = "1"
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["x", "ln"]
from ivit_amd import _lib
for cap in (0, 4, 8, 15):
    _lib.call("ivit_debug_ln_ablate", cap << 16)
    print("workgroup cap", cap * 256 or 512, flush=True)
    exec(open(os.path.join(ROOT, "scripts", "time_swin_kernels.py")).read())
