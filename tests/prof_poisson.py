"""Manual profiling driver (not collected by pytest): the default Poisson solve at WxH, a few times.   prof_poisson.py [W H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gdpt_amd as G
from test_poisson_oracle import lcg_fields
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
c, gx, gy = lcg_fields(w, h, seed=1)
for _ in range(6):
    out = G.fourierSolve(w, h, c, gx, gy, 0.04)
print(w, h, float(np.abs(out).mean()))
