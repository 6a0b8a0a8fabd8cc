"""Manual profiling target (not collected by pytest): the CG form of the screened-Poisson solve (GDPT_SOLVER_CG: cg_step_a /
cg_step_b, two launches per iteration) on a 512x512 and a 1280x720 right-hand side, for `rocprofv3 --kernel-trace --stats`.
    python tests/prof_cg.py        prints iterations and whole-solve time; the per-kernel times come from the profiler"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gdpt_amd as G
from test_poisson_oracle import lcg_fields
dev = torch.device("cuda", 0)
for w, h in ((512, 512), (1280, 720)):
    c, gx, gy = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lcg_fields(w, h, seed=1))
    out = torch.zeros_like(c)
    for rep in range(3):
        ps = G.poisson_solve_device(w, h, c.data_ptr(), gx.data_ptr(), gy.data_ptr(), out.data_ptr(), solver=G.SOLVER_CG, want_stats=True)
    n = w * h * 3
    print(f"{w}x{h}: CG {ps.iterations} iterations, {ps.solve_ms:.3f} ms whole solve; algorithmic bytes per iteration 64 N = {64 * n / 1e6:.1f} MB "
          f"(cg_step_a: 5 reads + 2 writes, cg_step_b: 5 reads + 2 writes of N = {n} doubles... see DESIGN 4.2)", flush=True)
