"""Manual timing (not collected by pytest): Poisson solve, own fp64 MFMA DCT vs the rocBLAS variant vs CG."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gdpt_amd as G
from test_poisson_oracle import lcg_fields
dev = torch.device("cuda", 0)
for w, h in ((512, 512), (1024, 1024), (1280, 720)):
    c, gx, gy = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in lcg_fields(w, h, seed=1))
    out = torch.zeros_like(c)
    res = {}
    shapes = {"own_bk16": dict(dct_bm=64, dct_bk=16), "own_bk32": dict(dct_bm=64, dct_bk=32), "own_bm32": dict(dct_bm=32)}
    for name, which in (("own_mfma", G.SOLVER_DCT_MFMA), ("own_bk16", G.SOLVER_DCT_MFMA), ("own_bk32", G.SOLVER_DCT_MFMA), ("own_bm32", G.SOLVER_DCT_MFMA), ("rocblas", G.SOLVER_DCT)):
        G.debug_knobs.reset()
        if name in shapes:
            G.debug_knobs.set(**shapes[name])
        for _ in range(12):
            G.poisson_solve_device(w, h, c.data_ptr(), gx.data_ptr(), gy.data_ptr(), out.data_ptr(), solver=which)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            G.poisson_solve_device(w, h, c.data_ptr(), gx.data_ptr(), gy.data_ptr(), out.data_ptr(), solver=which)
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 20, out.clone())
    d = (res["own_mfma"][1] - res["rocblas"][1]).abs().max().item()
    fl = 4 * 3 * (2.0 * w * w * h + 2.0 * h * h * w) / 2
    G.debug_knobs.reset()
    print(f"{w}x{h}: own MFMA {res['own_mfma'][0] * 1e3:.1f} us ({fl / res['own_mfma'][0] / 1e9:.1f} TFLOP/s counting the unfolded product; shapes 64xBK16 {res['own_bk16'][0] * 1e3:.1f} us, 64xBK32 {res['own_bk32'][0] * 1e3:.1f} us, 32xBK32 {res['own_bm32'][0] * 1e3:.1f} us), rocBLAS {res['rocblas'][0] * 1e3:.1f} us "
          f"({fl / res['rocblas'][0] / 1e9:.1f} TFLOP/s), max abs diff {d:.2e}", flush=True)
