"""Child of test_assembly_fused_with_the_solve_gives_the_bits_of_the_two_calls (not collected by pytest)."""
import os, sys
import numpy as np
import torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gdpt_amd as G

ok = True
for solver, which in (("own_mfma", G.SOLVER_DCT_MFMA), ("rocblas", G.SOLVER_DCT), ("cg", G.SOLVER_CG)):
    for w, h in ((2, 2), (3, 2), (33, 97), (64, 48), (320, 180), (512, 512)):
        rng = np.random.default_rng(w * 1000 + h)
        raw = [torch.from_numpy(rng.normal(size=(h, w, 3))).to(dev) for _ in range(5)]
        two = [torch.zeros((h, w, 3), dtype=torch.float64, device=dev) for _ in range(4)]
        one = [torch.full((h, w, 3), 7.0, dtype=torch.float64, device=dev) for _ in range(4)]
        G.assemble_device(w, h, [t.data_ptr() for t in raw], [t.data_ptr() for t in two[:3]])
        G.poisson_solve_device(w, h, two[0].data_ptr(), two[1].data_ptr(), two[2].data_ptr(), two[3].data_ptr(), solver=which)
        st = G.assemble_solve_device(w, h, [t.data_ptr() for t in raw], [t.data_ptr() for t in one[:3]], one[3].data_ptr(), solver=which, want_stats=True)
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(one, two)) and st.solver == which
        finite = bool(torch.isfinite(one[3]).all()) and float(one[3].abs().max()) > 0
        print(f"{solver} {w}x{h} equal: {same} finite: {finite}", flush=True)
        ok = ok and same and finite
print("ALL EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
