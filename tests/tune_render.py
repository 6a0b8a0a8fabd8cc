"""Manual tuning script (not collected by pytest): env knobs x bench of the render kernel on cbox 512x512x16."""
import os, sys, subprocess, json, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests"); sys.path.insert(0, %r + "/oracle")
import gdpt_amd as G, numpy as np
G.debug_knobs.from_env()      # GDPT_* of the parent -> test-only overrides (the library reads no environment)
sd = G.parse_scene(%r + "/scenes/cbox/cbox_gdpt.xml"); sc = G.Scene(sd)
ms = []
for i in range(4):
    bufs, st = sc.render(16, G.RNG_SAMPLE); ms.append(st.render_ms)
print("RESULT", min(ms), st.rays, st.bounces, float(bufs["img"].sum()), float(bufs["cx0"].sum()))
''' % (ROOT, ROOT, ROOT, ROOT)
configs = [{"GDPT_FORCE_EAGER": "1"}]
for wps in ("2", "3", "4"):
    configs.append({"GDPT_WPS": wps})
configs.append({"GDPT_WPS": "2", "GDPT_NO_LDS_SCENE": "1"})
for k in ("0", "1", "2", "3"):
    configs.append({"GDPT_WPS": "2", "GDPT_LOG2K": k})
for cfg in configs:
    env = dict(os.environ); env.update(cfg)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(cfg, line[0] if line else out.stderr[-300:], flush=True)
