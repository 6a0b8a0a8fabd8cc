"""Hardware-counter collection around a child process (rocprofv3), for bench.py's roofline block and profiles/.

bench.py cannot put rocprofv3 around itself, so it starts `rocprofv3 ... -- python3 bench.py --pmc-child ...` as a
CHILD process (never an exec: the parent has initialised the GPU) once per counter group — `--pmc` together with
`--kernel-trace` only, one group per pass, FETCH_SIZE and WRITE_SIZE in separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots) prescribes — and reads the CSVs back.

Units and gfx950 corrections applied in hbm_bytes(): FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE tallies
128-B requests at 64 B for wide coalesced reads, so the read side is doubled (an upper bound for narrower access
shapes; the guide calls other widths uncalibrated). SQ_* cycle counters are quad-cycles summed over waves.
"""
import csv
import glob
import os
import shutil
import subprocess
import sys
from collections import defaultdict

SQ_TIME = "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"
SQ_FLOPS = ("SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 "
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU")
TCC_HITS = "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
DEFAULT_GROUPS = ("FETCH_SIZE", "WRITE_SIZE", SQ_TIME, SQ_FLOPS)

NUM_SIMDS = 1024                    # 256 CUs x 4
NUM_SHADER_ENGINES = 32
VECTOR_FP64_PEAK_TFLOPS = 78.6      # MI355X vector fp64
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def rocprof():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def _short(name):
    """Kernel names without template arguments / parameter lists (stable keys across builds)."""
    n = name.strip('"')
    for cut in ("(", "<"):
        i = n.find(cut)
        if i > 0:
            n = n[:i]
    return n.replace("void ", "").strip()


def _run(pmc, child_argv, outdir, timeout):
    exe = rocprof()
    if exe is None:
        raise RuntimeError("rocprofv3 not found")
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir, exist_ok=True)
    cmd = [exe, "--kernel-trace"]
    if pmc:
        cmd += ["--pmc"] + pmc.split()
    cmd += ["-d", outdir, "-o", "run", "--output-format", "csv", "--", sys.executable] + list(child_argv)
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"rocprofv3 pass failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-400:]}")


def kernel_times(outdir):
    """{kernel: {"calls": n, "avg_us": t, "first_us": t0}} from a --kernel-trace pass. avg_us is the STEADY-STATE average: a
    kernel's first launch of the process (code-object load, cold caches and TLBs) is left out when there are others."""
    per = defaultdict(list)
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[_short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = {}
    for k, v in per.items():
        v.sort()
        d = [x[1] for x in v]
        steady = d[1:] if len(d) > 1 else d
        out[k] = {"calls": len(d), "avg_us": sum(steady) / len(steady), "first_us": d[0]}
    return out


def counter_means(outdir):
    """{kernel: {counter: mean over dispatches of the per-dispatch sum over XCD rows, "dispatches": n, "avg_us": t}}."""
    acc, span = defaultdict(float), {}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Dispatch_Id"], _short(r["Kernel_Name"]))
            acc[k + (r["Counter_Name"],)] += float(r["Counter_Value"])
            span[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per = defaultdict(lambda: defaultdict(list))
    for (d, k, c), v in acc.items():
        per[k][c].append(v)
    out = {}
    for k, cs in per.items():
        out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        out[k]["dispatches"] = max(len(v) for v in cs.values())
        t = [v for (d, kk), v in span.items() if kk == k]
        out[k]["avg_us_under_pmc"] = sum(t) / len(t)
    return out


def collect(child_argv, workdir, groups=DEFAULT_GROUPS, timeout=240):
    """One un-instrumented --kernel-trace pass (durations) + one --pmc pass per group. Returns
    {"times": kernel_times, "counters": {kernel: {counter: mean}}, "errors": [...]}; a failing pass is reported, not
    raised, so a bench run on a box without rocprofv3 still prints its line."""
    res = {"times": {}, "counters": defaultdict(dict), "errors": []}
    try:
        d = os.path.join(workdir, "trace")
        _run(None, child_argv, d, timeout)
        res["times"] = kernel_times(d)
    except Exception as e:          # noqa: BLE001 (reported in the JSON line)
        res["errors"].append(f"kernel-trace: {e}")
    for i, g in enumerate(groups):
        try:
            d = os.path.join(workdir, f"pmc{i}")
            _run(g, child_argv, d, timeout)
            for k, cs in counter_means(d).items():
                res["counters"][k].update(cs)
        except Exception as e:      # noqa: BLE001
            res["errors"].append(f"pmc[{g.split()[0]}..]: {e}")
    res["counters"] = dict(res["counters"])
    return res


def hbm_bytes(c):
    """Fabric-side bytes per launch from FETCH_SIZE / WRITE_SIZE (KB), with the gfx950 read-side correction."""
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


def valu_summary(c, launch_us):
    """What binds a VALU-heavy kernel, from the SQ counters of one launch.
    lane_util   = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): active lanes per issued vector instruction
    valu_share  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: share of a wave's life spent issuing vector instructions
    wait_share  = SQ_WAIT_ANY / SQ_WAVE_CYCLES: share parked in s_waitcnt / barriers
    fp64_tflops = (2 FMA + ADD + MUL + TRANS) * 64 * lane_util / time: useful vector fp64 rate"""
    out = {}
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        out["lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES: shader clocks a shader engine was busy, summed over the 32 engines = 32 x the launch in clocks;
        # SQ_ACTIVE_INST_VALU: quad-cycles (4 clocks) in which a wave had a vector instruction executing, summed over waves;
        # two waves of a SIMD never overlap there, so the quotient is the share of the launch the 1024 vector pipes were busy
        out["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (NUM_SIMDS * c["SQ_BUSY_CYCLES"] / NUM_SHADER_ENGINES)
        if "lane_util" in out:
            out["issue_slot_frac"] = out["valu_busy"] * out["lane_util"]     # lane-issue slots of the launch that carried a live lane
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        for key, name in (("SQ_ACTIVE_INST_VALU", "valu_share"), ("SQ_WAIT_ANY", "wait_share"),
                          ("SQ_WAIT_INST_ANY", "issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "active_share")):
            if key in c:
                out[name] = c[key] / wc
    if "SQ_INSTS_VALU_FMA_F64" in c and "lane_util" in out and launch_us:
        wave_instr = 2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0)
        out["fp64_tflops"] = wave_instr * 64.0 * out["lane_util"] / (launch_us * 1e-6) / 1e12
        w32 = 2 * c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0)
        out["fp32_tflops"] = w32 * 64.0 * out["lane_util"] / (launch_us * 1e-6) / 1e12
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) > 0:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    return out
