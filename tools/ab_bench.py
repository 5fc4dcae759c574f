#!/usr/bin/env python
"""A/B of two builds of libgags_hip.so on ONE box (boxes of the pool differ by several percent): runs bench.py's headline step
with each library in turn, several rounds, and prints the per-stage medians.
    python tools/ab_bench.py tools/tmp/libgags_base.so gags_amd/csrc/libgags_hip.so [--rounds 3] [--env GAGS_FWD_SPW=2]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
envs = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--env"]  # applied to the LAST library only
code = ("import sys, runpy; sys.path.insert(0, %r); import gags_amd._lib as L; L.LIB_PATH = %r; "
        "sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-heavy']; "
        "runpy.run_path(%r, run_name='__main__')")
res = {}
for r in range(rounds):
    for i, lib in enumerate(libs):
        env = dict(os.environ)
        if i == len(libs) - 1:
            env.update(dict(e.split("=", 1) for e in envs))
        out = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.abspath(lib), os.path.join(ROOT, "bench.py"))],
                             capture_output=True, text=True, env=env, cwd=ROOT)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(lib, "failed:", out.stderr[-400:])
            continue
        res.setdefault(lib, []).append(d)
for lib, ds in res.items():
    med = sorted(d["step_ms"]["median"] for d in ds)
    st = {k: sorted(d["stages_ms"][k] for d in ds)[len(ds) // 2] for k in ds[0]["stages_ms"]}
    print(json.dumps({"lib": lib, "step_ms_median_of_medians": med[len(med) // 2], "all": med, "stages_ms": st}))
