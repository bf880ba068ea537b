"""bench.py's cpu_baseline, plan A: the oracle orchestration over a live cv2 with the windows fanned
out (oracle/cv2_backend.py).  No cv2 exists in this image or on the GPU pool, so the leg is exercised in a subprocess whose
`cv2` is the throw-away stand-in that forwards to the C oracle (tests/golden/_cv2_standin): that proves the plumbing -- the
fan-out, the stage clock, the JSON fields -- and, because the stand-in IS the oracle, that the fanned-out run reproduces the
plain one bit for bit.  A result obtained through the stand-in is labelled kind "port", never "opencv"."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "golden", "_cv2_standin")

CODE = r"""
import json, sys
import numpy as np
import bench
from microaligner_amd import synthetic
from oracle import cv2_backend, register_oracle as RO
params = dict(num_pyr_lvl=2, num_iterations=2, tile_size=100, overlap=20, use_full_res_img=True, use_dog=True)
line = bench.cpu_baseline(300, params)
ref, mov = synthetic.make_pair(300, 300, 1)
flow, rep, warped = cv2_backend.register_over_cv2(ref, mov, workers=4, **params)
flow0, rep0 = RO.register(ref, mov, **params)
same = bool(np.array_equal(flow, flow0) and np.array_equal(warped, RO.warp(mov, flow0, 100, 20))
            and [r[3] for r in rep] == [r[3] for r in rep0] and np.allclose([r[1:3] for r in rep], [r[1:3] for r in rep0], rtol=0, atol=1e-12))   # NMI by scikit-learn here
print(json.dumps({"line": line, "same": same}))
"""


def test_cpu_baseline_runs_over_a_cv2_module_and_labels_a_standin_as_port():
    env = dict(os.environ, PYTHONPATH=STANDIN + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    line = out["line"]
    assert out["same"], "the fanned-out orchestration over the cv2 module must reproduce the plain oracle run"
    assert line["kind"] == "port" and line["opencv"]["standin"] is True and line["opencv"]["version"] == "4.5.5"
    assert line["unit"] == "Mpix/s" and line["value"] > 0 and 1 <= line["cores"] <= (os.cpu_count() or 1) and line["cores_limit"]
    assert {"pyramid", "dog", "farneback", "warp", "nmi", "final_warp"} <= set(line["stage_seconds"])


def test_cpu_baseline_falls_back_to_the_oracle_without_cv2():
    code = "import bench, json; print(json.dumps(bench.cpu_baseline(260, dict(num_pyr_lvl=1, tile_size=100, overlap=20))))"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["kind"] == "port" and "opencv" not in line
