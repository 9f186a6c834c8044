"""1-GPU proxy of the strong-scaling split (128/G walkers per GPU): bench.py at the per-rank batch
sizes of G = 1, 2, 4, 8 ranks for cfg 2 (and cfg 5 with --cfg5).  Prints one JSON object.
    python tools/batch_sweep.py [--cfg5] > profiles/r02_batch_sweep.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
plans = [("cfg2", [128, 64, 32, 16])]
if "--cfg5" in sys.argv:
    plans.append(("cfg5", [32, 16, 8, 4]))
out = {}
for cfg, batches in plans:
    rows = []
    for B in batches:
        steps = "3" if cfg == "cfg5" else "10"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--batch", str(B), "--steps", steps,
                            "--warmup", "2", "--cpu-sample", "0", "--no-structured", "--no-extra-legs"], capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append(dict(batch=B, ranks_equivalent=batches[0] // B, evals_per_s=d["value"], ms_per_step=d["ms_per_step"],
                         ms_per_eval=d["ms_per_step"] / B, panel_frac_of_peak=d["roofline"].get("panel_frac", d["roofline"]["frac"]), whole_path_frac=d["roofline"]["frac"]))
    base = rows[0]["ms_per_eval"]
    for r_ in rows:
        r_["per_eval_efficiency_vs_full_batch"] = base / r_["ms_per_eval"]
    out[cfg] = rows
print(json.dumps(out, indent=1))
