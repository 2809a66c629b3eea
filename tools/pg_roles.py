"""Developer aid: duration of k_param_grads in the headline step (per-launch timing of the C ABI: glam_prof_*), for the role-isolating
builds of layer.hip (-DGLAM_PG_ONLY=0..3 = A | B | C | D; GLAM_HIP_LIB=...)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-seconds", "0", "--large-batch", "0", "--steps", "2000", "--warmup", "100"],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
ks = d.get("roofline_kernels") or d.get("step_kernels") or {}
print(os.environ.get("GLAM_HIP_LIB", "default"), round(d["ms_per_step"] * 1000, 2), "us/step;",
      {k: round(v.get("avg_us", 0), 2) for k, v in ks.items() if "param" in k or "wgrad" in k})
