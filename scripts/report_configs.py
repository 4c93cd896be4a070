"""Runs bench.py on every BASELINE.json config that fits one GPU and collects the JSON lines
(SURVEY.md 8d "Report"): python scripts/report_configs.py [out.json]   (run on the GPU box)"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [
    ("configs[0] african_head / default / 800x800", ["--model", "african_head", "--pipeline", "default", "--size", "800"]),
    ("configs[1] diablo / phong / 2048x2048", ["--pipeline", "phong", "--size", "2048"]),
    ("configs[2] diablo / darboux / 4096x4096", ["--pipeline", "darboux", "--size", "4096"]),
    ("configs[3] diablo / shadow / 4096x4096", ["--pipeline", "shadow", "--size", "4096"]),
    ("configs[4] diablo x64 grid / specular / 8192x8192 (one GPU)", ["--pipeline", "specular", "--size", "8192", "--grid", "8",
                                                                      "--steps", "80", "--warmup", "8"]),
    ("metric: diablo / phong / 4096x4096", ["--pipeline", "phong", "--size", "4096"]),
]
out = {}
for name, flags in CONFIGS:
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--cpu-seconds", "3", "--no-configs"] + flags
    if "--steps" not in flags:
        cmd += ["--steps", "800", "--warmup", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not line:
        out[name] = {"error": (r.stderr or r.stdout)[-400:]}
        print(name, "FAILED", out[name]["error"], flush=True)
        continue
    j = json.loads(line[-1])
    out[name] = j
    print("%-62s %8.1f us/frame (per-frame calls %.1f, unfused %.1f)  k_tile %s us/frame x %d  frac %.3f  parity %s  mem %s MB  cpu %s" % (
        name, j["ms_per_step"] * 1e3, (j.get("per_frame_protocol") or {}).get("ms_per_step", 0.0) * 1e3,
        (j.get("per_frame_protocol") or {}).get("unfused_ms_per_step", 0.0) * 1e3,
        j["kernel_us_per_frame"].get("k_tile"), j["config"]["frames_per_launch"], j["roofline"]["frac"], j["parity_vs_oracle"]["ok"], j.get("device_memory_mb"),
        j["cpu_baseline"]["sample"].split(",")[-1].strip() if j.get("cpu_baseline") else "-"), flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "configs.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
