"""bench.py swept over one setting at a time (batch per GPU, T = 64, launch-group sizes): one line per run,
the JSON lines appended to gpurun_out/sweep_bench.jsonl.

usage: python tools/sweep_bench.py batch=1,8,32,128,256,512,1024,2048 tokens=64 vit=65536 prefill=32768 [model=7b]
"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FLAGS = {"batch": "--batch", "tokens": "--new-tokens", "vit": "--vit-chunk", "prefill": "--prefill-chunk"}
model = "7b"
out = ROOT / "gpurun_out" / "sweep_bench.jsonl"
out.parent.mkdir(exist_ok=True)
for arg in sys.argv[1:]:
    name, vals = arg.split("=")
    if name == "model":
        model = vals
        continue
    for v in vals.split(","):
        extra = []
        if name == "batch":  # enough steps for a timed region of a few seconds at every size
            extra = ["--steps", str(max(2, min(12, 1024 // int(v)))), "--warmup", "2"]
        else:
            extra = ["--steps", "2", "--warmup", "1"]
        cmd = [sys.executable, str(ROOT / "bench.py"), "--model", model, "--no-cpu-baseline", "--no-pil-leg", "--scorer-labels", "4096",
               FLAGS[name], v, *extra]
        res = subprocess.run(cmd, capture_output=True, text=True)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, v, "FAILED", res.stderr[-500:], flush=True)
            continue
        d = json.loads(line[0])
        with open(out, "a") as f:
            f.write(json.dumps({"sweep": f"{name}={v}", "model": model, **d}) + "\n")
        print(f"{model} {name}={v}: {d['value']:.1f} images/s  ({d['ms_per_step']:.1f} ms/step, {100 * d['mfma_frac_end_to_end']:.1f} % of the bf16 peak "
              f"end to end, GEMM {d['roofline']['achieved']:.0f} TFLOP/s, from host uint8 {d['images_per_s_from_host_uint8']:.1f})", flush=True)
