"""Compare split-f16 GEMM kernel variants (RLCF_X3_KERNEL is read once per process: one subprocess per variant)."""
import os, subprocess, sys
shapes = sys.argv[1:] or ["100864x2304x768", "100864x768x768", "100864x3072x768", "100864x768x3072", "403456x3072x768", "12608x3072x768"]
for v in os.environ.get("VARIANTS", "3,2,1").split(","):
    env = dict(os.environ, RLCF_X3_KERNEL=v)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gemm_bench.py"), *shapes], env=env, capture_output=True, text=True)
    print(f"--- variant {v}")
    print("\n".join(l for l in out.stdout.splitlines() if "prec=2" in l))
    if out.returncode: print(out.stderr[-2000:])
