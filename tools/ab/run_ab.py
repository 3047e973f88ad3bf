"""same-box A/B of libdtrl variants: python tools/ab/run_ab.py <lib> <config> <groups> -> one line"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib, cfg, g = sys.argv[1], sys.argv[2], sys.argv[3]
code = "import sys; sys.path.insert(0, %r); import deepterrainrl_amd as da; da.LIB_PATH = %r; import bench; sys.argv = ['bench.py', '--config', %r, '--steps', '60', '--warmup', '20', '--no-cpu-baseline', '--no-trained-leg', '--exchange-steps', '0']; bench.main()" % (REPO, lib, cfg)
env = dict(os.environ, DTRL_GROUPS=g)
out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()
d = json.loads(out[-1])
print("%s cfg %s G=%s: %.2f M (%.2f-%.2f) kernel %.3f ms" % (os.path.basename(lib), cfg, g, d["value"] / 1e6, d["value_min"] / 1e6, d["value_max"] / 1e6, d["roofline"]["kernel_avg_ms"]), flush=True)
