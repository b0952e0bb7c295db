"""A/B timing of libmgx variants in separate processes on one GPU: python tools/ab_time.py <script> <lib.so>... (rounds interleaved)."""
import os, subprocess, sys
script, libs = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for so in libs:
        env = dict(os.environ, MGX_LIB=os.path.join(root, so))
        r = subprocess.run([sys.executable] + script.split(), env=env, capture_output=True, text=True, timeout=200, cwd=root)
        out = [l for l in r.stdout.splitlines() if l.strip()]
        print(f"[{rnd}] {os.path.basename(so):28s} " + " | ".join(out[-3:]) + ("" if r.returncode == 0 else f" rc={r.returncode} {r.stderr[-300:]}"), flush=True)
