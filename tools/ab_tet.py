"""A/B: write-through (sc0 sc1) vs plain stores in the blocked tet kernel."""
import os, subprocess, sys
for pl in (1, 0, 1, 0):
    env = dict(os.environ, TETSIM_DEBUG_PLAIN_STORES=str(pl))
    out = subprocess.run([sys.executable, "tools/quick_time.py", "55", "fastonly"], env=env, capture_output=True, text=True).stdout
    print("plain_stores", pl, [l for l in out.splitlines() if "polar fast" in l])
