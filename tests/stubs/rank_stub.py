"""Stand-in for one rank of bench.py (tests/test_bench_launch.py): reports the environment the launcher gave it.
STUB_FAIL_RANK=r: that rank exits with STUB_FAIL_CODE after a moment; STUB_HANG=1: every other rank then sleeps 'for ever'."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
fail = int(os.environ.get("STUB_FAIL_RANK", "-1"))
if rank == fail:
    time.sleep(0.2)
    sys.stderr.write("stub rank %d failing\n" % rank)
    sys.exit(int(os.environ.get("STUB_FAIL_CODE", "3")))
if os.environ.get("STUB_HANG") == "1":
    time.sleep(600)
if rank == 0:
    print(json.dumps({"rank": rank, "world": world, "local_rank": int(os.environ["LOCAL_RANK"]), "addr": os.environ["MASTER_ADDR"],
                      "port": int(os.environ["MASTER_PORT"]), "argv": sys.argv[1:]}), flush=True)
else:
    print("noise from rank %d that must not reach the launcher's stdout" % rank, flush=True)
