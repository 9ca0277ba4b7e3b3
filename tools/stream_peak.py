import sys; sys.path.insert(0,'.')
from tetsim_amd import measure_stream_bandwidth
for size in (64<<20, 256<<20, 1<<30, 4<<30):
    print(size>>20, "MiB", {k: round(measure_stream_bandwidth(size, k, 10)) for k in ("copy","read","write")}, flush=True)
