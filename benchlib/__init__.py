"""bench.py in parts: launcher / rank adapters / headline / the legs beside it / the CPU baseline."""
