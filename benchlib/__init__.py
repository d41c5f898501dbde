"""bench.py's parts: one module per workload, the CPU baselines, the launcher and the one compact JSON line."""
