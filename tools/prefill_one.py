"""Time (and, under rocprofv3, profile) one prompt of P ids through the host model: python tools/prefill_one.py P [reps]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); hip = pkg.hipabi.load(); hip.init(0)
host = pkg.load_host()
cfg = host.default_config(4, 3)
m = host.model(cfg); m.load_synthetic(1234)
P = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
toks = host.synthetic_tokens(2048, seed=1)
m.set_fast_decode(False)
m.logits(toks[:P], 0, want=False); hip.sync()
t0 = time.perf_counter()
for _ in range(reps): m.logits(toks[:P], 0, want=False)
hip.sync()
print(f"P={P}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms per prompt", flush=True)
