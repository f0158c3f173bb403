"""bench.py with default execution options (``ops.DEFAULTS``: they seed every network built afterwards) overridden -- development aid:
EXP=wgrad_side_stream=False,fuse_3x3_backward=False python tools/exp_bench.py ..."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import _hip
if os.environ.get("IEAGAN_LIB"):          # A/B of two builds on the same box
    _hip.LIB_PATH = os.environ["IEAGAN_LIB"]
import ops
for kv in os.environ.get("EXP", "").split(","):
    if kv:
        k, v = kv.split("=")
        ops.DEFAULTS.update(**{k: eval(v)})
import bench
if os.environ.get("CFG"):                  # CFG=sn_prefetch=False,... : overrides of the benchmark's train configuration
    _base = bench.bench_config

    def _patched(*a, **k):
        cfg = _base(*a, **k)
        for kv in os.environ["CFG"].split(","):
            key, val = kv.split("=")
            cfg[key] = eval(val)
        return cfg
    bench.bench_config = _patched
bench.main()
