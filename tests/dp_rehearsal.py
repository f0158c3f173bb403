"""Run in a subprocess by test_networks_gpu.py: ONE full train(x, y) through the data-parallel code path (1-rank RCCL group, side-stream
all-reduce + update, D(x_real) evaluated before G(z) so G's exchange hides behind it) against the CPU oracle, which follows the
reference's order.  Prints the parity report as one JSON line.  Test infrastructure (imports oracle/ through parity_util)."""
import json
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29537")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from parity_util import step_parity  # noqa: E402
import parallel  # noqa: E402


def main():
    real_first = sys.argv[1] != "0" if len(sys.argv) > 1 else True
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    parallel.set_context(parallel.GradSync(overlap=True, force=True))
    # (sn_warm: the oracle keeps the reference's fake-then-real order; real-first hands each pass the OTHER spectral-norm iterate,
    #  which is a tolerance-level difference only once the power iteration has converged -- not on a random initial u0)
    rep = step_parity(resolution=64, H_base=1, state_check=True, dp_real_first=real_first, sn_warm=12 if real_first else 0)
    parallel.quiesce()
    out = {k: v for k, v in rep.items() if k not in ("losses", "ref_losses")}
    out["losses"] = {k: float(v) for k, v in rep["losses"].items()}
    out["ref_losses"] = {k: float(v) for k, v in rep["ref_losses"].items()}
    print(json.dumps(out))
    parallel.shutdown()


if __name__ == "__main__":
    main()
