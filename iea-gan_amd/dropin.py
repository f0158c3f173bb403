#!/usr/bin/env python3
"""Run one of the reference's own scripts on the MI355X path:

    python /path/to/repo/iea-gan_amd/dropin.py /path/to/IEA-GAN/train.py --dataroot ... --outputroot ... --device cuda

Python always puts the directory of the script it runs at ``sys.path[0]``, so with a plain ``python train.py`` inside
a reference checkout the reference's ``model.py`` / ``layers.py`` / ... win no matter what ``PYTHONPATH`` says.  This
launcher builds the search path explicitly -- ``[this package, the script's directory, ...]`` -- and then executes the
script as ``__main__``: ``import model, layers, train_fns, utils, loss, RRM, diff_aug, cr_diff_aug`` (reference
train.py:12-19) resolve to the MI355X implementations, while ``utils.configuration`` / ``utils.logging`` /
``utils.dataloader`` / ``utils.plot`` (host-side bookkeeping, not re-implemented) and ``config.json`` resolve to the
checkout's own files (``utils/__init__.py`` of this package appends the checkout's ``utils/`` directory to its
``__path__``).  The script's working directory is left alone (the reference opens ``config.json`` relative to it).
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"dropin: no such script: {script}")
    pkg = os.path.dirname(os.path.abspath(__file__))
    ref_root = os.path.dirname(script)
    sys.path[:] = [pkg, ref_root] + [p for p in sys.path[1:] if os.path.abspath(p or ".") not in (pkg, ref_root)]
    for name in ("model", "layers", "train_fns", "utils", "loss", "RRM", "diff_aug", "cr_diff_aug"):
        sys.modules.pop(name, None)          # nothing imported before the path was set may leak in
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
