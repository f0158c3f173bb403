"""Golden fixtures for the steps either side of the train step: event ingestion and the Frechet distance.

Development container only: imports the reference's own `utils/norm.py`, `utils/noise.py` and `mycleanfid/fid.py`
(read-only under /root/reference; packages the image lacks -- cv2, torchvision, cleanfid -- are replaced by empty
modules: none of the functions used here touches them), checks the oracle against them and writes data files.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_io.py

torchvision is absent, so three steps of the transform chain are restated from its documented behaviour, not imported:
`Pad((0,3,0,3))` = three zero rows above and below, `ToTensor()` = uint8 / 255, `Normalize((0.5,), (0.5,))` =
(x - 0.5) / 0.5.  `fn_lognorm255`, `UniformNoise` and `frechet_distance` are the reference's functions.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference"
for _n in ["cv2", "torchvision", "torchvision.transforms", "cleanfid", "cleanfid.downloads_helper", "cleanfid.inception_pytorch",
           "mycleanfid"]:
    _m = types.ModuleType(_n)
    _m.__path__ = []
    sys.modules[_n] = _m
sys.modules["cv2"].__getattr__ = lambda name: 0          # fid.py builds a table of cv2.INTER_* constants at import time
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["cleanfid.inception_pytorch"].InceptionV3 = object


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


import ieagan_oracle as O          # noqa: E402

R_norm = load("ref_norm", os.path.join(REF, "utils", "norm.py"))
R_noise = load("ref_noise", os.path.join(REF, "utils", "noise.py"))
R_fid = load("ref_fid", os.path.join(REF, "mycleanfid", "fid.py"))


def main():
    rng = np.random.default_rng(11)
    # ---- ingestion: 5 sensors of 10 x 16 pixels, sparse hits like PXD data, incl. 0 and 255
    ev = np.zeros((5, 10, 16), dtype=np.uint8)
    hit = rng.random(ev.shape) < 0.15
    ev[hit] = rng.integers(1, 256, size=int(hit.sum()), dtype=np.uint8)
    ev[0, 0, 0], ev[0, 0, 1] = 255, 0
    evt = torch.from_numpy(ev)
    exp = []
    draws = []
    for n in range(ev.shape[0]):
        t = torch.nn.functional.pad(evt[n].float() / 255.0, (0, 0, 3, 3)).unsqueeze(0)       # Pad + ToTensor
        t = R_norm.fn_lognorm255(t)
        torch.manual_seed(100 + n)
        u = torch.rand_like(t)                                                              # the draw UniformNoise makes
        torch.manual_seed(100 + n)
        t = R_noise.UniformNoise(scale=4e-3)(t)
        draws.append(u)
        exp.append((t - 0.5) / 0.5)                                                         # Normalize
    exp, u = torch.stack(exp), torch.stack(draws)
    got = O.ingest_event(evt, u)
    assert got.shape == exp.shape == (5, 1, 16, 16)
    assert (got - exp).abs().max().item() <= 1e-6
    np.savez_compressed(os.path.join(HERE, "op_ingest.npz"), ev=ev, u=u.numpy(), out=exp.numpy())
    # ---- Frechet distance: two feature clouds (d = 24) + a rank-deficient pair that needs the eps retry
    a = rng.standard_normal((200, 24)) * rng.uniform(0.5, 2.0, 24)
    b = rng.standard_normal((180, 24)) * rng.uniform(0.5, 2.0, 24) + 0.3
    c = np.zeros((6, 24)); c[:, :3] = rng.standard_normal((6, 3))                           # singular covariance
    cases = {}
    for name, (p, q) in {"ab": (a, b), "aa": (a, a), "ac": (a, c)}.items():
        m1, s1, m2, s2 = p.mean(0), np.cov(p, rowvar=False), q.mean(0), np.cov(q, rowvar=False)
        ref = float(R_fid.frechet_distance(m1, s1, m2, s2))
        mine = O.frechet_distance(m1, s1, m2, s2)
        assert abs(ref - mine) <= 1e-9 * max(1.0, abs(ref)), (name, ref, mine)
        cases[name] = ref
    np.savez_compressed(os.path.join(HERE, "op_frechet.npz"), a=a, b=b, c=c, **{"fd_" + k: np.float64(v) for k, v in cases.items()})
    print("wrote op_ingest.npz, op_frechet.npz", cases)
    # ---- export step after the path: model.generate (model.py:1130-1148) and utils.norm.denorm (utils/norm.py:34-46)
    # driven with a stand-in "generator" that returns a fixed image, so that only the export arithmetic is pinned:
    # values around the -0.26 threshold (both sides and exactly on it), the clamp at 255 and the [-1, 1] end points
    sys.path.insert(0, REF)
    import pandas  # noqa: F401  (layers.py imports it for an unused helper)
    import model as R_model
    img = torch.from_numpy(rng.uniform(-1.0, 1.0, (40, 1, 16, 24)).astype(np.float32))
    img[0, 0, 5, :6] = torch.tensor([-0.26, -0.2600001, -0.2599999, 1.0, -1.0, 0.999999])

    class Fixed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, z, y):
            assert z.shape == (40, 128) and y.tolist() == list(range(40))
            return img.clone()

    adu = R_model.generate(Fixed())
    den = R_norm.denorm(img.clone())
    assert adu.shape == (40, 10, 24) and den.shape == (40, 1, 10, 24)
    assert torch.equal(O.generate_export(img), adu), "oracle generate_export != reference model.generate"
    assert torch.equal(O.denorm(img), den), "oracle denorm != reference utils.norm.denorm"
    np.savez_compressed(os.path.join(HERE, "op_export.npz"), img=img.numpy(), adu=adu.numpy(), denorm=den.numpy())
    print("wrote op_export.npz")


if __name__ == "__main__":
    main()
