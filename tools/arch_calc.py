#!/usr/bin/env python3
"""Algorithmic work of one IEA-GAN train step, from the architecture tables alone.

Conv layers of G (reference model.py:86-95 / GBlock 16-71) and D (model.py:573-582 / DBlock 490-557, Attention
layers.py:262-300) at the shipped configuration (ch = 32, depth 2, 256x768, 40 sensors), with
    FLOPs = 2 * N * Cout * H * W * Cin * k^2            bytes = 2 * N * (Hs * Ws * Cin + H * W * Cout)   (bf16 in + out)
per layer (Hs, Ws: resolution the layer READS -- a fused up-sample reads the small map, a fused pool the large one).
Step composition (train_fns.py:23-205): 2 G forwards (one under no_grad) + 3 D forwards + 2 full D backwards (dgrad +
wgrad) + 1 dgrad-only D backward + 1 full G backward.  `bench.py` imports this module: the roofline fractions it
reports divide THESE bytes / FLOPs by measured kernel time; the launchers' own accounting (which adds the epilogue
operands a launch really reads -- ReLU masks, shortcut tensors) is reported next to it.

    python tools/arch_calc.py            # per-layer table + totals
"""
from __future__ import annotations

import json
import sys

G_IN, G_OUT = [16, 16, 8, 8, 4, 2], [16, 8, 8, 4, 2, 1]            # x ch, resolution 256 (model.py:86-95)
D_IN, D_OUT = [1, 2, 4, 8, 8, 16], [2, 4, 8, 8, 16, 16]            # model.py:573-582


def conv(name, n, cin, cout, h, w, k, hs=None, ws=None, net="G"):
    hs, ws = (h, w) if hs is None else (hs, ws)
    return dict(name=name, net=net, taps=k * k, cin=cin, cout=cout, h=h, w=w, flops=2.0 * n * cout * h * w * cin * k * k,
                bytes=2.0 * n * (hs * ws * cin + h * w * cout),
                bytes_out_res=2.0 * n * h * w * (cin + cout))      # SURVEY 8d's simplification: both operands at H x W


def generator_layers(ch=32, n=40, bw=4, h_base=3, depth=2):
    L, h, w = [], bw, bw * h_base
    for s, (ci, co) in enumerate(zip(G_IN, G_OUT)):
        for g in range(depth):
            cin, cout = ch * ci, ch * (co if g == depth - 1 else ci)
            up = g == depth - 1
            hid = cin // 4
            p = f"G.blocks.{s * depth + g}"
            L.append(conv(p + ".conv1", n, cin, hid, h, w, 1))
            h2, w2 = (2 * h, 2 * w) if up else (h, w)
            L.append(conv(p + ".conv2", n, hid, hid, h2, w2, 3, h, w))          # nearest x2 folded into the gather
            L.append(conv(p + ".conv3", n, hid, hid, h2, w2, 3))
            L.append(conv(p + ".conv4", n, hid, cout, h2, w2, 1))
            h, w = h2, w2
    L.append(conv("G.output_layer.2", n, ch * G_OUT[-1], 1, h, w, 3))
    return L


def discriminator_layers(ch=32, n=40, res=256, h_base=3, depth=2, attn_stage=2):
    L, h, w = [], res, res * h_base
    L.append(conv("D.input_conv", n, 1, ch * D_IN[0], h, w, 3, net="D"))
    for s, (ci, co) in enumerate(zip(D_IN, D_OUT)):
        for d in range(depth):
            cin, cout = ch * (ci if d == 0 else co), ch * co
            down = d == 0
            hid = cout // 4
            p = f"D.blocks.{s}.{d}"
            L.append(conv(p + ".conv1", n, cin, hid, h, w, 1, net="D"))
            L.append(conv(p + ".conv2", n, hid, hid, h, w, 3, net="D"))
            L.append(conv(p + ".conv3", n, hid, hid, h, w, 3, net="D"))
            h2, w2 = (h // 2, w // 2) if down else (h, w)
            L.append(conv(p + ".conv4", n, hid, cout, h2, w2, 1, h, w, net="D"))  # 2x2 average pool folded into the gather
            if cin != cout:
                L.append(conv(p + ".conv_sc", n, cin, cout - cin, h2, w2, 1, h, w, net="D"))
            h, w = h2, w2
        if s == attn_stage:                                                       # D_attn = "32": after stage 2 (32x96)
            c = ch * co
            for nm, o, i in (("theta", c // 8, c), ("phi", c // 8, c), ("g", c // 2, c), ("o", c, c // 2)):
                L.append(conv(f"D.blocks.{s}.{depth}.{nm}", n, i, o, h, w, 1, net="D"))
    return L


def attention_bmm_flops(ch=32, n=40, h=32, w=96):
    c = ch * D_OUT[2]
    lq, lk = h * w, h * w // 4
    return 2.0 * n * lq * lk * (c // 8) + 2.0 * n * lq * lk * (c // 2)


def summary(**kw):
    G, D = generator_layers(**{k: v for k, v in kw.items() if k in ("ch", "n")}), discriminator_layers(**{k: v for k, v in kw.items() if k in ("ch", "n")})
    gf, df = sum(l["flops"] for l in G), sum(l["flops"] for l in D)
    gb, db = sum(l["bytes"] for l in G), sum(l["bytes"] for l in D)
    # passes: forward = 1x, dgrad = 1x, wgrad = 1x of a layer's forward FLOPs; each reads + writes one in/out pair
    fwd = 2 * gf + 3 * df
    full_bwd = 2 * (2 * df) + 2 * gf
    dgrad_only = df
    bn_elems = sum(l["bytes"] / 2.0 * l["cin"] / (l["cin"] + l["cout"]) for l in G if l["name"] != "G.output_layer.2") \
        + G[-1]["bytes"] / 2.0 * G[-1]["cin"] / (G[-1]["cin"] + G[-1]["cout"])
    return dict(G_conv_gflop=gf / 1e9, D_conv_gflop=df / 1e9, D_attention_bmm_gflop=attention_bmm_flops() / 1e9,
                G_conv_gbytes=gb / 1e9, D_conv_gbytes=db / 1e9,
                G_conv_gbytes_out_res=sum(l["bytes_out_res"] for l in G) / 1e9, D_conv_gbytes_out_res=sum(l["bytes_out_res"] for l in D) / 1e9,
                step_conv_gflop=(fwd + full_bwd + dgrad_only) / 1e9,
                # conv in+out traffic of the step: every forward, dgrad and wgrad launch moves one in/out pair of its layer
                step_conv_gbytes=(2 * gb + 3 * db + 2 * (2 * db) + db + 2 * gb) / 1e9,
                G_bn_normalised_elements=bn_elems, layers=len(G) + len(D))


def per_pass_bytes(layers, kind):
    """Sum of 8(d) bytes over `layers` restricted to 1x1 / 3x3 (kind = 1 / 9), in GB."""
    return sum(l["bytes"] for l in layers if l["taps"] == kind) / 1e9


def step_family_gbytes():
    """8(d) bytes per STEP of the kernel families bench.py times: forward + dgrad launches of the 1x1 / 3x3 layers
    (single-channel convs D.input_conv / G.output_layer.2 excluded: they run in conv_c1.hip), and the wgrad launches."""
    G = [l for l in generator_layers() if l["name"] != "G.output_layer.2"]
    D = [l for l in discriminator_layers() if l["name"] != "D.input_conv"]
    out = {}
    for kind, tag in ((1, "conv1x1"), (9, "conv3x3")):
        g, d = per_pass_bytes(G, kind), per_pass_bytes(D, kind)
        out[tag + "_fwd_dgrad"] = 2 * g + 3 * d + (g + 2 * d + d)          # forwards + dgrads (G full bwd, 2 D full, 1 D dgrad-only)
        out[tag + "_wgrad"] = g + 2 * d
        # the same sum with every DGRAD launch taken as what it is: a convolution whose two operands both live at the layer's own
        # resolution (the gradient w.r.t. an up-sampled / pooled source is folded back to the source resolution by the kernel
        # behind it) -- this is the figure the launchers' per-launch 8(d) bytes add up to, to the byte
        go = sum(l["bytes_out_res"] for l in G if l["taps"] == kind) / 1e9
        do = sum(l["bytes_out_res"] for l in D if l["taps"] == kind) / 1e9
        out[tag + "_fwd_dgrad_as_launched"] = 2 * g + 3 * d + (go + 2 * do + do)
    # The first DBlock's conv1 and conv_sc run inside d_stem_fwd (all three D forwards) and d_stem_bwd (conv1's dgrad + wgrad in the two
    # full D backwards; conv_sc's backward is a conv1x1_bwd launch and the dgrad-only pass uses the generic launches): their share of the
    # 1x1 totals above never shows up in the conv1x1_* families bench.py times
    stem = [l for l in D if l["name"] in ("D.blocks.0.0.conv1", "D.blocks.0.0.conv_sc")]
    c1 = [l for l in stem if l["name"].endswith("conv1")]
    out["d_stem_1x1"] = (3 * sum(l["bytes"] for l in stem) + 2 * sum(l["bytes_out_res"] + l["bytes"] for l in c1)) / 1e9
    return out


def main():
    G, D = generator_layers(), discriminator_layers()
    print(f"{'layer':34s} {'k':>2s} {'Cin':>4s} {'Cout':>4s} {'HxW':>9s} {'GFLOP':>8s} {'MB':>8s} {'F/B':>6s}")
    for l in G + D:
        print(f"{l['name']:34s} {int(l['taps'] ** 0.5):2d} {l['cin']:4d} {l['cout']:4d} {l['h']:4d}x{l['w']:<4d} "
              f"{l['flops'] / 1e9:8.2f} {l['bytes'] / 1e6:8.1f} {l['flops'] / l['bytes']:6.1f}")
    s = summary()
    print(json.dumps(s, indent=1))
    print(json.dumps(step_family_gbytes(), indent=1))
    # the survey's figures: conv GFLOP G 401.52 / D 434.74, conv in+out GB G 6.03 / D 6.81, step conv 5083.6 GFLOP
    assert abs(s["G_conv_gflop"] - 401.52) < 0.05 and abs(s["D_conv_gflop"] - 434.74) < 0.05, s
    # (the survey counts both operands at the OUTPUT resolution; with the fused up-sample / pool the kernels read the source at
    #  its own resolution, which is what `bytes` and every roofline figure of bench.py use: G 5.68 GB, D 7.83 GB per forward)
    assert abs(s["G_conv_gbytes_out_res"] - 6.03) < 0.02 and abs(s["D_conv_gbytes_out_res"] - 6.81) < 0.02, s
    assert abs(s["step_conv_gflop"] - 5083.6) < 1.0, s
    return 0


if __name__ == "__main__":
    sys.exit(main())
