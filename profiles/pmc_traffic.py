"""HBM traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.

usage: python profiles/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> <out.json> [git head]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters tick in KiB and gfx950's FETCH_SIZE counts half of wide
coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Families are the names bench.py's per-kernel HIP-event
timing uses, so `roofline.traffic` can be looked up by name.
"""
import csv, json, re, sys
from collections import defaultdict


def family(name):
    n = re.sub(r"^void ", "", name)
    if n.startswith("conv_gather_kernel<1") or n.startswith("conv1x1_stream_kernel") or n.startswith("conv1x1_tile_kernel"): return "conv1x1_gather"      # (bench.py family names)
    if n.startswith("conv3x3_lds_kernel") or n.startswith("conv3x3_lds_fp8_kernel") or n.startswith("conv3x3_ws_kernel"): return "conv3x3_halo"
    if n.startswith("conv1x1_bwd_kernel"): return "conv1x1_bwd"
    if n.startswith("conv3x3_bwd_kernel"): return "conv3x3_bwd"
    if n.startswith("d_stem_fwd_kernel"): return "d_stem_fwd"
    if n.startswith("d_stem_bwd_kernel"): return "d_stem_bwd"
    if n.startswith("wgrad_reduce_kernel"): return "conv3x3_wgrad"
    if n.startswith("conv_gather_kernel<9"): return "conv3x3_gather"
    if n.startswith("conv3x3_halo_kernel"): return "conv3x3_halo"
    if n.startswith("conv_wgrad_kernel<9"): return "conv3x3_wgrad"
    if n.startswith("conv_wgrad_kernel<1"): return "conv1x1_wgrad"
    for key, fam in (("effgrad", "effgrad"), ("prologue_bwd", "prologue_bwd"), ("conv_Cto1", "conv_Cto1"), ("conv_1toC", "conv_1toC"),
                     ("wgrad_c1", "wgrad_c1"), ("nl_attn_fwd", "nl_attention_fwd"), ("nl_attn_bwd", "nl_attention_bwd"),
                     ("sn_bwd", "sn_backward"), ("sn_phase", "sn_forward"), ("ortho_", "ortho_grad"), ("adam", "adam_step"),
                     ("ema_kernel", "ema_update"), ("bn_finalize_fwd", "bn_finalize_fwd"), ("bn_finalize_bwd", "bn_finalize_bwd"),
                     ("res_bwd", "res_bwd"), ("diffaug", "diffaug"), ("maxpool2", "maxpool2"), ("gamma_residual", "gamma_residual"),
                     ("rrm_attn", "rrm_attention"), ("loss_block", "loss_block"), ("relu_sum_pool", "relu_sum_pool"),
                     ("nchw_to_nhwc", "nchw_to_nhwc"), ("nhwc_to_nchw", "nhwc_to_nchw"), ("slin_fwd", "slin_fwd"), ("slin_bwd", "slin_bwd"),
                     ("ln_fwd", "ln_fwd"), ("ln_bwd", "ln_bwd"), ("embed_norm", "embed_norm")):
        if key in n:
            return fam
    return "library (ATen/rocBLAS/RCCL)"


def load(path, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        tot[f] += float(r["Counter_Value"])
        disp[f].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


def main():
    fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    fe, nf = load(fetch_csv, "FETCH_SIZE")
    wr, nw = load(write_csv, "WRITE_SIZE")
    fams = {}
    for f in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, 0) + wr.get(k, 0))):
        b = (2.0 * fe.get(f, 0.0) + wr.get(f, 0.0)) * 1024.0
        n = max(nf.get(f, 0), nw.get(f, 0), 1)
        fams[f] = {"launches_per_step": n / steps, "hbm_GB_per_step": b / steps / 1e9, "hbm_bytes_per_launch": b / n}
    total = sum(v["hbm_GB_per_step"] for v in fams.values())
    head = sys.argv[5] if len(sys.argv) > 5 else None
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the same bench.py command; "
                       "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md",
               "git_head": head, "steps": steps, "hbm_GB_per_step_total": total, "families": fams}, open(out, "w"), indent=1)
    print(f"{total:.1f} GB/step over {len(fams)} families -> {out}")


if __name__ == "__main__":
    main()
