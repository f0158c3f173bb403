"""Autograd operators of the MI355X path: thin ``torch.autograd.Function`` shells around the C ABI.

Internal activation format: bf16, NHWC, contiguous.  Statistics buffers: fp32 ``[STAT_REPL, 2, C]``
(replicated per-channel sum / sum of squares).  PyTorch only provides device memory, streams and
the autograd tape here; all arithmetic on activations happens in libieagan_hip.so.

Fused operator ``conv`` = [BN apply (per-(n,c) scale/shift) + ReLU + nearest-x2 upsample | 2x2 avg
pool] -> 1x1/3x3 convolution with the spectrally normalised weight -> [+bias, + residual (own
resample / channel slice / concat), + batch statistics of the output].  Its backward folds the
batch-statistics path of the *following* BatchNorm into the out-grad (g_eff = dout + dsum + 2 out
dsumsq), so one BN layer costs one extra read in forward and no standalone pass over the tensor.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

import _hip as H
from _hip import STAT_REPL

BF16 = torch.bfloat16


def _kpad(k: int) -> int:
    return (k + 31) // 32 * 32


class _ZeroPool:
    """Zero-initialised fp32 scratch carved from a few large pre-filled chunks: the step needs ~900 small
    zeroed buffers (statistics replicas, column sums, split-K accumulators); one fill per 32 MiB chunk
    replaces one fill kernel per buffer.  A chunk is used once; it is released when its views die."""
    CHUNK = 8 << 20            # floats

    def __init__(self):
        self.buf, self.pos = {}, {}

    def take(self, n: int, device) -> torch.Tensor:
        # one chunk per (device, stream): a chunk is filled on the stream that is current when it is created, and only that
        # stream may carve it without further synchronisation
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        n_al = (n + 63) // 64 * 64
        if n_al > self.CHUNK // 4:
            return torch.zeros(n, dtype=torch.float32, device=device)
        if key not in self.buf or self.pos[key] + n_al > self.buf[key].numel():
            self.buf[key] = torch.zeros(self.CHUNK, dtype=torch.float32, device=device)
            self.pos[key] = 0
        o = self.pos[key]
        self.pos[key] = o + n_al
        return self.buf[key][o:o + n]

    def reset(self):
        """Drop the current chunks.  MUST bracket a HIP-graph capture: a chunk filled before the capture would be
        carved inside it without its fill being part of the graph (stale accumulators on replay), and a chunk
        allocated inside the capture belongs to the graph's private memory pool."""
        self.buf, self.pos = {}, {}


_ZEROS = _ZeroPool()


def reset_zero_pool():
    _ZEROS.reset()


def zeros(shape, device) -> torch.Tensor:
    """fp32 zeros of ``shape`` from the pool."""
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    return _ZEROS.take(n, device).view(shape)


def new_stats(C: int, device, events: int = 1, slots: int = STAT_REPL) -> torch.Tensor:
    """Zeroed per-event (sum, sumsq) accumulators: [E, slots, 2, C].  ``slots`` = the producing launch's blocks per event makes every slot
    a single-adder address (bit-reproducible sums; ``_conv_launch(stats=True)`` sizes it that way); the default is the legacy replica
    count (several adders per slot: the float-atomic order then decides the last bits)."""
    return zeros((events, slots, 2, C), device)


# When True (set by the train step around ``backward()``), gradients of spectrally normalised weights
# and of conv biases are accumulated by the HIP kernels straight into the existing ``param.grad`` (views
# of the flat gradient arena) and autograd is handed ``None`` -- no per-parameter accumulate kernels.
DIRECT_GRADS = False


class direct_grads:
    def __enter__(self):
        global DIRECT_GRADS
        self.prev, DIRECT_GRADS = DIRECT_GRADS, True

    def __exit__(self, *a):
        global DIRECT_GRADS
        DIRECT_GRADS = self.prev


# =====================================================================================================
# Execution options: which of the equivalent launch sequences the passes of ONE network use
# =====================================================================================================
class ExecOptions:
    """Launch-orchestration choices of one network -- properties of the network like ``conv_dtype`` (whose IEAGAN_CONV_FP8 bit rides in the
    descriptor of its layers), not process-wide switches: every ``SNBank`` (one per network) carries its own copy, seeded from ``DEFAULTS``
    when the bank is built; the autograd shells read them through ``opts_of(rec)``.  Tests and benchmarks change the options of the
    network they measure (``set_options(net, ...)``) -- e.g. bench.py serialises the weight gradients of its per-kernel timing pass
    without touching any other network of the process.

    wgrad_side_stream   weight-gradient launches of a training backward pass on a side stream (nothing in the pass reads dW before the
                        batched spectral-norm backward at its end, ``SNPass.flush`` joins)
    two_stage_wgrad     conv_wgrad: partial slabs + reduction launch instead of float atomics when the atomic volume is large
    use_tr_read         ds_read_b64_tr_b16 operand reads in conv_wgrad (False: scalar LDS reads, the test reference)
    fuse_bn_backward    BatchNorm-apply backward inside the dgrad epilogue (False: the stand-alone prologue_bwd pass)
    fuse_shortcut_grad  residual-shortcut gradients added inside the dx-producing kernel (False: autograd adds)
    fuse_1x1_backward / fuse_3x3_backward (+ *_min_pixels)   the whole-backward kernels conv1x1_bwd / conv3x3_bwd (False: separate launches)
    fuse_d_stem         D.input_conv + the first DBlock's conv1 / conv_sc / pooled shortcut in one launch each way
    b1_flags            ieagan_conv1x1_bwd_desc.flags (benchmarks: H.B1_OCC2 / H.B1_OCC3)
    fused_reduce_side_stream   the slab folds of the whole-backward kernels on the weight-gradient side stream (with wgrad_side_stream).
                        OFF: measured 36.75-37.0 vs 34.75-34.97 ms per step on one box -- the side stream then sits behind every fused kernel
                        of the large maps and the weight gradients queued after it start late
    fused_reduce_batched       the slab folds of the whole-backward kernels of a pass in ONE launch at the end of the pass (``SNPass.flush``:
                        nothing reads dW earlier) instead of one launch behind each of them"""
    FIELDS = dict(wgrad_side_stream=True, two_stage_wgrad=True, use_tr_read=True, fuse_bn_backward=True, fuse_shortcut_grad=True,
                  fuse_1x1_backward=True, fuse_1x1_min_pixels=1 << 16, fuse_3x3_backward=True, fuse_3x3_min_pixels=1 << 16,
                  fuse_d_stem=True, b1_flags=0, fused_reduce_side_stream=False, fused_reduce_batched=True)
    __slots__ = tuple(FIELDS)

    def __init__(self, like=None, **kw):
        src = like if like is not None else globals().get("DEFAULTS")
        for k, v in self.FIELDS.items():
            setattr(self, k, getattr(src, k) if src is not None else v)
        self.update(**kw)

    def update(self, **kw):
        for k, v in kw.items():
            if k not in self.FIELDS:
                raise KeyError(f"unknown execution option {k!r} (have: {sorted(self.FIELDS)})")
            setattr(self, k, v)
        return self

    def as_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS}


DEFAULTS = ExecOptions()          # seeds the options of every bank built from now on (tests: change, build, restore)


def opts_of(rec) -> "ExecOptions":
    """Options of the network the layer record ``rec`` belongs to (records without a bank -- identity convs -- take the defaults)."""
    ps = getattr(rec, "pass_", None)
    return ps.bank.opts if ps is not None else DEFAULTS


def opts_of_net(net) -> "ExecOptions":
    return (net if isinstance(net, SNBank) else net._prepare()["bank"]).opts


def set_options(net, **kw) -> "ExecOptions":
    """Change execution options of ONE network (a Generator / Discriminator, or anything with ``_prepare()``) or of a bank."""
    bank = net if isinstance(net, SNBank) else net._prepare()["bank"]
    return bank.opts.update(**kw)


# =====================================================================================================
# Spectral-norm bank: every SN layer of a network in three launches per forward
# =====================================================================================================
KIND_PLAIN, KIND_CONV, KIND_C1_IN, KIND_C1_OUT = 0, 1, 2, 3


class SNRecord:
    """What one forward pass knows about one spectrally normalised layer."""
    __slots__ = ("kind", "out", "inn", "taps", "cin", "kpad", "kpad2", "w_fwd", "w_bwd", "w_plain", "ctx", "name", "pass_",
                 "deferred")


class SNPass:
    """One forward pass of a network through its SN bank.  In training (gradients accumulated straight into the flat
    gradient arena) the conv layers' weight-gradient kernels of the matching backward pass write into ONE zeroed scratch
    arena owned by this object, and a single batched launch at the end of ``backward()`` maps all of them through
    d(W/sigma)/dW -- instead of one small launch chain per layer."""

    def __init__(self, bank, ctx):
        self.bank, self.ctx, self.arena, self._ok = bank, ctx, None, None
        self.side = None             # side stream carrying this pass's weight-gradient launches (joined in flush)
        self.reduces = []            # (slab workspace, dW view, S, Cout, Kpad, K) of the whole-backward kernels: folded in ONE launch by flush
        self.stream = torch.cuda.current_stream(ctx.device)      # the stream this pass runs on (autograd runs its backward there too)

    def usable(self) -> bool:
        if not DIRECT_GRADS or self.bank.owner is None or self.bank.bwd is None:
            return False
        if self._ok is None:
            self._ok = self.bank.owner.grads_attached()
        return self._ok

    def scratch(self, name, which, shape):
        """View of this pass's scratch arena for layer ``name`` ('w': weight gradient, 'b': bias column sums) or None."""
        if not self.usable() or name not in self.bank.bwd["rows"]:
            return None
        goff, coff = self.bank.bwd["rows"][name]
        off = goff if which == "w" else coff
        if off < 0:
            return None
        if self.arena is None:
            self.arena = zeros((self.bank.bwd["total"],), self.ctx.device)
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(self.flush)
        n = 1
        for d in shape:
            n *= int(d)
        return self.arena[off:off + n].view(shape)

    def flush(self):
        if self.arena is None:
            return
        cur = torch.cuda.current_stream()
        if self.stream != cur:
            cur.wait_stream(self.stream)
        if self.side is not None:
            cur.wait_stream(self.side)
            self.side = None
        if self.reduces:                # the slab sets the whole-backward kernels of this pass left behind (fused_reduce_batched)
            items = (H.ReduceItem * len(self.reduces))(*[H.ReduceItem(ws.data_ptr(), dwp.data_ptr(), S, Cout, kpad, K)
                                                        for ws, dwp, S, Cout, kpad, K in self.reduces])
            H.call("ieagan_wgrad_reduce_batched", items, len(self.reduces), H.stream())
            if cur != self.stream:
                for ws, *_ in self.reduces:
                    ws.record_stream(cur)
            self.reduces = []
        b = self.bank
        H.call("ieagan_sn_backward_batched", b.bwd["table"].data_ptr(), b.bwd["work"].data_ptr(), b.bwd["nwork"], b.arena.data_ptr(),
               self.ctx.data_ptr(), self.arena.data_ptr(), b.owner.grad.data_ptr(), H.stream())
        if cur != self.stream:                   # launched from another stream than the one whose pool owns these buffers
            self.arena.record_stream(cur)
            self.ctx.record_stream(cur)
        self.arena = None


def sn_scratch(rec, which, shape, device):
    """Zeroed fp32 accumulator for a layer's weight gradient / bias column sums: a slice of the pass's batched-backward
    arena when that path is active (then ``sn_backward`` is a no-op for the layer), else a private pool buffer."""
    ps = getattr(rec, "pass_", None)
    if ps is not None:
        v = ps.scratch(rec.name, which, shape)
        if v is not None:
            rec.deferred = True
            return v
    return zeros(shape, device)


class SNBank:
    """Layer table over a flat fp32 arena (parameters + buffers of one network).

    ``entries``: list of (name, kind, weight, u0, sv0) where the tensors are views into ``arena``.
    ``stack`` : names of kind-0 layers whose normalised weights must be laid out contiguously, in
    this order, as one [sum(out), in] matrix (the 96 ccbn gain/bias linears of G)."""

    ROWS = 32

    def __init__(self, arena: torch.Tensor, entries, stack=(), owner=None, biases=None):
        self.arena = arena
        self.opts = ExecOptions()               # this network's execution options (a copy of DEFAULTS as of now)
        self._prefetched = []
        self.owner, self.bwd = owner, None          # owner: the network's Arena (flat gradient buffer) -> batched backward
        self.names = [e[0] for e in entries]
        self.index = {n: i for i, n in enumerate(self.names)}
        dev = arena.device
        rows, blocks, cblocks = [], [], []
        ctx_off, pack_off, part_off = 0, 0, 0
        self.meta = []
        base = arena.data_ptr()

        def off(t):
            d = t.data_ptr() - base
            assert d % 4 == 0 and 0 <= d < arena.numel() * 4, "SN tensors must live in the arena"
            return d // 4

        order = list(stack) + [n for n in self.names if n not in set(stack)]
        self.stack_rows = {}
        srow = 0
        place = {}
        for n in order:      # pack offsets follow `order`; table rows follow `entries`
            i = self.index[n]
            _, kind, w, u, sv = entries[i]
            out = w.shape[0]
            inn = w.numel() // out
            taps = 9 if (w.dim() == 4 and w.shape[-1] == 3) else 1
            cin = w.shape[1] if w.dim() == 4 else inn
            kpad = kpad2 = 0
            p2 = 0
            if kind == KIND_CONV:
                kpad, kpad2 = _kpad(taps * cin), _kpad(taps * out)
                p1 = pack_off
                p2 = p1 + out * kpad * 2
                nbytes = out * kpad * 2 + cin * kpad2 * 2
            else:
                p1 = pack_off
                nbytes = out * inn * 4
            place[n] = (p1, p2, kpad, kpad2, taps, cin, out, inn)
            pack_off += (nbytes + 255) // 256 * 256
            if n in stack:
                self.stack_rows[n] = (srow, out)
                srow += out
        self.stack_total = srow
        for i, (n, kind, w, u, sv) in enumerate(entries):
            p1, p2, kpad, kpad2, taps, cin, out, inn = place[n]
            row = [0] * H.SN_FIELDS
            row[0:14] = [off(w), off(u), off(sv), out, inn, taps, cin, kind, ctx_off, p1, p2, kpad, kpad2, part_off]
            rows.append(row)
            self.meta.append((kind, out, inn, taps, cin, kpad, kpad2, ctx_off, p1, p2))
            ctx_off += (8 + 2 * out + 2 * inn + 7) // 8 * 8
            nchunk = (out + self.ROWS - 1) // self.ROWS
            part_off += nchunk * inn
            for r0 in range(0, out, self.ROWS):
                blocks.append([i, r0])
            for c0 in range(0, inn, 32):          # SN_CB columns per phase-1b block (csrc/sn.hip)
                cblocks.append([i, c0])
        self.ctx_size, self.pack_size, self.part_size = ctx_off, pack_off, part_off
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.blocks = torch.tensor(blocks, dtype=torch.int32, device=dev)
        self.cblocks = torch.tensor(cblocks, dtype=torch.int32, device=dev)
        self.nblocks, self.ncblocks = len(blocks), len(cblocks)
        if owner is not None:
            self._plan_backward(entries, biases or {})

    def _plan_backward(self, entries, biases):
        """Static layout of the per-pass scratch arena and the work list of ``ieagan_sn_backward_batched``."""
        offs = {id(p): o for p, o, _ in self.owner.param_slices}
        al = lambda n: (n + 63) // 64 * 64
        conv = [(i, e) for i, e in enumerate(entries) if e[1] != KIND_PLAIN]
        if not conv:
            return
        goff = al(len(conv))                         # [0, n_layers): the <gsn, W> accumulators
        rows, tab, work = {}, [], []
        for li, (i, (n, kind, w, u, sv)) in enumerate(conv):
            _, out, inn, taps, cin, kpad, _, coff, _, _ = self.meta[i]
            gsz = out * kpad if kind == KIND_CONV else out * inn
            b = biases.get(n)
            nb = 0 if (b is None or kind == KIND_C1_OUT) else out
            tab.append([offs[id(w)], out, inn, taps, cin, kind, kpad, coff, goff, -1, offs[id(b)] if nb else -1, nb])
            rows[n] = [goff, -1]
            goff += al(gsz)
            for c in range((out * inn + 2047) // 2048):
                work.append([li, c])
        for li, (i, (n, kind, w, u, sv)) in enumerate(conv):
            if tab[li][11]:
                tab[li][9] = rows[n][1] = goff
                goff += al(STAT_REPL * tab[li][11])
        dev = self.arena.device
        self.bwd = dict(rows={k: tuple(v) for k, v in rows.items()}, total=goff, nwork=len(work),
                        table=torch.tensor(tab, dtype=torch.int64, device=dev), work=torch.tensor(work, dtype=torch.int32, device=dev))

    def _launch(self, training: bool, eps: float):
        dev = self.arena.device
        ctx = torch.empty(self.ctx_size, dtype=torch.float32, device=dev)
        part = torch.empty(self.part_size, dtype=torch.float32, device=dev)
        pack = torch.empty(self.pack_size, dtype=torch.uint8, device=dev)

        def go():
            H.call("ieagan_sn_forward", self.table.data_ptr(), self.blocks.data_ptr(), self.nblocks, self.cblocks.data_ptr(),
                   self.ncblocks, self.arena.data_ptr(), ctx.data_ptr(), part.data_ptr(), pack.data_ptr(), float(eps),
                   int(training), H.stream())
        return (ctx, part, pack), go

    def prefetch(self, training: bool, eps: float, side):
        """Issue the power iteration of the NEXT forward pass of this network on the stream ``side`` now.  The spectral norms
        depend on the weights only, never on activations: one pass (~0.2 ms of small launches that leave most CUs idle) can run
        under the convolutions of whatever the main stream is doing.  Passes queue in order (every pass advances ``u``); the
        next ``run`` call hands out the oldest one after making the current stream wait for it.  The result buffers are
        allocated on the current stream, which is also where they are consumed and released."""
        main = torch.cuda.current_stream()
        bufs, go = self._launch(training, eps)
        ready = torch.cuda.Event()
        ready.record(main)                     # weights as of now (e.g. behind the optimizer step)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            go()
            done = torch.cuda.Event()
            done.record(side)
        self._prefetched.append((bool(training), float(eps), bufs, done))

    def discard_prefetched(self):
        """Drop queued passes (a step that ended early); the main stream still joins them."""
        while self._prefetched:
            torch.cuda.current_stream().wait_event(self._prefetched.pop(0)[3])

    def run(self, training: bool, eps: float):
        if self._prefetched:
            tr, e, (ctx, part, pack), done = self._prefetched.pop(0)
            if tr != bool(training) or e != float(eps):
                raise RuntimeError("a prefetched spectral-norm pass was issued for a different mode than the forward that consumes it")
            torch.cuda.current_stream().wait_event(done)
        else:
            (ctx, part, pack), go = self._launch(training, eps)
            go()
        recs = {}
        ps = SNPass(self, ctx)
        for n, (kind, out, inn, taps, cin, kpad, kpad2, coff, p1, p2) in zip(self.names, self.meta):
            r = SNRecord()
            r.kind, r.out, r.inn, r.taps, r.cin, r.kpad, r.kpad2 = kind, out, inn, taps, cin, kpad, kpad2
            r.name, r.pass_, r.deferred = n, ps, False
            r.ctx = ctx[coff:coff + 8 + 2 * out + 2 * inn]
            r.w_fwd = r.w_bwd = r.w_plain = None
            if kind == KIND_CONV:
                r.w_fwd = pack[p1:p1 + out * kpad * 2].view(BF16).view(out, kpad)
                r.w_bwd = pack[p2:p2 + cin * kpad2 * 2].view(BF16).view(cin, kpad2)
            elif kind == KIND_PLAIN:
                r.w_plain = pack[p1:p1 + out * inn * 4].view(torch.float32).view(out, inn)
            else:
                r.w_plain = pack[p1:p1 + out * inn * 4].view(torch.float32)      # [9][C]
            recs[n] = r
        recs["__ctx__"] = ctx
        if self.stack_total:
            first = self.meta[self.index[next(iter(self.stack_rows))]]
            inn = first[2]
            recs["__stack__"] = pack[first[8]:first[8] + self.stack_total * inn * 4].view(torch.float32).view(self.stack_total, inn)
        return recs


def _direct_target(p):
    g = p.grad if (DIRECT_GRADS and p is not None) else None
    return g if (g is not None and g.is_contiguous() and g.dtype == torch.float32) else None


def sn_backward(gsn: torch.Tensor, weight: torch.Tensor, rec: SNRecord, colsum=None, bias=None):
    """Gradient of the parameter ``weight`` from the gradient w.r.t. its normalised form (consumer layout),
    plus (optionally) the bias gradient folded from replicated column sums.  Returns (dW, dbias); an entry is
    None when it was accumulated directly into ``param.grad`` (see DIRECT_GRADS)."""
    if getattr(rec, "deferred", False):          # gsn / colsum live in the pass arena: handled by SNPass.flush
        return None, None
    tgt = _direct_target(weight)
    dW = tgt if tgt is not None else torch.empty_like(weight)
    btgt = _direct_target(bias) if colsum is not None else None
    dbias = None
    if colsum is not None:
        dbias = btgt if btgt is not None else torch.empty_like(bias)
    big = rec.out * rec.inn > 20000          # multi-block path: <gsn, W> accumulates into a zeroed scratch float
    scratch = zeros((1,), weight.device) if big else None
    H.call("ieagan_sn_backward", gsn.data_ptr(), weight.data_ptr(), rec.kind, rec.out, rec.inn, rec.taps, rec.cin,
           rec.kpad, rec.ctx.data_ptr(), H.ptr(scratch), dW.data_ptr(), int(tgt is not None), H.ptr(colsum), H.ptr(dbias),
           int(btgt is not None), H.stream())
    return (None if tgt is not None else dW), (None if (btgt is not None or colsum is None) else dbias)


class SNWeightFn(torch.autograd.Function):
    """W -> W / sigma for a linear / embedding layer whose consumer is a plain library GEMM."""

    @staticmethod
    def forward(ctx, weight, rec):
        ctx.rec = rec
        ctx.save_for_backward(weight)
        return rec.w_plain.view(weight.shape)

    @staticmethod
    def backward(ctx, g):
        (weight,) = ctx.saved_tensors
        return sn_backward(g.contiguous().float(), weight, ctx.rec)[0], None


class StackedSNLinearFn(torch.autograd.Function):
    """All ccbn gain/bias SNLinear layers of G as ONE GEMM: [N, cond] x [cond, sum C]; the backward through
    the 96 spectral norms is one launch (block per layer)."""

    @staticmethod
    def forward(ctx, y, wstack, recs, plan, *weights):
        ctx.recs, ctx.plan = recs, plan
        y = y.contiguous()
        ctx.save_for_backward(y, wstack, *weights)
        return _slin_fwd(y, wstack, None, None)[0]

    @staticmethod
    def backward(ctx, g):
        y, wstack, *weights = ctx.saved_tensors
        plan, recs = ctx.plan, ctx.recs
        bank, ar = plan["bank"], plan["arena"]
        dy, gst, _ = _slin_bwd(g.contiguous(), wstack, xn=y, want_db=False)     # gst: [sum C, cond] gradient w.r.t. the normalised rows
        direct = DIRECT_GRADS and ar.grads_attached()
        if direct:
            base, dst, out = ar.grad, plan["stack_dst_arena"], [None] * len(weights)
        else:
            base, dst = torch.empty_like(gst), plan["stack_dst_flat"]
            out, r0 = [], 0
            for w in weights:
                out.append(base[r0:r0 + w.shape[0]])
                r0 += w.shape[0]
        H.call("ieagan_sn_backward_stack", bank.table.data_ptr(), plan["stack_layers"].data_ptr(), plan["stack_row0"].data_ptr(),
               dst.data_ptr(), len(weights), gst.data_ptr(), bank.arena.data_ptr(), recs["__ctx__"].data_ptr(), base.data_ptr(),
               int(direct), H.stream())
        return (dy, None, None, None, *out)


# =====================================================================================================
# BatchNorm finalize: statistics -> per-(n,c) scale / shift
# =====================================================================================================
def _fin_scratch(which, *args, device):
    """Zeroed hand-off space of a bn_finalize launch that splits its fold over several workgroups (None: not needed)."""
    n = getattr(H.lib(), "ieagan_bn_finalize_" + which + "_scratch")(*args)
    return zeros((n,), device) if n > 0 else None


class GainBank:
    """The [N, sum C] matrix of all ccbn gains/biases of one generator forward, plus the shared
    gradient buffer its consumers write into (each BN owns disjoint columns; the last consumer to
    run its backward hands the complete buffer to autograd)."""

    def __init__(self, gb: torch.Tensor, consumers: int):
        self.gb, self.pending, self.grad = gb, consumers, None

    def grad_buffer(self):
        if self.grad is None:
            self.grad = zeros(self.gb.shape, self.gb.device)
        return self.grad


class BNFinalizeFn(torch.autograd.Function):
    """ccbn: scale = rstd*(1+gain[n,c]), shift = bias[n,c] - mean*scale  (layers.py:656-689)."""

    @staticmethod
    def forward(ctx, stats, gb, bank, col_gain, col_bias, C, run_mean, run_var, count, eps, momentum, training, events=1, link=None):
        """``count``: elements per channel of ONE event; ``stats`` [E, STAT_REPL, 2, C]."""
        ctx.link = link
        N, ld = gb.shape
        dev = gb.device
        scale = torch.empty(N, C, dtype=torch.float32, device=dev)
        shift = torch.empty(N, C, dtype=torch.float32, device=dev)
        mr = torch.empty(events, 2, C, dtype=torch.float32, device=dev)
        repl = int(stats.shape[1]) if stats is not None else 0
        H.call("ieagan_bn_finalize_fwd", H.ptr(stats), float(count), gb.data_ptr() + 4 * col_gain,
               gb.data_ptr() + 4 * col_bias, ld, 1, float(eps), float(momentum), int(training), run_mean.data_ptr(),
               run_var.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), N, C, events, repl,
               H.ptr(_fin_scratch("fwd", C, events, repl, device=dev) if training else None), H.stream())
        ctx.bank, ctx.cols, ctx.C, ctx.count, ctx.training, ctx.events, ctx.repl = bank, (col_gain, col_bias), C, count, training, events, repl
        ctx.has_stats = stats is not None
        ctx.save_for_backward(gb, mr)
        return scale, shift

    @staticmethod
    def backward(ctx, dscale, dshift):
        gb, mr = ctx.saved_tensors
        N, ld = gb.shape
        C = ctx.C
        bank = ctx.bank
        gbuf = bank.grad_buffer()
        E = ctx.events
        dstat = torch.empty(E, 2, C, dtype=torch.float32, device=gb.device)
        acc = ctx.link.acc if ctx.link is not None else None
        repl = 0
        if acc is not None:        # the consumer conv's dgrad folded the apply backward in: replicated per-image accumulators
            ctx.link.acc = None
            dscale, dshift, repl = acc, acc, int(acc.shape[1])
        else:
            dscale = dscale.contiguous() if dscale is not None else torch.zeros(N, C, device=gb.device)
            dshift = dshift.contiguous() if dshift is not None else torch.zeros(N, C, device=gb.device)
        H.call("ieagan_bn_finalize_bwd", dscale.data_ptr(), dshift.data_ptr(), gb.data_ptr() + 4 * ctx.cols[0], ld, 1,
               mr.data_ptr(), float(ctx.count), int(ctx.training), gbuf.data_ptr() + 4 * ctx.cols[0],
               gbuf.data_ptr() + 4 * ctx.cols[1], ld, dstat.data_ptr(), N, C, E, repl,
               H.ptr(_fin_scratch("bwd", N, C, E, repl, device=gb.device)), H.stream())
        bank.pending -= 1
        dgb = gbuf if bank.pending == 0 else None
        dstats = dstat.unsqueeze(1).expand(E, ctx.repl, 2, C) if (ctx.has_stats and ctx.training) else None
        return dstats, dgb, None, None, None, None, None, None, None, None, None, None, None, None


class BNFinalizePlainFn(torch.autograd.Function):
    """layers.bn: per-channel gain / bias parameters (layers.py:728-742)."""

    @staticmethod
    def forward(ctx, stats, gain, bias, run_mean, run_var, count, eps, momentum, training, events=1, n_images=1):
        """One event: per-channel scale / shift [C].  E > 1 events: one row per image, [N, C] (every image takes the
        statistics of its own event)."""
        C = gain.numel()
        dev = gain.device
        rows = 1 if events == 1 else n_images
        shape = (C,) if rows == 1 else (rows, C)
        scale = torch.empty(shape, dtype=torch.float32, device=dev)
        shift = torch.empty(shape, dtype=torch.float32, device=dev)
        mr = torch.empty(events, 2, C, dtype=torch.float32, device=dev)
        repl = int(stats.shape[1]) if stats is not None else 0
        H.call("ieagan_bn_finalize_fwd", H.ptr(stats), float(count), gain.data_ptr(), bias.data_ptr(), 0, 0, float(eps),
               float(momentum), int(training), run_mean.data_ptr(), run_var.data_ptr(), scale.data_ptr(),
               shift.data_ptr(), mr.data_ptr(), rows, C, events, repl,
               H.ptr(_fin_scratch("fwd", C, events, repl, device=dev) if training else None), H.stream())
        ctx.count, ctx.training, ctx.has_stats, ctx.events, ctx.rows, ctx.repl = count, training, stats is not None, events, rows, repl
        ctx.save_for_backward(gain, mr)
        return scale, shift

    @staticmethod
    def backward(ctx, dscale, dshift):
        gain, mr = ctx.saved_tensors
        C = gain.numel()
        dev = gain.device
        E, rows = ctx.events, ctx.rows
        dgain, dbias = torch.empty_like(gain), torch.empty_like(gain)
        dstat = torch.empty(E, 2, C, dtype=torch.float32, device=dev)
        shape = (C,) if rows == 1 else (rows, C)
        dscale = dscale.contiguous() if dscale is not None else torch.zeros(shape, device=dev)
        dshift = dshift.contiguous() if dshift is not None else torch.zeros(shape, device=dev)
        H.call("ieagan_bn_finalize_bwd", dscale.data_ptr(), dshift.data_ptr(), gain.data_ptr(), 0, 0, mr.data_ptr(),
               float(ctx.count), int(ctx.training), dgain.data_ptr(), dbias.data_ptr(), 0, dstat.data_ptr(), rows, C, E, 0,
               None, H.stream())
        dstats = dstat.unsqueeze(1).expand(E, ctx.repl, 2, C) if (ctx.has_stats and ctx.training) else None
        return dstats, dgain, dbias, None, None, None, None, None, None, None, None


# =====================================================================================================
# Fused convolution
# =====================================================================================================
def _conv_launch(x, Cx, Hs, Ws, rs, scale, shift, nstride, relu, N, Hc, Wc, Cin, Cout, taps, kpad, w, bias,
                 ra, Cra, Ca, ra_rs, rb, Crb, mask, out, stats, ra_scale=1.0, npe=0, flags=0, bnb=None):
    """``bnb`` = (scale, shift, nstride, relu): BatchNorm-apply backward fused into this (dgrad) launch -- ``mask`` is then the
    BatchNorm input x and ``stats`` the per-image accumulators [N, slots, 2, Cout] (``npe`` = 1), see include/ieagan_hip.h.
    ``stats``: None | a caller-zeroed tensor [groups, slots, 2, Cout] (tests; block b of a group adds into slot b % slots) | True: the
    buffer is allocated here with slots = the blocks per statistics group this very launch will use (ieagan_conv_stats_slots: the
    library plans the dispatch without launching) -- every slot then has ONE adder and the sums are bit-reproducible.  Returns the
    statistics tensor."""
    bs, bt, bn, br = (H.ptr(bnb[0]), H.ptr(bnb[1]), int(bnb[2]), int(bool(bnb[3]))) if bnb is not None else (None, None, 0, 0)
    alloc = stats is True
    d = H.ConvDesc(N, Hc, Wc, Cin, Cout, taps, kpad, H.src_desc(x, Cx, Hs, Ws, rs, scale, shift, nstride, relu),
                   H.ptr(w), H.ptr(bias), H.ptr(ra), Cra, Ca, ra_rs, float(ra_scale), H.ptr(rb), Crb, H.ptr(mask),
                   H.ptr(out), 16 if alloc else H.ptr(stats), int(npe), int(flags), bs, bt, bn, br,
                   0 if (alloc or stats is None) else int(stats.shape[1]))
    if alloc:
        slots = H.lib().ieagan_conv_stats_slots(d)
        if slots <= 0:
            raise RuntimeError(f"ieagan_conv_stats_slots failed ({slots}): {H.lib().ieagan_last_error().decode()}")
        stats = zeros((N // npe if npe else 1, slots, 2, Cout), out.device)
        d.stats, d.stats_slots = stats.data_ptr(), slots
    H.call("ieagan_conv_forward", d, H.stream())
    return stats


# Weight-gradient launches of a training backward pass go to a side stream (ExecOptions.wgrad_side_stream): nothing in the pass reads dW
# before the batched spectral-norm backward at its end (SNPass.flush joins), so they run under the dgrad chain of this and the following
# layers -- the mid / small feature maps (<= 240 blocks on 256 CUs, long float-atomic tails) leave most of the chip idle on their own.
_WGRAD_STREAMS = {}


def wgrad_stream(device) -> "torch.cuda.Stream":
    """Side stream for the weight-gradient launches issued from the current stream."""
    key = (torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream)
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _WGRAD_STREAMS[key]


def release_device_caches():
    """Drop the module-level device objects (zero-pool chunks, side streams, placeholder tensors): part of an orderly shutdown
    (train_fns ``train.close()``); everything is re-created on demand."""
    _ZEROS.reset()
    _WGRAD_STREAMS.clear()
    _PLACEHOLDERS.clear()


class BNLink:
    """Side channel between a conv's backward and the backward of the BatchNorm finalize that produced its prologue scale /
    shift: when the dgrad kernel folds the BatchNorm-apply backward in, the per-image sums (d shift, d scale) arrive as replicated
    accumulators [N, BNB_REPL, 2, C]; autograd is handed shape-correct placeholders and the finalize backward reads ``acc``."""
    __slots__ = ("acc",)

    def __init__(self):
        self.acc = None

class ResLink:
    """Hand-off of a shortcut gradient between two convs of one residual block.

    The block input x feeds both the first conv (through its prologue) and the last conv's residual
    operand.  Instead of letting autograd materialise and add two full-size gradients for x, the last
    conv's backward *deposits* its out-grad here and the first conv's backward adds it inside the kernel
    that produces dx anyway (prologue_bwd, or the dgrad epilogue when the prologue is a bare ReLU)."""
    __slots__ = ("g", "C", "Ca", "mode", "ready")

    def __init__(self):
        self.g, self.C, self.Ca, self.mode, self.ready = None, 0, 0, 0, False

    def deposit(self, g, C, Ca, mode):
        self.g, self.C, self.Ca, self.mode, self.ready = g, C, Ca, mode, True

    def take(self):
        if not self.ready:
            raise RuntimeError("shortcut gradient requested before it was produced (autograd order changed?)")
        out = (self.g, self.C, self.Ca, self.mode)
        self.g, self.ready = None, False
        return out


class SumLink(ResLink):
    """Fan-in of the gradients of one activation that feeds ``expected`` fused consumers (non-local block: theta, phi, g and
    the residual all read the block input).  Each consumer's backward adds the running sum inside the kernel that produces its own
    contribution and deposits the result; the last one to arrive hands the total to autograd, the others return None --
    instead of ``expected`` full-size tensors and ``expected - 1`` library add kernels."""
    __slots__ = ("expected", "seen")

    def __init__(self, expected):
        super().__init__()
        self.expected, self.seen = expected, 0

    def arrive(self, g, C):
        """Register this consumer's (already accumulated) gradient ``g``; returns the total if it is the last one, else None."""
        self.seen += 1
        if self.seen == self.expected:
            self.g, self.ready, self.seen = None, False, 0
            return g
        self.deposit(g, C, C, 0)
        return None


_PLACEHOLDERS = {}


def _placeholder(like):
    """A zero tensor of ``like``'s shape that is never read (autograd insists on shape-correct gradients)."""
    key = (tuple(like.shape), str(like.device))
    if key not in _PLACEHOLDERS:
        _PLACEHOLDERS[key] = torch.zeros(like.shape, dtype=torch.float32, device=like.device)
    return _PLACEHOLDERS[key]


# The whole backward of a 1x1 convolution on a large map in ONE launch (csrc/conv1x1_bwd.hip): effgrad + dgrad + prologue backward +
# wgrad + bias column sums, every operand tile read once (ExecOptions.fuse_1x1_backward; tests compare it with the separate launches).


def _fused_bwd_launch(name, d, rec, ws, dwp, Cout, K, operands):
    """Launch a whole-backward kernel whose partial dW slabs sit in ``ws``.  ``ExecOptions.fused_reduce_side_stream`` (off by default: it
    measured 2 ms slower) sends the slab fold (``wgrad_reduce``, ~10 us + a launch boundary per layer) to the weight-gradient side stream:
    nothing reads dW before the end of the pass (``SNPass.flush`` joins the side stream)."""
    o = opts_of(rec)
    side_ok = ws is not None and o.wgrad_side_stream and o.fused_reduce_side_stream and getattr(rec, "deferred", False)
    pass_ = getattr(rec, "pass_", None)
    batch_ok = (not side_ok and ws is not None and o.fused_reduce_batched and getattr(rec, "deferred", False) and pass_ is not None
                and pass_.arena is not None)
    if side_ok or batch_ok:
        d.flags |= H.BWD_NO_REDUCE
    H.call(name, d, H.stream())
    if batch_ok:                        # folded with the other slab sets of the pass in SNPass.flush
        pass_.reduces.append((ws, dwp, ws.numel() // (Cout * rec.kpad), Cout, rec.kpad, K))
        return
    if side_ok:
        dev = ws.device
        fork = torch.cuda.Event()
        fork.record(torch.cuda.current_stream())
        side = wgrad_stream(dev)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            H.call("ieagan_wgrad_reduce", ws.data_ptr(), dwp.data_ptr(), ws.numel() // (Cout * rec.kpad), Cout, rec.kpad, K, H.stream())
        for t in (ws,) + tuple(operands):           # read by the side stream after this node has handed them back to the pool
            if t is not None:
                t.record_stream(side)
        rec.pass_.side = side


def _fused_1x1_eligible(ctx, dout, dstats):
    """Can ``ConvFn.backward`` take the fused 1x1 backward for this node?  (shape instantiated, big map, both gradients wanted, a
    shortcut-gradient link the kernel can add in place.)"""
    rec = ctx.rec
    o = opts_of(rec)
    if not o.fuse_1x1_backward:
        return False
    taps, rs, relu, Ca, ra_rs, nstride, Hc, Wc = ctx.cfg
    need = ctx.needs_input_grad
    has_bias, has_aff, has_ra, has_rb = ctx.has
    res_out, res_in = ctx.links
    x = ctx.saved_tensors[0]
    N = x.shape[0]
    if taps != 1 or rs not in (0, 2) or not need[0] or not need[1] or Wc % 32 != 0 or N * Hc * Wc < o.fuse_1x1_min_pixels:
        return False
    if not H.lib().ieagan_conv1x1_bwd_supported(rec.cin, rec.out, rs, int(has_aff)):
        return False
    if isinstance(res_in, SumLink) or (has_aff and (rs != 0 or ctx.events > 1 and nstride == 0)):
        return False
    if res_in is not None and res_in.ready:
        if rs == 2 and not (res_out is None and not relu and not has_aff):
            return False            # a pooled source takes a link only as conv_sc (plain da + add at the pooled resolution, re-deposited)
        if rs == 0 and res_in.mode not in (0, 1, 2):
            return False
    elif rs == 2 and res_out is None and not relu and res_in is not None:
        return False                # conv_sc whose link was not produced: leave the bookkeeping to the generic path
    if rs == 2 and rec.cin != 16 and (relu or res_in is None or not res_in.ready):
        return False                # dx at source resolution of a pooled source exists for Cin = 16 only
    st = dout.stride()
    ok_stride = dout.is_contiguous() or (st[3] == 1 and st[2] % 8 == 0 and st[2] >= rec.out and st[1] == Wc * st[2] and
                                         st[0] == Hc * Wc * st[2] and dout.data_ptr() % 16 == 0)
    return ok_stride


def _conv1x1_backward_fused(ctx, dout, dstats):
    x, weight, scale, shift, out = ctx.saved_tensors
    rec = ctx.rec
    taps, rs, relu, Ca, ra_rs, nstride, Hc, Wc = ctx.cfg
    has_bias, has_aff, has_ra, has_rb = ctx.has
    res_out, res_in = ctx.links
    N, Hs, Ws, Cx = x.shape
    Cout, Cin = rec.out, rec.cin
    dev = x.device
    need = ctx.needs_input_grad
    g = dout
    Cg = Cout if dout.is_contiguous() else dout.stride()[2]
    eff = dstats is not None
    # g_eff has other consumers than this layer's own gradients when the shortcut operands need it: the kernel then stores it
    shortcut_needs_g = (has_ra and need[5]) or (has_rb and need[6])
    geff = torch.empty(N, Hc, Wc, Cout, dtype=BF16, device=dev) if (eff and shortcut_needs_g) else None
    dstat = dstats[:, 0].contiguous() if eff else None
    colsum = sn_scratch(rec, "b", (STAT_REPL, Cout), dev) if (has_bias and need[2]) else None
    dwp = sn_scratch(rec, "w", (Cout, rec.kpad), dev)
    # ---- shortcut-gradient link into dx
    lg = lC = lCa = lmode = None
    if res_in is not None and res_in.ready:
        lg, lC, lCa, lmode = res_in.take()
    conv_sc = rs == 2 and lg is not None            # D block conv_sc: plain da at the pooled resolution + the identity part, re-deposited
    out_mode = 1 if conv_sc else 0
    if conv_sc:
        lmode = 0                                   # the deposited gradient lives at the block OUTPUT (= pooled) resolution
    dx = torch.empty((N, Hc, Wc, Cin) if out_mode == 1 else (N, Hs, Ws, Cin), dtype=BF16, device=dev)
    d = H.Conv1x1BwdDesc(N, Hc, Wc, Cin, Cout, rec.kpad, rec.kpad2, H.src_desc(x, Cx, Hs, Ws, rs, scale, shift, nstride, relu),
                         g.data_ptr(), Cg, H.ptr(out) if eff else None, H.ptr(dstat), N // ctx.events, H.ptr(geff), rec.w_bwd.data_ptr(),
                         H.ptr(lg), lC or 0, lCa or 0, lmode or 0, dx.data_ptr(), out_mode, 16 if has_aff else None, dwp.data_ptr(), None, H.ptr(colsum),
                         opts_of(rec).b1_flags, 0)
    acc = None
    if has_aff:         # per-image BatchNorm accumulators: one slot per block of the image (single adder: bit-reproducible)
        d.bn_slots = H.lib().ieagan_conv1x1_bwd_slots(d)
        if d.bn_slots <= 0:
            raise RuntimeError(f"ieagan_conv1x1_bwd_slots failed: {H.lib().ieagan_last_error().decode()}")
        acc = zeros((N, d.bn_slots, 2, Cin), dev)
        d.bn_acc = acc.data_ptr()
    ws_n = H.lib().ieagan_conv1x1_bwd_workspace(d)
    ws = None
    if ws_n > 0:
        ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
        d.partials = ws.data_ptr()
    _fused_bwd_launch("ieagan_conv1x1_bwd", d, rec, ws, dwp, Cout, Cin, ())
    # ---- residual operands (as the generic path, on g_eff)
    gl = geff if eff else g
    d_ra = d_rb = None
    if has_ra and need[5]:
        rshape = ctx.ra_shape
        if res_out is not None:
            res_out.deposit(gl, gl.stride()[2] if not gl.is_contiguous() else Cout, Ca, ra_rs)
        elif ra_rs == 0:
            if Ca == Cout and rshape[-1] == Cout:
                d_ra = gl
            else:
                d_ra = torch.zeros(rshape, dtype=BF16, device=dev)
                d_ra[..., :Ca] = gl[..., :Ca]
        else:
            d_ra = torch.empty(rshape, dtype=BF16, device=dev)
            H.call("ieagan_res_bwd", gl.contiguous().data_ptr(), Cout, d_ra.data_ptr(), rshape[-1], Ca, ra_rs, N, rshape[1], rshape[2],
                   H.stream())
    if has_rb and need[6]:
        d_rb = gl[..., Ca:]
    dscale = dshift = None
    if has_aff:
        bn_link = getattr(scale, "_bn_link", None)
        if bn_link is not None:
            bn_link.acc = acc
            dscale = dshift = _placeholder(scale)
        else:                       # stand-alone use (tests): fold the replicated per-image accumulators here
            sums = acc.sum(1)
            dshift, dscale = sums[:, 0], sums[:, 1]
            if nstride == 0:
                dshift, dscale = dshift.sum(0), dscale.sum(0)
    if conv_sc:
        res_in.deposit(dx, Cin, Cin, 2)
        dx = None
    dW, dbias = sn_backward(dwp, weight, rec, colsum, ctx.bias_ref)
    return dx, dW, dbias, dscale, dshift, d_ra, d_rb, None, None, None, None, None, None, None, None, None, None, None


# The whole backward of a 3x3 convolution with Cin = Cout = 16 / 32 on a large map in ONE launch (csrc/conv3x3_bwd.hip): effgrad on load + dgrad
# with the prologue backward (ReLU mask / BatchNorm apply / 2x2 sum of an up-sampled source) in its store phase + wgrad + bias column sums
# from the same LDS tiles (ExecOptions.fuse_3x3_backward; tests compare it with the separate launches).
def _fused_3x3_eligible(ctx, dout, dstats):
    rec = ctx.rec
    o = opts_of(rec)
    if not o.fuse_3x3_backward:
        return False
    taps, rs, relu, Ca, ra_rs, nstride, Hc, Wc = ctx.cfg
    need = ctx.needs_input_grad
    has_bias, has_aff, has_ra, has_rb = ctx.has
    res_out, res_in = ctx.links
    x = ctx.saved_tensors[0]
    N = x.shape[0]
    if taps != 9 or rec.cin != rec.out or not need[0] or not need[1] or has_ra or has_rb or res_out is not None or res_in is not None:
        return False
    if N * Hc * Wc < o.fuse_3x3_min_pixels or rec.kpad != rec.kpad2 or x.shape[3] != rec.cin:
        return False
    if not H.lib().ieagan_conv3x3_bwd_supported(rec.cin, rs, int(has_aff), int(bool(relu)), int(dstats is not None), Hc, Wc):
        return False
    if has_aff and ctx.events > 1 and nstride == 0:
        return False
    st = dout.stride()
    return dout.is_contiguous() or (st[3] == 1 and st[2] % 8 == 0 and st[2] >= rec.out and st[1] == Wc * st[2] and
                                    st[0] == Hc * Wc * st[2] and dout.data_ptr() % 16 == 0)


def _conv3x3_backward_fused(ctx, dout, dstats):
    x, weight, scale, shift, out = ctx.saved_tensors
    rec = ctx.rec
    taps, rs, relu, Ca, ra_rs, nstride, Hc, Wc = ctx.cfg
    has_bias, has_aff, has_ra, has_rb = ctx.has
    N, Hs, Ws, Cx = x.shape
    C = rec.cin
    dev = x.device
    need = ctx.needs_input_grad
    Cg = C if dout.is_contiguous() else dout.stride()[2]
    eff = dstats is not None
    dstat = dstats[:, 0].contiguous() if eff else None
    colsum = sn_scratch(rec, "b", (STAT_REPL, C), dev) if (has_bias and need[2]) else None
    dwp = sn_scratch(rec, "w", (C, rec.kpad), dev)
    dx = torch.empty(N, Hs, Ws, C, dtype=BF16, device=dev)
    d = H.Conv3x3BwdDesc(N, Hc, Wc, C, rec.kpad, H.src_desc(x, Cx, Hs, Ws, rs, scale, shift, nstride, relu), dout.data_ptr(), Cg,
                         H.ptr(out) if eff else None, H.ptr(dstat), N // ctx.events, rec.w_bwd.data_ptr(), dx.data_ptr(), 16 if has_aff else None,
                         dwp.data_ptr(), None, H.ptr(colsum), 0, 0)
    ws = torch.empty(H.lib().ieagan_conv3x3_bwd_workspace(d), dtype=torch.float32, device=dev)
    d.partials = ws.data_ptr()
    acc = None
    if has_aff:         # per-image BatchNorm accumulators: one slot per block of the image (single adder: bit-reproducible)
        d.bn_slots = H.lib().ieagan_conv3x3_bwd_slots(d)
        if d.bn_slots <= 0:
            raise RuntimeError(f"ieagan_conv3x3_bwd_slots failed: {H.lib().ieagan_last_error().decode()}")
        acc = zeros((N, d.bn_slots, 2, C), dev)
        d.bn_acc = acc.data_ptr()
    _fused_bwd_launch("ieagan_conv3x3_bwd", d, rec, ws, dwp, C, 9 * C, ())
    dscale = dshift = None
    if has_aff:
        bn_link = getattr(scale, "_bn_link", None)
        if bn_link is not None:
            bn_link.acc = acc
            dscale = dshift = _placeholder(scale)
        else:                       # stand-alone use (tests): fold the replicated per-image accumulators here
            sums = acc.sum(1)
            dshift, dscale = sums[:, 0], sums[:, 1]
            if nstride == 0:
                dshift, dscale = dshift.sum(0), dscale.sum(0)
    dW, dbias = sn_backward(dwp, weight, rec, colsum, ctx.bias_ref)
    return dx, dW, dbias, dscale, dshift, None, None, None, None, None, None, None, None, None, None, None, None, None


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, scale, shift, ra, rb, rec, taps, rs, relu, Ca, ra_rs, want_stats, res_out, res_in, events=1, flags=0):
        N, Hs, Ws, Cx = x.shape
        Cout, Cin = rec.out, rec.cin
        assert x.dtype == BF16 and x.is_contiguous() and Cx == Cin, (x.dtype, x.shape, Cin)
        assert N % events == 0, (N, events)
        Hc, Wc = (2 * Hs, 2 * Ws) if rs == 1 else (Hs // 2, Ws // 2) if rs == 2 else (Hs, Ws)
        out = torch.empty(N, Hc, Wc, Cout, dtype=BF16, device=x.device)
        nstride = 0 if (scale is None or scale.dim() == 1) else scale.shape[1]
        # per-layer kernel-selection flags of the conv descriptor (H.CONV_FP8: e4m3 MFMA operands, conv_dtype='fp8' of the owning
        # network -- BASELINE configs[4]); they only mean something to the C = 64 / 128 3x3 kernels and ride along to the dgrad launch
        flags = int(flags) if (taps == 9 and Cin in (64, 128)) else 0
        stats = _conv_launch(x, Cx, Hs, Ws, rs, scale, shift, nstride, relu, N, Hc, Wc, Cin, Cout, taps, rec.kpad, rec.w_fwd,
                             bias, ra, ra.shape[-1] if ra is not None else 0, Ca, ra_rs, rb,
                             rb.shape[-1] if rb is not None else 0, None, out, True if want_stats else None, npe=N // events, flags=flags)
        ctx.events, ctx.flags = events, flags
        ctx.rec, ctx.cfg = rec, (taps, rs, relu, Ca, ra_rs, nstride, Hc, Wc)
        ctx.ra_shape = ra.shape if ra is not None else None
        ctx.has = (bias is not None, scale is not None, ra is not None, rb is not None)
        ctx.links = (res_out, res_in)
        ctx.bias_ref = bias
        ctx.save_for_backward(x, weight, scale, shift, out if want_stats else None)
        return out, stats

    @staticmethod
    def backward(ctx, dout, dstats):
        if _fused_1x1_eligible(ctx, dout, dstats):
            return _conv1x1_backward_fused(ctx, dout, dstats)
        if _fused_3x3_eligible(ctx, dout, dstats):
            return _conv3x3_backward_fused(ctx, dout, dstats)
        x, weight, scale, shift, out = ctx.saved_tensors
        rec = ctx.rec
        opts = opts_of(rec)
        taps, rs, relu, Ca, ra_rs, nstride, Hc, Wc = ctx.cfg
        has_bias, has_aff, has_ra, has_rb = ctx.has
        res_out, res_in = ctx.links
        N, Hs, Ws, Cx = x.shape
        Cout, Cin = rec.out, rec.cin
        dev = x.device
        need = ctx.needs_input_grad
        # The out-grad may be a channel slice of a wider NHWC tensor (concat shortcut of a D block): the dgrad / wgrad kernels
        # take a row stride, so only the statistics / effgrad paths need a packed copy.
        Cg = Cout
        st = dout.stride()
        sliced = (not dout.is_contiguous() and st[3] == 1 and st[2] % 8 == 0 and st[2] >= Cout and st[1] == Wc * st[2]
                  and st[0] == Hc * Wc * st[2] and dout.data_ptr() % 16 == 0)
        if sliced and dstats is None and not (has_bias and need[2] and not need[1]) and not (has_ra and need[5]) and not (has_rb and need[6]):
            g, Cg = dout, st[2]
        else:
            g = dout.contiguous()
        P = N * Hc * Wc
        # ---- fold the batch-statistics path of the following BN into the out-grad; bias gradient
        dbias = None
        colsum_in_wgrad = False
        if dstats is not None:
            geff = torch.empty_like(g)
            colsum = None
            if has_bias:
                colsum = sn_scratch(rec, "b", (STAT_REPL, Cout), dev) if (need[1] and need[2]) else zeros((STAT_REPL, Cout), dev)
            H.call("ieagan_effgrad", g.data_ptr(), out.data_ptr(), dstats[:, 0].contiguous().data_ptr(), geff.data_ptr(),
                   H.ptr(colsum), P, Cout, ctx.events, H.stream())
            g = geff
        elif has_bias and need[2]:
            colsum = sn_scratch(rec, "b", (STAT_REPL, Cout), dev) if need[1] else zeros((STAT_REPL, Cout), dev)
            if need[1]:
                colsum_in_wgrad = True      # the wgrad kernel stages every g tile anyway: it takes the column sums along
            else:
                H.call("ieagan_effgrad", g.data_ptr(), None, None, None, colsum.data_ptr(), P, Cout, 1, H.stream())
        # ---- residual operands
        d_ra = d_rb = None
        if has_ra and need[5]:
            rshape = ctx.ra_shape
            if res_out is not None:
                res_out.deposit(g, Cout, Ca, ra_rs)            # consumed in-kernel by the block's first conv
            elif ra_rs == 0:
                if Ca == Cout and rshape[-1] == Cout:
                    d_ra = g
                else:
                    d_ra = torch.zeros(rshape, dtype=BF16, device=dev)
                    d_ra[..., :Ca] = g[..., :Ca]
            else:
                d_ra = torch.empty(rshape, dtype=BF16, device=dev)
                H.call("ieagan_res_bwd", g.data_ptr(), Cout, d_ra.data_ptr(), rshape[-1], Ca, ra_rs, N, rshape[1], rshape[2],
                       H.stream())
        if has_rb and need[6]:
            d_rb = g[..., Ca:]
        # ---- weight-gradient accumulator; fork point of the side stream (g and the zeroed accumulators are ready here)
        dwp = fork = None
        if need[1]:
            dwp = sn_scratch(rec, "w", (Cout, rec.kpad), dev)
            if (opts.wgrad_side_stream and rec.deferred and d_ra is not g and d_rb is None):
                fork = torch.cuda.Event()
                fork.record(torch.cuda.current_stream())
        # ---- data gradient
        dx = dscale = dshift = None
        if need[0] or (has_aff and (need[3] or need[4])):
            da = torch.empty(N, Hc, Wc, Cin, dtype=BF16, device=dev)
            fuse_mask = relu and not has_aff and rs == 0
            plain = not relu and not has_aff
            lg = lC = lCa = lmode = None
            fan_in = res_in if isinstance(res_in, SumLink) else None
            if res_in is not None and not res_in.ready:
                res_in = None          # the shortcut operand needed no gradient (e.g. a detached block input) / first of a fan-in
            if res_in is not None:
                lg, lC, lCa, lmode = res_in.take()
            bn_link = getattr(scale, "_bn_link", None) if has_aff else None
            if (opts.fuse_bn_backward and has_aff and rs == 0 and bn_link is not None and (Hs * Ws) % 128 == 0 and Cin % 8 == 0
                    and (res_in is None or lmode in (0, 1))):
                # BatchNorm apply + ReLU backward inside the dgrad epilogue: dx is written directly, the per-(n, c) sums go to
                # replicated per-image accumulators that bn_finalize_bwd folds (no da tensor, no stand-alone pass over da / x)
                dx = torch.empty(N, Hs, Ws, Cin, dtype=BF16, device=dev)
                up = res_in is not None and lmode == 1          # shortcut gradient at double resolution: 2x2 SUM = 4 * average
                acc = _conv_launch(g, Cg, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, Cout, Cin, taps, rec.kpad2, rec.w_bwd, None,
                                   lg, lC or 0, lCa or 0, 2 if up else 0, None, 0, x, dx, True, ra_scale=4.0 if up else 1.0, npe=1,
                                   bnb=(scale, shift, nstride, relu), flags=ctx.flags)
                bn_link.acc = acc
                dscale = dshift = _placeholder(scale)
            elif res_in is not None and (fuse_mask or (plain and rs == 0)):
                # bare-ReLU / no prologue (D blocks): the dgrad epilogue masks the main path and adds the
                # shortcut gradient (0.25 * nearest-expand when the shortcut was average-pooled)
                _conv_launch(g, Cg, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, Cout, Cin, taps, rec.kpad2, rec.w_bwd, None,
                             lg, lC, lCa, 1 if lmode == 2 else 0, None, 0, x if fuse_mask else None, da, None,
                             ra_scale=0.25 if lmode == 2 else 1.0, flags=ctx.flags)
                dx = da
            elif res_in is not None and plain and rs == 2 and res_out is None:
                # conv_sc of a D block (pooled, un-activated input): chain -- add the deposited shortcut
                # gradient at the pooled resolution and re-deposit the sum for the block's first conv
                _conv_launch(g, Cg, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, Cout, Cin, taps, rec.kpad2, rec.w_bwd, None,
                             lg, lC, lCa, 0, None, 0, None, da, None)
                res_in.deposit(da, Cin, Cin, 2)
                dx = None
            else:
                _conv_launch(g, Cg, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, Cout, Cin, taps, rec.kpad2, rec.w_bwd, None,
                             None, 0, 0, 0, None, 0, x if fuse_mask else None, da, None, flags=ctx.flags)
                if (fuse_mask or (plain and rs == 0)) and res_in is None:
                    dx = da
                else:
                    dx = torch.empty(N, Hs, Ws, Cin, dtype=BF16, device=dev)
                    acc = None
                    if has_aff:     # per-image {sum d, sum d x} in one slot per block (single writer: bit-reproducible), folded by bn_finalize_bwd
                        acc = zeros((N, H.PROLOGUE_BWD_SLOTS, 2, Cin), dev)
                    rmode = 0
                    if res_in is not None:
                        if lmode == 2:
                            raise RuntimeError("pooled shortcut gradient cannot be added by prologue_bwd")
                        rmode = lmode
                    H.call("ieagan_prologue_bwd", da.data_ptr(), x.data_ptr(), Cx, H.ptr(scale), H.ptr(shift), nstride, int(relu),
                           rs, dx.data_ptr(), H.ptr(acc), None, N, Hs, Ws, Cin, H.ptr(lg), lC or 0, lCa or 0, rmode,
                           H.PROLOGUE_BWD_SLOTS if has_aff else 0, H.stream())
                    if has_aff:
                        if bn_link is not None:
                            bn_link.acc = acc
                            dscale = dshift = _placeholder(scale)
                        else:                       # stand-alone use (tests): fold the per-image slots here
                            sums = acc.sum(1)
                            dshift, dscale = sums[:, 0], sums[:, 1]
                            if nstride == 0:
                                dshift, dscale = dshift.sum(0), dscale.sum(0)
            if fan_in is not None:
                if not (plain and rs == 0 and dx is da):
                    raise RuntimeError("a fan-in gradient link needs a plain same-resolution conv")
                dx = fan_in.arrive(dx, Cin)
        # ---- weight gradient (skipped entirely when the parameter is frozen, e.g. D in the G phase)
        dW = None
        if need[1]:
            d = H.WgradDesc(N, Hc, Wc, Cin, Cout, taps, rec.kpad,
                            H.src_desc(x, Cx, Hs, Ws, rs, scale, shift, nstride, relu), g.data_ptr(), Cg, dwp.data_ptr(), 0, 0, None,
                            H.ptr(colsum) if colsum_in_wgrad else None)

            def launch():
                # large weight x many pixel splits: the blocks store partial slabs and a second launch folds them (two-stage
                # accumulation; the float-atomic tail was the longest phase of these launches)
                ws_n = H.lib().ieagan_conv_wgrad_workspace(d, int(opts.use_tr_read)) if opts.two_stage_wgrad else 0
                if ws_n > 0:
                    ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
                    d.partials = ws.data_ptr()
                H.call("ieagan_conv_wgrad", d, int(opts.use_tr_read), H.stream())

            if fork is not None:
                side = wgrad_stream(dev)
                with torch.cuda.stream(side):
                    side.wait_event(fork)
                    launch()
                for t in (x, g, scale, shift):          # read by the side stream after this node has returned them to the pool
                    if t is not None:
                        t.record_stream(side)
                rec.pass_.side = side
            else:
                launch()
            dW, dbias = sn_backward(dwp, weight, rec, colsum if (has_bias and need[2]) else None, ctx.bias_ref)
        elif has_bias and need[2]:
            dbias = colsum.sum(0)
        return dx, dW, dbias, dscale, dshift, d_ra, d_rb, None, None, None, None, None, None, None, None, None, None, None


def conv(x, weight, bias, rec, taps, *, scale=None, shift=None, relu=False, rs=0, ra=None, Ca=0, ra_rs=0, rb=None,
         want_stats=False, res_out=None, res_in=None, events=1, flags=0):
    """``events``: the batch holds that many events of N / events images each; the statistics of the output are taken per
    event ([E, STAT_REPL, 2, Cout]).  ``flags``: ieagan_conv_desc.flags of this layer (H.CONV_FP8)."""
    return ConvFn.apply(x, weight, bias, scale, shift, ra, rb, rec, taps, rs, relu, Ca, ra_rs, want_stats, res_out, res_in, events, flags)


# =====================================================================================================
# single-channel-image convolutions
# =====================================================================================================
class InputConvFn(torch.autograd.Function):
    """D.input_conv: fp32 image [N,1,H,W] -> bf16 NHWC [N,H,W,C]  (model.py:730, 905)."""

    @staticmethod
    def forward(ctx, img, weight, bias, rec):
        N, _, Hh, Ww = img.shape
        C = rec.out
        img = img.contiguous().float()
        out = torch.empty(N, Hh, Ww, C, dtype=BF16, device=img.device)
        H.call("ieagan_conv_1toC", img.data_ptr(), None, rec.w_plain.data_ptr(), H.ptr(bias), out.data_ptr(), N, Hh, Ww, C, 0,
               H.stream())
        ctx.rec, ctx.bias_ref = rec, bias
        ctx.save_for_backward(img, weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        img, weight = ctx.saved_tensors
        rec = ctx.rec
        N, _, Hh, Ww = img.shape
        C = rec.out
        dev = img.device
        g = dout.contiguous()
        need = ctx.needs_input_grad
        dimg = dW = dbias = colsum = None
        if need[2]:
            colsum = sn_scratch(rec, "b", (STAT_REPL, C), dev) if need[1] else zeros((STAT_REPL, C), dev)
            H.call("ieagan_effgrad", g.data_ptr(), None, None, None, colsum.data_ptr(), N * Hh * Ww, C, 1, H.stream())
        if need[0]:
            dimg = torch.empty(N, 1, Hh, Ww, dtype=torch.float32, device=dev)
            H.call("ieagan_conv_Cto1", g.data_ptr(), None, None, 0, 0, rec.w_plain.data_ptr(), None, dimg.data_ptr(), 0, N, Hh,
                   Ww, C, 1, H.stream())
        if need[1]:
            dw = sn_scratch(rec, "w", (9, C), dev)
            H.call("ieagan_wgrad_c1", img.data_ptr(), None, g.data_ptr(), None, None, 0, 0, dw.data_ptr(), N, Hh, Ww, C, 0,
                   H.stream())
            dW, dbias = sn_backward(dw, weight, rec, colsum, ctx.bias_ref)
        elif colsum is not None:
            dbias = colsum.sum(0)
        return dimg, dW, dbias, None


class DStemFn(torch.autograd.Function):
    """Input side of the first discriminator block: img -> (h1 = conv1(h0), p0 = AvgPool2d(h0), sc = conv_sc(p0)) with
    h0 = input_conv(img) recomputed on chip (reference model.py:905 + 534-557).  ``link``: the ResLink the block's conv4 deposits its
    out-gradient into (the pooled identity shortcut's share of d p0)."""

    @staticmethod
    def forward(ctx, img, w_in, b_in, w1, b1, wsc, bsc, rec_in, rec1, recsc, link):
        N, _, Hh, Ww = img.shape
        img = img.contiguous().float()
        dev = img.device
        h1 = torch.empty(N, Hh, Ww, 16, dtype=BF16, device=dev)
        p0 = torch.empty(N, Hh // 2, Ww // 2, 32, dtype=BF16, device=dev)
        sc = torch.empty(N, Hh // 2, Ww // 2, 32, dtype=BF16, device=dev)
        d = H.DStemDesc(img.data_ptr(), N, Hh, Ww, rec_in.w_plain.data_ptr(), b_in.data_ptr(), rec1.w_fwd.data_ptr(), b1.data_ptr(),
                        recsc.w_fwd.data_ptr(), bsc.data_ptr(), h1.data_ptr(), p0.data_ptr(), sc.data_ptr(), None, None, None, None, None, None, None)
        H.call("ieagan_d_stem_fwd", d, H.stream())
        ctx.recs, ctx.link = (rec_in, rec1, recsc), link
        ctx.params = (w_in, b_in, w1, b1, wsc, bsc)
        ctx.save_for_backward(img, p0)
        return h1, p0, sc

    @staticmethod
    def backward(ctx, dh1, dp0, dsc):
        img, p0 = ctx.saved_tensors
        rec_in, rec1, recsc = ctx.recs
        w_in, b_in, w1, b1, wsc, bsc = ctx.params
        need = ctx.needs_input_grad
        N, _, Hh, Ww = img.shape
        Hp, Wp = Hh // 2, Ww // 2
        dev = img.device
        want_w = any(need[1:7])
        dh1 = dh1.contiguous()
        # ---- the identity-shortcut share of d p0: deposited by conv4 (channels [0, 32) of its out-gradient) or handed over by autograd
        lg, lC, lCa = dp0, 32, 32
        if ctx.link is not None and ctx.link.ready:
            lg, lC, lCa, _ = ctx.link.take()
        # ---- conv_sc backward: d p0 = dsc Wsc + shortcut share (+ its weight / bias gradients)
        st = dsc.stride()
        Cg = 32 if dsc.is_contiguous() else st[2]
        dpt = torch.empty(N, Hp, Wp, 32, dtype=BF16, device=dev)
        dWsc = dbsc = None
        if want_w:
            dwp = sn_scratch(recsc, "w", (32, recsc.kpad), dev)
            colsum = sn_scratch(recsc, "b", (STAT_REPL, 32), dev)
            d = H.Conv1x1BwdDesc(N, Hp, Wp, 32, 32, recsc.kpad, recsc.kpad2, H.src_desc(p0, 32, Hp, Wp, 0, None, None, 0, False), dsc.data_ptr(), Cg,
                                 None, None, N, None, recsc.w_bwd.data_ptr(), H.ptr(lg), lC if lg is not None else 0, lCa if lg is not None else 0, 0,
                                 dpt.data_ptr(), 1, None, dwp.data_ptr(), None, colsum.data_ptr(), opts_of(recsc).b1_flags, 0)
            ws_n = H.lib().ieagan_conv1x1_bwd_workspace(d)
            if ws_n > 0:
                ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
                d.partials = ws.data_ptr()
            H.call("ieagan_conv1x1_bwd", d, H.stream())
            dWsc, dbsc = sn_backward(dwp, wsc, recsc, colsum, bsc)
        else:
            _conv_launch(dsc, Cg, Hp, Wp, 0, None, None, 0, False, N, Hp, Wp, 32, 32, 1, recsc.kpad2, recsc.w_bwd, None, lg, lC if lg is not None else 0,
                         lCa if lg is not None else 0, 0, None, 0, None, dpt, None)
        dW_in = db_in = dW1 = db1 = dimg = None
        if want_w:
            dw_in = sn_scratch(rec_in, "w", (9, 32), dev)
            cs_in = sn_scratch(rec_in, "b", (STAT_REPL, 32), dev)
            dw1 = sn_scratch(rec1, "w", (16, rec1.kpad), dev)
            cs1 = sn_scratch(rec1, "b", (STAT_REPL, 16), dev)
            d = H.DStemDesc(img.data_ptr(), N, Hh, Ww, rec_in.w_plain.data_ptr(), b_in.data_ptr(), None, None, None, None, None, None, None,
                            dh1.data_ptr(), dpt.data_ptr(), rec1.w_bwd.data_ptr(), dw_in.data_ptr(), cs_in.data_ptr(), dw1.data_ptr(), cs1.data_ptr())
            H.call("ieagan_d_stem_bwd", d, H.stream())
            dW_in, db_in = sn_backward(dw_in, w_in, rec_in, cs_in, b_in)
            dW1, db1 = sn_backward(dw1, w1, rec1, cs1, b1)
        if need[0]:
            # the pass that trains G: d img = input_conv^T (dh1 W1 + 0.25 expand(d p0)) through the generic launches
            dh0 = torch.empty(N, Hh, Ww, 32, dtype=BF16, device=dev)
            _conv_launch(dh1, 16, Hh, Ww, 0, None, None, 0, False, N, Hh, Ww, 16, 32, 1, rec1.kpad2, rec1.w_bwd, None, dpt, 32, 32, 1, None, 0, None,
                         dh0, None, ra_scale=0.25)
            dimg = torch.empty(N, 1, Hh, Ww, dtype=torch.float32, device=dev)
            H.call("ieagan_conv_Cto1", dh0.data_ptr(), None, None, 0, 0, rec_in.w_plain.data_ptr(), None, dimg.data_ptr(), 0, N, Hh, Ww, 32, 1,
                   H.stream())
        return dimg, dW_in, db_in, dW1, db1, dWsc, dbsc, None, None, None, None


class OutputConvFn(torch.autograd.Function):
    """G.output_layer: BN apply + ReLU + 3x3 conv (C -> 1) + tanh -> fp32 [N,1,H,W]  (model.py:379-387, 487)."""

    @staticmethod
    def forward(ctx, h, scale, shift, weight, bias, rec):
        N, Hh, Ww, C = h.shape
        y = torch.empty(N, 1, Hh, Ww, dtype=torch.float32, device=h.device)
        ns = 0 if scale.dim() == 1 else C           # per-image rows when the batch holds several events
        H.call("ieagan_conv_Cto1", h.data_ptr(), scale.data_ptr(), shift.data_ptr(), ns, 1, rec.w_plain.data_ptr(), H.ptr(bias),
               y.data_ptr(), 1, N, Hh, Ww, C, 0, H.stream())
        ctx.rec = rec
        ctx.save_for_backward(h, scale, shift, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, scale, shift, weight, y = ctx.saved_tensors
        rec = ctx.rec
        N, Hh, Ww, C = h.shape
        dev = h.device
        need = ctx.needs_input_grad
        dpre = (dy.contiguous().float() * (1.0 - y * y)).contiguous()          # tanh'
        dh = dscale = dshift = dW = dbias = None
        if need[4]:
            dbias = dpre.sum().view(1)
        if need[0] or need[1] or need[2]:
            # dgrad of the 3x3 conv with the BatchNorm-apply + ReLU backward in its store phase: the gradient w.r.t. the activated
            # tensor (0.5 GB at 256x768) is never written (was: conv_1toC -> da, then prologue_bwd over da and h)
            dh = torch.empty_like(h)
            ns = 0 if scale.dim() == 1 else C
            # {sum d, sum d x} in single-adder slots (bit-reproducible), folded here in a fixed order
            slots = H.lib().ieagan_conv_1toC_bnb_slots(N, Hh, Ww, ns)
            acc = zeros((N if ns else 1, slots, 2, C), dev)
            H.call("ieagan_conv_1toC_bnb", dpre.data_ptr(), None, rec.w_plain.data_ptr(), h.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                   ns, 1, dh.data_ptr(), acc.data_ptr(), None, N, Hh, Ww, C, 1, slots, H.stream())
            sums = acc.sum(1)
            dshift, dscale = sums[:, 0].reshape(shift.shape), sums[:, 1].reshape(scale.shape)
        if need[3]:
            dw = sn_scratch(rec, "w", (9, C), dev)
            H.call("ieagan_wgrad_c1", dpre.data_ptr(), None, h.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                   0 if scale.dim() == 1 else C, 1, dw.data_ptr(), N, Hh, Ww, C, 1, H.stream())
            dW = sn_backward(dw, weight, rec)[0]
        return dh, dscale, dshift, dW, dbias, None


def output_conv_export(h, scale, shift, rec, bias):
    """G.output_layer fused with the export of ``model.generate``: BN apply + ReLU + conv + tanh + threshold +
    256^((x+1)/2) - 1 + clamp + crop -> detector units, fp32 [N, H-6, W] (inference only, no autograd)."""
    N, Hh, Ww, C = h.shape
    out = torch.empty(N, Hh - 6, Ww, dtype=torch.float32, device=h.device)
    H.call("ieagan_conv_Cto1", h.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0 if scale.dim() == 1 else C, 1, rec.w_plain.data_ptr(), H.ptr(bias),
           out.data_ptr(), 2, N, Hh, Ww, C, 0, H.stream())
    return out


# =====================================================================================================
# module-boundary layout changes
# =====================================================================================================
class ToNHWCFn(torch.autograd.Function):
    """fp32 NCHW -> bf16 NHWC (+ batch statistics of the result)."""

    @staticmethod
    def forward(ctx, x, want_stats, events=1):
        N, C, Hh, Ww = x.shape
        x = x.contiguous().float()
        out = torch.empty(N, Hh, Ww, C, dtype=BF16, device=x.device)
        # one statistics slot per (image of the event, 32-pixel block): a single adder each (bit-reproducible sums)
        stats = new_stats(C, x.device, events, (N // events) * ((Hh * Ww + 31) // 32)) if want_stats else None
        H.call("ieagan_nchw_to_nhwc", x.data_ptr(), out.data_ptr(), H.ptr(stats), N, C, Hh * Ww, N // events, H.stream())
        ctx.events = events
        ctx.save_for_backward(out if want_stats else None)
        return out, stats

    @staticmethod
    def backward(ctx, dout, dstats):
        (out,) = ctx.saved_tensors
        g = dout.contiguous()
        N, Hh, Ww, C = g.shape
        if dstats is not None:
            geff = torch.empty_like(g)
            H.call("ieagan_effgrad", g.data_ptr(), out.data_ptr(), dstats[:, 0].contiguous().data_ptr(), geff.data_ptr(), None,
                   N * Hh * Ww, C, ctx.events, H.stream())
            g = geff
        dx = torch.empty(N, C, Hh, Ww, dtype=torch.float32, device=g.device)
        H.call("ieagan_nhwc_to_nchw", g.data_ptr(), dx.data_ptr(), N, C, Hh * Ww, H.stream())
        return dx, None, None


class ToNCHWFn(torch.autograd.Function):
    """bf16 NHWC -> fp32 NCHW."""

    @staticmethod
    def forward(ctx, x):
        N, Hh, Ww, C = x.shape
        out = torch.empty(N, C, Hh, Ww, dtype=torch.float32, device=x.device)
        H.call("ieagan_nhwc_to_nchw", x.contiguous().data_ptr(), out.data_ptr(), N, C, Hh * Ww, H.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Hh, Ww = g.shape
        dx = torch.empty(N, Hh, Ww, C, dtype=BF16, device=g.device)
        H.call("ieagan_nchw_to_nhwc", g.contiguous().float().data_ptr(), dx.data_ptr(), None, N, C, Hh * Ww, 0, H.stream())
        return dx


def channel_stats(x: torch.Tensor) -> torch.Tensor:
    """(sum, sumsq) of a bf16 NHWC tensor (no autograd; for tensors not produced by a conv)."""
    C = x.shape[-1]
    st = new_stats(C, x.device)[0]
    H.call("ieagan_channel_stats", x.data_ptr(), st.data_ptr(), x.numel() // C, C, H.stream())
    return st


# =====================================================================================================
# non-local attention core
# =====================================================================================================
class NLAttentionFn(torch.autograd.Function):
    """o = softmax(theta^T phi) g with a streaming softmax on MFMA (no [N, Lq, Lk] tensor; layers.py:291-299)."""

    @staticmethod
    def forward(ctx, q, k, v):
        N, Lq, dqk = q.shape
        Lk, dv = k.shape[1], v.shape[2]
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o = torch.empty(N, Lq, dv, dtype=BF16, device=q.device)
        lse = torch.empty(N, Lq, dtype=torch.float32, device=q.device)
        H.call("ieagan_nl_attention_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), N, Lq, Lk, dqk,
               dv, H.stream())
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        N, Lq, dqk = q.shape
        Lk, dv = k.shape[1], v.shape[2]
        do = do.contiguous()
        delta = torch.empty(N, Lq, dtype=torch.float32, device=q.device)
        dq, dk, dvv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        H.call("ieagan_nl_attention_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
               delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(), N, Lq, Lk, dqk, dv, H.stream())
        return dq, dk, dvv


# =====================================================================================================
# small fused kernels: RRM attention core, loss block, D-head pooling
# =====================================================================================================
class RRMAttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v per (batch, head) on the packed projection [B,S,H*3*hd] -> [B,S,H*hd];
    the S x S affinity lives in LDS (RRM.py:10-16, 46-58)."""

    @staticmethod
    def forward(ctx, qkv, num_heads):
        B, S, E3 = qkv.shape
        hd = E3 // (3 * num_heads)
        qkv = qkv.contiguous().float()
        out = torch.empty(B, S, num_heads * hd, dtype=torch.float32, device=qkv.device)
        att = torch.empty(B, num_heads, S, S, dtype=torch.float32, device=qkv.device)
        H.call("ieagan_rrm_attention_fwd", qkv.data_ptr(), out.data_ptr(), att.data_ptr(), B, S, num_heads, hd, H.stream())
        ctx.num_heads = num_heads
        ctx.save_for_backward(qkv, att)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, dout, _datt):
        qkv, att = ctx.saved_tensors
        B, S, E3 = qkv.shape
        hd = E3 // (3 * ctx.num_heads)
        dqkv = torch.empty_like(qkv)
        H.call("ieagan_rrm_attention_bwd", qkv.data_ptr(), att.data_ptr(), dout.contiguous().float().data_ptr(), dqkv.data_ptr(), B, S,
               ctx.num_heads, hd, H.stream())
        return dqkv, None


def _slin_fwd(x, w, b, res, ln_w=None, ln_b=None, relu=False, eps=1e-5):
    """y = [relu](LN?(x) @ w^T + b) [+ res] in one launch (csrc/rrm_fused.hip); returns (y, xhat, rstd)."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    xhat = rstd = None
    if ln_w is not None:
        xhat = torch.empty(M, K, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    H.call("ieagan_slin_fwd", x.data_ptr(), w.data_ptr(), H.ptr(b), H.ptr(res), y.data_ptr(), H.ptr(ln_w), H.ptr(ln_b), H.ptr(xhat), H.ptr(rstd),
           M, K, N, int(relu), float(eps), H.stream())
    return y, xhat, rstd


def _slin_bwd(dy, w, xn=None, xhat=None, ln_w=None, ln_b=None, ymask=None, want_dx=True, want_dw=True, want_db=True):
    return _slin_bwd_impl(dy, w, xn, xhat, ln_w, ln_b, ymask, want_dx, want_dw, want_db)


def _slin_bwd_impl(dy, w, xn, xhat, ln_w, ln_b, ymask, want_dx, want_dw, want_db):
    """(dx, dw, db) of one linear layer in one launch: dx = dy' w, dw = dy'^T xn, db = column sums of dy' (dy' = dy masked where
    ``ymask`` <= 0; xn = the GEMM input, or xhat * ln_w + ln_b)."""
    M, N = dy.shape
    K = w.shape[1]
    dev = dy.device
    split = want_dx and N >= 2048 and N % 16 == 0          # long reduction over n: partial tiles are added into a zeroed dx
    dx = (zeros((M, K), dev) if split else torch.empty(M, K, dtype=torch.float32, device=dev)) if want_dx else None
    dw = torch.empty(N, K, dtype=torch.float32, device=dev) if want_dw else None
    db = torch.empty(N, dtype=torch.float32, device=dev) if want_db else None
    if dx is None and dw is None and db is None:
        return None, None, None
    H.call("ieagan_slin_bwd", dy.data_ptr(), H.ptr(ymask), H.ptr(xn), H.ptr(xhat), H.ptr(ln_w), H.ptr(ln_b), w.data_ptr(), H.ptr(dx), H.ptr(dw),
           H.ptr(db), M, K, N, int(split), H.stream())
    return dx, dw, db


def _ln_bwd(dy, xhat, rstd, w, dres=None, want_param=True, l2_beta=None):
    """LayerNorm backward (+ the residual-path gradient ``dres``); ``l2_beta``: the forward ended with F.normalize."""
    M, K = dy.shape
    dx = torch.empty_like(dy)
    dg = zeros((K,), dy.device) if want_param else None
    dbeta = zeros((K,), dy.device) if want_param else None
    H.call("ieagan_ln_bwd", dy.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), w.data_ptr(), H.ptr(l2_beta), H.ptr(dres), dx.data_ptr(), H.ptr(dg),
           H.ptr(dbeta), M, K, H.stream())
    return dx, dg, dbeta


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of [M, K] rows in one launch; ``l2norm``: followed by F.normalize(., dim=1) (the
    discriminator's embedding head, model.py:920-935)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, l2norm=False):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous().float()
        M, K = x2.shape
        y = torch.empty_like(x2)
        xhat = torch.empty_like(x2)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        H.call("ieagan_ln_fwd", x2.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), M, K, float(eps),
               int(l2norm), H.stream())
        ctx.shape, ctx.l2norm = shape, l2norm
        ctx.save_for_backward(xhat, rstd, w, b)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, w, b = ctx.saved_tensors
        g = dy.reshape(-1, ctx.shape[-1]).contiguous().float()
        need = ctx.needs_input_grad
        dx, dg, dbeta = _ln_bwd(g, xhat, rstd, w, None, need[1] or need[2], b if ctx.l2norm else None)
        return dx.view(ctx.shape), dg, dbeta, None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for a small fp32 linear layer (nn.Linear / SNLinear of the G entry and the D head, reference model.py:467, 475,
    915, 920) in one launch each way; ``rec``: the layer's SNRecord (the GEMM then takes W / sigma of this pass and the weight
    gradient goes through the spectral-norm backward) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, rec):
        w_eff = weight if rec is None else rec.w_plain.view(weight.shape)
        x2 = x.contiguous().float()
        y, _, _ = _slin_fwd(x2, w_eff, bias, None)
        ctx.rec = rec
        ctx.save_for_backward(x2, w_eff, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        x2, w_eff, weight = ctx.saved_tensors
        need = ctx.needs_input_grad
        dx, dw, db = _slin_bwd(g.contiguous().float(), w_eff, xn=x2, want_dx=need[0], want_dw=need[1], want_db=need[2])
        if dw is not None and ctx.rec is not None:
            dw = sn_backward(dw, weight, ctx.rec)[0]
        return dx, dw, db, None


class EmbedNormFn(torch.autograd.Function):
    """F.normalize(SNEmbedding(y), dim=1): the unit-sphere class proxies of the discriminator head (model.py:916, 933)."""

    @staticmethod
    def forward(ctx, idx, weight, rec):
        wn = rec.w_plain.view(weight.shape)
        M, D = idx.numel(), weight.shape[1]
        idx = idx.contiguous().long()
        out = torch.empty(M, D, dtype=torch.float32, device=weight.device)
        inv = torch.empty(M, dtype=torch.float32, device=weight.device)
        H.call("ieagan_embed_norm_fwd", idx.data_ptr(), wn.data_ptr(), out.data_ptr(), inv.data_ptr(), M, D, H.stream())
        ctx.rec = rec
        ctx.save_for_backward(idx, out, inv, weight)
        return out

    @staticmethod
    def backward(ctx, dp):
        idx, out, inv, weight = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None
        gsn = zeros(tuple(weight.shape), weight.device)
        H.call("ieagan_embed_norm_bwd", idx.data_ptr(), out.data_ptr(), inv.data_ptr(), dp.contiguous().float().data_ptr(), gsn.data_ptr(),
               idx.numel(), weight.shape[1], H.stream())
        return None, sn_backward(gsn, weight, ctx.rec)[0], None


class RRMBlockFn(torch.autograd.Function):
    """One pre-LN encoder block of the Relational Reasoning Module (reference RRM.py:66-109) on [B, S <= 64, E] tokens, stage-wise
    fused: [LN1 + qkv] -> [attention core] -> [o proj + residual] -> [LN2 + FFN1 + ReLU] -> [FFN2 + residual]: five launches
    forward, seven backward (+ one per spectrally normalised weight).  ``params`` = (norm1.w, norm1.b, qkv.w, qkv.b, o.w, o.b,
    norm2.w, norm2.b, ffn1.w, ffn1.b, ffn2.w, ffn2.b); ``recs``: the four SNRecords of the linear layers (SNLinear flavour, D) or
    None (nn.Linear, G)."""

    @staticmethod
    def forward(ctx, x, heads, eps, recs, *params):
        n1w, n1b, wq, bq, wo, bo, n2w, n2b, w1, b1, w2, b2 = params
        B, S, E = x.shape
        M = B * S
        x2d = x.reshape(M, E).contiguous().float()
        eff = [wq, wo, w1, w2] if recs is None else [r.w_plain.view(w.shape) for r, w in zip(recs, (wq, wo, w1, w2))]
        qkv, xhat1, rstd1 = _slin_fwd(x2d, eff[0], bq, None, n1w, n1b, False, eps)
        hd = E // heads
        vals = torch.empty(M, E, dtype=torch.float32, device=x.device)
        att = torch.empty(B, heads, S, S, dtype=torch.float32, device=x.device)
        H.call("ieagan_rrm_attention_fwd", qkv.data_ptr(), vals.data_ptr(), att.data_ptr(), B, S, heads, hd, H.stream())
        xa, _, _ = _slin_fwd(vals, eff[1], bo, x2d)
        h, xhat2, rstd2 = _slin_fwd(xa, eff[2], b1, None, n2w, n2b, True, eps)
        out, _, _ = _slin_fwd(h, eff[3], b2, xa)
        ctx.recs, ctx.cfg = recs, (B, S, E, heads, hd)
        ctx.save_for_backward(xhat1, rstd1, qkv, att, vals, xhat2, rstd2, h, *eff, *params)
        return out.view(B, S, E)

    @staticmethod
    def backward(ctx, dout):
        xhat1, rstd1, qkv, att, vals, xhat2, rstd2, h, eq, eo, e1, e2, *params = ctx.saved_tensors
        n1w, n1b, wq, bq, wo, bo, n2w, n2b, w1, b1, w2, b2 = params
        B, S, E, heads, hd = ctx.cfg
        M = B * S
        need = ctx.needs_input_grad[4:]               # aligned with ``params``
        g = dout.reshape(M, E).contiguous().float()
        dh, dw2, db2 = _slin_bwd(g, e2, xn=h, want_dw=need[10], want_db=need[11])
        dxn2, dw1, db1 = _slin_bwd(dh, e1, xhat=xhat2, ln_w=n2w, ln_b=n2b, ymask=h, want_dw=need[8], want_db=need[9])
        dxa, dn2w, dn2b = _ln_bwd(dxn2, xhat2, rstd2, n2w, g, need[6] or need[7])
        dvals, dwo, dbo = _slin_bwd(dxa, eo, xn=vals, want_dw=need[4], want_db=need[5])
        dqkv = torch.empty_like(qkv)
        H.call("ieagan_rrm_attention_bwd", qkv.data_ptr(), att.data_ptr(), dvals.data_ptr(), dqkv.data_ptr(), B, S, heads, hd, H.stream())
        dxn1, dwq, dbq = _slin_bwd(dqkv, eq, xhat=xhat1, ln_w=n1w, ln_b=n1b, want_dw=need[2], want_db=need[3])
        dx, dn1w, dn1b = _ln_bwd(dxn1, xhat1, rstd1, n1w, dxa, need[0] or need[1])
        dws = [dwq, dwo, dw1, dw2]
        if ctx.recs is not None:                      # gradient of W / sigma -> gradient of W (incl. the sigma term), one launch per layer
            for k, (r, w) in enumerate(zip(ctx.recs, (wq, wo, w1, w2))):
                if dws[k] is not None:
                    dws[k] = sn_backward(dws[k], w, r)[0]
        return (dx.view(B, S, E), None, None, None, dn1w, dn1b, dws[0], dbq, dws[1], dbo, dn2w, dn2b, dws[2], db1, dws[3], db2)


class LossBlockFn(torch.autograd.Function):
    """All losses of one phase, value and gradient, in one launch (loss.py:8-44, 79-132).
    Returns (total, terms[8]); only ``total`` is differentiable."""

    @staticmethod
    def forward(ctx, dfake, dreal, e, p, er, weights, temperature, events=1):
        import ctypes
        ref = next(t for t in (dfake, dreal, e) if t is not None)
        dev = ref.device
        if ref.shape[0] % events != 0:
            raise ValueError(f"loss_block: {ref.shape[0]} rows are not {events} whole events")
        n = ref.shape[0] // events
        d = e.shape[1] if e is not None else 0
        c = lambda t: None if t is None else t.contiguous().float()
        dfake, dreal, e, p, er = c(dfake), c(dreal), c(e), c(p), c(er)
        vals = torch.empty(events, 8, dtype=torch.float32, device=dev)
        g_df = torch.empty_like(dfake) if dfake is not None else None
        g_dr = torch.empty_like(dreal) if dreal is not None else None
        g_e = torch.empty_like(e) if e is not None else None
        g_p = torch.empty_like(p) if p is not None else None
        w = (ctypes.c_float * 6)(*[float(x) for x in weights])
        # one workgroup per event (the Grams / hinge means are intra-event); total = mean over the events
        H.call("ieagan_loss_block_events", H.ptr(dfake), H.ptr(dreal), H.ptr(e), H.ptr(p), H.ptr(er), w, float(temperature),
               vals.data_ptr(), H.ptr(g_df), H.ptr(g_dr), H.ptr(g_e), H.ptr(g_p), n, d, int(events), H.stream())
        ctx.save_for_backward(g_df, g_dr, g_e, g_p)
        ctx.events = int(events)
        terms = vals.mean(0) if events > 1 else vals[0].clone()
        ctx.mark_non_differentiable(terms)
        return (vals[:, 0].mean() if events > 1 else vals[0, 0]), terms

    @staticmethod
    def backward(ctx, gtotal, _gterms):
        g_df, g_dr, g_e, g_p = ctx.saved_tensors
        sc = gtotal / float(ctx.events)
        m = lambda t: None if t is None else t * sc
        return m(g_df), m(g_dr), m(g_e), m(g_p), None, None, None, None


def loss_block(dfake=None, dreal=None, e=None, p=None, er=None, w_hinge_real=0.0, w_hinge_fake=0.0, w_hinge_gen=0.0,
               w_contra=0.0, w_unif=0.0, w_iea=0.0, temperature=1.0, events=1):
    """(total, terms) with terms = [total, hinge_real, hinge_fake, hinge_gen, contrastive, uniformity, iea, 0].  ``events`` > 1: the
    tensors hold that many events of equal size; every term is evaluated per event (one workgroup each, one launch) and averaged."""
    return LossBlockFn.apply(dfake, dreal, e, p, er, (w_hinge_real, w_hinge_fake, w_hinge_gen, w_contra, w_unif, w_iea), temperature, events)


class ReluSumPoolFn(torch.autograd.Function):
    """D head: sum over (h, w) of relu(x), bf16 NHWC -> fp32 [N, C]  (model.py:912)."""

    @staticmethod
    def forward(ctx, x):
        N, Hh, Ww, C = x.shape
        out = torch.empty(N, C, dtype=torch.float32, device=x.device)
        H.call("ieagan_relu_sum_pool", x.data_ptr(), out.data_ptr(), N, Hh * Ww, C, H.stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dh):
        (x,) = ctx.saved_tensors
        N, Hh, Ww, C = x.shape
        dx = torch.empty_like(x)
        H.call("ieagan_relu_sum_pool_bwd", x.data_ptr(), dh.contiguous().float().data_ptr(), dx.data_ptr(), N, Hh * Ww, C, H.stream())
        return dx


class MaxPool2Fn(torch.autograd.Function):
    """F.max_pool2d(., [2, 2]) on a bf16 NHWC map (non-local block, layers.py:288-289)."""

    @staticmethod
    def forward(ctx, x):
        N, Hh, Ww, C = x.shape
        x = x.contiguous()
        out = torch.empty(N, Hh // 2, Ww // 2, C, dtype=x.dtype, device=x.device)
        idx = torch.empty(N, Hh // 2, Ww // 2, C, dtype=torch.uint8, device=x.device)
        H.call("ieagan_maxpool2_fwd", x.data_ptr(), out.data_ptr(), idx.data_ptr(), N, Hh, Ww, C, H.stream())
        ctx.save_for_backward(idx)
        ctx.shape = (N, Hh, Ww, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        N, Hh, Ww, C = ctx.shape
        dout = dout.contiguous()
        dx = torch.empty(N, Hh, Ww, C, dtype=dout.dtype, device=dout.device)
        H.call("ieagan_maxpool2_bwd", dout.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, Hh, Ww, C, H.stream())
        return dx


class GammaResidualFn(torch.autograd.Function):
    """out = gamma * o + x on bf16 maps with the learnable scalar read from device memory (layers.py:300)."""

    @staticmethod
    def forward(ctx, o, x, gamma, link=None):
        o, x = o.contiguous(), x.contiguous()
        out = torch.empty_like(x)
        H.call("ieagan_gamma_residual_fwd", o.data_ptr(), x.data_ptr(), gamma.data_ptr(), out.data_ptr(), x.numel(), H.stream())
        ctx.save_for_backward(o, gamma)
        ctx.link = link
        return out

    @staticmethod
    def backward(ctx, d):
        o, gamma = ctx.saved_tensors
        d = d.contiguous()
        d_o = torch.empty_like(o)
        part = zeros((STAT_REPL,), o.device)
        H.call("ieagan_gamma_residual_bwd", d.data_ptr(), o.data_ptr(), gamma.data_ptr(), d_o.data_ptr(), part.data_ptr(), o.numel(),
               H.stream())
        d_x = d
        if ctx.link is not None:      # fan-in of the block input's gradients (SumLink): the residual path's share is d itself
            if ctx.link.ready:        # (normally this node runs first and only deposits)
                d_x = d + ctx.link.take()[0]
            d_x = ctx.link.arrive(d_x, d.shape[-1])
        return d_o, d_x, part.sum().reshape(gamma.shape), None


# =====================================================================================================
# augmentation
# =====================================================================================================
class DiffAugFn(torch.autograd.Function):
    """DiffAugment('color,translation,cutout') on single-channel fp32 events (diff_aug.py:10-109)."""

    @staticmethod
    def forward(ctx, x, bright, contrast, tx, ty, ox, oy):
        N, C, Hh, Ww = x.shape
        x = x.contiguous()
        out = torch.empty_like(x)
        sums = zeros((N, H.AUG_SLOTS), x.device)
        H.call("ieagan_diffaug_fwd", x.data_ptr(), bright.data_ptr(), contrast.data_ptr(), tx.data_ptr(), ty.data_ptr(),
               ox.data_ptr(), oy.data_ptr(), sums.data_ptr(), out.data_ptr(), N, Hh, Ww, H.stream())
        ctx.save_for_backward(contrast, tx, ty, ox, oy)
        return out

    @staticmethod
    def backward(ctx, g):
        contrast, tx, ty, ox, oy = ctx.saved_tensors
        N, C, Hh, Ww = g.shape
        g = g.contiguous()
        gx = torch.empty_like(g)
        gs = zeros((N, H.AUG_SLOTS), g.device)
        H.call("ieagan_diffaug_bwd", g.data_ptr(), contrast.data_ptr(), tx.data_ptr(), ty.data_ptr(), ox.data_ptr(),
               oy.data_ptr(), gs.data_ptr(), gx.data_ptr(), N, Hh, Ww, H.stream())
        return gx, None, None, None, None, None, None


def cr_diffaug(x, flip_u, tx, ty):
    N, C, Hh, Ww = x.shape
    x = x.contiguous()
    out = torch.empty_like(x)
    H.call("ieagan_cr_diffaug", x.data_ptr(), flip_u.data_ptr(), tx.data_ptr(), ty.data_ptr(), out.data_ptr(), N, Hh, Ww,
           H.stream())
    return out
