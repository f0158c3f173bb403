"""The IEA-GAN train step (one D update + one G update) for MI355X.

Entry point kept from reference ``train_fns.py:20-206``:
``GAN_training_function(G, D, GD, z_, y_, ema, state_dict, config, device) -> train(x, y) -> dict`` with the
keys ``G_loss, D_loss_real, D_loss_fake, unif_loss_d, iea_loss``.  Phases, loss composition, call order of
the fake / real discriminator passes and the optimiser placement follow the reference, including its quirk
that ``G.optim.step()`` only runs when ``clip_norm`` is not None.  Differences, all in
places where the reference raises: loss terms that are switched off are reported as 0.0 instead of an
UnboundLocalError (9-Q2), and consistency regularisation with ``split_D`` runs a third discriminator pass on
the augmented real event (9-Q3).  The five losses are read back with ONE host synchronisation.

``config['events_per_step']`` = E > 1 (BASELINE configs[3]; the reference consumes exactly one event per step, SURVEY
9-Q5): ``x`` / ``y`` hold E events of ``batch_size`` sensors, batched on the leading dimension through every kernel --
BatchNorm statistics, the RRM tokens and the loss Grams stay per event, gradients are the mean over the events, the
BatchNorm running statistics receive the mean of the E per-event momentum updates.

Execution modes
* eager: every launch issued from Python (reference behaviour, needed for explicit-noise parity tests);
* ``config['hip_graph']``, one GPU: the WHOLE step is captured once and replayed as one HIP graph;
* data parallel (``parallel.GradSync`` context): D's gradient all-reduce + Adam run on a side stream; with
  ``hip_graph`` the step is replayed as three graphs (D forward/backward | G forward/backward | G update) with
  the RCCL all-reduces issued eagerly between them -- collectives are never captured.
"""
from __future__ import annotations

import torch

import loss
import ops
import parallel
import utils
from cr_diff_aug import CR_DiffAug

KEYS = ("G_loss", "D_loss_real", "D_loss_fake", "unif_loss_d", "iea_loss")


def dummy_training_function():
    def train(x, y):
        return {}
    return train


def _clip(net, max_norm):
    """clip_grad_norm_ on the flat gradient arena, without a host round trip."""
    ar = net.__dict__.get("_arena")
    if ar is not None and ar.grads_attached():
        total = torch.linalg.vector_norm(ar.grad)
        ar.grad.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    else:
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)


class _Replay:
    """Capture a zero-argument callable once (after ``warmup`` eager calls) and replay it as a HIP graph."""

    def __init__(self, fn, warmup=2):
        self.fn, self.warmup, self.calls, self.graph, self.out = fn, warmup, 0, None, None

    def __call__(self):
        if self.graph is None:
            if self.calls < self.warmup:
                self.calls += 1
                return self.fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            ops.reset_zero_pool()          # every zero-filled scratch chunk must be filled INSIDE the graph ...
            # thread_local: the RCCL watchdog thread polls its events while we capture; only this thread's calls are checked
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.out = self.fn()
            ops.reset_zero_pool()          # ... and eager code must never carve from graph-owned memory
            self.graph = g
        self.graph.replay()
        return self.out

    def reset(self):
        """Forget the captured graph (its static inputs were re-allocated): warm up and capture again."""
        self.calls, self.graph, self.out = 0, None, None


def test(G, D, G_ema, state_dict, config, test_log):
    """FID bookkeeping of the reference (train_fns.py:209-233): the metric itself needs the PXD Inception network of
    ``mycleanfid`` (weights not distributed with the reference) and is outside the accelerated path.  When
    a reference checkout with ``mycleanfid`` is importable the reference's call is made with this package's generator;
    otherwise the iteration is logged without a score instead of aborting the training run."""
    try:
        from mycleanfid import fid
    except Exception as e:            # cleanfid / cv2 / the Inception blob are missing
        print(f"FID skipped at itr {state_dict['itr']}: mycleanfid is not importable ({type(e).__name__}: {e})")
        test_log.log(itr=int(state_dict["itr"]), FID=None)
        return
    print("Gathering inception metrics...")
    FID = fid.compute_fid(gen=G, dataset_name="pxd_sim_test_com", dataset_res=256, batch_size=40, mode="clean",
                          dataset_split="custom", z_dim=128, num_gen=config["num_incep_images"], trunc=None,
                          device=config["device"])
    print(f"The FID score is {FID}")
    if config["which_best"] == "FID" and FID < state_dict["best_FID"]:
        print("%s improved over previous best, saving checkpoint..." % config["which_best"])
        state_dict["save_best_num"] = (state_dict["save_best_num"] + 1) % config["num_best_copies"]
    state_dict["best_FID"] = min(state_dict["best_FID"], FID)
    test_log.log(itr=int(state_dict["itr"]), FID=float(FID))


def GAN_training_function(G, D, GD, z_, y_, ema, state_dict, config, device):
    if config["pos_collected_numerator"]:
        raise NotImplementedError("pos_collected_numerator=True is not part of the MI355X path (reference default: False)")
    bs = config["batch_size"]
    if config.get("conv_dtype", "bf16") not in ("bf16", "fp8"):
        raise NotImplementedError("conv_dtype must be 'bf16' or 'fp8'")
    for net in (G, D):          # the operand precision is a property of the NETWORKS (per-layer descriptor flags), set at construction
        if getattr(net, "conv_dtype", "bf16") != config.get("conv_dtype", "bf16"):
            raise ValueError(f"config['conv_dtype'] = {config.get('conv_dtype', 'bf16')!r} but the network was built with "
                             f"{getattr(net, 'conv_dtype', 'bf16')!r}: pass the same config to Generator / Discriminator")
    E = max(int(config.get("events_per_step", 1) or 1), 1)
    if E > 1 and not (config["split_D"] and config["num_D_accumulations"] == 1 and config["num_G_accumulations"] == 1):
        raise NotImplementedError("events_per_step > 1 needs split_D and no gradient accumulation: the E events of a step are "
                                  "batched on the leading dimension (per-event BatchNorm / RRM / loss Grams), see DESIGN section 7")
    chunk = bs * E                  # images one pass consumes: E events of batch_size sensors each
    contra = config["conditional_strategy"] == "Contra"
    t = 1.0
    sync = parallel.get_context()
    if sync is not None:
        # the main stream must not evaluate a network before its (side-stream) update has landed
        D.register_forward_pre_hook(lambda m, inp: sync.wait("D"))
        G.register_forward_pre_hook(lambda m, inp: sync.wait("G"))
    st = {"x": None, "y": None, "noise": None, "emb_real": None, "counter": 0}     # state shared by the phases

    def explicit(key):
        """Explicit draws of a parity test: one dict per step, or a list of E per-event dicts (concatenated in event order)."""
        noise = st["noise"]
        if isinstance(noise, (list, tuple)):
            vals = [n[key] for n in noise]
            if isinstance(vals[0], dict):
                return {k: torch.cat([v[k] for v in vals]) for k in vals[0]}
            return torch.cat(vals)
        return noise[key]

    def sample(phase):
        import diff_aug as _da
        if st["noise"] is None:
            z_.sample_()
            return
        z_.copy_(explicit("z_" + phase).to(z_.device))
        G.__dict__["_next_rdof"] = explicit("rdof_" + phase).to(z_.device)
        if config["diff_aug"]:
            _da.NEXT_DRAWS.append(explicit("aug_" + phase))

    def per_event(**kw):
        """The fused loss block evaluated per event (the contrastive / uniformity / IEA Grams are intra-event) and averaged:
        (total, terms).  One launch, one workgroup per event."""
        return ops.loss_block(**kw, events=E)

    # Spectral-norm passes ahead of time on a side stream (ops.SNBank.prefetch): 4 of the 5 passes of a step depend on weights
    # that are final long before the forward that consumes them.  Single-GPU default step only (split_D, no accumulation): the
    # number of forward passes per phase is then known -- G: 1 + 1, D: 2 (+1 with Con_reg) + 1.
    sn_known = bool(config.get("sn_prefetch", True)) and config["split_D"] and config["num_D_steps"] == 1 and \
        config["num_D_accumulations"] == 1 and config["num_G_accumulations"] == 1 and contra
    sn_ahead = sn_known and sync is None
    # Data parallel: the step is three graphs with eager all-reduces between them, and a pass issued in one graph cannot be handed to
    # the next -- only the passes that are issued AND consumed inside the D phase go ahead (3 of the step's 5).
    sn_local = sn_known and sync is not None
    # Data parallel: D(x_real) is evaluated BEFORE G(z) in the D phase (GD(real_first=True)), and only then does the main stream wait
    # for G's side-stream all-reduce + ortho + Adam + EMA of the previous step -- they run under the real pass instead of in front of
    # the step.  Deviation from the reference's fake-then-real order: the two passes see swapped spectral-norm iterates
    # (tolerance level, 9-Q6); single-GPU runs keep the reference order.
    real_first = sync is not None and bool(config.get("dp_real_first", True)) and config["split_D"] and config["num_D_accumulations"] == 1

    def prefetch_sn(net, passes, local=False):
        if not (sn_ahead or (local and sn_local)) or not next(net.parameters()).is_cuda:
            return
        if st.get("sn_stream") is None:
            st["sn_stream"] = torch.cuda.Stream()
        bank = net._prepare()["bank"]
        for _ in range(passes):
            bank.prefetch(net.training, net.SN_eps, st["sn_stream"])

    # ------------------------------------------------------------------------------------------------ D phase
    def check_shapes(x, y):
        """One pass consumes ``chunk`` = batch_size * events_per_step images (the reference: one event); ``x`` holds one
        chunk per G accumulation (``torch.split(x, chunk)`` indexed by the G-phase counter, train_fns.py:34-37, 182).  A short batch
        would silently run ``loss_block`` on empty per-event slices: refuse it here."""
        n_acc = max(int(config["num_G_accumulations"]), 1)
        if x.shape[0] % chunk != 0 or x.shape[0] // chunk < n_acc or y.shape[0] != x.shape[0]:
            raise ValueError(f"train(x, y): expected a multiple (>= {n_acc}) of {chunk} images and as many labels per step "
                             f"(batch_size {bs} x events_per_step {E}), got x {tuple(x.shape)}, y {tuple(y.shape)}")
        if z_.shape[0] < chunk:
            raise ValueError(f"z_ holds {z_.shape[0]} latent rows, one pass needs {chunk} (prepare_z_y(batch_size * events_per_step, ...))")

    def d_real_forward():
        """Data parallel, real_first: D(x_real) alone -- nothing here touches G (whose update may still be running on the side
        stream).  Returns a dummy tensor; the outputs (with their autograd tape) wait in ``st['real_out']``."""
        x, y = st["x"], st["y"]
        prefetch_sn(D, 1, local=True)
        D.optim.zero_grad()
        if config["toggle_grads"]:
            utils.toggle_grad(D, True)
        st["real_out"] = D(torch.split(x, chunk)[0], torch.split(y, chunk)[0])
        return torch.zeros((), device=x.device)

    def d_forward_backward():
        """Zero both gradient arenas, accumulate D's gradient (train_fns.py:24-130); returns [real, fake, unif_d]."""
        x, y = st["x"], st["y"]
        real_out = st.pop("real_out", None) if real_first else None
        if real_first and real_out is None:     # not segmented: the real pass first, in here
            d_real_forward()
            real_out = st.pop("real_out")
        if real_first:
            sync.wait("G")                      # G's exchange + update (side stream) ran under the real pass: join before G is touched
        if sn_local:
            prefetch_sn(G, 1, local=True)       # this phase's generator pass
        else:
            prefetch_sn(G, 2)                   # both generator passes of the step: G's weights only change in g_update
        if real_first:
            prefetch_sn(D, 2 if config["Con_reg"] else 1, local=True)      # D(fake)[, D(real_aug)]
        else:
            prefetch_sn(D, 3 if config["Con_reg"] else 2, local=True)      # D(fake), D(real)[, D(real_aug)]: D's weights change in d_update
        G.optim.zero_grad()
        if not real_first:
            D.optim.zero_grad()
        x_aug = None
        if config["Con_reg"]:
            x_aug = CR_DiffAug(x, draws=explicit("cr") if st["noise"] is not None else None)
        xs, ys = torch.split(x, chunk), torch.split(y, chunk)
        xa = torch.split(x_aug, chunk) if x_aug is not None else None
        c = st["counter"] = 0
        zero = torch.zeros((), device=x.device)
        if config["toggle_grads"]:
            utils.toggle_grad(D, True)
            utils.toggle_grad(G, False)
        D_loss_real = D_loss_fake = unif_loss_d = zero
        for _ in range(config["num_D_accumulations"]):
            sample("d")
            joint_aug = config["Con_reg"] and not config["split_D"]
            outs = GD(z_[:chunk], ys[c], xs[c], ys[c], xa[c] if joint_aug else None, contra=contra, train_G=False,
                      split_D=config["split_D"], diff_aug=config["diff_aug"], real_out=real_out)
            aug_out = None
            if contra:
                if len(outs) == 8:
                    *outs, emb_aug, D_real_aug = outs
                    aug_out = (emb_aug, D_real_aug)
                _, _, D_fake, proxy_real, emb_real, D_real = outs
                if config["Con_reg"] and aug_out is None:          # split_D: third pass on the augmented event
                    _, emb_aug, D_real_aug = D(xa[c], ys[c])
                    aug_out = (emb_aug, D_real_aug)
                st["emb_real"] = emb_real.detach()      # detached: must not keep the D-phase autograd graph alive across steps
            else:
                proxy_real = emb_real = None
                if len(outs) == 3:
                    D_fake, D_real, D_real_aug = outs
                    aug_out = (None, D_real_aug)
                else:
                    D_fake, D_real = outs
                    if config["Con_reg"]:
                        aug_out = (None, D(xa[c], ys[c]))
            # every D-phase loss term in ONE fused launch (value + gradient): hinge real / fake, 2C, uniformity
            use_c = contra and config["contra_lambda"] != 0
            use_u = contra and bool(config["Uniformity_loss"])
            D_loss, terms = per_event(dfake=D_fake, dreal=D_real, e=emb_real if (use_c or use_u) else None,
                                      p=proxy_real if use_c else None, w_hinge_real=1.0, w_hinge_fake=1.0,
                                      w_contra=config["contra_lambda"] if use_c else 0.0,
                                      w_unif=config["unif_lambda"] if use_u else 0.0, temperature=t)
            D_loss_real, D_loss_fake = terms[1], terms[2]
            if use_u:
                unif_loss_d = terms[5]
            if aug_out is not None:
                consistency = loss.l2_loss(D_real, aug_out[1])
                if aug_out[0] is not None:
                    consistency = consistency + loss.l2_loss(emb_real, aug_out[0])
                D_loss = D_loss + config["cr_lambda"] * consistency
            with ops.direct_grads():
                (D_loss / float(config["num_D_accumulations"])).backward()
        return torch.stack([D_loss_real.detach().reshape(()), D_loss_fake.detach().reshape(()), unif_loss_d.detach().reshape(())])

    def d_update():
        if config["D_ortho"] > 0.0:
            utils.ortho(D, config["D_ortho"])
        if config["clip_norm"] is not None:
            _clip(D, config["clip_norm"])
        D.optim.step()

    # ------------------------------------------------------------------------------------------------ G phase
    def g_forward_backward():
        """train_fns.py:142-182; returns [G_loss, iea]."""
        ys = torch.split(st["y"], chunk)
        zero = torch.zeros((), device=st["y"].device)
        if config["toggle_grads"]:
            utils.toggle_grad(D, False)
            utils.toggle_grad(G, True)
        G.optim.zero_grad()
        if sn_local:
            sync.wait("D")                      # D's update (side stream) precedes the spectral-norm pass of the D evaluation below,
            prefetch_sn(D, 1, local=True)       # which then runs under the generator forward
        iea_loss = zero
        for _ in range(config["num_G_accumulations"]):
            c = st["counter"]
            sample("g")
            if contra:
                proxy_fake, emb_fake, D_fake = GD(z_, ys[c], x_aug=None, contra=True, train_G=True,
                                                  split_D=config["split_D"], diff_aug=config["diff_aug"])
                use_c = config["contra_lambda"] != 0
                use_i = bool(config["IEA_loss"])
                use_u = use_i and bool(config["Uniformity_loss"])      # nested under IEA_loss, as in the reference (:171-178)
                need_e = use_c or use_i or use_u
                G_loss, terms = per_event(dfake=D_fake, e=emb_fake if need_e else None, p=proxy_fake if use_c else None,
                                          er=st["emb_real"] if use_i else None, w_hinge_gen=1.0,
                                          w_contra=config["contra_lambda"] if use_c else 0.0,
                                          w_unif=config["unif_lambda"] if use_u else 0.0,
                                          w_iea=config["IEA_lambda"] if use_i else 0.0, temperature=t)
                if use_i:
                    iea_loss = terms[6]
            else:
                D_fake = GD(z_, y_, x_aug=None, contra=False, train_G=True, split_D=config["split_D"],
                            diff_aug=config["diff_aug"])
                G_loss = loss.loss_hinge_gen(D_fake.reshape(-1))
            G_loss = G_loss / float(config["num_G_accumulations"])
            with ops.direct_grads():
                G_loss.backward()
            st["counter"] = c + 1
        return torch.stack([G_loss.detach().reshape(()), iea_loss.detach().reshape(())])

    def g_update():
        """ortho reg, (clip + Adam only when clip_norm is not None -- reference quirk 9-Q1), EMA."""
        if config["G_ortho"] > 0.0:
            utils.ortho(G, config["G_ortho"], blacklist=[p for p in G.shared.parameters()])
        if config["clip_norm"] is not None:
            _clip(G, config["clip_norm"])
            G.optim.step()
        if config["ema"]:
            ema.update(state_dict["itr"])
        return torch.zeros((), device=st["y"].device)

    def reduce(net, key, then=None):
        """Data parallel: average the flat gradient arena over the ranks on the side stream, ``then`` (the update) behind
        it.  D's exchange + Adam overlap the G-phase generator forward; G's exchange + ortho + Adam + EMA overlap whatever
        the main stream and the host do until the generator is evaluated again (ingestion of the next event, loss
        read-back, logging) -- the main stream re-joins in ``sync.wait`` at the top of the next step."""
        if sync is not None:
            sync.reduce_then(key, net._arena.grad, then)
        elif then is not None:
            then()

    def step_tensor(run_d, run_g, run_gu, replayed=False, run_dr=None):
        """[G_loss, D_loss_real, D_loss_fake, unif_loss_d, iea_loss] as one device tensor (no host sync)."""
        for _ in range(config["num_D_steps"]):
            if sync is not None:
                # D's previous all-reduce + Adam (side stream) still read the gradient arena that this iteration zeroes
                # first thing: order the main stream behind them here (eager: the forward pre-hook fires too late for
                # zero_grad(); replayed: a graph runs no Python hooks at all)
                sync.wait("D")
                if not real_first:
                    sync.wait("G")      # likewise G's update of the previous step (it zeroes / reads G's gradient arena)
                elif run_dr is not None:
                    run_dr()            # segmented replay: D(x_real) as its own graph, G's update still in flight beside it
                    sync.wait("G")
            dv = run_d()
            reduce(D, "D", d_update)
            if not replayed:
                prefetch_sn(D, 1)               # the G-phase discriminator pass sees the updated weights: overlaps G's forward
        if replayed and sync is not None:
            sync.wait("D")
        gv = run_g()
        reduce(G, "G", run_gu)  # the ortho term is deterministic in W: added after the gradient mean
        return torch.stack([gv[0], dv[0], dv[1], dv[2], gv[1]])

    def whole_step():
        try:
            return step_tensor(d_forward_backward, g_forward_backward, g_update)     # (real_first: d_forward_backward runs the real pass itself)
        except BaseException:
            # a step that ends early must not leave spectral-norm passes queued: the next forward would silently consume a pass
            # computed from older weights / an older u (every later step shifted by one pass)
            for net in (G, D):
                plan = getattr(net, "_plan", None)
                if plan is not None:
                    plan["bank"].discard_prefetched()
            raise

    use_graph = bool(config.get("hip_graph", False))
    whole = _Replay(whole_step)
    seg = (_Replay(d_forward_backward), _Replay(g_forward_backward), _Replay(g_update))
    seg_real = _Replay(d_real_forward) if real_first else None        # data parallel + graphs: D(x_real) is a graph of its own

    def push_device_state():
        for net in (G, D):
            net.optim.push_hyper()
        if config["ema"]:
            ema.push_decay(state_dict["itr"])

    def train(x, y, noise=None):
        """The reference's ``train(x, y) -> dict of 5 python floats`` (ONE host sync, at the end)."""
        check_shapes(x, y)
        if not use_graph or noise is not None:
            st["x"], st["y"], st["noise"] = x, y, noise
            vals = whole_step()
            st["noise"] = None
            return dict(zip(KEYS, vals.tolist()))
        if st.get("static") is None or st["x"].shape != x.shape or st["y"].shape != y.shape:
            st["x"], st["y"], st["static"] = x.clone(), y.clone(), True     # static inputs of the captured graphs
            for r in (whole, *seg, *([seg_real] if seg_real is not None else [])):     # graphs captured against the previous static buffers / shapes are stale
                r.reset()
        st["x"].copy_(x)
        st["y"].copy_(y)
        st["noise"] = None
        if sync is not None and (G.optim.hyper_dirty() or D.optim.hyper_dirty() or (config["ema"] and ema.decay_dirty(state_dict["itr"]))):
            # the previous step's side-stream Adam / EMA still READ the lr / decay block that the push below rewrites on the
            # main stream: join them first, or which step sees a new value is timing-dependent per rank (replicas diverge).
            # Only when there IS something to push (a scheduler step, the EMA start iteration): an unconditional join here would
            # put G's exchange back in front of the step.
            sync.wait("D")
            sync.wait("G")
        push_device_state()                               # lr / decay changes reach the graphs through device memory
        vals = whole() if sync is None else step_tensor(*seg, replayed=True, run_dr=seg_real)
        return dict(zip(KEYS, vals.tolist()))

    def step_eager_tensor(x, y):
        """One eager step returning the loss tensor (bench.py: per-kernel timing pass)."""
        st["x"], st["y"], st["noise"] = x, y, None
        st["static"] = None
        return whole_step()

    def close():
        """Release what the step holds on the device, in a defined order, while the HIP runtime is still alive: join the side
        streams (gradient exchange, spectral-norm prefetch, weight gradients), drop the captured graphs (their executables and the
        private memory pools they pin), the queued spectral-norm passes, the static input copies and the cached scratch.  Without
        it these objects die during interpreter teardown, in whatever order the module globals are cleared -- a captured graph
        destroyed after its streams / the process group is the exit-time fault bench.py used to mask with os._exit.  ``train`` may
        be called again afterwards (graphs are re-captured)."""
        import gc
        cuda = torch.cuda.is_available()
        if sync is not None and cuda:
            sync.wait_all()
        if cuda:
            torch.cuda.synchronize()
        for r in (whole, *seg, *([seg_real] if seg_real is not None else [])):
            r.reset()
        for net in (G, D):
            plan = getattr(net, "_plan", None)
            if plan is not None:
                plan["bank"].discard_prefetched()
        st.update(x=None, y=None, static=None, noise=None, emb_real=None, sn_stream=None)
        st.pop("real_out", None)
        ops.release_device_caches()
        gc.collect()
        if cuda:
            torch.cuda.synchronize()

    train.step_tensor = step_eager_tensor
    train.close = close
    return train
