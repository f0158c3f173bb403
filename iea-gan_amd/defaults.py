"""Default hyper-parameters of the IEA-GAN train step, grouped by the component that reads them.

Values are those the reference ships for its 40-sensor 256x768 PXD configuration (reference
``config.json``); a user's own ``config.json`` / CLI overrides are merged on top exactly as before
(``train.py``).  Keys that exist only in this implementation are listed last and default to the
reference's behaviour.
"""

def default_config() -> dict:
    c = {}
    # -- event geometry / labels
    c.update(resolution=256, H_base=3, bottom_width=4, n_classes=40, batch_size=40, G_batch_size=0)
    # -- generator
    c.update(G_ch=32, G_depth=2, G_param="SN", G_shared=True, shared_dim=128, dim_z=128, z_var=1.0,
             hier=True, G_activation="inplace_relu", G_attn="0", G_init="ortho", G_kernel_size=3,
             norm_style="bn", cross_replica=False, mybn=False, RRM_prx_G=True, n_head_G=2, rdof_dim=4,
             normalized_proxy_G=False, prior_embed=False, z_dist="normal", truncated_threshold=1.0)
    # -- discriminator
    c.update(D_ch=32, D_depth=2, D_param="SN", D_wide=True, D_activation="inplace_relu", D_attn="32",
             D_init="ortho", D_kernel_size=3, attn_type="sa", conditional_strategy="Contra",
             hypersphere_dim=1024, nonlinear_embed=False, normalize_embed=True, RRM_prx_D=False,
             RRM_embed=True, n_head=4, inv_stereographic=False)
    # -- numerics
    c.update(BN_eps=1e-5, SN_eps=1e-6, adam_eps=1e-6, num_G_SVs=1, num_D_SVs=1, num_G_SV_itrs=1,
             num_D_SV_itrs=1, G_fp16=False, D_fp16=False, G_mixed_precision=False,
             D_mixed_precision=False)
    # -- optimisation
    c.update(G_lr=5e-5, D_lr=5e-5, G_B1=0.0, D_B1=0.0, G_B2=0.999, D_B2=0.999, amsgrad=False,
             ada_belief=False, num_D_steps=1, num_D_accumulations=1, num_G_accumulations=1,
             split_D=True, toggle_grads=True, G_ortho=1e-4, D_ortho=0.0, clip_norm=None,
             sched_version="default", ema=True, ema_decay=0.9999, use_ema=True, ema_start=10000)
    # -- losses / regularisers
    c.update(pos_collected_numerator=False, contra_lambda=1.0, Angle=False, angle_lambda=1.0,
             IEA_loss=True, IEA_lambda=1.0, Uniformity_loss=True, unif_lambda=0.1, diff_aug=True,
             Con_reg=False, cr_lambda=10, pixel_reg=False, px_lambda=1.0, latent_op=False,
             latent_reg_weight=300)
    # -- run control / bookkeeping (not read by the hot path)
    c.update(seed=3651, num_workers=8, pin_memory=False, shuffle=True, augment=0,
             use_multiepoch_sampler=False, debug=False, model="IEAGAN", num_epochs=4, parallel=False,
             accumulate_stats=False, num_standing_accumulations=16, G_eval_mode=True, save_every=1000,
             test_every=1000, num_save_copies=2, num_best_copies=2, skip_init=False, logstyle="%3.3e",
             sv_log_interval=10, log_interval=100, run_name="BGd_2718", resume=False, add_blur=False,
             add_noise=True, add_style=False, pbar="tqdm", which_best="FID", stop_after=100000,
             trunc_z=0.5, denoise=False, metric_log_name="metric_log.jsonl",
             reinitialize_metric_logs=False, reinitialize_parameter_logs=False, num_incep_images=16000,
             load_optim=True)
    # -- keys that exist only here (defaults = reference behaviour)
    c.update(hip_graph=False)            # replay the train step from captured HIP graphs (one graph; three in data-parallel runs)
    c.update(conv_dtype="bf16")          # 'fp8': e4m3 MFMA operands in the forward of the C >= 64 3x3 layers (BASELINE configs[4])
    c.update(events_per_step=1)          # E events of batch_size images per GPU and step (BASELINE configs[3])
    # data-parallel runs: evaluate D(x_real) before G(z) -> D(fake) so that G's gradient exchange hides under the real pass (DESIGN
    # section 7).  The two D passes then see swapped spectral-norm iterates relative to the reference / a single-GPU run
    # (tolerance-level): `--dp_real_first false` keeps the reference order on N GPUs; the value is written to the run
    # metadata with the rest of the configuration.
    c.update(dp_real_first=True)
    c.update(sn_prefetch=True)           # spectral-norm passes issued ahead of time on a side stream (single-GPU default step)
    return c
