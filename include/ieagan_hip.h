/* C ABI of libieagan_hip.so -- the MI355X (gfx950) kernels behind the IEA-GAN G+D train step.
 *
 * Conventions: every entry point is `extern "C"`, takes plain pointers and
 * sizes, enqueues its work on the given hipStream_t (passed as void*), never allocates device
 * memory, never synchronises and never changes the current device.  The caller (the PyTorch caching
 * allocator on the Python side) owns every buffer.  Return value: 0 on success, negative on error;
 * `ieagan_last_error()` returns a thread-local description.  Functions are re-entrant (forward runs
 * on the Python main thread, backward on the autograd thread).
 *
 * Activations are bf16 NHWC; parameters, statistics and accumulators are fp32.  The reference has no
 * native interface (it is pure PyTorch): each entry point cites the reference ATen op sequence it
 * replaces (paths relative to the reference repository root).
 */
#ifndef IEAGAN_HIP_H
#define IEAGAN_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* ieagan_last_error(void);
#define IEAGAN_ABI_VERSION 11       /* bumped whenever a struct layout or a signature in this header changes */
int ieagan_abi_version(void);        /* == IEAGAN_ABI_VERSION of the header the library was built from */

/* ---- profiling hooks (bench.py): per-kernel HIP-event timing on the launch stream ---- */
int ieagan_prof_enable(int on);          /* 0 off | 1 time every launch (hipEvents) | 2 + shape tags */
int ieagan_prof_reset(void);
/* Writes up to `cap` records; returns the record count.  `bytes` = what the launches had to move by the launcher's own
 * accounting (operands incl. ReLU masks / shortcut tensors an epilogue really reads); `bytes_min` = the layer-granular
 * minimum of SURVEY 8d (conv: 2 * N * (Hs*Ws*Cin + H*W*Cout), tools/arch_calc.py), equal to `bytes` where no such
 * distinction exists. */
typedef struct { char name[96]; long launches; double ms; double flops; double bytes; double bytes_min; } ieagan_prof_rec;
int ieagan_prof_collect(ieagan_prof_rec* out, int cap);

/* ---- source operand of a convolution with its fused prologue --------------------------------
 * Replaces, in front of F.conv2d: ccbn / bn apply (layers.py:656-689, 728-742), ReLU,
 * F.interpolate(scale_factor=2) (model.py:60-65) and nn.AvgPool2d(2) (model.py:553-554, 535-536). */
typedef struct {
    const void* x;        /* bf16 [N, Hs, Ws, Cx]                                              */
    int Cx, Hs, Ws;
    int rs;               /* 0 same resolution | 1 nearest x2 upsample of x | 2 2x2 average pool */
    const float* scale;   /* per-(n,c) BN scale  rstd*(1+gain)   or NULL                        */
    const float* shift;   /* per-(n,c) BN shift  bias - mean*scale                              */
    int aff_nstride;      /* image stride of scale/shift (0: per-channel only)                  */
    int relu;
} ieagan_src_desc;

/* ---- implicit-GEMM convolution (forward, and dgrad with the transposed pack) -----------------
 * Replaces SNConv2d.forward = F.conv2d(x, W/sigma, bias, 1, pad) (layers.py:197-206) together with
 * the residual add / channel-dropped shortcut / concat shortcut of GBlock and DBlock
 * (model.py:60-71, 534-557) and the statistics pass of the following batch norm. */
typedef struct {
    int N, H, W;          /* output (= conv) resolution                                         */
    int Cin, Cout, taps;  /* taps 1 (1x1, pad 0) or 9 (3x3, pad 1)                              */
    int Kpad;             /* weight-pack row length, multiple of 32                             */
    ieagan_src_desc src;
    const void* w;        /* bf16 [Cout][Kpad], k = tap*Cin + c  (from ieagan_sn_forward)       */
    const float* bias;    /* [Cout] or NULL                                                     */
    const void* ra;       /* residual A: bf16, channels [0,Ca) of a tensor with Cra channels    */
    int Cra, Ca, ra_rs;   /* ra_rs 0 same | 1 residual at half res | 2 residual at double res   */
    float ra_scale;       /* out += ra_scale * resample(ra)  (1 in forward; 4 / 0.25 when the    */
                          /* epilogue adds a shortcut GRADIENT: 2x2 sum / 0.25 * expand)         */
    const void* rb;       /* residual B: bf16 [N,H,W,Crb] feeding channels [Ca, Cout)           */
    int Crb;
    const void* mask;     /* bf16 [N,H,W,Cout]: output zeroed where mask <= 0, or NULL          */
    void* out;            /* bf16 [N,H,W,Cout]                                                  */
    float* stats;         /* fp32 [E][32][2][Cout] replicated (sum, sumsq) per event, accumulated; or NULL */
    int n_per_event;      /* images per event: E = N / n_per_event statistics groups (BatchNorm statistics are
                           * intra-event); 0 or N = one event.  E > 1 needs n_per_event*H*W % 128 == 0 */
    int flags;            /* IEAGAN_CONV_* bits below (kernel selection overrides for tests / benchmarks)        */
    /* BatchNorm-apply backward fused into a dgrad launch (replaces the stand-alone pass over da / x that
     * ieagan_prologue_bwd makes): with bnb_scale != NULL the accumulator tile da is NOT stored; per element
     *   pre = x*scale[n,c] + shift[n,c];  d = (bnb_relu && pre <= 0) ? 0 : da;  out = d*scale[n,c] (+ residual A)
     * where x is the tensor passed as `mask` (the BatchNorm input = the forward conv's source), and per image
     *   stats[n][r][0][c] += sum d      (d shift)      stats[n][r][1][c] += sum d*x   (d scale)
     * i.e. `stats` is fp32 [N][8][2][Cout] (8 replicas) and n_per_event must be 1.  Reference: autograd of
     * F.batch_norm(...)*(1+gain)+bias followed by ReLU (layers.py:656-689). */
    const float* bnb_scale;
    const float* bnb_shift;
    int bnb_nstride;      /* image stride of bnb_scale / bnb_shift (0: per-channel rows shared by all images)  */
    int bnb_relu;
    /* Slots per statistics group: `stats` is fp32 [E][stats_slots][2][Cout] (BatchNorm-backward mode: [N][stats_slots][2][Cout]); block b of
     * a group adds into slot b % stats_slots.  With stats_slots == ieagan_conv_stats_slots(d) every slot has exactly ONE adder, so the
     * sums (and everything computed from them) are bit-reproducible run to run -- the order of float atomics on a shared address is not.
     * 0: the legacy replica counts (32; 8 in BatchNorm-backward mode). */
    int stats_slots;
} ieagan_conv_desc;
#define IEAGAN_CONV_FORCE_GATHER 1
#define IEAGAN_CONV_FP8 4              /* forward and dgrad launches of the C = 64 / 128 3x3 layers with OCP e4m3 MFMA operands (per-slice
                                        * weight scale, per-tile activation scale, block-scaled K = 128 MFMA, fp32 accumulate): BASELINE
                                        * configs[4]; tensors stay bf16 */
#define IEAGAN_CONV_FP8_NOSCALE 8      /* with IEAGAN_CONV_FP8 (benchmarks): the non-scaled K = 32 fp8 MFMA instead of the block-scaled K = 128 form */
#define IEAGAN_CONV_NO_LDS_WEIGHTS 2   /* tests / benchmarks: C = 64 / 128 3x3 layers through conv3x3_halo instead of conv3x3_lds */
int ieagan_conv_forward(const ieagan_conv_desc* d, void* stream);
/* blocks per statistics group of the launch ieagan_conv_forward(d) would make (the kernel selection and its grid are decided exactly as in
 * the real call, nothing is launched); < 0 on error.  Size `stats` with it and pass it as stats_slots. */
int ieagan_conv_stats_slots(const ieagan_conv_desc* d);

/* ---- weight gradient: dWp[Cout][Kpad] += G^T A   (autograd of F.conv2d w.r.t. weight) -------- */
typedef struct {
    int N, H, W;
    int Cin, Cout, taps, Kpad;
    ieagan_src_desc src;  /* the forward's A operand (same prologue)                            */
    const void* g;        /* bf16 gradient w.r.t. the conv output, channels [0,Cout) of Cg      */
    int Cg;
    float* dw;            /* fp32 [Cout][Kpad], caller zeroes                                   */
    int tiles_per_block;  /* filled by the launcher                                             */
    int pad_rows;         /* filled by the launcher                                             */
    float* partials;      /* optional workspace of ieagan_conv_wgrad_workspace(d) floats: the pixel-split blocks then STORE
                           * their partial dW slabs there and a second launch folds them into dw -- float atomics retire at
                           * ~0.7 TB/s here, which made the accumulation tail the longest phase of every large-dW launch */
    float* colsum;        /* optional fp32 [32][Cout], caller-zeroed replicas: += column sums of g (the
                           * bias gradient), taken from the g tiles the kernel stages anyway           */
} ieagan_wgrad_desc;
int ieagan_conv_wgrad(const ieagan_wgrad_desc* d, int use_tr_read, void* stream);
/* floats of `partials` workspace the launch can use (0: the direct atomic accumulation is the better choice) */
long ieagan_conv_wgrad_workspace(const ieagan_wgrad_desc* d, int use_tr_read);

/* ---- the whole backward of a 1x1 convolution on a large feature map in one launch (conv1x1_bwd.hip) -------------------
 * Replaces, per layer and backward pass, the launches ieagan_effgrad -> ieagan_conv_forward (as dgrad) [-> ieagan_prologue_bwd]
 * -> ieagan_conv_wgrad, i.e. autograd of  F.conv2d(relu(bn(x)) | avg_pool(relu(x)), W / sigma, b)  for kernel_size 1 (reference
 * layers.py:197-206, 656-689; model.py:54-71, 541-557): every operand tile is read once.
 *   g_eff = g + dsum[e][c] + 2 y dsumsq[e][c]               (y / dstat NULL: g_eff = g; geff_out: g_eff is also stored, bf16 [N,H,W,Cout])
 *   da    = g_eff W            (conv resolution)             w_bwd = the transposed pack [Cin][Kpad2], k = cout
 *   out_mode 0:  dx[N,Hs,Ws,Cin] = prologue'(x) (.) resample^T(da) + shortcut gradient;  with the affine prologue dx = d * scale[n,c] and
 *                bn_acc[N][8][2][Cin] += {sum d, sum d*x} per image (d = ReLU-masked da), as ieagan_conv_desc.bnb_*
 *   out_mode 1:  dx[N,H,W,Cin]   = da + shortcut gradient   (plain prologue only; the pooled conv_sc of a D block, whose consumer expands)
 *   dw[Cout][Kpad] += g_eff^T a,  a = the forward's A operand (prologue + 2x2 average pool applied);  colsum[32][Cout] += column sums of g_eff
 * Shortcut gradient lg: channels [0, lCa) of a tensor with lC channels; lmode 0 same resolution as dx, 1 = 2x2 SUM of a tensor at double
 * resolution, 2 = 0.25 * nearest expand of a tensor at half resolution.  Supported shapes: ieagan_conv1x1_bwd_supported; W % 32 == 0. */
typedef struct {
    int N, H, W;              /* conv (output) resolution                                                            */
    int Cin, Cout, Kpad, Kpad2;
    ieagan_src_desc src;      /* the forward's source operand with its prologue (rs 0 or 2)                          */
    const void* g;            /* bf16 out-grad, channels [0,Cout) of Cg                                              */
    int Cg;
    const void* y;            /* bf16 [N,H,W,Cout] forward output (effgrad) or NULL                                  */
    const float* dstat;       /* fp32 [E][2][Cout] (dsum, dsumsq) or NULL                                            */
    int n_per_event;          /* images per event (0: one event)                                                     */
    void* geff_out;           /* optional bf16 [N,H,W,Cout]                                                          */
    const void* w_bwd;        /* bf16 [Cin][Kpad2]                                                                   */
    const void* lg;           /* shortcut gradient or NULL                                                           */
    int lC, lCa, lmode;
    void* dx;                 /* bf16, see out_mode; NULL: no data gradient                                          */
    int out_mode;
    float* bn_acc;            /* fp32 [N][bn_slots][2][Cin], caller-zeroed (affine prologue)                         */
    float* dw;                /* fp32 [Cout][Kpad], accumulated; NULL: no weight gradient                            */
    float* partials;          /* optional workspace of ieagan_conv1x1_bwd_workspace(d) floats (two-stage accumulation) */
    float* colsum;            /* optional fp32 [32][Cout] caller-zeroed replicas (bias gradient)                     */
    int flags;                /* IEAGAN_B1_* (benchmarks: force the 2 / 3 blocks-per-CU build of the kernel)         */
    int bn_slots;             /* bn_acc is [N][bn_slots][2][Cin]; == ieagan_conv1x1_bwd_slots(d): one adder per slot (bit-reproducible); 0: 8 */
} ieagan_conv1x1_bwd_desc;
#define IEAGAN_B1_OCC2 1
#define IEAGAN_B1_OCC3 2
#define IEAGAN_B1_TP32 4          /* 32-pixel wave tiles everywhere */
#define IEAGAN_BWD_NO_REDUCE 16   /* ieagan_conv1x1_bwd / ieagan_conv3x3_bwd: leave the partial dW slabs in `partials` -- the caller folds them with
                                   * ieagan_wgrad_reduce (e.g. on a side stream: nothing in a backward pass reads dW before its end) */
int ieagan_conv1x1_bwd(const ieagan_conv1x1_bwd_desc* d, void* stream);
long ieagan_conv1x1_bwd_workspace(const ieagan_conv1x1_bwd_desc* d);
int ieagan_conv1x1_bwd_supported(int Cin, int Cout, int rs, int affine);
int ieagan_conv1x1_bwd_slots(const ieagan_conv1x1_bwd_desc* d);      /* blocks per image of the launch */
/* second stage of every two-stage weight-gradient accumulation: dw[Cout][Kpad] += sum over the S slabs of `partials` ([S][Cout][Kpad]; columns
 * k >= K are padding).  S = workspace floats / (Cout * Kpad). */
int ieagan_wgrad_reduce(const float* partials, float* dw, int S, int Cout, int Kpad, int K, void* stream);
/* the same for n slab sets in one launch: the whole-backward kernels of a backward pass leave their slabs (IEAGAN_BWD_NO_REDUCE), the pass
 * folds them together at its end (nothing reads dW earlier) */
typedef struct {
    const float* partials;    /* [S][Cout][Kpad] */
    float* dw;                /* [Cout][Kpad], accumulated */
    int S, Cout, Kpad, K;
} ieagan_reduce_item;
int ieagan_wgrad_reduce_batched(const ieagan_reduce_item* items, int n, void* stream);

/* ---- the whole backward of a 3x3 convolution with Cin = Cout = C in {16, 32} on a large feature map in one launch (conv3x3_bwd.hip) ----
 * Replaces, per layer and backward pass, ieagan_effgrad -> ieagan_conv_forward (as dgrad) [-> ieagan_prologue_bwd] -> ieagan_conv_wgrad, i.e.
 * autograd of  F.conv2d(relu(bn(x)) | relu(x) | F.interpolate(relu(bn(x)), 2), W / sigma, b, padding=1)  (reference layers.py:197-206,
 * 656-689; model.py:54-71 GBlock conv2 / conv3, 541-557 DBlock conv2 / conv3): g (and y) are read once with a one-pixel halo, x once.
 *   g_eff = g + dsum[e][c] + 2 y dsumsq[e][c]   inside the image, 0 outside   (y / dstat NULL: g_eff = g)
 *   da[q] = sum_tap g_eff[q - off(tap)] W[.][tap, .]                            w_bwd = the flipped / transposed pack [C][Kpad]
 *   dx[N,Hs,Ws,C] = prologue'(x) (.) resample^T(da): ReLU mask; with the affine prologue dx = d * scale[n,c] and
 *                   bn_acc[N][8][2][C] += {sum d, sum d*x} per image (as ieagan_conv_desc.bnb_*); rs = 1: 2x2 sum of da first
 *   dw[C][Kpad] += sum_q g_eff[q - off(tap)]^T a[q]  (a = the forward's activated, resampled input);  colsum[32][C] += column sums of g_eff
 * The prologue must include the ReLU (every 3x3 layer of the path does).  H % 8 == 0, W % 32 == 0 (conv resolution). */
typedef struct {
    int N, H, W;              /* conv (output) resolution                                                            */
    int C, Kpad;              /* Cin = Cout = C; Kpad = pad32(9 C), the row length of both weight packs               */
    ieagan_src_desc src;      /* the forward's source operand with its prologue (rs 0 or 1, relu = 1)                */
    const void* g;            /* bf16 out-grad, channels [0,C) of Cg                                                 */
    int Cg;
    const void* y;            /* bf16 [N,H,W,C] forward output (effgrad) or NULL                                     */
    const float* dstat;       /* fp32 [E][2][C] (dsum, dsumsq) or NULL                                               */
    int n_per_event;          /* images per event (0: one event)                                                     */
    const void* w_bwd;        /* bf16 [C][Kpad]                                                                      */
    void* dx;                 /* bf16 [N,Hs,Ws,C]                                                                    */
    float* bn_acc;            /* fp32 [N][bn_slots][2][C], caller-zeroed (affine prologue), else NULL                */
    float* dw;                /* fp32 [C][Kpad], accumulated                                                         */
    float* partials;          /* workspace of ieagan_conv3x3_bwd_workspace(d) floats: one dW slab per block          */
    float* colsum;            /* optional fp32 [32][C] caller-zeroed replicas (bias gradient)                        */
    int flags;                /* 0 | IEAGAN_BWD_NO_REDUCE                                                            */
    int bn_slots;             /* bn_acc is [N][bn_slots][2][C]; == ieagan_conv3x3_bwd_slots(d): one adder per slot (bit-reproducible); 0: 8 */
} ieagan_conv3x3_bwd_desc;
int ieagan_conv3x3_bwd(const ieagan_conv3x3_bwd_desc* d, void* stream);
long ieagan_conv3x3_bwd_workspace(const ieagan_conv3x3_bwd_desc* d);
int ieagan_conv3x3_bwd_supported(int C, int rs, int affine, int relu, int effgrad, int H, int W);
int ieagan_conv3x3_bwd_slots(const ieagan_conv3x3_bwd_desc* d);      /* blocks per image of the launch */

/* ---- element-wise companions (bn_elem.hip) ---------------------------------------------------- */
/* g_eff = dout + dsum[e][c] + 2*out*dsumsq[e][c]; colsum[32][C] += column sums (bias gradient).
 * Autograd of F.batch_norm's batch statistics folded into the producer's out-grad.  P pixels in E events of P / E
 * consecutive pixels each, dstat fp32 [E][2][C]. */
int ieagan_effgrad(const void* dout, const void* out, const float* dstat, void* geff, float* colsum,
                   long P, int C, int E, void* stream);
/* backward of the fused prologue: dx, and per-(n,c) d scale / d shift.
 * radd (optional): gradient of a shortcut that read the same x (channels [0,Ca) of a tensor with Cr
 * channels; rmode 0 same resolution, 1 = 2x2 sum of a tensor at double resolution), added into dx.
 * slots == 0: dscale / dshift fp32 rows n * nstride, caller-zeroed, accumulated with float atomics (order-dependent last bits);
 * slots >= IEAGAN_PROLOGUE_BWD_SLOTS: dscale points at per-image accumulators fp32 [N][slots][2][C] ({sum d, sum d*x}, caller-zeroed; the
 * layout ieagan_bn_finalize_bwd takes with acc_repl = slots), every block stores into its own slot: bit-reproducible; dshift unused. */
#define IEAGAN_PROLOGUE_BWD_SLOTS 64
int ieagan_prologue_bwd(const void* da, const void* x, int Cx, const float* scale, const float* shift,
                        int nstride, int relu, int rs, void* dx, float* dscale, float* dshift,
                        int N, int Hs, int Ws, int C, const void* radd, int Cr, int Ca, int rmode, int slots, void* stream);
/* ccbn / bn statistics -> scale/shift (+ running-stat update), layers.py:656-689, 728-742.
 * E events of N / E images: stats [E][repl][2][C] (count = elements per channel of ONE event; the slots are folded in a fixed order), mean_rstd [E][2][C],
 * dstat [E][2][C]; the running statistics receive the mean of the E per-event momentum updates.  ld == 0 (plain bn):
 * per-channel gain / bias; scale / shift then have rows = 1 for E == 1 and rows = N (one per image) for E > 1, and
 * dgain / dbias are summed over the rows. */
int ieagan_bn_finalize_fwd(const float* stats, float count, const float* gain, const float* bias, int ld,
                           int plus_one, float eps, float momentum, int training, float* run_mean,
                           float* run_var, float* scale, float* shift, float* mean_rstd, int N, int C, int E,
                           int repl, float* scratch, void* stream);
/* `scratch`: ieagan_bn_finalize_fwd_scratch(C, E, repl) caller-ZEROED floats (0: none needed), or NULL.  With many slots the fold is split
 * over several workgroups which hand their partial sums over through it (ticket of the last arriver); without it one workgroup per 32
 * channels folds all the slots (same result bit for bit, slower for repl > 128). */
long ieagan_bn_finalize_fwd_scratch(int C, int E, int repl);
/* acc_repl > 0: dshift is ignored and dscale points at the replicated per-image accumulators [rows][acc_repl][2][C]
 * ({sum d, sum d*x}) a BatchNorm-backward dgrad launch produced (ieagan_conv_desc.bnb_*); they are folded here. */
int ieagan_bn_finalize_bwd(const float* dscale, const float* dshift, const float* gain, int ld, int plus_one,
                           const float* mean_rstd, float count, int training, float* dgain, float* dbias,
                           int ldd, float* dstat, int N, int C, int E, int acc_repl, float* scratch, void* stream);
/* `scratch`: ieagan_bn_finalize_bwd_scratch(N, C, E, acc_repl) caller-ZEROED floats (0: none needed), or NULL (one workgroup per 32 channels). */
long ieagan_bn_finalize_bwd_scratch(int N, int C, int E, int acc_repl);
int ieagan_res_bwd(const void* g, int Cg, void* dr, int Cr, int Ca, int mode, int N, int Hr, int Wr, void* stream);
/* stats (optional): fp32 [E][S][2][C] caller-zeroed (sum, sumsq) slots per event, E = N / n_per_event (0: one event),
 * S = (N / E) * ceil(HW / 32): one slot per (image of the event, 32-pixel block) -- a single adder each, bit-reproducible */
int ieagan_nchw_to_nhwc(const float* in, void* out, float* stats, int N, int C, int HW, int n_per_event, void* stream);
int ieagan_nhwc_to_nchw(const void* in, float* out, int N, int C, int HW, void* stream);
int ieagan_channel_stats(const void* x, float* stats, long P, int C, void* stream);

/* ---- single-channel-image convolutions (conv_c1.hip): D.input_conv (model.py:730, 905) and
 * G.output_layer = bn + relu + conv + tanh (model.py:379-387, 487) -------------------------------- */
int ieagan_conv_1toC(const float* img, const float* tanh_y, const float* w, const float* bias, void* out,
                     int N, int H, int W, int C, int flip, void* stream);
/* the same as a dgrad with the BatchNorm-apply + ReLU backward of G.output_layer's prologue folded into the store phase (model.py:379-387):
 * dx[n,h,w,c] = d * scale[n*nstride + c], d = (relu && !(x*scale+shift > 0)) ? 0 : conv;  dshift += sum d, dscale += sum d * x (caller-zeroed,
 * rows n * nstride; nstride 0: one row for the batch).  The gradient w.r.t. the activated tensor is never materialised.
 * slots >= ieagan_conv_1toC_bnb_slots(N, H, W, nstride): dscale points at accumulators fp32 [rows][slots][2][C] ({sum d, sum d*x}; rows = N,
 * or 1 with nstride 0; caller-zeroed) with ONE adder per address (bit-reproducible sums); dshift unused.  slots 0: float atomics on the rows. */
int ieagan_conv_1toC_bnb(const float* img, const float* tanh_y, const float* w, const void* x, const float* scale, const float* shift,
                         int nstride, int relu, void* dx, float* dscale, float* dshift, int N, int H, int W, int C, int flip, int slots,
                         void* stream);
int ieagan_conv_1toC_bnb_slots(int N, int H, int W, int nstride);
/* tanh_out: 0 = linear, 1 = tanh, 2 = tanh + the detector-unit export of model.generate (model.py:1139-1147:
 * threshold(-0.26 -> -1), 256^((r+1)/2) - 1, clamp to [0, 255], rows 3 .. H-4 only): out is then fp32 [N, H-6, W]. */
int ieagan_conv_Cto1(const void* x, const float* scale, const float* shift, int nstride, int relu,
                     const float* w, const float* bias, float* out, int tanh_out, int N, int H, int W, int C,
                     int flip, void* stream);
int ieagan_wgrad_c1(const float* img, const float* tanh_y, const void* t, const float* scale, const float* shift,
                    int nstride, int relu, float* dw, int N, int H, int W, int C, int flip, void* stream);

/* ---- input side of the first discriminator block in one launch each way (d_stem.hip) ---------------------------------------
 * Replaces D.input_conv (3x3, 1 -> 32, model.py:905) + the three reads of its output by the first DBlock (conv1 1x1 32 -> 16, conv_sc on
 * AvgPool2d, the pooled identity shortcut: model.py:534-557, first block: no pre-activation):
 *   fwd:  img fp32 [N,H,W] -> h1 = conv1(h0) bf16 [N,H,W,16], p0 = AvgPool2d(h0) bf16 [N,H/2,W/2,32], sc = conv_sc(p0) bf16 [N,H/2,W/2,32],
 *         h0 = input_conv(img) recomputed on chip, never stored;
 *   bwd:  dh0 = dh1 W1 + 0.25 expand(dp0) (LDS only) -> dw_in [9][32] += , db_in [32][32 replicas] +=, dw1 [16][32] +=, db1 [32][16] +=
 *         (weight gradients only: the pass that needs d img keeps the generic launches).
 * w_in fp32 [9][32] (tap-major, / sigma), w1 / wsc the bf16 forward packs [Cout][32], w1_bwd the transposed pack [32][32].  H % 8 == 0, W % 32 == 0. */
typedef struct {
    const float* img;
    int N, H, W;
    const float* w_in;
    const float* b_in;
    const void* w1;
    const float* b1;
    const void* wsc;
    const float* bsc;
    void* h1;
    void* p0;
    void* sc;
    const void* dh1;
    const void* dp0;
    const void* w1_bwd;
    float* dw_in;
    float* db_in;
    float* dw1;
    float* db1;
} ieagan_d_stem_desc;
int ieagan_d_stem_fwd(const ieagan_d_stem_desc* d, void* stream);
int ieagan_d_stem_bwd(const ieagan_d_stem_desc* d, void* stream);

/* ---- batched spectral norm (sn.hip): layers.SN.W_ / power_iteration (layers.py:89-165) -------- */
int ieagan_sn_forward(const long* table, const int* blocks, int nblocks, const int* cblocks, int ncblocks,
                      float* params, float* ctx, float* part, void* pack, float eps, int training, void* stream);
/* dW (=|+=) d(W/sigma) applied to gsn; optionally dbias (=|+=) fold of the replicated column sums */
int ieagan_sn_backward(const float* gsn, const float* W, int kind, int out, int in, int taps, int cin, int kpad,
                       const float* ctx, float* inner_scratch, float* dW, int accumulate, const float* colsum,
                       float* dbias, int bias_accumulate, void* stream);
int ieagan_sn_backward_stack(const long* table, const int* layers, const long* row0, const long* dst, int nlayers,
                             const float* gst, const float* params, const float* ctx, float* grad_base, int accumulate,
                             void* stream);
/* All conv layers of a network at the end of one backward pass (two launches).  The pass's wgrad kernels accumulated
 * d/d(W/sigma) (consumer layout) and the replicated bias column sums into ONE caller-zeroed scratch arena:
 *   btab  int64[12] per layer: {weight offset (same in params and grad), out, in, taps, cin, kind, kpad, ctx offset,
 *                               gsn offset in scratch, colsum offset in scratch or -1, bias offset in grad or -1, bias length}
 *   work  int32[2] per block : {layer, chunk of 2048 weight elements};  scratch[0 .. nlayers) = <gsn, W> accumulators
 * grad (the flat gradient arena) is accumulated into. */
int ieagan_sn_backward_batched(const long* btab, const int* work, int nwork, const float* params, const float* ctx,
                               float* scratch, float* grad, void* stream);

/* ---- non-local self-attention core (attention.hip): softmax(theta^T phi) applied to g, layers.py:291-299,
 * streaming softmax (no [N, Lq, Lk] tensor).  Q [N,Lq,dqk], K [N,Lk,dqk], V [N,Lk,dv], O [N,Lq,dv] bf16;
 * LSE, delta [N,Lq] fp32.  No 1/sqrt(d) scaling (as the reference). ------------------------------------ */
int ieagan_nl_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int N, int Lq, int Lk,
                            int dqk, int dv, void* stream);
int ieagan_nl_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                            float* delta, void* dQ, void* dK, void* dV, int N, int Lq, int Lk, int dqk, int dv, void* stream);

/* ---- small single-workgroup kernels (small_ops.hip) --------------------------------------------------
 * RRM attention core: softmax(q k^T / sqrt(hd)) v per (batch, head), S <= 64 tokens, affinity in LDS
 * (RRM.py:10-16, 46-58).  qkv [B,S,H,3*hd] (packed projection), out [B,S,H*hd], att [B,H,S,S], all fp32. */
int ieagan_rrm_attention_fwd(const float* qkv, float* out, float* att, int B, int S, int H, int hd, void* stream);
int ieagan_rrm_attention_bwd(const float* qkv, const float* att, const float* dout, float* dqkv, int B, int S, int H, int hd,
                             void* stream);
/* Stage-wise fused linear layers of the Relational Reasoning Module and the heads (rrm_fused.hip), fp32, exact on the fp32 matrix cores.
 * Replaces nn.LayerNorm + nn.Linear / SNLinear + ReLU + residual add chains (reference RRM.py:85-109, 118-125; model.py:915-921):
 *   slin_fwd:  Y[M,N] = [relu]( LN?(X)[M,K] W[N,K]^T + b ) [+ R]     LN: rows of X normalised with gamma / beta (eps), xhat [M,K] / rstd [M] saved
 *   slin_bwd:  dY' = dY masked where Ymask <= 0;  dX[M,K] = dY' W,  dW[N,K] = dY'^T Xn,  db[N] = column sums of dY'   (Xn NULL: xhat*ln_g+ln_b)
 *   ln_fwd:    nn.LayerNorm rows (+ F.normalize(., dim=1) with l2norm);   ln_bwd: its backward + dRes (residual path), dg / dbeta accumulated
 * K a multiple of 4 (<= 2048 forward); dx_zeroed: dX is pre-zeroed, long reductions over n may then be split and added atomically;
 * l2_beta: the LayerNorm was followed by F.normalize -- dY is the gradient w.r.t. the normalised rows. */
int ieagan_slin_fwd(const float* X, const float* W, const float* b, const float* R, float* Y, const float* ln_g, const float* ln_b,
                    float* xhat, float* rstd, int M, int K, int N, int relu, float eps, void* stream);
int ieagan_slin_bwd(const float* dY, const float* Ymask, const float* Xn, const float* xhat, const float* ln_g, const float* ln_b,
                    const float* W, float* dX, float* dW, float* db, int M, int K, int N, int dx_zeroed, void* stream);
int ieagan_ln_fwd(const float* X, const float* g, const float* b, float* Y, float* xhat, float* rstd, int M, int K, float eps, int l2norm,
                  void* stream);
int ieagan_ln_bwd(const float* dY, const float* xhat, const float* rstd, const float* g, const float* l2_beta, const float* dRes, float* dX,
                  float* dg, float* dbeta, int M, int K, void* stream);
/* class proxies of the discriminator head: F.normalize(F.embedding(y, Wn), dim=1) (model.py:916, 933) and the scatter of its gradient
 * into the caller-zeroed dW [classes, D] */
int ieagan_embed_norm_fwd(const long* y, const float* Wn, float* out, float* inv, int M, int D, void* stream);
int ieagan_embed_norm_bwd(const long* y, const float* p, const float* inv, const float* dp, float* dW, int M, int D, void* stream);
/* all losses of one phase, value and gradient, one launch (loss.py:8-44, 79-132); weights6 (host) = weights of
 * {hinge_real, hinge_fake, hinge_gen, contrastive, uniformity, IEA}; vals8 (device) = {total, the six terms, 0} */
int ieagan_loss_block(const float* dfake, const float* dreal, const float* e, const float* p, const float* er,
                      const float* weights6, float temperature, float* vals8, float* g_dfake, float* g_dreal, float* g_e,
                      float* g_p, int n, int d, void* stream);
/* the same for a batch of `events` events of n rows each (one workgroup per event; all tensors hold events * n rows, vals8 is
 * [events][8], the gradients are those of each event's own total) */
int ieagan_loss_block_events(const float* dfake, const float* dreal, const float* e, const float* p, const float* er,
                      const float* weights6, float temperature, float* vals8, float* g_dfake, float* g_dreal, float* g_e,
                      float* g_p, int n, int d, int events, void* stream);
/* D head: global sum pool of relu(h) (model.py:912) */
int ieagan_relu_sum_pool(const void* x, float* out, int N, int HW, int C, void* stream);
int ieagan_relu_sum_pool_bwd(const void* x, const float* dh, void* dx, int N, int HW, int C, void* stream);
/* Non-local block glue (reference layers.py:288-300).  maxpool2: F.max_pool2d(., [2,2]) on a bf16 NHWC map [N,H,W,C]
 * -> [N,H/2,W/2,C]; idx = one byte per output element (window position 0..3 of the first maximum).
 * gamma_residual: out = gamma[0] * o + x (bf16, n elements, gamma a device scalar); backward d_o = gamma * d and
 * dgamma[32 replicas, caller-zeroed] += sum(d * o); the gradient w.r.t. x is d itself. */
int ieagan_maxpool2_fwd(const void* x, void* out, void* idx, int N, int H, int W, int C, void* stream);
int ieagan_maxpool2_bwd(const void* dout, const void* idx, void* dx, int N, int H, int W, int C, void* stream);
int ieagan_gamma_residual_fwd(const void* o, const void* x, const float* gamma, void* out, long n, void* stream);
int ieagan_gamma_residual_bwd(const void* d, const void* o, const float* gamma, void* d_o, float* dgamma, long n,
                              void* stream);

/* ---- augmentation + optimiser (aug_optim.hip) -------------------------------------------------- */
#define IEAGAN_AUG_SLOTS 128      /* sums / gsums of the two DiffAugment entry points: fp32 [N][IEAGAN_AUG_SLOTS], caller-zeroed (per-image partial
                                   * sums, one writer per slot, folded in a fixed order: bit-reproducible image means) */
int ieagan_diffaug_fwd(const float* x, const float* bright, const float* contrast, const long* tx, const long* ty,
                       const long* ox, const long* oy, float* sums, float* out, int N, int H, int W, void* stream);
int ieagan_diffaug_bwd(const float* gout, const float* contrast, const long* tx, const long* ty, const long* ox,
                       const long* oy, float* gsums, float* gx, int N, int H, int W, void* stream);
int ieagan_cr_diffaug(const float* x, const float* flip_u, const long* tx, const long* ty, float* out, int N,
                      int H, int W, void* stream);
/* Event ingestion (reference utils/dataloader.py:66-77, utils/norm.py:9-20, utils/noise.py:29-32): uint8 [N,Hin,W] ->
 * zero-pad `pad` rows top and bottom -> log(u + 1)/log(256) -> + scale * noise (explicit U[0,1) draws, fp32 [N,Hin+2pad,W],
 * or NULL) -> (v - 0.5)/0.5 = fp32 [N, 1, Hin + 2 pad, W]. */
int ieagan_event_ingest(const void* ev_u8, const float* noise, float* out, int N, int Hin, int W, int pad, float scale,
                        void* stream);
/* hp: device float[8] = {lr, beta1, beta2, eps, step, grad scale, -, -}; the launcher increments step and
 * derives the bias corrections on the device, so a captured HIP graph of the step stays valid. */
int ieagan_adam_step(float* p, const float* g, float* m, float* v, long n, float* hp, void* stream);
int ieagan_ema_update(float* tgt, const float* src, long n, const float* decay_dev, void* stream);

/* ---- orthogonal regularisation (ortho.hip) -------------------------------------------------------
 * Replaces utils.ortho (reference utils/__init__.py:843-859) for ALL weight matrices of a network in one call:
 * grad[l] += 2*strength * ((W_l W_l^T) (.) (1 - I)) W_l, fp32, on the flat parameter / gradient arenas.
 *   table       int64[4] per layer: {weight offset (floats) in flat == in grad, R = shape[0], K = numel / R, Gram offset}
 *               Gram of layer l is [M,M] with M = min(R,K) (W W^T when R <= K, else W^T W), packed in `gram`
 *   gram_tiles  int32[4] per work item: {layer, tile_i, tile_j, split} -- 64x64 Gram tiles, reduce range
 *               [split*ieagan_ortho_ksplit(), +ieagan_ortho_ksplit()) of the long dimension
 *   apply_tiles int32[4] per work item: {layer, tile_i (rows of W), tile_j (columns of W), 0}
 *   gram        caller-owned scratch of gram_floats floats (zeroed by the call) */
int ieagan_ortho_ksplit(void);
int ieagan_ortho_grad(const float* flat, float* grad, const long* table, const int* gram_tiles, int n_gram_tiles,
                      const int* apply_tiles, int n_apply_tiles, float* gram, long gram_floats, float strength,
                      void* stream);

/* ---- self-test of the transposed LDS read used by conv_wgrad (tests only) ---------------------- */
int ieagan_selftest_tr_read(const void* in_bf16_64x16, void* out_bf16_64x8, void* stream);

#ifdef __cplusplus
}
#endif
#endif
