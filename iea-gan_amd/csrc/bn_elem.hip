// Bandwidth-bound companions of the conv kernels (bf16 NHWC, 16 bytes = 8 channels per lane):
//   effgrad        g_eff = dout + dsum[c] + 2*out*dsumsq[c]   (batch-stat path of BN folded into the
//                  producer's out-grad) + per-channel column sums (bias grads)
//   prologue_bwd   backward of the fused conv prologue (BN apply + ReLU + upsample/pool): dx and the
//                  per-(n,c) reductions d scale / d shift
//   bn_finalize    (sum, sumsq, gain(y), bias(y)) -> per-(n,c) scale/shift, running-stat update;
//                  and its backward -> d gain, d bias, d sum, d sumsq
//   res_bwd        gradient of the residual operand of the conv epilogue (2x2 sum / 0.25 expand)
//   layout         NCHW fp32 <-> NHWC bf16 at the module boundary (+ statistics)
// Replaces F.batch_norm / ccbn / relu / interpolate / AvgPool2d backward of the reference
// (layers.py:656-689, 728-742; model.py:54-71, 541-557).
#include "common.h"
#include "../../include/ieagan_hip.h"

// ------------------------------------------------------------------------------------------------
// block-level reduction of 8 per-thread partials by channel group; thread t owns channel group
// (t % groups); result is added into dst[replica][c] with float atomics.
// ------------------------------------------------------------------------------------------------
// Only the first `active` = (256 / groups) * groups threads of a block carry data, so that a thread's
// channel group stays fixed over a grid-stride loop for ANY channel count that is a multiple of 8.
__device__ __forceinline__ int active_threads(int groups) { return (256 / groups) * groups; }

template <bool STORE = false>
__device__ __forceinline__ void reduce_groups_atomic(float (&part)[8], int groups, float* dst, float* red /*[256][8]*/) {
    const int t = threadIdx.x;
    const int active = active_threads(groups);
#pragma unroll
    for (int i = 0; i < 8; ++i) red[t * 8 + i] = part[i];
    __syncthreads();
    for (int o = t; o < groups * 8; o += 256) {
        const int g = o >> 3, i = o & 7;
        float s = 0.f;
        for (int u = g; u < active; u += groups) s += red[u * 8 + i];
        if (STORE) dst[g * 8 + i] = s;          // the block owns dst (its slot): plain store, bit-reproducible
        else atomicAdd(dst + g * 8 + i, s);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// effgrad: P pixels x C channels.  dstat = [2][C] (dsum, dsumsq) or nullptr.  colsum = [STAT_REPL][C].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void effgrad_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                      const float* __restrict__ dstat, bf16* __restrict__ geff,
                                                      float* __restrict__ colsum, long P, int C) {
    __shared__ float red[256 * 8];
    const int groups = C >> 3;                 // divides 256 (C in {16..512}, multiple of 16 ... checked on host)
    const int cg = threadIdx.x % groups;
    // blockIdx.y = event: P pixels of this event, its own (dsum, dsumsq) row
    const long ev_off = (long)blockIdx.y * P * C;
    dout += ev_off;
    if (out) out += ev_off;
    if (geff) geff += ev_off;
    if (dstat) dstat += (long)blockIdx.y * 2 * C;
    const long chunks = P * groups;
    float ds[8], dq[8], part[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ds[i] = dstat ? dstat[cg * 8 + i] : 0.f;
        dq[i] = dstat ? 2.f * dstat[C + cg * 8 + i] : 0.f;
        part[i] = 0.f;
    }
    const int act = active_threads(groups);
    for (long idx = (long)blockIdx.x * act + threadIdx.x; idx < chunks && threadIdx.x < act; idx += (long)gridDim.x * act) {
        const bf16x8 d = *(const bf16x8*)(dout + idx * 8);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf2f(d[i]);
        if (dstat) {
            const bf16x8 o = *(const bf16x8*)(out + idx * 8);
            bf16x8 w;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = v[i] + ds[i] + bf2f(o[i]) * dq[i];
                w[i] = f2bf(v[i]);
            }
            *(bf16x8*)(geff + idx * 8) = w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) part[i] += v[i];
    }
    if (colsum) reduce_groups_atomic(part, groups, colsum + (long)((blockIdx.x + blockIdx.y) % STAT_REPL) * C, red);
}

extern "C" int ieagan_effgrad(const void* dout, const void* out, const float* dstat, void* geff, float* colsum,
                              long P, int C, int E, void* stream) {
    CHECK_ARG(C % 8 == 0 && C <= 2048, "effgrad: C=%d must be a multiple of 8, <= 2048", C);
    CHECK_ARG(dstat == nullptr || (out != nullptr && geff != nullptr), "effgrad: dstat needs out and geff");
    if (E < 1) E = 1;
    CHECK_ARG(P % E == 0 && E <= 65535, "effgrad: %ld pixels are not %d whole events", P, E);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("effgrad", 0.0, (dstat ? 6.0 : 2.0) * P * C, st);
    const long Pe = P / E;
    long blocks = (Pe * (C / 8) + 255) / 256;
    const long cap = E > 1 ? (4096 / E > 64 ? 4096 / E : 64) : 2048;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(effgrad_kernel, dim3((unsigned)blocks, (unsigned)E), dim3(256), 0, st, (const bf16*)dout, (const bf16*)out,
                       dstat, (bf16*)geff, colsum, Pe, C);
    CHECK_LAUNCH("effgrad");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// prologue_bwd: iterate over SOURCE pixels of one image (grid.y = n).
//   da  [N, H, W, C]    gradient w.r.t. the conv input (conv resolution)
//   x   [N, Hs, Ws, Cx] the source the prologue read (channels [0,C))
//   dx  [N, Hs, Ws, C]
// ------------------------------------------------------------------------------------------------
template <bool AFF, bool RELU, int RS>
__global__ __launch_bounds__(256) void prologue_bwd_kernel(const bf16* __restrict__ da, const bf16* __restrict__ x, int Cx,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int nstride, bf16* __restrict__ dx, float* __restrict__ dscale,
                                                           float* __restrict__ dshift, int Hs, int Ws, int C,
                                                           const bf16* __restrict__ radd, int Cr, int Ca, int rmode, int slots) {
    __shared__ float red[256 * 8];
    const int n = blockIdx.y;
    const int groups = C >> 3;
    const int cg = threadIdx.x % groups;
    const int H = (RS == 1) ? 2 * Hs : (RS == 2) ? Hs / 2 : Hs;
    const int W = (RS == 1) ? 2 * Ws : (RS == 2) ? Ws / 2 : Ws;
    float sc[8], sh[8], p_ds[8], p_dt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = AFF ? scale[(long)n * nstride + cg * 8 + i] : 1.f;
        sh[i] = AFF ? shift[(long)n * nstride + cg * 8 + i] : 0.f;
        p_ds[i] = p_dt[i] = 0.f;
    }
    const long chunks = (long)Hs * Ws * groups;
    const int act = active_threads(groups);
    for (long idx = (long)blockIdx.x * act + threadIdx.x; idx < chunks && threadIdx.x < act; idx += (long)gridDim.x * act) {
        const long sp = idx / groups;        // source pixel within the image (idx % groups == cg)
        const int hs = (int)(sp / Ws), ws = (int)(sp - (long)hs * Ws);
        float d[8];
        if (RS == 0) {
            const bf16x8 v = *(const bf16x8*)(da + (((long)n * H + hs) * W + ws) * C + cg * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = bf2f(v[i]);
        } else if (RS == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 v = *(const bf16x8*)(da + (((long)n * H + 2 * hs + (q >> 1)) * W + 2 * ws + (q & 1)) * C + cg * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] += bf2f(v[i]);
            }
        } else {
            const bf16x8 v = *(const bf16x8*)(da + (((long)n * H + (hs >> 1)) * W + (ws >> 1)) * C + cg * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = 0.25f * bf2f(v[i]);
        }
        const bf16x8 xv = *(const bf16x8*)(x + (((long)n * Hs + hs) * Ws + ws) * Cx + cg * 8);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xf = bf2f(xv[i]);
            const float pre = AFF ? xf * sc[i] + sh[i] : xf;
            float di = d[i];
            if (RELU && !(pre > 0.f)) di = 0.f;
            p_ds[i] += di * xf;
            p_dt[i] += di;
            d[i] = AFF ? di * sc[i] : di;
        }
        if (radd != nullptr && cg * 8 < Ca) {
            // gradient of the shortcut that also read x (the consumer conv's residual operand): added here
            // so that autograd never has to sum two full-size tensors.  rmode 0: same resolution;
            // rmode 1: the shortcut was nearest-upsampled -> sum its 2x2 gradients.
            if (rmode == 0) {
                const bf16x8 r = *(const bf16x8*)(radd + (((long)n * Hs + hs) * Ws + ws) * Cr + cg * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] += bf2f(r[i]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 r = *(const bf16x8*)(radd + (((long)n * (2 * Hs) + 2 * hs + (q >> 1)) * (2 * Ws) + 2 * ws + (q & 1)) * Cr + cg * 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[i] += bf2f(r[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(d[i]);
        *(bf16x8*)(dx + (((long)n * Hs + hs) * Ws + ws) * C + cg * 8) = o;
    }
    if (AFF) {
        if (slots > 0) {        // dscale = acc [N][slots][2][C]: this block's own slot {sum d, sum d x} (the layout of the dgrad epilogue's accumulators)
            float* slot = dscale + (((long)n * slots + blockIdx.x) * 2) * C;
            reduce_groups_atomic<true>(p_dt, groups, slot, red);
            reduce_groups_atomic<true>(p_ds, groups, slot + C, red);
        } else {
            reduce_groups_atomic(p_ds, groups, dscale + (long)n * nstride, red);
            reduce_groups_atomic(p_dt, groups, dshift + (long)n * nstride, red);
        }
    }
}

extern "C" int ieagan_prologue_bwd(const void* da, const void* x, int Cx, const float* scale, const float* shift,
                                   int nstride, int relu, int rs, void* dx, float* dscale, float* dshift, int N, int Hs,
                                   int Ws, int C, const void* radd, int Cr, int Ca, int rmode, int slots, void* stream) {
    CHECK_ARG(radd == nullptr || (Ca % 8 == 0 && Ca <= C && Ca <= Cr && (rmode == 0 || rmode == 1)), "prologue_bwd: bad shortcut-gradient operand");
    CHECK_ARG(C % 8 == 0 && C <= 2048, "prologue_bwd: C=%d unsupported", C);
    CHECK_ARG(rs >= 0 && rs <= 2, "prologue_bwd: bad rs");
    CHECK_ARG(rs != 2 || (Hs % 2 == 0 && Ws % 2 == 0), "prologue_bwd: pooled source needs even size");
    CHECK_ARG(scale == nullptr || (shift && dscale && (dshift || slots > 0)), "prologue_bwd: affine needs shift/dscale/dshift");
    CHECK_ARG(slots == 0 || slots >= IEAGAN_PROLOGUE_BWD_SLOTS, "prologue_bwd: slots must be 0 or >= %d", IEAGAN_PROLOGUE_BWD_SLOTS);
    hipStream_t st = (hipStream_t)stream;
    const double px = (double)N * Hs * Ws * C;
    ProfScope prof("prologue_bwd", 0.0, 2.0 * px * (rs == 1 ? 6.0 : rs == 2 ? 2.25 : 3.0), st);
    long per = ((long)Hs * Ws * (C / 8) + 255) / 256;
    if (per > IEAGAN_PROLOGUE_BWD_SLOTS) per = IEAGAN_PROLOGUE_BWD_SLOTS;
    if (per < 1) per = 1;
    dim3 grid((unsigned)per, N);
    const bool aff = scale != nullptr;
#define PB(A, R, S) hipLaunchKernelGGL((prologue_bwd_kernel<A, R, S>), grid, dim3(256), 0, st, (const bf16*)da, (const bf16*)x, Cx, \
                                       scale, shift, nstride, (bf16*)dx, dscale, dshift, Hs, Ws, C, (const bf16*)radd, Cr, Ca, rmode, slots)
#define PB_RS(A, R)            \
    if (rs == 0) PB(A, R, 0);  \
    else if (rs == 1) PB(A, R, 1); \
    else PB(A, R, 2)
    if (aff && relu) { PB_RS(true, true); }
    else if (aff) { PB_RS(true, false); }
    else if (relu) { PB_RS(false, true); }
    else { PB_RS(false, false); }
#undef PB_RS
#undef PB
    CHECK_LAUNCH("prologue_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bn_finalize
//   stats     [E][R][2][C] (sum, sumsq) slots of the normalised tensor (R = the producer's blocks per event: single-adder slots, see
//             ieagan_conv_desc.stats_slots), count = elements per channel of one event
//   gain/bias per-(n,c) conditional terms with row stride ld (ccbn: scale uses 1+gain) or per-channel
//             parameters with ld == 0 (plain bn: scale uses gain)
//   training: batch statistics + running-stat update (momentum, unbiased variance for the running
//             update, as F.batch_norm); eval: running statistics
// A block = CH channels x {sum, sumsq} columns x PH slot phases: lane j of a phase reads column j of slot r -- consecutive lanes, consecutive
// floats.  Every sum has a FIXED order (phase ph takes its slots sequentially, the phase partials and the group partials are added in
// index order): bit-reproducible for any R.
// Large R (the producers of the big feature maps run 512 ... 2048 blocks): the slots are split over G block rows (blockIdx.y), each row
// publishes its partial sums (sc1 stores), and the row whose ticket comes last folds the G partials and does the rest -- one CU pulling
// 2 R C floats on its own took 10-14 us per launch.  Hand-off as MI355X_MICROARCH.md prescribes for "the workgroup whose add came last":
// sc1 payload stores, every storing wave's s_waitcnt vmcnt(0), workgroup barrier, ONE agent-scope returning add per workgroup, sc1 loads
// of the payload by the last workgroup behind a barrier its adding wave joins.
// ------------------------------------------------------------------------------------------------
#define BNF_CH 32
#define BNF_MAXE 16
#define BNF_SLOTS_PER_ROW 256      // slots one block row folds at most before the work is split (G = ceil(R / 256), <= 16): 16 loads per thread, one batch

__device__ __forceinline__ void sc1_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sc1_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ticket of block column `cb`: true for the block row whose add came last (all threads of the block get the same answer)
__device__ __forceinline__ bool last_row_arrives(int* tick, int cb, int G, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's payload stores have left
    __syncthreads();
    if (threadIdx.x == 0) *s_flag = (atomicAdd(tick + cb, 1) == G - 1) ? 1 : 0;
    __syncthreads();
    return *s_flag != 0;
}

static inline int bnf_rows(int R) {
    int g = (R + BNF_SLOTS_PER_ROW - 1) / BNF_SLOTS_PER_ROW;
    return g < 1 ? 1 : (g > 16 ? 16 : g);
}
static inline long bnf_tick_floats(int C) { return ((cdiv(C, 16) + 63) / 64) * 64; }

template <int CH>
__global__ __launch_bounds__(1024) void bn_finalize_fwd_kernel(const float* __restrict__ stats, float count, const float* __restrict__ gain,
                                       const float* __restrict__ bias, int ld, int plus_one, float eps, float momentum,
                                       int training, float* __restrict__ run_mean, float* __restrict__ run_var,
                                       float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_rstd,
                                       int N, int C, int E, int R, float* __restrict__ scratch, long tick_floats) {
    constexpr int COLS = 2 * CH, PH = 1024 / COLS;
    __shared__ float part[PH][COLS];
    __shared__ float tot[BNF_MAXE][COLS];
    __shared__ float ms[2][CH];
    __shared__ int s_flag;
    const int t = threadIdx.x, ph = t / COLS, j = t % COLS, cj = j % CH, which = j / CH;
    const int cb = blockIdx.x, g = blockIdx.y, G = gridDim.y;
    const int c0 = cb * CH, c = c0 + cj;
    const bool valid = c < C;
    const int npe = N / E;
    const int rows = (ld == 0 && E == 1) ? 1 : N;
    // the fan-out operands of the FIRST event (gain / bias rows: independent of the statistics) are requested before the slot fold, so
    // that their memory round trip runs under it instead of behind it (two items per thread cover 64 images x 32 channels)
    constexpr int FO = 2;
    float pg[FO], pb[FO];
    {
        const int cnt0 = ((rows == 1) ? 1 : npe) * CH;
#pragma unroll
        for (int k = 0; k < FO; ++k) {
            const int idx = min(t + k * 1024, cnt0 - 1);
            const int n = ((rows == 1) ? 0 : 0) + idx / CH, cc = min(c0 + idx % CH, C - 1);
            pg[k] = gain[(long)n * ld + cc];
            pb[k] = bias[(long)n * ld + cc];
        }
    }
    if (training) {
        const int Rg = (R + G - 1) / G, r_lo = g * Rg, r_hi = min(R, r_lo + Rg);
        for (int e = 0; e < E; ++e) {
            float acc = 0.f;
            if (valid && r_lo < r_hi) {
                // 16 independent loads in flight, then added in slot order
                const float* p = stats + (long)e * R * 2 * C + (long)which * C + c;
                for (int r0 = r_lo + ph; r0 < r_hi; r0 += 16 * PH) {
                    float buf[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int r = r0 + k * PH;
                        buf[k] = p[(long)min(r, r_hi - 1) * 2 * C];
                        if (r >= r_hi) buf[k] = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc += buf[k];
                }
            }
            part[ph][j] = acc;
            __syncthreads();
            if (t < COLS) {
                float sacc = 0.f;
#pragma unroll
                for (int k = 0; k < PH; ++k) sacc += part[k][t];
                if (G == 1) tot[e][t] = sacc;
                else if (valid) sc1_store(scratch + tick_floats + ((long)e * G + g) * 2 * C + (long)which * C + c, sacc);
            }
            __syncthreads();
        }
        if (G > 1) {
            if (!last_row_arrives((int*)scratch, cb, G, &s_flag)) return;
            if (t < COLS && valid) {
                for (int e = 0; e < E; ++e) {
                    // all G partials requested before the first add (a load -> add -> load chain is one memory round trip per row)
                    float buf[16];
#pragma unroll
                    for (int gg = 0; gg < 16; ++gg)
                        buf[gg] = sc1_load(scratch + tick_floats + ((long)e * G + min(gg, G - 1)) * 2 * C + (long)which * C + c);
                    float sacc = 0.f;
#pragma unroll
                    for (int gg = 0; gg < 16; ++gg) sacc += (gg < G) ? buf[gg] : 0.f;
                    tot[e][t] = sacc;
                }
            }
            __syncthreads();
        }
    } else if (g != 0) {
        return;
    }
    float upd_mean = 0.f, upd_var = 0.f;             // (threads t < CH)
    for (int e = 0; e < E; ++e) {
        if (t < CH && valid) {
            float mean, var;
            if (training) {
                const float s1 = tot[e][t], s2 = tot[e][CH + t];
                mean = s1 / count;
                var = fmaxf(s2 / count - mean * mean, 0.f);
                upd_mean += mean;
                upd_var += var * (count / fmaxf(count - 1.f, 1.f));
            } else {
                mean = run_mean[c];
                var = run_var[c];
            }
            const float rstd = rsqrtf(var + eps);
            ms[0][t] = mean;
            ms[1][t] = rstd;
            mean_rstd[(long)e * 2 * C + c] = mean;
            mean_rstd[(long)e * 2 * C + C + c] = rstd;
        }
        __syncthreads();
        // fan-out over the images of this event: thread -> (image, channel), CH consecutive channels per image row
        const int n0 = (rows == 1) ? 0 : e * npe, n1 = (rows == 1) ? 1 : (e + 1) * npe;
        auto fan = [&](int idx, float gv, float bv) {
            const int n = n0 + idx / CH, cc = idx % CH;
            const float sc = ms[1][cc] * (gv + (plus_one ? 1.f : 0.f));
            scale[(long)n * C + c0 + cc] = sc;
            shift[(long)n * C + c0 + cc] = bv - ms[0][cc] * sc;
        };
        const int cnt = (n1 - n0) * CH;
        if (e == 0) {                                   // the operands requested up front (registers: static indices only)
#pragma unroll
            for (int k = 0; k < FO; ++k) {
                const int idx = t + k * 1024;
                if (idx < cnt && c0 + idx % CH < C) fan(idx, pg[k], pb[k]);
            }
        }
        for (int idx = t + (e == 0 ? FO * 1024 : 0); idx < cnt; idx += 1024) {
            const int n = n0 + idx / CH, cc = idx % CH;
            if (c0 + cc < C) fan(idx, gain[(long)n * ld + c0 + cc], bias[(long)n * ld + c0 + cc]);
        }
        __syncthreads();
    }
    if (training && t < CH && valid) {     // mean of the E per-event momentum updates (E = 1: F.batch_norm's update)
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * upd_mean / (float)E;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * upd_var / (float)E;
    }
}

extern "C" long ieagan_bn_finalize_fwd_scratch(int C, int E, int repl) {
    if (repl == 0) repl = STAT_REPL;
    const int G = bnf_rows(repl);
    return G <= 1 ? 0 : bnf_tick_floats(C) + (long)(E < 1 ? 1 : E) * G * 2 * C;
}

extern "C" int ieagan_bn_finalize_fwd(const float* stats, float count, const float* gain, const float* bias, int ld,
                                      int plus_one, float eps, float momentum, int training, float* run_mean,
                                      float* run_var, float* scale, float* shift, float* mean_rstd, int N, int C, int E,
                                      int repl, float* scratch, void* stream) {
    CHECK_ARG(!training || stats != nullptr, "bn_finalize: training mode needs batch statistics");
    CHECK_ARG(repl >= 0, "bn_finalize: bad slot count %d", repl);
    if (repl == 0) repl = STAT_REPL;
    if (E < 1) E = 1;
    CHECK_ARG(N % E == 0 && E <= BNF_MAXE, "bn_finalize: %d images are not %d (<= %d) whole events", N, E, BNF_MAXE);
    int G = training ? bnf_rows(repl) : 1;
    if (scratch == nullptr) G = 1;                 // no hand-off space: one block row folds everything (correct, slower for large R)
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("bn_finalize_fwd", 0.0, 0.0, st);
    const long tkf = bnf_tick_floats(C);
    if (C <= 16)
        hipLaunchKernelGGL((bn_finalize_fwd_kernel<16>), dim3(cdiv(C, 16), G), dim3(1024), 0, st, stats, count, gain, bias, ld, plus_one, eps,
                           momentum, training, run_mean, run_var, scale, shift, mean_rstd, N, C, E, repl, scratch, tkf);
    else
        hipLaunchKernelGGL((bn_finalize_fwd_kernel<BNF_CH>), dim3(cdiv(C, BNF_CH), G), dim3(1024), 0, st, stats, count, gain, bias, ld, plus_one, eps,
                           momentum, training, run_mean, run_var, scale, shift, mean_rstd, N, C, E, repl, scratch, tkf);
    CHECK_LAUNCH("bn_finalize_fwd");
    return 0;
}

//   dscale/dshift [rows][C] (R == 0) or per-image slot accumulators [N][R][2][C] ({sum d, sum d x}, R > 0)
//   ->  dgain/dbias [rows][ld-strided slice] (ld==0: [C], summed over the rows), dstat [E][2][C]
// Block = 32 channels x {d shift, d scale} per 64-lane phase, 16 image phases; the images of an event are split over G block rows (each row
// owns whole images: their d gain / d bias rows leave directly), the per-channel sums over the images (d rstd, d mean; ld == 0: d gain,
// d bias) are published per row and folded by the row whose ticket comes last.  Every sum in a fixed order.
#define BNF_NB 64          // images per LDS pass
#define BNF_IMG_PER_ROW 8  // images one block row takes at most before the work is split (G <= 8)
static inline int bnb_rows(int npe, int R, int rows) {
    if (rows == 1 || R == 0) return 1;
    int g = (npe + BNF_IMG_PER_ROW - 1) / BNF_IMG_PER_ROW;
    return g < 1 ? 1 : (g > 8 ? 8 : g);
}

__global__ __launch_bounds__(1024) void bn_finalize_bwd_kernel(const float* __restrict__ dscale, const float* __restrict__ dshift,
                                       const float* __restrict__ gain, int ld, int plus_one, const float* __restrict__ mean_rstd,
                                       float count, int training, float* __restrict__ dgain, float* __restrict__ dbias, int ldd,
                                       float* __restrict__ dstat, int N, int C, int E, int R, float* __restrict__ scratch, long tick_floats) {
    __shared__ float v[BNF_NB][2 * BNF_CH];        // [image][{dt | ds} x channel], then {drstd | dmean} contributions
    __shared__ float w[BNF_NB][2 * BNF_CH];        // ld == 0: {d gain | d bias} contributions
    __shared__ int s_flag;
    const int t = threadIdx.x, ph = t >> 6, j = t & 63, cj = j & (BNF_CH - 1), which = j >> 5;
    const int cb = blockIdx.x, g = blockIdx.y, G = gridDim.y;
    const int c0 = cb * BNF_CH, c = c0 + cj;
    const bool valid = c < C;
    const int npe = N / E;
    const int rows = (ld == 0 && E == 1 && R == 0) ? 1 : N;
    float dg_sum = 0.f, db_sum = 0.f;              // ld == 0 (threads t < BNF_CH): per-channel parameters, summed over rows / events
    float drstd_e[BNF_MAXE], dmean_e[BNF_MAXE];    // (threads t < BNF_CH; G == 1: this block's sums ARE the totals)
#pragma unroll
    for (int e = 0; e < BNF_MAXE; ++e) drstd_e[e] = dmean_e[e] = 0.f;
#pragma unroll
    for (int e = 0; e < BNF_MAXE; ++e) {
        if (e >= E) break;
        int n0 = (rows == 1) ? 0 : e * npe, n1 = (rows == 1) ? 1 : (e + 1) * npe;
        if (G > 1) {                                // this row's images of the event
            const int ipr = (npe + G - 1) / G;
            n0 = min(e * npe + g * ipr, n1);
            n1 = min(n0 + ipr, n1);
        }
        float drstd = 0.f, dmean = 0.f, dg_e = 0.f, db_e = 0.f;
        for (int nb = n0; nb < n1; nb += BNF_NB) {
            const int ne = min(nb + BNF_NB, n1);
            // 1. {dt, ds} of every (image, channel): slots folded sequentially
            for (int n = nb + ph; n < ne; n += 16) {
                float acc = 0.f;
                if (valid) {
                    if (R > 0) {
                        const float* p = dscale + ((long)n * R * 2 + which) * C + c;      // slot layout: [0] = sum d (d shift), [1] = sum d x (d scale)
                        for (int r0 = 0; r0 < R; r0 += 16) {        // 16 independent loads in flight, added in slot order
                            float buf[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k) {
                                buf[k] = p[(long)min(r0 + k, R - 1) * 2 * C];
                                if (r0 + k >= R) buf[k] = 0.f;
                            }
#pragma unroll
                            for (int k = 0; k < 16; ++k) acc += buf[k];
                        }
                    } else {
                        acc = (which ? dscale : dshift)[(long)n * C + c];
                    }
                }
                v[n - nb][j] = acc;
            }
            __syncthreads();
            // 2. per (image, channel): conditional-parameter gradients and the contributions to d rstd / d mean
            for (int idx = t; idx < (ne - nb) * BNF_CH; idx += 1024) {
                const int nl = idx / BNF_CH, cc = idx % BNF_CH, n = nb + nl;
                float a_ = 0.f, b_ = 0.f, g_ = 0.f, h_ = 0.f;
                if (c0 + cc < C) {
                    const float mean = mean_rstd[(long)e * 2 * C + c0 + cc], rstd = mean_rstd[(long)e * 2 * C + C + c0 + cc];
                    const float dt = v[nl][cc], ds = v[nl][BNF_CH + cc];
                    const float gg = gain[(long)n * ld + c0 + cc] + (plus_one ? 1.f : 0.f);
                    const float ee = ds - dt * mean;          // d/d(scale) with shift = bias - mean*scale folded in
                    if (ld != 0) {
                        dgain[(long)n * ldd + c0 + cc] = ee * rstd;
                        dbias[(long)n * ldd + c0 + cc] = dt;
                    } else {
                        g_ = ee * rstd;
                        h_ = dt;
                    }
                    a_ = ee * gg;
                    b_ = -dt * rstd * gg;
                }
                w[nl][cc] = g_;
                w[nl][BNF_CH + cc] = h_;
                // (v[nl][cc] / v[nl][BNF_CH + cc] were read by this thread only: overwrite in place)
                v[nl][cc] = a_;
                v[nl][BNF_CH + cc] = b_;
            }
            __syncthreads();
            // 3. channel sums over the images of this pass, in image order
            if (t < BNF_CH) {
                for (int nl = 0; nl < ne - nb; ++nl) {
                    drstd += v[nl][t];
                    dmean += v[nl][BNF_CH + t];
                    dg_e += w[nl][t];
                    db_e += w[nl][BNF_CH + t];
                }
            }
            __syncthreads();
        }
        if (t < BNF_CH && valid) {
            if (G == 1) {
                drstd_e[e] = drstd;
                dmean_e[e] = dmean;
                dg_sum += dg_e;
                db_sum += db_e;
            } else {
                float* pay = scratch + tick_floats + (((long)e * G + g) * 4) * C + c;
                sc1_store(pay, drstd);
                sc1_store(pay + C, dmean);
                sc1_store(pay + 2 * C, dg_e);
                sc1_store(pay + 3 * C, db_e);
            }
        }
    }
    if (G > 1) {
        if (!last_row_arrives((int*)scratch, cb, G, &s_flag)) return;
        if (t < BNF_CH && valid) {
#pragma unroll
            for (int e = 0; e < BNF_MAXE; ++e) {
                if (e >= E) break;
                float buf[8][4];                    // all G (<= 8) partial rows requested before the first add
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    const float* pay = scratch + tick_floats + (((long)e * G + min(gg, G - 1)) * 4) * C + c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) buf[gg][q] = sc1_load(pay + q * C);
                }
                float a_ = 0.f, b_ = 0.f;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    if (gg < G) {
                        a_ += buf[gg][0];
                        b_ += buf[gg][1];
                        dg_sum += buf[gg][2];
                        db_sum += buf[gg][3];
                    }
                }
                drstd_e[e] = a_;
                dmean_e[e] = b_;
            }
        }
    }
    if (t < BNF_CH && valid) {
        if (dstat) {
#pragma unroll
            for (int e = 0; e < BNF_MAXE; ++e) {
                if (e >= E) break;
                float* de = dstat + (long)e * 2 * C;
                if (training) {
                    const float mean = mean_rstd[(long)e * 2 * C + c], rstd = mean_rstd[(long)e * 2 * C + C + c];
                    const float dvar = -0.5f * rstd * rstd * rstd * drstd_e[e];
                    de[C + c] = dvar / count;                          // d sumsq
                    de[c] = (dmean_e[e] - 2.f * mean * dvar) / count;  // d sum
                } else {
                    de[c] = 0.f;
                    de[C + c] = 0.f;
                }
            }
        }
        if (ld == 0) {
            dgain[c] = dg_sum;
            dbias[c] = db_sum;
        }
    }
}

extern "C" long ieagan_bn_finalize_bwd_scratch(int N, int C, int E, int acc_repl) {
    if (E < 1) E = 1;
    const int G = bnb_rows(N / E, acc_repl, N);
    return G <= 1 ? 0 : bnf_tick_floats(C) + (long)E * G * 4 * C;
}

extern "C" int ieagan_bn_finalize_bwd(const float* dscale, const float* dshift, const float* gain, int ld, int plus_one,
                                      const float* mean_rstd, float count, int training, float* dgain, float* dbias,
                                      int ldd, float* dstat, int N, int C, int E, int acc_repl, float* scratch, void* stream) {
    if (E < 1) E = 1;
    CHECK_ARG(N % E == 0 && E <= BNF_MAXE, "bn_finalize_bwd: %d images are not %d (<= %d) whole events", N, E, BNF_MAXE);
    CHECK_ARG(acc_repl >= 0 && acc_repl <= 4096, "bn_finalize_bwd: bad replica count %d", acc_repl);
    const int rows = (ld == 0 && E == 1 && acc_repl == 0) ? 1 : N;
    int G = bnb_rows(N / E, acc_repl, rows);
    if (scratch == nullptr) G = 1;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("bn_finalize_bwd", 0.0, 0.0, st);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, BNF_CH), G), dim3(1024), 0, st, dscale, dshift, gain, ld, plus_one,
                       mean_rstd, count, training, dgain, dbias, ldd, dstat, N, C, E, acc_repl, scratch, bnf_tick_floats(C));
    CHECK_LAUNCH("bn_finalize_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// res_bwd: gradient of the conv epilogue's residual operand A (channels [0, Ca) of a tensor with
// Cr channels): mode 1 -> residual lived at half resolution (sum the 2x2 block), mode 2 -> at double
// resolution (0.25 * expand).  Channels [Ca, Cr) get zeros.  g: [N,H,W,Cg].  dr: [N,Hr,Wr,Cr].
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void res_bwd_kernel(const bf16* __restrict__ g, int Cg, bf16* __restrict__ dr, int Cr, int Ca,
                                                      int N, int Hr, int Wr) {
    const int groups = Cr >> 3;
    const long chunks = (long)N * Hr * Wr * groups;
    const int H = (MODE == 1) ? 2 * Hr : Hr / 2, W = (MODE == 1) ? 2 * Wr : Wr / 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < chunks; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % groups);
        const long sp = idx / groups;
        const int ws = (int)(sp % Wr);
        const long t = sp / Wr;
        const int hs = (int)(t % Hr), n = (int)(t / Hr);
        bf16x8 o = zero8();
        if (cg * 8 < Ca) {
            if (MODE == 1) {
                float d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 v = *(const bf16x8*)(g + (((long)n * H + 2 * hs + (q >> 1)) * W + 2 * ws + (q & 1)) * Cg + cg * 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[i] += bf2f(v[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = f2bf(d[i]);
            } else {
                const bf16x8 v = *(const bf16x8*)(g + (((long)n * H + (hs >> 1)) * W + (ws >> 1)) * Cg + cg * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = f2bf(0.25f * bf2f(v[i]));
            }
        }
        *(bf16x8*)(dr + idx * 8) = o;
    }
}

extern "C" int ieagan_res_bwd(const void* g, int Cg, void* dr, int Cr, int Ca, int mode, int N, int Hr, int Wr, void* stream) {
    CHECK_ARG(mode == 1 || mode == 2, "res_bwd: mode must be 1 (upsampled residual) or 2 (pooled residual)");
    CHECK_ARG(Cr % 8 == 0 && Ca % 8 == 0 && Ca <= Cr && Ca <= Cg, "res_bwd: bad channel counts");
    CHECK_ARG(mode != 2 || (Hr % 2 == 0 && Wr % 2 == 0), "res_bwd: pooled residual needs even size");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("res_bwd", 0.0, 2.0 * N * Hr * Wr * (double)Cr * (mode == 1 ? 5.0 : 1.25), st);
    long blocks = ((long)N * Hr * Wr * (Cr / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (mode == 1) hipLaunchKernelGGL((res_bwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)g, Cg, (bf16*)dr, Cr, Ca, N, Hr, Wr);
    else hipLaunchKernelGGL((res_bwd_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)g, Cg, (bf16*)dr, Cr, Ca, N, Hr, Wr);
    CHECK_LAUNCH("res_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// module-boundary layout changes.  NCHW fp32 [N,C,HW] <-> NHWC bf16 [N,HW,C] through an LDS tile of
// 32 pixels x 32 channels so that both sides stay coalesced.  Optional (sum, sumsq) statistics of the
// bf16-rounded values for a BatchNorm that consumes the converted tensor.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, bf16* __restrict__ out,
                                                           float* __restrict__ stats, int C, int HW, int npe) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? in[((long)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;   // channel c0+tx over the pixels this thread writes
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < C) {
            const bf16 v = f2bf(tile[tx][j]);
            out[((long)n * HW + p) * C + c] = v;
            const float f = bf2f(v);
            s1 += f;
            s2 += f * f;
        }
    }
    if (stats != nullptr) {
        // one slot per (image of the event, 32-pixel block): the 8 pixel phases (ty) of a channel are folded through LDS in a fixed
        // order, the slot has a single adder -> bit-reproducible statistics
        __syncthreads();
        tile[ty][tx] = s1;
        tile[8 + ty][tx] = s2;
        __syncthreads();
        if (ty == 0 && c0 + tx < C) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a1 += tile[j][tx];
                a2 += tile[8 + j][tx];
            }
            const int ipe = npe > 0 ? npe : (int)gridDim.z;
            const int event = n / ipe;
            const long slot = (long)(n - event * ipe) * gridDim.x + blockIdx.x;
            float* st = stats + ((long)event * ipe * gridDim.x + slot) * 2 * C;
            atomicAdd(st + c0 + tx, a1);
            atomicAdd(st + C + c0 + tx, a2);
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16* __restrict__ in, float* __restrict__ out, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        tile[j][tx] = (c < C && p < HW) ? bf2f(in[((long)n * HW + p) * C + c]) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (c < C && p < HW) out[((long)n * C + c) * HW + p] = tile[tx][j];
    }
}

extern "C" int ieagan_nchw_to_nhwc(const float* in, void* out, float* stats, int N, int C, int HW, int n_per_event, void* stream) {
    CHECK_ARG(n_per_event >= 0 && (n_per_event == 0 || N % n_per_event == 0), "nchw_to_nhwc: N=%d is not a whole number of events of %d images", N, n_per_event);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("nchw_to_nhwc", 0.0, 6.0 * N * C * (double)HW, st);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), N), dim3(256), 0, st, in, (bf16*)out, stats, C, HW, n_per_event);
    CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}

extern "C" int ieagan_nhwc_to_nchw(const void* in, float* out, int N, int C, int HW, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("nhwc_to_nchw", 0.0, 6.0 * N * C * (double)HW, st);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), N), dim3(256), 0, st, (const bf16*)in, out, C, HW);
    CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}

// per-channel (sum, sumsq) of an NHWC bf16 tensor -> replicated stats (used when the producer was not
// one of the conv kernels).
__global__ __launch_bounds__(256) void stats_kernel(const bf16* __restrict__ x, float* __restrict__ stats, long P, int C) {
    __shared__ float red[256 * 8];
    const int groups = C >> 3;
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    const long chunks = P * groups;
    const int act = active_threads(groups);
    for (long idx = (long)blockIdx.x * act + threadIdx.x; idx < chunks && threadIdx.x < act; idx += (long)gridDim.x * act) {
        const bf16x8 v = *(const bf16x8*)(x + idx * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float f = bf2f(v[i]);
            s1[i] += f;
            s2[i] += f * f;
        }
    }
    float* st = stats + (long)(blockIdx.x % STAT_REPL) * 2 * C;
    reduce_groups_atomic(s1, groups, st, red);
    reduce_groups_atomic(s2, groups, st + C, red);
}

extern "C" int ieagan_channel_stats(const void* x, float* stats, long P, int C, void* stream) {
    CHECK_ARG(C % 8 == 0 && C <= 2048, "channel_stats: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("channel_stats", 0.0, 2.0 * P * C, st);
    long blocks = (P * (C / 8) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, stats, P, C);
    CHECK_LAUNCH("channel_stats");
    return 0;
}
