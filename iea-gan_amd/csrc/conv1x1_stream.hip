// conv1x1_stream: the 1x1 convolutions on LARGE feature maps (SNConv2d 1x1 of GBlock / DBlock, model.py:34-45, 512-532) as a
// streaming kernel.  These layers are pure HBM traffic (arithmetic intensity 10-130 FLOP/B): what matters is how many
// bytes every CU keeps in flight, not the MFMA schedule.  Compared with conv_gather (one 128-pixel tile per block, every load
// exposed, weights / BatchNorm table / statistics flush paid per tile):
//   * a block walks `gpb` consecutive pixel groups; the A fragments of group g+1 and the epilogue operands of group g (ReLU mask,
//     same-resolution shortcut tensors) are requested BEFORE the MFMAs of group g, so every wave always has 2-8 KB in flight;
//   * the weight fragments (K = Cin <= 128) live in registers for the whole block;
//   * the per-image BatchNorm scale / shift table is staged in LDS once per image, the statistics of the next BatchNorm are
//     flushed once per block (per event) instead of once per 128 pixels.
// Same operator contract as conv_gather for taps == 1, src.rs == 0 (prologue: per-(n,c) affine + ReLU; epilogue: bias, ReLU mask,
// residual A with its own resample / channel slice, residual B, per-event statistics).
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"

template <bool AFF, bool RELU, int NT, int KS, int MT>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(ConvArgs a, int gpb, int gpe, int nblk, int bpe) {
    constexpr int MTS = (NT == 1) ? 2 : 1;            // m-tiles per epilogue pass (a pass must fill whole waves with 16-byte chunks)
    constexpr int CPP = NT * 2;                       // 8-channel chunks per pixel
    constexpr int ITER = 16 * MTS * CPP / 64;         // chunks per lane per pass
    constexpr int EP = MT / MTS;                      // epilogue passes per group
    constexpr int LDW = NT * 16 + 4;                  // padded transpose row (floats)
    constexpr int GP = 4 * MT * 16;                   // pixels per block and group
    static_assert(MT % MTS == 0 && ITER >= 1, "bad tile shape");
    static_assert(4 * 16 * MTS * LDW >= 4 * STATS_SX_FLOATS || true, "");
    __shared__ __attribute__((aligned(16))) float epi[(4 * 16 * MTS * LDW > 4 * STATS_SX_FLOATS) ? 4 * 16 * MTS * LDW : 4 * STATS_SX_FLOATS];
    __shared__ float red[4 * NT * 16 * 2];
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * AFF_MAXC : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W;
    const long M = (long)a.N * HW;
    const int Cin = a.Cin;
    // XCD-aware order (blocks b and b + 8 share an XCD): every XCD walks one contiguous run of pixel groups.  The n-tile columns of a
    // layer with more output channels than one block takes (gy > 1) are the FAST index of that order: the gy blocks that read the same
    // pixels then run at the same time on the same XCD and share the activations through its L2 -- as blockIdx.y they were whole
    // grid passes apart and every column streamed the input from memory again (5.7 GB per step over the 1x1 layers).
    const int gy = (a.Cout + 16 * NT - 1) / (16 * NT);
    int bid = blockIdx.x;
    {
        const int tot = nblk * gy, q = tot / 8, r = tot % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n_base = (bid % gy) * NT * 16;
    bid /= gy;
    // a block stays inside ONE statistics group (event; gpe pixel groups, bpe blocks): a single flush at its end
    const int event = bid / bpe;
    const int g0 = event * gpe + (bid - event * bpe) * gpb;
    const int g1 = min(g0 + gpb, (event + 1) * gpe);
    if (g0 >= g1) return;

    // ---- weights: [Cout][Kpad] -> B fragments in registers (lane: column lr of n-tile nt, k = ks*32 + lg*8 ..)
    const bool col_ok = (n_base + (NT - 1) * 16 + lr) < a.Cout;          // only NT == 1 can have a half-empty n-tile
    bf16x8 bfrag[KS][NT];
    {
        const bf16* wrow = (const bf16*)a.w + (long)(col_ok ? n_base + lr : 0) * a.Kpad + lg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bfrag[ks][nt] = *(const bf16x8*)(wrow + (long)nt * 16 * a.Kpad + ks * 32);
                if (NT == 1 && !col_ok) bfrag[ks][nt] = zero8();
            }
    }
    // ---- fixed chunk geometry of this lane in the epilogue
    const int cc = lane % CPP;
    const int co0 = n_base + cc * 8;
    const bool ch_ok = co0 < a.Cout;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = (a.bias && ch_ok) ? a.bias[co0 + i] : 0.f;
    const bool has_mask = a.mask != nullptr;
    const bool bnb = !AFF && !RELU && a.bnb_scale != nullptr;
    const bool ra_same = a.ra != nullptr && a.ra_rs == 0 && co0 < a.Ca;
    const bool ra_up = a.ra != nullptr && a.ra_rs == 1 && co0 < a.Ca;
    const bool ra_other = a.ra != nullptr && a.ra_rs == 2 && co0 < a.Ca;
    const bool has_rb = a.rb != nullptr && !(a.ra != nullptr && co0 < a.Ca) && co0 >= a.Ca;
    const bf16* res_ptr = ra_same ? (const bf16*)a.ra + co0 : (has_rb ? (const bf16*)a.rb + (co0 - a.Ca) : nullptr);
    const long res_stride = ra_same ? a.Cra : a.Crb;
    const float res_scale = (ra_same || ra_up) ? a.ra_scale : 1.f;
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    float* wlds = epi + wave * 16 * MTS * LDW;

    bf16x8 cur[MT][KS], nxt[MT][KS];
    auto load_A = [&](int g, bf16x8(&r)[MT][KS]) {
        const long wbase = (long)g * GP + wave * (MT * 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long m = wbase + mt * 16 + lr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 32 + lg * 8;
                r[mt][ks] = zero8();
                if (m < M && k < Cin) r[mt][ks] = *(const bf16x8*)((const bf16*)a.src.x + m * a.src.Cx + k);
            }
        }
    };
    // Software pipeline of depth one over the pixel groups: at the top of iteration g EVERY operand of group g (A fragments, ReLU
    // mask, residual chunks) is already in registers -- requested during iteration g-1 -- and one explicit wait retires them;
    // then everything of group g+1 is requested and group g is computed without touching the vector-memory counter again.
    // (Requesting this group's epilogue operands and the next group's A fragments at the same point, as before, made the
    // compiler drain the counter -- vmcnt(0) -- in front of the first MFMA: the loads sit under per-lane conditions, so it cannot
    // count how many younger loads may stay in flight.)
    bf16x8 pm[EP][ITER], pr[EP][ITER], pm_n[EP][ITER], pr_n[EP][ITER];   // pr: same-resolution residual operand (A or B, by chunk)
    auto load_epi = [&](int g, bf16x8(&m_)[EP][ITER], bf16x8(&r_)[EP][ITER]) {
        const long gb = (long)g * GP;
        const int n_im = (int)(gb / HW);
        const long wb = gb + wave * (MT * 16);
#pragma unroll
        for (int ep = 0; ep < EP; ++ep)
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const long m = wb + ep * (MTS * 16) + (it * 64 + lane) / CPP;
                const bool ok = m < M && ch_ok;
                if (has_mask && ok) m_[ep][it] = *(const bf16x8*)((const bf16*)a.mask + m * a.Cout + co0);
                // (else-if: two conditional loads into the same registers made the second one wait for every outstanding load)
                if (res_ptr != nullptr && ok) r_[ep][it] = *(const bf16x8*)(res_ptr + m * res_stride);
                else if (ra_up && ok) {     // shortcut at half resolution (nearest x2): pixel (n, h, w) reads (n, h/2, w/2)
                    const int rem = (int)(m - (long)n_im * HW);
                    const int h = rem / W, w = rem - h * W;
                    r_[ep][it] = *(const bf16x8*)((const bf16*)a.ra + (((long)n_im * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * a.Cra + co0);
                }
            }
    };
    load_A(g0, cur);
    load_epi(g0, pm, pr);
    int aff_n = -1;
    float rsc[8], rsh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rsc[i] = rsh[i] = 0.f;
    for (int g = g0; g < g1; ++g) {
        const long gbase = (long)g * GP;
        const int n_img = (int)(gbase / HW);                       // launcher: HW % GP == 0 -> one image per group
        if (AFF && n_img != aff_n) {                              // block-uniform
            __syncthreads();
            stage_aff(aff_s, a.src, n_img, Cin);
            aff_n = n_img;
            __syncthreads();
            if (KS == 1) {          // Cin <= 32: a lane's channel chunk never changes -- its scale / shift stay in registers for the image
                const int c0 = (lg * 8) % AFF_MAXC;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rsc[i] = aff_s[c0 + i];
                    rsh[i] = aff_s[AFF_MAXC + c0 + i];
                }
            }
        }
        const long wbase = gbase + wave * (MT * 16);
        // group g's operands (requested one iteration ago) have landed.  NOT vmcnt(0): on gfx9 the counter also holds the STORES of
        // group g-1, which were issued after those loads -- draining them put the write latency of every group on the critical path
        // of the next one.  The EP * ITER store instructions of a group are always issued (whole groups only: HW % GP == 0).
        __builtin_amdgcn_s_waitcnt(0x0F70 | (EP * ITER));
        if (g + 1 < g1) {
            load_A(g + 1, nxt);
            load_epi(g + 1, pm_n, pr_n);
        }
        // ---- compute + epilogue, one pass of MTS m-tiles at a time (K is tiny: the accumulators of a pass die right away)
#pragma unroll
        for (int ep = 0; ep < EP; ++ep) {
            f32x4 acc[MTS][NT];
#pragma unroll
            for (int ms = 0; ms < MTS; ++ms) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[ms][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 af = cur[ep * MTS + ms][ks];
                    if (AFF || RELU) {
                        if (!AFF) {
                            af = relu8(af);
                        } else {
                            float v[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = bf2f(af[i]);
                            if (KS == 1) {          // (four LDS reads and their wait per m-tile otherwise)
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    v[i] = v[i] * rsc[i] + rsh[i];
                                    if (RELU) v[i] = fmaxf(v[i], 0.f);
                                }
                            } else {
                                xform8<AFF, RELU>(v, a.src, n_img, (ks * 32 + lg * 8) % AFF_MAXC, aff_s);
                            }
#pragma unroll
                            for (int i = 0; i < 8; ++i) af[i] = f2bf(v[i]);
                            if (ks * 32 + lg * 8 >= Cin) af = zero8();     // K padding must stay zero after the shift
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[ms][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[ks][nt], acc[ms][nt], 0, 0, 0);
                }
            }
            // transpose through the wave-private LDS buffer: lane then owns 8 channels of one pixel
#pragma unroll
            for (int ms = 0; ms < MTS; ++ms)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) wlds[(ms * 16 + lg * 4 + r) * LDW + nt * 16 + lr] = acc[ms][nt][r];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int row = (it * 64 + lane) / CPP;
                const long m = wbase + ep * (MTS * 16) + row;
                const f32x4 lo = *(const f32x4*)(wlds + row * LDW + cc * 8);
                const f32x4 hi = *(const f32x4*)(wlds + row * LDW + cc * 8 + 4);
                if (!(m < M && ch_ok)) continue;
                float v[8] = {lo[0] + bv[0], lo[1] + bv[1], lo[2] + bv[2], lo[3] + bv[3], hi[0] + bv[4], hi[1] + bv[5], hi[2] + bv[6], hi[3] + bv[7]};
                if (bnb) {                  // BatchNorm-apply backward folded in (see ieagan_conv_desc.bnb_*): pm holds x
                    const long so = (long)n_img * a.bnb_nstride + co0;
                    const f32x4 sc0 = *(const f32x4*)(a.bnb_scale + so), sc1 = *(const f32x4*)(a.bnb_scale + so + 4);
                    const f32x4 sh0 = *(const f32x4*)(a.bnb_shift + so), sh1 = *(const f32x4*)(a.bnb_shift + so + 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float xf = bf2f(pm[ep][it][i]);
                        const float sc = i < 4 ? sc0[i & 3] : sc1[i & 3], sh = i < 4 ? sh0[i & 3] : sh1[i & 3];
                        const float d = (a.bnb_relu && !(xf * sc + sh > 0.f)) ? 0.f : v[i];
                        s1[i] += d;
                        s2[i] += d * xf;
                        v[i] = d * sc;
                    }
                } else if (has_mask) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (bf2f(pm[ep][it][i]) > 0.f) ? v[i] : 0.f;
                }
                if (res_ptr != nullptr || ra_up) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += res_scale * bf2f(pr[ep][it][i]);
                } else if (ra_other) {              // operand lives at double resolution: 2x2 average (not prefetched)
                    const int rem = (int)(m - (long)n_img * HW);
                    const int h = rem / W, w = rem - h * W;
                    const bf16* p = (const bf16*)a.ra + (((long)n_img * (2 * H) + 2 * h) * (2 * W) + 2 * w) * a.Cra + co0;
                    const long rs_ = (long)2 * W * a.Cra;
                    const bf16x8 t0 = *(const bf16x8*)p, t1 = *(const bf16x8*)(p + a.Cra), t2 = *(const bf16x8*)(p + rs_),
                                 t3 = *(const bf16x8*)(p + rs_ + a.Cra);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += a.ra_scale * 0.25f * (bf2f(t0[i]) + bf2f(t1[i]) + bf2f(t2[i]) + bf2f(t3[i]));
                }
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    o[i] = f2bf(v[i]);
                    if (!bnb) {
                        s1[i] += v[i];
                        s2[i] += v[i] * v[i];
                    }
                }
                *(bf16x8*)((bf16*)a.out + m * a.Cout + co0) = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the next pass overwrites the wave's buffer
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) cur[mt][ks] = nxt[mt][ks];
#pragma unroll
        for (int ep = 0; ep < EP; ++ep)
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                pm[ep][it] = pm_n[ep][it];
                pr[ep][it] = pr_n[ep][it];
            }
    }
    if (a.stats != nullptr) {
        __syncthreads();          // every wave has left its transpose buffer: it now serves as the fold scratch
        stats_flush<NT>(a, s1, s2, n_base, red, epi, bid, event);
    }
}

// Launch.  Returns 1 when the layer was taken, 0 when it must go to conv_gather (small maps, unusual shapes).
template <bool AFF, bool RELU, int NT, int KS, int MT>
static void stream_launch_one(const ConvArgs& a, hipStream_t st) {
    constexpr int GP = 4 * MT * 16;
    const long M = (long)a.N * a.H * a.W;
    const int ngroups = (int)(M / GP);                  // exact: H*W % GP == 0 (checked by the caller)
    const int n_events = (a.stats != nullptr && a.n_per_event > 0) ? a.N / a.n_per_event : 1;
    const int gpe = ngroups / n_events;
    // ~8 blocks per CU and n-tile column; at least 4 groups per block so that the prefetch pipeline has something to overlap
    int gpb = (ngroups + 2047) / 2048;
    if (gpb < 4) gpb = 4;
    if (gpb > 32) gpb = 32;
    if (gpb > gpe) gpb = gpe;
    const int bpe = (gpe + gpb - 1) / gpb;
    const int nblk = bpe * n_events;
    CONV_PLAN_POINT(bpe, )
    hipLaunchKernelGGL((conv1x1_stream_kernel<AFF, RELU, NT, KS, MT>), dim3(nblk * ((a.Cout + 16 * NT - 1) / (16 * NT))), dim3(256), 0, st, a,
                       gpb, gpe, nblk, bpe);
}

template <bool AFF, bool RELU>
static int stream_dispatch(const ConvArgs& a, hipStream_t st) {
    // (Cin -> KS k-steps, MT m-tiles per wave and group) x (Cout -> NT n-tiles per block)
#define SL(NTV, KSV, MTV) { stream_launch_one<AFF, RELU, NTV, KSV, MTV>(a, st); return 1; }
#define BY_NT(KSV, MTV)                                   \
    if (a.Cout % 64 == 0) SL(4, KSV, MTV)                  \
    else if (a.Cout % 32 == 0) SL(2, KSV, MTV)             \
    else SL(1, KSV, MTV)
    if (a.Cin <= 16) {      // 16 -> 64 keeps 2 m-tiles: 4 would spill (8 prefetched epilogue chunks per operand and lane)
        if (a.Cout % 64 == 0) SL(4, 1, 2)
        BY_NT(1, 4)
    }
    else if (a.Cin <= 32) { BY_NT(1, 2) }
    else if (a.Cin <= 64) { BY_NT(2, 2) }
    else {                  // Cin = 128: 4 k-steps of weights in registers leave room for at most 2 n-tiles
        if (a.Cout % 32 == 0) SL(2, 4, 2)
        SL(1, 4, 2)
    }
#undef BY_NT
#undef SL
    return 0;
}

int conv1x1_stream_launch(const ConvArgs& a, hipStream_t st) {
    if (a.taps != 1 || a.src.rs != 0 || a.Cin > 128 || a.Kpad != ((a.Cin + 31) / 32) * 32) return 0;
    // wide expansions go to conv1x1_tile (one LDS copy of 128 pixels serves 64 couts; here every 32 / 64 couts re-fetch the fragments):
    // 128 -> 256 @32x96 52.9 -> 42.0 us, 128 -> 128 29.2 -> 24.8, 64 -> 256 37.1 -> 33.4 (32 -> 128 / 256 and 128 -> 64 stay: equal or slower there)
    if ((a.Cin == 128 && a.Cout >= 128) || (a.Cin == 64 && a.Cout >= 256)) return 0;
    const long HW = (long)a.H * a.W, M = (long)a.N * HW;
    const int mt = (a.Cin <= 16 && a.Cout % 64 != 0) ? 4 : 2;
    const int gp = 4 * mt * 16;
    if (HW % gp != 0 || M / gp < 480) return 0;                    // >= 480 pixel groups (32x96 maps and larger)
    if (a.src.scale != nullptr && a.Cin > AFF_MAXC) return 0;
    if (a.ra && a.ra_rs == 1 && (a.H % 2 || a.W % 2)) return 0;
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (aff && relu) return stream_dispatch<true, true>(a, st);
    if (aff) return stream_dispatch<true, false>(a, st);
    if (relu) return stream_dispatch<false, true>(a, st);
    return stream_dispatch<false, false>(a, st);
}
