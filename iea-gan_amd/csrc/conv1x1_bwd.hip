// conv1x1_bwd: the WHOLE backward of a 1x1 convolution on a large feature map in ONE launch -- data gradient, weight gradient,
// bias gradient, the batch-statistics term of the following BatchNorm (effgrad) on the way in and the backward of the fused
// prologue (ReLU mask / BatchNorm apply / 2x2 average pool) on the way out.  Replaces, per layer and backward pass, the chain
//     effgrad (2R + 1W)  ->  conv_forward as dgrad (R g, R x as mask, W dx)  [-> prologue_bwd (R da, R x, W dx)]  ->  conv_wgrad (R x, R g)
// of autograd(F.conv2d(relu(bn(x)) | avg_pool(relu(x)), W / sigma, b)) (reference layers.py:197-206, 656-689; model.py:54-71, 541-557).
// These layers are pure HBM traffic (10-130 FLOP/B): every operand tile (g, the conv output y for effgrad, x, the shortcut gradient)
// is read ONCE, dx is written once.
//
// Structure: a block is four AUTONOMOUS waves.  A wave owns tiles of 32 consecutive conv pixels; per tile it
//   1. retires the raw 16-byte chunks of g / y / x it requested one tile ago, forms g_eff = g + dsum[c] + 2 y dsumsq[c] and writes the
//      g_eff and x (pooled + activated for a pooled source) tiles into its PRIVATE LDS region (rows padded to an odd multiple of 32 B),
//   2. requests the shortcut-gradient chunks of this tile and the operands of its next tile,
//   3. weight gradient: dW[Cout][Cin] += g_eff^T a  -- the 32 pixels are exactly one MFMA K step; both operands are read K(pixel)-major
//      with ds_read_b64_tr_b16 (the prologue is applied to the B fragment: a lane holds 8 pixels of ONE channel, so scale / shift are
//      two scalars); accumulators stay in registers over all tiles of the wave; an all-ones B tile yields the bias column sums,
//   4. data gradient: da = g_eff W  (A fragments = 16-byte rows of the g tile, B = the transposed weight pack in registers), transposed
//      through a wave-private LDS buffer so that every lane owns 8 channels of one pixel; the prologue backward reads x from the LDS
//      tile (no second global read), adds the shortcut gradient and stores dx with 16-byte stores.
// No block barrier inside the tile loop: waves drift apart and cover each other's memory latency.  At the end the four partial dW
// are folded through LDS atomics and leave the block as ONE slab store (two-stage accumulation) or one round of global atomics.
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"
#include <stdio.h>

typedef ieagan_conv1x1_bwd_desc Bwd1Args;


__host__ __device__ constexpr int b1_stride(int C) { return ((C / 16) % 2 == 0) ? C + 16 : C; }      // elements; odd multiple of 32 bytes

// K(pixel)-major fragment: rows pix0 .. pix0+3 and pix0+8 .. pix0+11 of column (col0 + lr) of a [pixel][channel] 16-bit LDS image
// (see conv_wgrad.hip: frag_T).  EXEC must be all ones.
__device__ __forceinline__ bf16x8 b1_frag_T(const bf16* lds, int stride_elems, int pix0, int col0, int lr) {
    const int q = lr >> 2, p = lr & 3;
    const bf16* p0 = lds + (pix0 + q) * stride_elems + col0 + 4 * p;
    const bf16* p1 = p0 + 8 * stride_elems;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hh[0]; f[5] = hh[1]; f[6] = hh[2]; f[7] = hh[3];
    return f;
}

// TP: conv-resolution pixels per wave tile (32 or 64).  64 keeps twice the bytes in flight per wave -- these launches wait on memory
// (rocprofv3: 40-70 % of the wave cycles parked in s_waitcnt at two waves per SIMD) -- where the LDS tiles of two blocks still fit a CU.
template <int CIN, int COUT, int RS, bool AFF, int TP>
struct B1Cfg {
    static constexpr int GS = b1_stride(COUT), XS = b1_stride(CIN);
    static constexpr int CPG = COUT / 8, CPX = CIN / 8;           // 16-byte chunks per pixel
    static constexpr int GCH = TP * CPG / 64;                   // g chunks per lane and tile
    static constexpr int XIT = TP * CPX / 64;                   // x items per lane and tile (an item = one chunk; 2x2 chunks when pooled)
    static constexpr int KS_D = (COUT + 31) / 32, NT_D = CIN / 16; // dgrad: k-steps over cout, n-tiles over cin
    static constexpr int MT_W = COUT / 16, NJ_W = CIN / 16;        // wgrad: m-tiles over cout, n-tiles over cin
    static constexpr int EROWS = (CIN == 16) ? 32 : 16;            // pixel rows per epilogue pass (a pass must fill whole waves)
    static constexpr int EPASS = TP / EROWS;
    static constexpr int EIT = EROWS * CPX / 64;                   // items per lane and pass
    static constexpr int LDW = NT_D * 16 + 4;                      // padded fp32 transpose row
    static constexpr int G_BYTES = TP * GS * 2, X_BYTES = TP * XS * 2, E_BYTES = EROWS * LDW * 4;
    static constexpr int WAVE_BYTES = G_BYTES + X_BYTES + E_BYTES;
    static constexpr int FOLD_BYTES = (COUT * CIN + COUT) * 4;     // block-end fold of the four partial dW (+ column sums)
    static constexpr int SX_BYTES = 4 * STATS_SX_FLOATS * 4;
    static constexpr int SMEM = (4 * WAVE_BYTES > FOLD_BYTES ? 4 * WAVE_BYTES : FOLD_BYTES) > SX_BYTES
                                    ? (4 * WAVE_BYTES > FOLD_BYTES ? 4 * WAVE_BYTES : FOLD_BYTES) : SX_BYTES;
    static_assert(XIT == EPASS * EIT, "staging items and epilogue items must be the same (lane, index) mapping");
    static_assert(64 % CPG == 0 && 64 % CPX == 0, "a lane's channel chunk must not change between its items");
};

template <int CIN, int COUT, int RS, bool AFF, int OCC, int TP>
__global__ __launch_bounds__(256, OCC) void conv1x1_bwd_kernel(Bwd1Args a, int tpi, int tpb, int bpi, int nblk) {
    typedef B1Cfg<CIN, COUT, RS, AFF, TP> K;
    __shared__ __attribute__((aligned(16))) char smem[K::SMEM];
    __shared__ float red[4 * K::NT_D * 16 * 2];
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * CIN : 8];
    __shared__ __attribute__((aligned(32))) float eff_s[2 * COUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W;
    // XCD-aware order: blocks b and b + 8 share an XCD (and its L2): every XCD walks one contiguous run of tiles
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n = bid / bpi;                                      // a block stays inside ONE image (per-image BatchNorm accumulators)
    const int t0 = (bid - n * bpi) * tpb;
    const int t1 = min(t0 + tpb, tpi);
    bf16* lds_g = (bf16*)(smem + wave * K::WAVE_BYTES);
    bf16* lds_x = (bf16*)(smem + wave * K::WAVE_BYTES + K::G_BYTES);
    float* wlds = (float*)(smem + wave * K::WAVE_BYTES + K::G_BYTES + K::X_BYTES);

    const bool relu = a.src.relu != 0;
    const bool eff = a.y != nullptr;
    const bool want_w = a.dw != nullptr;
    const int lmode = a.lg != nullptr ? a.lmode : -1;

    // ---- block prologue: BatchNorm table of this image, dgrad weight fragments, effgrad terms of this lane's g chunk
    if (AFF) {
        for (int i = threadIdx.x; i < CIN; i += 256) {
            aff_s[i] = a.src.scale[(long)n * a.src.aff_nstride + i];
            aff_s[CIN + i] = a.src.shift[(long)n * a.src.aff_nstride + i];
        }
    }
    if (eff) {
        const int ev = (a.n_per_event > 0) ? n / a.n_per_event : 0;
        for (int i = threadIdx.x; i < COUT; i += 256) {
            eff_s[i] = a.dstat[(long)ev * 2 * COUT + i];
            eff_s[COUT + i] = 2.f * a.dstat[(long)ev * 2 * COUT + COUT + i];
        }
    }
    __syncthreads();
    typedef const __attribute__((address_space(3))) float* lds_cf;
    bf16x8 bfrag[K::KS_D][K::NT_D];                               // B[k = cout][col = cin]: rows of the transposed pack [Cin][Kpad2]
#pragma unroll
    for (int ks = 0; ks < K::KS_D; ++ks)
#pragma unroll
        for (int nt = 0; nt < K::NT_D; ++nt) {
            const int k = ks * 32 + lg * 8;
            bfrag[ks][nt] = (k < COUT) ? *(const bf16x8*)((const bf16*)a.w_bwd + (long)(nt * 16 + lr) * a.Kpad2 + k) : zero8();
        }
    const int ccg = lane % K::CPG, ccx = lane % K::CPX;          // this lane's channel chunk in g / in x (fixed over its items)
    typedef const __attribute__((address_space(3))) f32x4* lds_cf4;
    const float* dsl = eff_s + ccg * 8;                           // {dsum, 2 dsumsq} rows of this lane's g chunk (LDS)
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    f32x4 accw[K::MT_W][K::NJ_W], accc[K::MT_W];
#pragma unroll
    for (int mt = 0; mt < K::MT_W; ++mt) {
        accc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nj = 0; nj < K::NJ_W; ++nj) accw[mt][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;

    const long img_px = (long)n * HW;                             // first conv pixel of this image
    const int Hs = a.src.Hs, Ws = a.src.Ws;

    // ---- raw operand requests of one tile (no transform)
    bf16x8 rg[K::GCH], ry[K::GCH], rx[K::XIT][RS == 2 ? 4 : 1];
    auto request = [&](int t) {
        const long p0 = img_px + (long)t * TP;
#pragma unroll
        for (int j = 0; j < K::GCH; ++j) {
            const int px = (j * 64 + lane) / K::CPG;
            rg[j] = *(const bf16x8*)((const bf16*)a.g + (p0 + px) * a.Cg + ccg * 8);
            if (eff) ry[j] = *(const bf16x8*)((const bf16*)a.y + (p0 + px) * COUT + ccg * 8);
        }
        if (RS == 0) {
#pragma unroll
            for (int j = 0; j < K::XIT; ++j) {
                const int px = (j * 64 + lane) / K::CPX;
                rx[j][0] = *(const bf16x8*)((const bf16*)a.src.x + (p0 + px) * a.src.Cx + ccx * 8);
            }
        } else {
            const int tl = t * TP;                             // W % 32 == 0: a tile lies inside one conv row
            const int h = tl / W, w0 = tl - h * W;
#pragma unroll
            for (int j = 0; j < K::XIT; ++j) {
                const int px = (j * 64 + lane) / K::CPX;
                const bf16* p = (const bf16*)a.src.x + (((long)n * Hs + 2 * h) * Ws + 2 * (w0 + px)) * a.src.Cx + ccx * 8;
                rx[j][0] = *(const bf16x8*)p;
                rx[j][1] = *(const bf16x8*)(p + a.src.Cx);
                rx[j][2] = *(const bf16x8*)(p + (long)Ws * a.src.Cx);
                rx[j][3] = *(const bf16x8*)(p + (long)Ws * a.src.Cx + a.src.Cx);
            }
        }
    };

    if (t0 + wave < t1) request(t0 + wave);
    for (int t = t0 + wave; t < t1; t += 4) {
        const long p0 = img_px + (long)t * TP;
        const int tl = t * TP;
        const int th = tl / W, tw0 = tl - th * W;                 // conv coordinates of the tile's first pixel
        // ---- 1. stage the tile (this wave's private LDS: in-order LDS operations, no block barrier)
        constexpr bool KEEPX = RS == 2 && CIN == 16;              // pooled source with dx at source resolution (launcher: only Cin = 16)
        bf16x8 xk[KEEPX ? K::XIT : 1][KEEPX ? 4 : 1];             // ... the raw 2x2 chunks stay in registers for the epilogue
#pragma unroll
        for (int j = 0; j < K::GCH; ++j) {
            const int px = (j * 64 + lane) / K::CPG;
            bf16x8 wv = rg[j];
            if (eff) {
                const f32x4 d0 = *(lds_cf4)(dsl), d1 = *(lds_cf4)(dsl + 4), q0 = *(lds_cf4)(dsl + COUT), q1 = *(lds_cf4)(dsl + COUT + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    wv[i] = f2bf(bf2f(rg[j][i]) + (i < 4 ? d0[i & 3] : d1[i & 3]) + bf2f(ry[j][i]) * (i < 4 ? q0[i & 3] : q1[i & 3]));
                if (a.geff_out != nullptr) *(bf16x8*)((bf16*)a.geff_out + (p0 + px) * COUT + ccg * 8) = wv;
            }
            *(bf16x8*)(lds_g + px * K::GS + ccg * 8) = wv;
        }
#pragma unroll
        for (int j = 0; j < K::XIT; ++j) {
            const int px = (j * 64 + lane) / K::CPX;
            if (RS == 0) {
                *(bf16x8*)(lds_x + px * K::XS + ccx * 8) = rx[j][0];
            } else {
                float acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (KEEPX) xk[j][q] = rx[j][q];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = bf2f(rx[j][q][i]);
                        acc[i] += relu ? fmaxf(v, 0.f) : v;
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = f2bf(0.25f * acc[i]);
                *(bf16x8*)(lds_x + px * K::XS + ccx * 8) = o;
            }
        }
        // ---- 2. requests: shortcut gradient of this tile first (older in the memory counter), then the next tile's operands
        bf16x8 rl[K::XIT];
        const bool l_here = lmode >= 0 && ccx * 8 < a.lCa;
        if (lmode == 0 || lmode == 2) {
#pragma unroll
            for (int j = 0; j < K::XIT; ++j) {
                const int px = (j * 64 + lane) / K::CPX;
                rl[j] = zero8();
                if (l_here) {
                    if (lmode == 0) rl[j] = *(const bf16x8*)((const bf16*)a.lg + (p0 + px) * a.lC + ccx * 8);
                    else rl[j] = *(const bf16x8*)((const bf16*)a.lg + (((long)n * (H >> 1) + (th >> 1)) * (W >> 1) + ((tw0 + px) >> 1)) * a.lC + ccx * 8);
                }
            }
        }
        if (t + 4 < t1) request(t + 4);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- 3. weight gradient (+ bias column sums): one K step of 32 pixels
        if (want_w) {
#pragma unroll
          for (int kk = 0; kk < TP / 32; ++kk) {
            const int pix0 = kk * 32 + (lg >> 1) * 16 + (lg & 1) * 4;
            bf16x8 afr[K::MT_W];
#pragma unroll
            for (int mt = 0; mt < K::MT_W; ++mt) afr[mt] = b1_frag_T(lds_g, K::GS, pix0, mt * 16, lr);
#pragma unroll
            for (int nj = 0; nj < K::NJ_W; ++nj) {
                bf16x8 b = b1_frag_T(lds_x, K::XS, pix0, nj * 16, lr);
                if (RS == 0) {                                    // forward prologue on the fragment: 8 pixels of channel nj*16 + lr
                    if (AFF) {
                        const float sc = *(lds_cf)(aff_s + nj * 16 + lr), sh = *(lds_cf)(aff_s + CIN + nj * 16 + lr);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float v = bf2f(b[i]) * sc + sh;
                            b[i] = f2bf(relu ? fmaxf(v, 0.f) : v);
                        }
                    } else if (relu) {
                        b = relu8(b);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < K::MT_W; ++mt) accw[mt][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[mt], b, accw[mt][nj], 0, 0, 0);
            }
            if (a.colsum != nullptr) {
#pragma unroll
                for (int mt = 0; mt < K::MT_W; ++mt) accc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[mt], ones, accc[mt], 0, 0, 0);
            }
          }
        }
        // ---- 4. data gradient, EROWS pixel rows per pass
        if (a.dx != nullptr) {
#pragma unroll
            for (int ep = 0; ep < K::EPASS; ++ep) {
                constexpr int MTS = K::EROWS / 16;
#pragma unroll
                for (int ms = 0; ms < MTS; ++ms) {
                    f32x4 acc[K::NT_D];
#pragma unroll
                    for (int nt = 0; nt < K::NT_D; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < K::KS_D; ++ks) {
                        const int k = ks * 32 + lg * 8;
                        const bf16x8 af = (k < COUT) ? *(const bf16x8*)(lds_g + (ep * K::EROWS + ms * 16 + lr) * K::GS + k) : zero8();
#pragma unroll
                        for (int nt = 0; nt < K::NT_D; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[ks][nt], acc[nt], 0, 0, 0);
                    }
#pragma unroll
                    for (int nt = 0; nt < K::NT_D; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) wlds[(ms * 16 + lg * 4 + r) * K::LDW + nt * 16 + lr] = acc[nt][r];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < K::EIT; ++it) {
                    const int j = ep * K::EIT + it;               // == the staging item index of this (lane, it)
                    const int row = (it * 64 + lane) / K::CPX;    // row inside the pass
                    const int px = ep * K::EROWS + row;           // pixel inside the tile
                    const f32x4 lo = *(const f32x4*)(wlds + row * K::LDW + ccx * 8);
                    const f32x4 hi = *(const f32x4*)(wlds + row * K::LDW + ccx * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (KEEPX && a.out_mode == 0) {
                        // pooled source: dx at SOURCE resolution = relu'(x) * 0.25 * da, four pixels per item
                        const long sp = (((long)n * Hs + 2 * th) * Ws + 2 * (tw0 + px)) * CIN + ccx * 8;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bf16x8 o;
#pragma unroll
                            for (int i = 0; i < 8; ++i) o[i] = f2bf((!relu || bf2f(xk[KEEPX ? j : 0][KEEPX ? q : 0][i]) > 0.f) ? 0.25f * v[i] : 0.f);
                            *(bf16x8*)((bf16*)a.dx + sp + (long)(q >> 1) * Ws * CIN + (q & 1) * CIN) = o;
                        }
                        continue;
                    }
                    if (a.out_mode == 0 && (AFF || relu)) {
                        const bf16x8 xv = *(const bf16x8*)(lds_x + px * K::XS + ccx * 8);        // RS == 0: the raw x tile is still in LDS
                        f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = c0, h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
                        if (AFF) {
                            c0 = *(lds_cf4)(aff_s + ccx * 8);
                            c1 = *(lds_cf4)(aff_s + ccx * 8 + 4);
                            h0 = *(lds_cf4)(aff_s + CIN + ccx * 8);
                            h1 = *(lds_cf4)(aff_s + CIN + ccx * 8 + 4);
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float xf = bf2f(xv[i]);
                            const float sc = i < 4 ? c0[i & 3] : c1[i & 3], sh = i < 4 ? h0[i & 3] : h1[i & 3];
                            const float pre = AFF ? xf * sc + sh : xf;
                            const float d = (relu && !(pre > 0.f)) ? 0.f : v[i];
                            if (AFF) {
                                s1[i] += d;                       // -> d shift
                                s2[i] += d * xf;                  // -> d scale
                                v[i] = d * sc;
                            } else {
                                v[i] = d;
                            }
                        }
                    }
                    if (l_here) {
                        if (lmode == 0) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] += bf2f(rl[j][i]);
                        } else if (lmode == 2) {                  // shortcut was average-pooled: 0.25 * nearest expand of its gradient
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] += 0.25f * bf2f(rl[j][i]);
                        } else {                                  // shortcut was up-sampled: 2x2 SUM of its gradient at double resolution
                            const bf16* p = (const bf16*)a.lg + (((long)n * (2 * H) + 2 * th) * (2 * W) + 2 * (tw0 + px)) * a.lC + ccx * 8;
                            const long rs_ = (long)2 * W * a.lC;
                            const bf16x8 q0 = *(const bf16x8*)p, q1 = *(const bf16x8*)(p + a.lC), q2 = *(const bf16x8*)(p + rs_),
                                         q3 = *(const bf16x8*)(p + rs_ + a.lC);
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] += 4.f * (0.25f * (bf2f(q0[i]) + bf2f(q1[i]) + bf2f(q2[i]) + bf2f(q3[i])));
                        }
                    }
                    bf16x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                    *(bf16x8*)((bf16*)a.dx + (p0 + px) * CIN + ccx * 8) = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the next pass / tile overwrites the wave's buffers
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

    // ---- block end: BatchNorm accumulators (sum d, sum d*x per image), then the fold of the four partial dW
    __syncthreads();
    if (AFF && a.bn_acc != nullptr) {
        // fold the lanes that share a chunk through the (now free) LDS, one atomic per channel and block: same scheme as stats_flush
        float* sx_all = (float*)smem;
        constexpr int CPP = K::CPX, SH = 64 / CPP;
        float* sx = sx_all + wave * STATS_SX_FLOATS;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sx[i * 64 + (lane ^ i)] = w ? s2[i] : s1[i];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 8 * CPP) {
                const int cc = lane % CPP, i = lane / CPP;
                float tt = 0.f;
#pragma unroll
                for (int k = 0; k < SH; ++k) tt += sx[i * 64 + ((cc + CPP * k) ^ i)];
                red[(wave * CIN + cc * 8 + i) * 2 + w] = tt;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        const int tt = threadIdx.x;
        if (tt < CIN) {
            float x1 = 0.f, x2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                x1 += red[(wv * CIN + tt) * 2 + 0];
                x2 += red[(wv * CIN + tt) * 2 + 1];
            }
            // slot = this block's index inside its image when bn_slots == blocks per image (one adder per address: bit-reproducible)
            const int R = a.bn_slots > 0 ? a.bn_slots : BNB_REPL;
            float* st = a.bn_acc + ((long)n * R + bid % R) * 2 * CIN;
            atomicAdd(st + tt, x1);
            atomicAdd(st + CIN + tt, x2);
        }
        __syncthreads();
    }
    if (want_w) {
        float* T = (float*)smem;                                  // [COUT][CIN] + [COUT]
        for (int i = threadIdx.x; i < COUT * CIN + COUT; i += 256) T[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < K::MT_W; ++mt) {
#pragma unroll
            for (int nj = 0; nj < K::NJ_W; ++nj)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&T[(mt * 16 + lg * 4 + r) * CIN + nj * 16 + lr], accw[mt][nj][r]);
            if (a.colsum != nullptr && lr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&T[COUT * CIN + mt * 16 + lg * 4 + r], accc[mt][r]);
            }
        }
        __syncthreads();
        if (a.partials != nullptr) {
            float* slab = a.partials + (long)blockIdx.x * COUT * a.Kpad;
            for (int i = threadIdx.x; i < COUT * CIN; i += 256) slab[(i / CIN) * a.Kpad + (i % CIN)] = T[i];
        } else {
            for (int i = threadIdx.x; i < COUT * CIN; i += 256) atomicAdd(a.dw + (long)(i / CIN) * a.Kpad + (i % CIN), T[i]);
        }
        if (a.colsum != nullptr && threadIdx.x < COUT)
            atomicAdd(a.colsum + (long)(blockIdx.x % STAT_REPL) * COUT + threadIdx.x, T[COUT * CIN + threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------------------
struct B1Plan {
    int tpi, tpb, bpi, nblk, tp;
    bool occ3;
    long ws_elems;
};

static bool b1_shape_ok(int cin, int cout, int rs, bool aff) {
    if (rs == 2) return !aff && ((cin == 16 && cout == 64) || (cin == 32 && cout == 32));
    if (rs != 0) return false;
    return (cin == 32 && cout == 16) || (cin == 64 && cout == 16) || (cin == 64 && cout == 32) || (cin == 16 && cout == 64) ||
           (cin == 16 && cout == 32) || (cin == 32 && cout == 64) || (cin == 32 && cout == 32);
}

// blocks per CU (= waves per SIMD): three only where 168 registers hold the kernel without spills AND it measured faster
// (measured at N = 40, tools/bwd1x1_bench.py: never faster than two blocks with 64-pixel tiles; kept for benchmarks, IEAGAN_B1_OCC3)
static bool b1_occ3(int cin, int cout, int rs, bool aff) { return false; }
// 64-pixel wave tiles where two blocks of four such waves still fit the 160 KB of a CU
static int b1_tp(int cin, int cout, int rs) {
    if (rs == 2) return cin == 32 ? 64 : 32;           // (Cin = 16 keeps the raw 2x2 chunks in registers: spills at 64)
    return (cin * cout == 2048) ? 32 : 64;             // 64 <-> 32 channels: LDS (two blocks per CU) / registers
}

static int b1_plan(const Bwd1Args& a, B1Plan& p) {
    CHECK_ARG(a.src.rs == 0 || a.src.rs == 2, "conv1x1_bwd: resample mode %d", a.src.rs);
    CHECK_ARG(b1_shape_ok(a.Cin, a.Cout, a.src.rs, a.src.scale != nullptr), "conv1x1_bwd: shape %d -> %d (rs %d, affine %d) is not instantiated", a.Cin, a.Cout,
              a.src.rs, a.src.scale != nullptr);
    const bool occ3 = (a.flags & IEAGAN_B1_OCC2) ? false : ((a.flags & IEAGAN_B1_OCC3) ? true : b1_occ3(a.Cin, a.Cout, a.src.rs, a.src.scale != nullptr));
    int TP = (a.flags & IEAGAN_B1_TP32) || occ3 ? 32 : b1_tp(a.Cin, a.Cout, a.src.rs);
    if (a.W % TP != 0) TP = 32;                       // a tile must lie inside one row
    CHECK_ARG(a.W % TP == 0, "conv1x1_bwd: W %% 32 required (W = %d)", a.W);
    if (a.src.rs == 0) CHECK_ARG(a.H == a.src.Hs && a.W == a.src.Ws, "conv1x1_bwd: geometry mismatch");
    if (a.src.rs == 2) CHECK_ARG(2 * a.H == a.src.Hs && 2 * a.W == a.src.Ws, "conv1x1_bwd: pool geometry mismatch");
    CHECK_ARG(a.Kpad % 32 == 0 && a.Kpad >= a.Cin && a.Kpad2 % 32 == 0 && a.Kpad2 >= a.Cout, "conv1x1_bwd: bad weight-pack row lengths");
    CHECK_ARG(a.Cg >= a.Cout && a.Cg % 8 == 0 && a.src.Cx >= a.Cin && a.src.Cx % 8 == 0, "conv1x1_bwd: bad channel strides");
    CHECK_ARG((a.src.scale == nullptr) == (a.src.shift == nullptr), "conv1x1_bwd: scale / shift must come together");
    CHECK_ARG((a.y == nullptr) == (a.dstat == nullptr), "conv1x1_bwd: the effgrad operands y / dstat must come together");
    CHECK_ARG(a.geff_out == nullptr || a.y != nullptr, "conv1x1_bwd: geff_out without effgrad");
    CHECK_ARG(a.n_per_event >= 0 && (a.n_per_event == 0 || a.N % a.n_per_event == 0), "conv1x1_bwd: N is not a whole number of events");
    CHECK_ARG(a.out_mode == 0 || a.out_mode == 1, "conv1x1_bwd: out_mode");
    CHECK_ARG(a.out_mode == 0 || a.src.scale == nullptr, "conv1x1_bwd: plain da output cannot carry the BatchNorm backward");
    if (a.src.rs == 2 && a.dx != nullptr) CHECK_ARG(a.out_mode == 1 || a.Cin == 16, "conv1x1_bwd: dx at source resolution of a pooled source needs Cin = 16");
    if (a.bn_acc != nullptr) CHECK_ARG(a.src.scale != nullptr && a.out_mode == 0 && a.dx != nullptr, "conv1x1_bwd: bn_acc needs the affine prologue and dx");
    if (a.src.scale != nullptr && a.dx != nullptr) CHECK_ARG(a.bn_acc != nullptr, "conv1x1_bwd: the affine prologue needs bn_acc");
    if (a.lg != nullptr) {
        CHECK_ARG(a.lmode >= 0 && a.lmode <= 2 && a.lC % 8 == 0 && a.lCa % 8 == 0 && a.lCa <= a.lC && a.lCa <= a.Cin, "conv1x1_bwd: bad shortcut-gradient operand");
        CHECK_ARG(!(a.src.rs == 2 && a.out_mode == 0), "conv1x1_bwd: no shortcut gradient on a pooled source");
        if (a.lmode == 2) CHECK_ARG(a.H % 2 == 0 && a.W % 2 == 0, "conv1x1_bwd: half-resolution shortcut gradient needs even H, W");
    }
    CHECK_ARG(a.dx != nullptr || a.dw != nullptr, "conv1x1_bwd: nothing to compute");
    CHECK_ARG((a.flags & ~(IEAGAN_B1_OCC2 | IEAGAN_B1_OCC3 | IEAGAN_B1_TP32 | IEAGAN_BWD_NO_REDUCE)) == 0, "conv1x1_bwd: unknown flag bits 0x%x", a.flags);
    CHECK_ARG(a.bn_slots >= 0, "conv1x1_bwd: bad bn_slots %d", a.bn_slots);
    CHECK_ARG(a.colsum == nullptr || a.dw != nullptr, "conv1x1_bwd: colsum rides on the weight gradient");
    const int tpi = a.H * a.W / TP;
    // ONE round of persistent blocks: every resident slot (256 CUs x blocks per CU) gets one block, whole blocks per image, the four
    // waves of a block interleave its tiles.  (A grid of 1040 blocks on 512 slots ran three rounds, the last one 3 % full.)
    const int slots = 256 * (occ3 ? 3 : 2);
    int bpi = slots / a.N;
    if (bpi < 1) bpi = 1;
    int tpb = (tpi + bpi - 1) / bpi;
    tpb = (tpb + 3) / 4 * 4;
    if (tpb < 8) tpb = 8;
    bpi = (tpi + tpb - 1) / tpb;
    p.tp = TP;
    p.occ3 = occ3;
    p.tpi = tpi;
    p.tpb = tpb;
    p.bpi = bpi;
    p.nblk = bpi * a.N;
    const long direct_bytes = (long)p.nblk * a.Cout * a.Cin * 4;
    p.ws_elems = (a.dw != nullptr && direct_bytes >= (8L << 20)) ? (long)p.nblk * a.Cout * a.Kpad : 0;
    return 0;
}

extern "C" long ieagan_conv1x1_bwd_workspace(const ieagan_conv1x1_bwd_desc* d) {
    B1Plan p;
    if (d == nullptr || b1_plan(*d, p) != 0) return 0;
    return p.ws_elems;
}

extern "C" int ieagan_conv1x1_bwd_slots(const ieagan_conv1x1_bwd_desc* d) {
    B1Plan p;
    if (d == nullptr) return IEAGAN_EINVAL;
    const int rc = b1_plan(*d, p);
    return rc != 0 ? rc : p.bpi;
}

extern "C" int ieagan_conv1x1_bwd_supported(int Cin, int Cout, int rs, int affine) { return b1_shape_ok(Cin, Cout, rs, affine != 0) ? 1 : 0; }

int wgrad_reduce_launch(const float* part, float* dw, int S, int Cout, int Kpad, int K, hipStream_t st);      // conv_wgrad.hip

extern "C" int ieagan_conv1x1_bwd(const ieagan_conv1x1_bwd_desc* d, void* stream) {
    CHECK_ARG(d != nullptr && d->g != nullptr && d->src.x != nullptr && d->w_bwd != nullptr, "conv1x1_bwd: null pointer");
    Bwd1Args a = *d;
    B1Plan p;
    const int prc = b1_plan(a, p);
    if (prc != 0) return prc;
    if (a.partials != nullptr && p.ws_elems == 0) a.partials = nullptr;       // small dW: direct atomics
    hipStream_t st = (hipStream_t)stream;
    const double P = (double)a.N * a.H * a.W, Ps = (double)a.N * a.src.Hs * a.src.Ws;
    const double flops = 2.0 * P * (double)a.Cout * a.Cin * ((a.dx ? 1 : 0) + (a.dw ? 1 : 0));
    // algorithmic bytes (layer-granular): dgrad R g + W dx, wgrad R x + R g -- what the replaced launches were charged
    // (the dgrad of a pooled layer is charged at the layer's own resolution, as tools/arch_calc.py's "as launched" figures do: the fold
    //  back to the source resolution used to be prologue_bwd's traffic)
    const double bytes_min = 2.0 * ((a.dx ? P * a.Cout + P * a.Cin : 0.0) + (a.dw ? Ps * a.Cin + P * a.Cout : 0.0));
    double bytes = 2.0 * (P * a.Cout + Ps * a.Cin + (a.dx ? (a.out_mode == 0 ? Ps : P) * a.Cin : 0.0));        // what this launch moves
    if (a.y) bytes += 2.0 * P * a.Cout * (a.geff_out ? 2 : 1);
    if (a.lg) bytes += 2.0 * P * a.lCa * (a.lmode == 1 ? 4.0 : (a.lmode == 2 ? 0.25 : 1.0));
    char tag[64] = "";
    if (prof_tags_on())
        snprintf(tag, sizeof(tag), "ci%d co%d %dx%d rs%d a%d r%d eff%d l%d w%d", a.Cin, a.Cout, a.H, a.W, a.src.rs, a.src.scale != nullptr, a.src.relu,
                 a.y != nullptr, a.lg ? a.lmode : -1, a.dw != nullptr);
    ProfScope prof("conv1x1_bwd", flops, bytes, st, tag, bytes_min);
    const bool occ3 = p.occ3;
#define B1_LAUNCH(CI, CO, RSV, AF)                                                                                                   \
    {                                                                                                                                \
        if (occ3) hipLaunchKernelGGL((conv1x1_bwd_kernel<CI, CO, RSV, AF, 3, 32>), dim3(p.nblk), dim3(256), 0, st, a, p.tpi, p.tpb, p.bpi, p.nblk); \
        else if (p.tp == 32) hipLaunchKernelGGL((conv1x1_bwd_kernel<CI, CO, RSV, AF, 2, 32>), dim3(p.nblk), dim3(256), 0, st, a, p.tpi, p.tpb, p.bpi, p.nblk); \
        else hipLaunchKernelGGL((conv1x1_bwd_kernel<CI, CO, RSV, AF, 2, 64>), dim3(p.nblk), dim3(256), 0, st, a, p.tpi, p.tpb, p.bpi, p.nblk);      \
    }
#define B1_CASE(CI, CO)                                                      \
    if (a.Cin == CI && a.Cout == CO) {                                       \
        if (a.src.scale != nullptr) B1_LAUNCH(CI, CO, 0, true)               \
        else B1_LAUNCH(CI, CO, 0, false)                                     \
    } else
    if (a.src.rs == 2) {
        if (a.Cin == 16 && a.Cout == 64) B1_LAUNCH(16, 64, 2, false)
        else B1_LAUNCH(32, 32, 2, false)
    } else {
        B1_CASE(32, 16) B1_CASE(64, 16) B1_CASE(64, 32) B1_CASE(16, 64) B1_CASE(16, 32) B1_CASE(32, 64) B1_CASE(32, 32) {}
    }
#undef B1_CASE
#undef B1_LAUNCH
    CHECK_LAUNCH("conv1x1_bwd");
    if (a.partials != nullptr && !(a.flags & IEAGAN_BWD_NO_REDUCE)) return wgrad_reduce_launch(a.partials, a.dw, p.nblk, a.Cout, a.Kpad, a.Cin, st);
    return 0;
}
