// Implicit-GEMM convolutions on MFMA (gfx950), bf16 NHWC activations, fp32 accumulate.
//
//   conv_gather : out[M = N*H*W, Cout] = A[M, taps*Cin] * Wp^T,  A gathered on the fly with the
//                 fused prologue (class-conditional BN apply, ReLU, nearest x2 upsample or 2x2 average
//                 pool of the source) and the fused epilogue (bias, residual add with its own
//                 resample / channel slice, ReLU-mask multiply, per-channel sum / sum-of-squares for
//                 the next BatchNorm).  Used for forward and, with the transposed/flipped weight pack,
//                 for dgrad.  Replaces F.conv2d + F.batch_norm + relu + F.interpolate + AvgPool2d +
//                 the residual add of the reference blocks (model.py:54-71, 541-557; layers.py:197-206).
//   conv_wgrad  : dWp[Cout, taps*Cin] += G^T * A over pixel tiles; both operands staged in LDS in
//                 their natural NHWC layout and read K(pixel)-major with ds_read_b64_tr_b16.
//
// MFMA: v_mfma_f32_16x16x32_bf16.  Lane l: A[row l&15][k 8*(l>>4)+j], B[k 8*(l>>4)+j][col l&15],
// D[row 4*(l>>4)+r][col l&15].
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"
#include <stdio.h>

// ------------------------------------------------------------------------------------------------
// A-operand gather with fused prologue.  Returns 8 consecutive channels [c, c+8) of the (virtual)
// conv-input pixel (n, hh, ww) at conv resolution; zero outside the image (conv padding is applied
// AFTER the activation, as in the reference where the activated tensor is what gets padded).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// conv_gather: block = 4 waves, each wave 2 m-tiles (32 pixels) x NT n-tiles (16*NT channels);
// A fragments gathered straight from global memory (any H, W; 1x1 and small 3x3 layers).
// ------------------------------------------------------------------------------------------------
template <int TAPS, bool AFF, bool RELU, int RS, int NT, bool LAFF, bool BNB = false>
__global__ __launch_bounds__(256, 4) void conv_gather_kernel(ConvArgs a) {
    // NT >= 2: the epilogue transposes 16 pixel rows at a time (half the LDS -> more resident blocks for this latency-bound kernel)
    constexpr int EROWS = (NT >= 2) ? 16 : 32;
    __shared__ __attribute__((aligned(16))) float epi[4 * EROWS * EpiLds<NT>::LDW];
    __shared__ float red[4 * NT * 16 * 2];
    __shared__ __attribute__((aligned(32))) float aff_s[LAFF ? 2 * AFF_MAXC : 8];   // LAFF: all 128 pixels of a block share one image
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W;
    const long M = (long)a.N * HW;
    // Split-K form (launcher: tiny feature maps, NT == 1): the four waves of a block share ONE 32-pixel tile and take every fourth
    // K step each -- a 4x12 / 8x24 map gives 15 / 60 blocks of 128 pixels whose single wave walks up to 36 dependent
    // load -> MFMA steps with nothing else resident to hide them; split, the chain is a quarter as long and there are 4x the blocks.
    const bool sk = NT == 1 && (a.flags & CONV_INTERNAL_SPLITK) != 0;
    const int ppb = sk ? 32 : 128;                          // pixels per block
    const long m_base = sk ? (long)blockIdx.x * 32 : ((long)blockIdx.x * 4 + wave) * 32;
    const int n_base = blockIdx.y * NT * 16;

    int pn[2], ph[2], pw[2];
    bool pv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long m = m_base + mt * 16 + lr;
        pv[mt] = m < M;
        const long mm = pv[mt] ? m : 0;
        pn[mt] = (int)(mm / HW);
        const int rem = (int)(mm - (long)pn[mt] * HW);
        ph[mt] = rem / W;
        pw[mt] = rem - ph[mt] * W;
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (LAFF) {
        stage_aff(aff_s, a.src, (int)(((long)blockIdx.x * ppb) / HW), a.Cin);
        __syncthreads();
    }
    const int ksteps = a.Kpad >> 5;
    const bool col_ok = (n_base + (NT - 1) * 16 + lr) < a.Cout;     // only NT == 1 can have a half-empty n-tile
    const bf16* wrow = (const bf16*)a.w + (long)(col_ok ? n_base + lr : 0) * a.Kpad + lg * 8;
    // (the bias stays a load inside the epilogue here: requested before the K loop it costs 8-12 VGPRs over the whole kernel and
    //  one to two resident waves per SIMD on most variants -- one tile per block, the other blocks hide that latency)
    for (int ks = sk ? wave : 0; ks < ksteps; ks += sk ? 4 : 1) {
        const int k = ks * 32 + lg * 8;
        int tap = 0, c = k;
        if (TAPS == 9) {
            tap = k / a.Cin;
            c = k - tap * a.Cin;
        }
        const bool kval = (TAPS == 9) ? (tap < 9) : (k < a.Cin);
        const int dy = (TAPS == 9) ? (tap / 3 - 1) : 0;
        const int dx = (TAPS == 9) ? (tap - (tap / 3) * 3 - 1) : 0;
        bf16x8 bfrag[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bfrag[nt] = *(const bf16x8*)(wrow + (long)nt * 16 * a.Kpad + ks * 32);
            if (NT == 1 && !col_ok) bfrag[nt] = zero8();
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const bf16x8 af = gather8<AFF, RELU, RS>(a.src, H, W, pn[mt], ph[mt] + dy, pw[mt] + dx, c, pv[mt] && kval,
                                                     LAFF ? aff_s : nullptr);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    if (NT == 1 && sk) {           // fold the four partial accumulators into wave 0 through the (not yet used) epilogue buffer
        float* part = epi;         // [3 waves][8 values][64 lanes]: 6 KB of the 10 KB buffer
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[((wave - 1) * 8 + mt * 4 + r) * 64 + lane] = acc[mt][0][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][0][r] += part[(w * 8 + mt * 4 + r) * 64 + lane];
        }
        // (wave 0 reads before it writes its own region of the buffer below: same wave, in order)
    }
    const bool epi_wave = !sk || wave == 0;
    const bool need_hw = (a.ra != nullptr && a.ra_rs != 0) || (BNB && a.bnb_scale != nullptr);
    auto pix = [&](int row, long& m, int& n, int& h, int& w) -> bool {
        m = m_base + row;
        n = h = w = 0;
        if (m >= M) return false;
        if (need_hw) {
            n = (int)(m / HW);
            const int rem = (int)(m - (long)n * HW);
            h = rem / W;
            w = rem - h * W;
        }
        return true;
    };
    if constexpr (NT >= 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            auto pixh = [&](int row, long& m, int& n, int& h, int& w) -> bool { return pix(row + 16 * half, m, n, h, w); };
            const f32x4(&sub)[1][NT] = *reinterpret_cast<const f32x4(*)[1][NT]>(&acc[half]);
            conv_epilogue<BNB, NT, 1, (RS != 2)>(a, sub, epi + wave * EROWS * EpiLds<NT>::LDW, n_base, pixh, s1, s2);      // (pooled-source variants: no registers to spare)
        }
    } else {
        if (epi_wave) conv_epilogue<BNB, NT, 2, (RS != 2)>(a, acc, epi + wave * EROWS * EpiLds<NT>::LDW, n_base, pix, s1, s2);
    }
    static_assert(4 * EROWS * EpiLds<NT>::LDW >= 4 * STATS_SX_FLOATS, "epilogue buffer doubles as the statistics scratch");
    if (a.stats != nullptr) {
        __syncthreads();          // every wave is done with its part of the epilogue buffer, which now serves as fold scratch
        // every 128-pixel block lies inside one event (launcher: n_per_event * H * W % 128 == 0 when E > 1)
        const int event = (a.n_per_event > 0) ? (int)(((long)blockIdx.x * ppb) / ((long)a.n_per_event * HW)) : 0;
        stats_flush<NT>(a, s1, s2, n_base, red, epi, blockIdx.x, event);
    }
}

template <int TAPS, bool AFF, bool RELU, int RS, bool BNB = false>
static int launch_gather_nt(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    const long M = (long)a.N * a.H * a.W;
    unsigned gx = (unsigned)((M + 127) / 128);
    // tiny feature maps (4x12 ... 16x48): split the output channels over more blocks to cover the chip
    const bool small4 = (long)gx * (a.Cout / 64) < 512, small2 = (long)gx * (a.Cout / 32) < 512;
    // affine table in LDS when every 128-pixel block lies inside one image (all but the tiniest maps)
    const bool laff = AFF && ((long)a.H * a.W) % 128 == 0 && a.Cin <= AFF_MAXC;
#define GATHER_LAUNCH(NTV, GY)                                                                                                  \
    {                                                                                                                           \
        CONV_PLAN_POINT((int)(gx / ((a.stats != nullptr && a.n_per_event > 0) ? (unsigned)(a.N / a.n_per_event) : 1u)), 0)       \
        if (AFF && laff) hipLaunchKernelGGL((conv_gather_kernel<TAPS, AFF, RELU, RS, NTV, AFF, BNB>), dim3(gx, GY), dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((conv_gather_kernel<TAPS, AFF, RELU, RS, NTV, false, BNB>), dim3(gx, GY), dim3(256), 0, st, a);      \
    }
    if (a.Cout % 64 == 0 && !small4) GATHER_LAUNCH(4, a.Cout / 64)
    else if (a.Cout % 32 == 0 && !(small4 && small2 && a.Cout % 64 == 0)) GATHER_LAUNCH(2, a.Cout / 32)
    else {
        // tiny maps with a long K loop: split K over the waves of a block (conv_gather_kernel: sk); per-event statistics need
        // whole 32-pixel blocks per event
        const bool ev_ok = !(a.stats != nullptr && a.n_per_event > 0 && a.n_per_event < a.N) || ((long)a.n_per_event * a.H * a.W) % 32 == 0;
        if (M <= 32768 && (a.Kpad >> 5) >= 8 && ev_ok && !(a.flags & IEAGAN_CONV_FORCE_GATHER)) {
            a.flags |= CONV_INTERNAL_SPLITK;
            gx = (unsigned)((M + 31) / 32);
        }
        GATHER_LAUNCH(1, (a.Cout + 15) / 16)
    }
#undef GATHER_LAUNCH
    return 0;
}

template <int TAPS, int RS>
static int launch_gather_pro(const ConvArgs& a, hipStream_t st) {
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (aff && relu) return launch_gather_nt<TAPS, true, true, RS>(a, st);
    if (aff) return launch_gather_nt<TAPS, true, false, RS>(a, st);
    if (relu) return launch_gather_nt<TAPS, false, true, RS>(a, st);
    if (RS == 0 && a.bnb_scale != nullptr) return launch_gather_nt<TAPS, false, false, (RS == 0 ? 0 : RS), (RS == 0)>(a, st);   // BatchNorm-backward epilogue
    return launch_gather_nt<TAPS, false, false, RS>(a, st);
}

// ------------------------------------------------------------------------------------------------
// conv3x3_halo: output tile 8 rows x 32 columns.  The (8+2) x (32+2) input halo is staged ONCE in LDS with
// the prologue (BN apply, ReLU, upsample source mapping) already applied, so the nine taps read transformed
// bf16 pixels with ds_read_b128 instead of re-gathering and re-transforming from L1.
// LDS pixel stride = Cin*2 + 16 bytes: 16 consecutive pixels land on 16 distinct 16-byte bank slots.
// Wave w owns tile rows 2w, 2w+1 (4 m-tiles of 16 pixels).
//
// PF > 0 (small Cin, the HBM-bound layers): a block walks `tpb` consecutive tiles; while tile t is in the
// MFMA loop the raw halo of tile t+1 is already in flight into PF x 16-byte registers per thread (the
// weights sit in LDS, so nothing in the loop waits on the vector-memory counter), i.e. global-memory
// latency is hidden inside the block instead of relying on occupancy alone.
// PF == 0: one tile per block, halo staged directly, weights from global/L1 (large Cin, MFMA-bound).
// ------------------------------------------------------------------------------------------------
#define HT_H 8
#define HT_W 32

template <bool AFF, bool RELU, int RS, int NT, int PF, int CIN, bool BNB = false>
__global__ __launch_bounds__(256, ((CIN >= 64 && PF == 0) ? 2 : (CIN == 16 ? 4 : (CIN == 32 ? 3 : 1)))) void conv3x3_halo_kernel(ConvArgs a, int tiles_w, int tiles_h, int tpe, int tpb, int nblk, int bpe) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    __shared__ float red[4 * NT * 16 * 2];
    constexpr int AW = HT_W + 2, AH = HT_H + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W;
    const int Cin = (CIN > 0) ? CIN : a.Cin;              // compile-time for the prefetching variants: no runtime divisions
    const int PS = Cin * 2 + 16;                           // bytes per halo pixel
    const int WS = ((CIN > 0) ? ((9 * CIN + 31) / 32) * 32 : a.Kpad) * 2 + 16;   // bytes per weight row in LDS (PF > 0)
    const int n_base = blockIdx.y * NT * 16;
    const int chunks = Cin >> 3;
    const int total = AH * AW * chunks;
    char* smem = smem_all + (PF > 0 ? NT * 16 * WS : 0);   // halo (and, after the MFMA loop, epilogue) region
    // XCD-aware order: blocks b and b+8 share an XCD (and its L2), so give each XCD a contiguous run of
    // tiles -- neighbouring tiles then share their halo rows / weights through one L2.
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    // a block stays inside ONE event (tpe tiles, bpe blocks per event): its statistics go to that event's accumulators with a
    // single flush at the end (a flush inside the tile loop costs the C = 16 / 32 variants 40-60 spilled VGPRs)
    const int event = bid / bpe;
    const int t0 = event * tpe + (bid - event * bpe) * tpb;
    const int t1 = min(t0 + tpb, (event + 1) * tpe);
    const bool col_ok = (n_base + (NT - 1) * 16 + lr) < a.Cout;

    if (PF > 0) {   // weights -> LDS once per block: compile-time trip count (Kpad = 9 * CIN rounded up to 32), all loads of a batch first
        constexpr int wchunks = (((9 * (CIN > 0 ? CIN : 16) + 31) / 32) * 32) >> 3;
        constexpr int WIT = (NT * 16 * wchunks + 255) / 256, WB = 6;
#pragma unroll
        for (int b0 = 0; b0 < WIT; b0 += WB) {
            bf16x8 wv[WB];
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                const int idx = min((int)threadIdx.x + (b0 + j) * 256, NT * 16 * wchunks - 1);
                const int row = idx / wchunks, kc = idx - row * wchunks;
                wv[j] = *(const bf16x8*)((const bf16*)a.w + (long)min(n_base + row, a.Cout - 1) * a.Kpad + kc * 8);
                if (n_base + row >= a.Cout) wv[j] = zero8();
            }
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                const int idx = threadIdx.x + (b0 + j) * 256;
                if (b0 + j < WIT && idx < NT * 16 * wchunks) *(bf16x8*)(smem_all + (idx / wchunks) * WS + (idx % wchunks) * 16) = wv[j];
            }
        }
    }
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * AFF_MAXC : 8];
    const float* affp = AFF ? aff_s : nullptr;             // Cin <= AFF_MAXC is checked by the launcher
    int aff_n = -1;
    bf16x8 raw[PF > 0 ? PF : 1];
    unsigned okmask = 0;
    auto tile_coords = [&](int t, int& n, int& h0, int& w0) {
        n = t / (tiles_w * tiles_h);
        const int trem = t - n * tiles_w * tiles_h;
        h0 = (trem / tiles_w) * HT_H;
        w0 = (trem % tiles_w) * HT_W;
    };
    auto load_tile = [&](int t) {      // raw 16-byte loads of the halo of tile t (no transform yet)
        int n, h0, w0;
        tile_coords(t, n, h0, w0);
        okmask = 0;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int hp = idx / chunks, cc = idx - hp * chunks;
            const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
            const bool ok = idx < total && hh >= 0 && hh < H && ww >= 0 && ww < W;
            if (ok) {
                const int sh_ = (RS == 1) ? (hh >> 1) : hh, sw_ = (RS == 1) ? (ww >> 1) : ww;
                raw[j] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
                okmask |= 1u << j;
            }
        }
    };
    auto store_tile = [&](int t) {     // prologue transform + LDS write of the prefetched halo
        int n, h0, w0;
        tile_coords(t, n, h0, w0);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int idx = threadIdx.x + j * 256;
            if (idx >= total) continue;
            const int hp = idx / chunks, cc = idx - hp * chunks;
            bf16x8 o = zero8();
            if (okmask & (1u << j)) {
                if (!AFF && RELU) {
                    o = relu8(raw[j]);
                } else if (AFF || RELU) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = bf2f(raw[j][i]);
                    xform8<AFF, RELU>(v, a.src, n, cc * 8, affp);
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                } else {
                    o = raw[j];
                }
            }
            *(bf16x8*)(smem + hp * PS + cc * 16) = o;
        }
    };

    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    int pbase[4];   // byte offset of this lane's pixel in m-tile mt at tap (0,0): row 2*wave + (mt>>1), col (mt&1)*16 + lr
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) pbase[mt] = ((2 * wave + (mt >> 1)) * AW + (mt & 1) * 16 + lr) * PS;
    const int ksteps = a.Kpad >> 5;
    const int KP = (CIN > 0) ? ((9 * CIN + 31) / 32) * 32 : a.Kpad;     // == a.Kpad (checked by the launcher)
    const bf16* wrow = (const bf16*)a.w + (long)(col_ok ? n_base + lr : 0) * KP + lg * 8;

    if (PF > 0 && t0 < t1) load_tile(t0);
    for (int t = t0; t < t1; ++t) {
        int n, h0, w0;
        tile_coords(t, n, h0, w0);
        if (AFF && n != aff_n) {        // block-uniform: (re)load this image's scale / shift rows
            __syncthreads();
            stage_aff(aff_s, a.src, n, Cin);
            aff_n = n;
            __syncthreads();
        }
        if (PF > 0) {
            store_tile(t);
        } else if constexpr (CIN >= 32) {
            // known channel count: the halo is fetched in batches of SB independent 16-byte loads per thread (all in flight
            // together), then transformed and written to LDS -- not one load -> wait -> store round trip per chunk
            constexpr int CH = CIN / 8, TOT = AH * AW * CH, ITERS = (TOT + 255) / 256, SB = 6;
#pragma unroll
            for (int b0 = 0; b0 < ITERS; b0 += SB) {
                bf16x8 rawb[SB];
                unsigned okb = 0;
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int idx = threadIdx.x + (b0 + j) * 256;
                    if (b0 + j >= ITERS || idx >= TOT) continue;
                    const int hp = idx / CH, cc = idx - hp * CH;
                    const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
                    if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                        const int sh_ = (RS == 1) ? (hh >> 1) : hh, sw_ = (RS == 1) ? (ww >> 1) : ww;
                        rawb[j] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
                        okb |= 1u << j;
                    }
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int idx = threadIdx.x + (b0 + j) * 256;
                    if (b0 + j >= ITERS || idx >= TOT) continue;
                    const int hp = idx / CH, cc = idx - hp * CH;
                    bf16x8 o = zero8();
                    if (okb & (1u << j)) {
                        if (!AFF && RELU) {
                            o = relu8(rawb[j]);
                        } else if (AFF || RELU) {
                            float v[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = bf2f(rawb[j][i]);
                            xform8<AFF, RELU>(v, a.src, n, cc * 8, affp);
#pragma unroll
                            for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                        } else {
                            o = rawb[j];
                        }
                    }
                    *(bf16x8*)(smem + hp * PS + cc * 16) = o;
                }
            }
        } else {
            for (int idx = threadIdx.x; idx < total; idx += 256) {
                const int hp = idx / chunks, cc = idx - hp * chunks;
                const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
                const bf16x8 v = gather8<AFF, RELU, RS>(a.src, H, W, n, hh, ww, cc * 8, true, affp);
                *(bf16x8*)(smem + hp * PS + cc * 16) = v;
            }
        }
        __syncthreads();
        if (PF > 0 && t + 1 < t1) load_tile(t + 1);       // in flight during the MFMA loop below

        f32x4 acc[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto kstep = [&](int ks) {
            const int k = ks * 32 + lg * 8;
            int tap = k / Cin;
            const int c = k - tap * Cin;
            const bool kval = tap < 9;
            if (!kval) tap = 0;
            const int toff = ((tap / 3) * AW + (tap - (tap / 3) * 3)) * PS + c * 2;
            bf16x8 bfrag[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (PF > 0) bfrag[nt] = *(const bf16x8*)(smem_all + (nt * 16 + lr) * WS + k * 2);
                else bfrag[nt] = *(const bf16x8*)(wrow + (long)nt * 16 * KP + ks * 32);
                if (NT == 1 && !col_ok) bfrag[nt] = zero8();
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                bf16x8 af = *(const bf16x8*)(smem + pbase[mt] + toff);
                if (!kval) af = zero8();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[nt], acc[mt][nt], 0, 0, 0);
            }
        };
        if constexpr (CIN >= 32) {
            // MFMA-bound layers: straight-line K loop with an explicit one-step-ahead software pipeline -- the weight
            // fragments (global / L1) and the pixel fragments (LDS) of step ks+1 are requested before the 4*NT MFMAs of
            // step ks issue; the scheduling barrier keeps the compiler from hoisting more loads (and registers) than that.
            constexpr int KS = (9 * CIN) / 32;
            constexpr int PSC = CIN * 2 + 16;
            bf16x8 bq[2][NT], aq[2][4];
            // weight address = uniform base of the n-tile (SGPR pair) + one 32-bit lane offset + immediate (ks * 64 bytes):
            // no per-(nt, ks) pointer registers
            const char* wnt[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wnt[nt] = (const char*)a.w + (long)(n_base + nt * 16) * (KP * 2);
            const unsigned wlane = (unsigned)(lr * KP + lg * 8) * 2u;
            auto ldb = [&](int ks, bf16x8(&b)[NT]) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (PF > 0) b[nt] = *(const bf16x8*)(smem_all + (nt * 16 + lr) * WS + (ks * 32 + lg * 8) * 2);     // LDS-resident weights
                    else b[nt] = *(const bf16x8*)(wnt[nt] + wlane + (unsigned)(ks * 64));
                }
            };
            auto lda = [&](int ks, bf16x8(&x)[4]) {
                const int tap = (ks * 32) / CIN, c0 = (ks * 32) % CIN;
                const int toff = ((tap / 3) * AW + (tap % 3)) * PSC + c0 * 2;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) x[mt] = *(const bf16x8*)(smem + pbase[mt] + lg * 16 + toff);
            };
            ldb(0, bq[0]);
            lda(0, aq[0]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                    ldb(ks + 1, bq[(ks + 1) & 1]);
                    lda(ks + 1, aq[(ks + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);      // requests first, then this step's MFMAs
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ks & 1][mt], bq[ks & 1][nt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (CIN >= 32) {      // channel count known at compile time (C = 16 measured faster as a rolled loop): straight-line K loop, every offset an immediate
#pragma unroll
            for (int ks = 0; ks < (9 * CIN + 31) / 32; ++ks) kstep(ks);
        } else {
            for (int ks = 0; ks < ksteps; ++ks) kstep(ks);
        }
        __syncthreads();                                   // halo consumed: its LDS becomes the epilogue buffer
        float* epi = (float*)smem + wave * EpiLds<NT>::FLOATS;
#pragma unroll
        for (int half = 0; half < 2; ++half) {             // tile row 2*wave + half: 32 pixels
            const int hh = h0 + 2 * wave + half;
            auto pix = [&](int row, long& m, int& nn, int& h, int& w) -> bool {
                nn = n;
                h = hh;
                w = w0 + row;
                m = ((long)n * H + h) * W + w;
                return h < H && w < W;
            };
            const f32x4(&sub)[2][NT] = *reinterpret_cast<const f32x4(*)[2][NT]>(&acc[2 * half]);
            conv_epilogue<BNB, NT, 2, false>(a, sub, epi, n_base, pix, s1, s2);
        }
        __syncthreads();      // every wave has left its epilogue buffer: the region is the next tile's halo / the fold scratch
    }
    if (a.stats != nullptr && t0 < t1) stats_flush<NT>(a, s1, s2, n_base, red, (float*)smem, bid, event);
}

template <bool AFF, bool RELU, int RS, bool BNB = false>
static int launch_halo_nt(const ConvArgs& a, hipStream_t st) {
    const int tiles_w = (a.W + HT_W - 1) / HT_W, tiles_h = (a.H + HT_H - 1) / HT_H;
    const int ntiles = a.N * tiles_w * tiles_h;
    // statistics groups: events of n_per_event images (forward: per-event BatchNorm statistics; BatchNorm-backward epilogue: every
    // image its own group); without statistics the whole batch is one group
    const int n_events = (a.stats != nullptr && a.n_per_event > 0) ? a.N / a.n_per_event : 1;
    const int tpe = ntiles / n_events;
    const size_t halo = (size_t)(HT_H + 2) * (HT_W + 2) * (a.Cin * 2 + 16);
    int tpb = ntiles / 2048;
    if (tpb < 1) tpb = 1;
    if (tpb > 8) tpb = 8;
#define HALO_LAUNCH(NTV, PFV, CINV)                                                                              \
    {                                                                                                        \
        size_t lds = halo;                                                                                   \
        const size_t epi = (size_t)4 * EpiLds<NTV>::FLOATS * 4;                                              \
        if (epi > lds) lds = epi;                                                                            \
        if (lds < 4 * STATS_SX_FLOATS * 4) lds = 4 * STATS_SX_FLOATS * 4;                                    \
        int tp = (PFV) > 0 ? tpb : 1;                                                                        \
        if ((PFV) > 0 && (CINV) >= 64) { tp = (ntiles + 255) / 256; if (tp > 8) tp = 8; }   /* one persistent block per CU */ \
        if ((PFV) > 0) lds += (size_t)NTV * 16 * (a.Kpad * 2 + 16);                                          \
        const int bpe = (tpe + tp - 1) / tp;                                                                 \
        const int nblk = bpe * n_events;                                                                     \
        CONV_PLAN_POINT(bpe, 0)                                                                              \
        hipLaunchKernelGGL((conv3x3_halo_kernel<AFF, RELU, RS, NTV, PFV, CINV, BNB>), dim3(nblk, (a.Cout + 16 * NTV - 1) / (16 * NTV)), \
                           dim3(256), lds, st, a, tiles_w, tiles_h, tpe, tp, nblk, bpe);                     \
    }
    // prefetching variants: Cin = Cout = 16 / 32 (PF = ceil(340 * Cin/8 / 256) = 3 / 6)
    // small feature maps: fewer channels per block so that the grid still covers the 256 CUs
    const bool small = ntiles * ((a.Cout + 63) / 64) < 512;
    const bool kp_ok = a.Kpad == ((9 * a.Cin + 31) / 32) * 32;       // the compile-time-Cin variants assume the canonical pack
    if (!kp_ok) {
        if (a.Cout % 64 == 0 && !small) HALO_LAUNCH(4, 0, 0)
        else if (a.Cout % 32 == 0 && !(small && ntiles * (a.Cout / 32) < 512)) HALO_LAUNCH(2, 0, 0)
        else HALO_LAUNCH(1, 0, 0)
    }
    else if (a.Cin == 16 && a.Cout == 16) HALO_LAUNCH(1, 3, 16)
    else if (a.Cin == 32 && a.Cout == 32) HALO_LAUNCH(2, 6, 32)
    // (C = 64 with LDS-resident weights + prefetch was measured SLOWER -- 195-245 vs 270-320 TFLOP/s: one block of
    //  4 waves per CU leaves the ds_read -> MFMA latency exposed; it needs a hand-pipelined K loop first.)
    // C = 64 on large maps: weights resident in LDS (one persistent block per CU, next halo prefetched into registers):
    // 478 vs 343 TFLOP/s at 64x192 -- the per-wave 16-byte weight fetches through L1 were the limiter
    else if (a.Cin == 64 && a.Cout == 64 && ntiles >= 1024) HALO_LAUNCH(4, 11, 64)
    else if (a.Cin == 64 && a.Cout % 64 == 0 && !small) HALO_LAUNCH(4, 0, 64)
    else if (a.Cin == 128 && a.Cout % 64 == 0 && !small) HALO_LAUNCH(4, 0, 128)
    else if (a.Cin == 128 && a.Cout % 32 == 0 && !(small && ntiles * (a.Cout / 32) < 512)) HALO_LAUNCH(2, 0, 128)
    else if (a.Cin == 64 && a.Cout % 32 == 0 && !(small && ntiles * (a.Cout / 32) < 512)) HALO_LAUNCH(2, 0, 64)
    else if (a.Cout % 64 == 0 && !small) HALO_LAUNCH(4, 0, 0)
    else if (a.Cout % 32 == 0 && !(small && ntiles * (a.Cout / 32) < 512)) HALO_LAUNCH(2, 0, 0)
    else HALO_LAUNCH(1, 0, 0)
#undef HALO_LAUNCH
    return 0;
}

template <int RS>
static int launch_halo_pro(const ConvArgs& a, hipStream_t st) {
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (aff && relu) return launch_halo_nt<true, true, RS>(a, st);
    if (aff) return launch_halo_nt<true, false, RS>(a, st);
    if (relu) return launch_halo_nt<false, true, RS>(a, st);
    if (RS == 0 && a.bnb_scale != nullptr) return launch_halo_nt<false, false, (RS == 0 ? 0 : RS), (RS == 0)>(a, st);             // BatchNorm-backward epilogue
    return launch_halo_nt<false, false, RS>(a, st);
}

int conv_gather_launch(const ConvArgs& a, hipStream_t st) {
    CHECK_ARG(a.taps == 1 || a.taps == 9, "conv: taps must be 1 or 9 (got %d)", a.taps);
    CHECK_ARG((a.flags & ~(IEAGAN_CONV_FORCE_GATHER | IEAGAN_CONV_NO_LDS_WEIGHTS | IEAGAN_CONV_FP8 | IEAGAN_CONV_FP8_NOSCALE)) == 0, "conv: unknown flag bits 0x%x", a.flags);
    CHECK_ARG(a.Cin % 8 == 0 && a.Cout % 8 == 0, "conv: Cin/Cout must be multiples of 8 (%d,%d)", a.Cin, a.Cout);
    CHECK_ARG(a.Kpad % 32 == 0 && a.Kpad >= a.taps * a.Cin, "conv: bad Kpad %d", a.Kpad);
    CHECK_ARG(a.src.rs >= 0 && a.src.rs <= 2, "conv: bad resample mode %d", a.src.rs);
    CHECK_ARG(a.src.Cx % 8 == 0, "conv: source channel stride must be a multiple of 8");
    CHECK_ARG((a.src.scale == nullptr) == (a.src.shift == nullptr), "conv: scale/shift must come together");
    if (a.src.rs == 1) CHECK_ARG(a.H == 2 * a.src.Hs && a.W == 2 * a.src.Ws, "conv: upsample geometry mismatch");
    if (a.src.rs == 2) CHECK_ARG(2 * a.H == a.src.Hs && 2 * a.W == a.src.Ws, "conv: pool geometry mismatch");
    if (a.src.rs == 0) CHECK_ARG(a.H == a.src.Hs && a.W == a.src.Ws, "conv: geometry mismatch");
    if (a.ra) CHECK_ARG(a.Ca <= a.Cout && a.Ca <= a.Cra && a.Ca % 8 == 0 && a.Cra % 8 == 0, "conv: residual channel slice out of range");
    if (a.ra && a.ra_rs == 1) CHECK_ARG(a.H % 2 == 0 && a.W % 2 == 0, "conv: upsampled residual needs even H, W");
    if (a.rb) CHECK_ARG(a.Crb % 8 == 0 && a.Crb >= a.Cout - a.Ca, "conv: bad residual-B channel count");
    CHECK_ARG(a.n_per_event >= 0 && (a.n_per_event == 0 || a.N % a.n_per_event == 0), "conv: N=%d is not a whole number of events of %d images", a.N, a.n_per_event);
    if (a.bnb_scale != nullptr)
        CHECK_ARG(a.src.scale == nullptr && a.src.relu == 0 && a.src.rs == 0 && a.bnb_shift != nullptr && a.mask != nullptr && a.stats != nullptr && a.n_per_event == 1 && a.bias == nullptr && a.Cout % 8 == 0 &&
                  (a.bnb_nstride == 0 || a.bnb_nstride >= a.Cout), "conv: BatchNorm-backward epilogue needs a plain prologue, shift, x (mask), per-image accumulators (n_per_event = 1), no bias");
    if (a.stats != nullptr && a.n_per_event > 0 && a.n_per_event < a.N)
        CHECK_ARG(((long)a.n_per_event * a.H * a.W) % 128 == 0, "conv: per-event statistics need n_per_event*H*W %% 128 == 0 (%d*%d*%d)", a.n_per_event, a.H, a.W);
    const double flops = 2.0 * a.N * a.H * a.W * (double)a.Cout * a.taps * a.Cin;
    // algorithmic bytes: source + output + the epilogue operands (ReLU mask, residual slices at their own resolution)
    double bytes = 2.0 * a.N * ((double)a.src.Hs * a.src.Ws * a.Cin + (double)a.H * a.W * a.Cout);
    const double bytes_min = bytes;      // SURVEY 8d / tools/arch_calc.py: source + output only
    if (a.mask) bytes += 2.0 * a.N * (double)a.H * a.W * a.Cout;
    if (a.ra) bytes += 2.0 * a.N * (double)a.H * a.W * a.Ca * (a.ra_rs == 1 ? 0.25 : (a.ra_rs == 2 ? 4.0 : 1.0));
    if (a.rb) bytes += 2.0 * a.N * (double)a.H * a.W * (a.Cout - a.Ca);
    const bool halo = a.taps == 9 && a.src.rs != 2 && a.W >= 16 && a.H >= 4 && a.Cin % 16 == 0 && !(a.flags & IEAGAN_CONV_FORCE_GATHER) &&
                      (size_t)(HT_H + 2) * (HT_W + 2) * (a.Cin * 2 + 16) <= 150 * 1024 && a.Cin <= AFF_MAXC;
    char tag[64] = "";
    if (prof_tags_on())
        snprintf(tag, sizeof(tag), "ci%d co%d %dx%d rs%d a%d r%d res%d%s", a.Cin, a.Cout, a.H, a.W, a.src.rs, a.src.scale != nullptr,
                 a.src.relu, a.ra ? a.ra_rs + 1 : 0, a.mask ? " mask" : "");
    ProfScope prof(a.taps == 9 ? (halo ? "conv3x3_halo" : "conv3x3_gather") : "conv1x1_gather", flops, bytes, st, tag, bytes_min);
    int rc;
    if (halo) {
        // C = 64 / 128: weights + halo resident in LDS (conv3x3_lds.hip); everything else: conv3x3_halo
        rc = (a.flags & IEAGAN_CONV_NO_LDS_WEIGHTS) ? 0 : conv3x3_lds_launch(a, st);
        if (rc < 0) return rc;
        // C = 16 / 32 on large maps: wave-specialised producer / consumer kernel (conv3x3_ws.hip)
        if (rc == 0 && !(a.flags & IEAGAN_CONV_NO_LDS_WEIGHTS)) rc = conv3x3_ws_launch(a, st);
        if (rc < 0) return rc;
        if (rc == 0) rc = (a.src.rs == 0) ? launch_halo_pro<0>(a, st) : launch_halo_pro<1>(a, st);
        else rc = 0;
    } else if (a.taps == 9) {
        if (a.src.rs == 0) rc = launch_gather_pro<9, 0>(a, st);
        else if (a.src.rs == 1) rc = launch_gather_pro<9, 1>(a, st);
        else { ieagan_set_error("conv: 3x3 with pooled source is not instantiated"); return IEAGAN_EINVAL; }
    } else {
        if (a.src.rs == 0) {
            // large maps: the streaming kernel (prefetched operands, weights in registers); otherwise one tile per block
            if (!(a.flags & IEAGAN_CONV_FORCE_GATHER) && conv1x1_stream_launch(a, st)) rc = 0;
            else if (!(a.flags & IEAGAN_CONV_FORCE_GATHER) && conv1x1_tile_launch(a, st)) rc = 0;      // operands through LDS in full rows
            else rc = launch_gather_pro<1, 0>(a, st);
        }
        else if (a.src.rs == 2) {
            if (!(a.flags & IEAGAN_CONV_FORCE_GATHER) && conv1x1_tile_launch(a, st)) rc = 0;
            else rc = launch_gather_pro<1, 2>(a, st);
        }
        else { ieagan_set_error("conv: 1x1 with upsampled source is not instantiated"); return IEAGAN_EINVAL; }
    }
    if (conv_plan_ctx().on) return rc;
    CHECK_LAUNCH("conv_forward");
    return rc;
}

