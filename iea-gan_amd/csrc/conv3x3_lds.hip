// conv3x3_lds: the 3x3 convolutions with Cin = Cout = 64 / 128 (SNConv2d 3x3 of the 16x48 ... 64x192 stages of G and D,
// model.py:37-42, 515-520; forward and -- with the flipped / transposed pack -- dgrad).  These are the only layers of the
// network at or above the bf16 ridge (288-576 FLOP/B), so the kernel is built around the MFMA pipe:
//   * the whole weight slice of the block ([NT*16 couts][9*CIN] bf16, 73.7 KB) AND the input halo live in LDS, both UNPADDED
//     with an XOR swizzle of their 16-byte chunks (a padded image does not fit 160 KB next to the weights); every MFMA operand
//     is one conflict-free ds_read_b128.  (Measured on the previous design: the per-wave 16-byte weight fetches through L1 ran
//     at ~16 B/clk/CU and left the MFMAs waiting -- K loop alone 60 of 106 us at 64x192.)
//   * the weights are loaded once per (persistent) block, the halo of the NEXT tile is requested into registers before the K loop
//     of the current one; the K loop runs from LDS only, software-pipelined one k-step ahead;
//   * C = 64: 8 waves share one 8x32-pixel tile x 64 couts (two waves per SIMD: one wave's LDS reads overlap the other's
//     MFMAs), ONE round of persistent blocks; C = 128: 4 waves, 8x16 pixels x 32 couts -- small tiles because these layers live
//     on 8x24 / 16x48 maps and need >= 256 blocks to fill the chip.
// Prologue (per-(n,c) affine + ReLU, nearest x2 upsample of the source) and epilogue (bias, ReLU mask, residual, statistics,
// BatchNorm-backward mode) are those of conv3x3_halo (shared code in conv_common.h).
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"

// 16-byte chunk swizzle of an LDS image with CPR chunks per row.  Rows are CPR*16 bytes apart, so the 16 lanes of a fragment
// read (same chunk of 16 consecutive rows) would hit only one or two of the sixteen 16-byte bank slots.  The chunk index is
// XOR-ed with a per-row key:  CPR % 16 == 8 -> the row parity already selects one half of the slots, permute the low 3 chunk
// bits with key = (r >> 1) & 7;  CPR % 16 == 0 -> permute 4 bits with key = r & 15.  `r` only has to enumerate the 16 rows of
// a fragment read bijectively: the weight image uses the row index, the halo image the COLUMN of the pixel inside its halo row
// (16 consecutive pixels of a row), which keeps every read address = per-lane register + compile-time immediate.
// (CPR % 16 == 4, the fp8 images of 64-channel rows: two row bits select a quarter of the slots, key = (r >> 2) & 3 permutes 2 bits.)
template <int CPR>
__device__ __forceinline__ int swz_key(int r) {
    static_assert(CPR % 16 == 8 || CPR % 16 == 0 || CPR % 16 == 4, "swizzle derived for rows of 4, 8 or 16 (mod 16) chunks");
    return (CPR % 16 == 8) ? ((r >> 1) & 7) : (CPR % 16 == 4 ? ((r >> 2) & 3) : (r & 15));
}
template <int CPR>
__device__ __forceinline__ int swz(int key, int chunk) {
    constexpr int G = (CPR % 16 == 8) ? 8 : (CPR % 16 == 4 ? 4 : 16);
    return (chunk & ~(G - 1)) | ((chunk ^ key) & (G - 1));
}

// LDS image of one block (weight slice + halo): above 80 KB only ONE block fits a CU, whatever the register budget says
// (the halo region also carries the per-wave epilogue transpose buffers: sized for the larger of the two)
template <int CIN, int NT, int TH, int TW, int NW = 0>
constexpr int lds_image_bytes() {
    const int halo = (TH + 2) * (TW + 2) * CIN * 2, epi = NW * EpiLds<NT>::FLOATS * 4;
    return NT * 16 * 9 * CIN * 2 + (halo > epi ? halo : epi);
}

// FULL: H % TH == 0 and W % TW == 0 -- every pixel of every tile exists, the epilogue's stores are unconditional and the compiler can
// count them when the prefetched halo of the next tile is consumed (instead of draining the counter, stores included).
template <bool AFF, bool RELU, int RS, int CIN, int NT, int TH, int TW, int NW, bool BNB, bool FULL = false>
__global__ __launch_bounds__(NW * 64, (lds_image_bytes<CIN, NT, TH, TW, NW>() > 80 * 1024 ? 1 : NW / 4)) void conv3x3_lds_kernel(ConvArgs a, int tiles_w, int tiles_h, int tpe, int tpb, int nblk, int bpe) {
    constexpr int AW = TW + 2, AH = TH + 2;
    constexpr int K = 9 * CIN;
    constexpr int KS = K / 32;
    constexpr int CH = CIN / 8;                 // 16-byte chunks per halo pixel
    constexpr int WCH = K / 8;                  // 16-byte chunks per weight row
    constexpr int MTW = (TH * TW / 16) / NW;    // m-tiles (16 pixels of one tile row) per wave
    constexpr int MPR = TW / 16;                // m-tiles per tile row
    constexpr int NTHR = NW * 64;
    static_assert(MTW >= 2 && MTW % 2 == 0 && (TH * TW / 16) % NW == 0, "each wave owns an even number of m-tiles");
    constexpr int W_BYTES = NT * 16 * K * 2;
    constexpr int HALO_BYTES = AH * AW * CIN * 2;
    constexpr int EPI_BYTES = NW * EpiLds<NT>::FLOATS * 4;
    static_assert(NW * STATS_SX_FLOATS * 4 <= (HALO_BYTES > EPI_BYTES ? HALO_BYTES : EPI_BYTES), "epilogue / fold scratch reuse the halo region");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    char* wsm = smem_all;                       // weights
    char* hsm = smem_all + W_BYTES;             // halo, later the epilogue transpose buffers
    __shared__ float red[NW * NT * 16 * 2];
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * AFF_MAXC : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W;
    // XCD-aware order, the n-tile column (C = 128: four blocks of 32 couts per tile range) as the fast index: the blocks that read
    // the same halos run side by side on one XCD and share them through its L2 (conv1x1_stream.hip)
    const int gy = (a.Cout + NT * 16 - 1) / (NT * 16);
    int bid = blockIdx.x;
    {
        const int tot = nblk * gy, q = tot / 8, r = tot % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n_base = (bid % gy) * NT * 16;
    bid /= gy;
    const int event = bid / bpe;
    const int t0 = event * tpe + (bid - event * bpe) * tpb;
    const int t1 = min(t0 + tpb, (event + 1) * tpe);
    if (t0 >= t1) return;

    // ---- weights -> LDS (swizzled), requested once per block
    {
        constexpr int TOT = NT * 16 * WCH;
#pragma unroll
        for (int idx = threadIdx.x; idx < TOT; idx += NTHR) {
            const int row = idx / WCH, kc = idx - row * WCH;
            bf16x8 v = zero8();
            if (n_base + row < a.Cout) v = *(const bf16x8*)((const bf16*)a.w + (long)(n_base + row) * a.Kpad + kc * 8);
            *(bf16x8*)(wsm + (row * WCH + swz<WCH>(swz_key<WCH>(row), kc)) * 16) = v;
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    int aff_n = -1;
    // The raw halo of tile t+1 is requested into registers right before the K loop of tile t and written to LDS (prologue applied)
    // after tile t's epilogue: with one block per CU (117 / 120 KB of LDS) nothing else hides a tile's load latency.  The requests
    // are unconditional (coordinates clamped into the image, the padding ring zeroed through a mask; past the block's last tile
    // the last tile is requested again and dropped) so that the compiler can count what is in flight behind them.
    // (C = 128: 16x48 29.7 -> 27.2 us, 8x24 20.3 -> 16.8 us.  C = 64 took this form in round 4 with the 8x32 tile: the 16x32 tile it
    //  replaced sat at the 256-register cap of an 8-wave block and spilled once the prefetch registers were added; 8x32 halves the
    //  accumulators -- 64x192 forward 63.6 -> 56.1 us, with BatchNorm prologue 70.8 -> 58.0, masked dgrad 73.1 -> 66.0 on one box.)
    // (the 4-wave C = 64 form of the small maps runs two blocks per CU, which overlap each other: with the prefetch 20.6 -> 22.8 us at 32x96)
    constexpr bool XPF = CIN >= 128 || NW == 8;
    constexpr int HTOT = AH * AW * CH, HIT = XPF ? (HTOT + NTHR - 1) / NTHR : 1;
    bf16x8 rawn[HIT];
    unsigned okn = 0;
    auto tile_of = [&](int t, int& n, int& h0, int& w0) {
        n = t / (tiles_w * tiles_h);
        const int trem = t - n * tiles_w * tiles_h;
        h0 = (trem / tiles_w) * TH;
        w0 = (trem % tiles_w) * TW;
    };
    auto request = [&](int t) {
        int n, h0, w0;
        tile_of(t, n, h0, w0);
        okn = 0;
#pragma unroll
        for (int j = 0; j < HIT; ++j) {
            const int idx = min((int)threadIdx.x + j * NTHR, HTOT - 1);
            const int hp = idx / CH, cc = idx - hp * CH;
            const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
            const int hc = min(max(hh, 0), H - 1), wc = min(max(ww, 0), W - 1);
            const int sh_ = (RS == 1) ? (hc >> 1) : hc, sw_ = (RS == 1) ? (wc >> 1) : wc;
            rawn[j] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
            if ((int)threadIdx.x + j * NTHR < HTOT && hh >= 0 && hh < H && ww >= 0 && ww < W) okn |= 1u << j;
        }
    };
    if constexpr (XPF) request(t0);
    for (int t = t0; t < t1; ++t) {
        int n, h0, w0;
        tile_of(t, n, h0, w0);
        if (AFF && n != aff_n) {
            __syncthreads();
            stage_aff(aff_s, a.src, n, CIN);
            aff_n = n;
        }
        __syncthreads();              // previous tile's epilogue has left the halo region; the BatchNorm table is in place
        if constexpr (XPF) {
        // ---- halo -> LDS with the prologue applied
    #pragma unroll
            for (int j = 0; j < HIT; ++j) {
                const int idx = threadIdx.x + j * NTHR;
                if (idx >= HTOT) continue;
                const int hp = idx / CH, cc = idx - hp * CH;
                bf16x8 o = zero8();
                if (okn & (1u << j)) {
                    if (!AFF && RELU) {
                        o = relu8(rawn[j]);
                    } else if (AFF || RELU) {
                        float v[8];
    #pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = bf2f(rawn[j][i]);
                        xform8<AFF, RELU>(v, a.src, n, cc * 8, aff_s);
    #pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                    } else {
                        o = rawn[j];
                    }
                }
                *(bf16x8*)(hsm + (hp * CH + swz<CH>(swz_key<CH>(hp % AW), cc)) * 16) = o;
            }
        } else {
        // ---- halo -> LDS with the prologue applied: batches of SB independent loads per thread
            {
                constexpr int TOT = AH * AW * CH, ITERS = (TOT + NTHR - 1) / NTHR, SB = 10;
    #pragma unroll
                for (int b0 = 0; b0 < ITERS; b0 += SB) {
                    bf16x8 rawb[SB];
                    unsigned okb = 0;
    #pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        const int idx = threadIdx.x + (b0 + j) * NTHR;
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
                        const int idx = threadIdx.x + (b0 + j) * NTHR;
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
                                xform8<AFF, RELU>(v, a.src, n, cc * 8, aff_s);
    #pragma unroll
                                for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                            } else {
                                o = rawb[j];
                            }
                        }
                        *(bf16x8*)(hsm + (hp * CH + swz<CH>(swz_key<CH>(hp % AW), cc)) * 16) = o;
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (XPF) request(min(t + 1, t1 - 1));  // in flight during the K loop and the epilogue below
        // ---- K loop from LDS: wave owns m-tiles [wave*MTW, +MTW) (tile row mt / MPR, columns (mt % MPR)*16 ..)
        f32x4 acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Fragment addresses = per-lane register (swizzled chunk offset) + immediate (tap / k-step / n-tile displacement):
        //   A: pixel (row mt/MPR + dy, col (mt%MPR)*16 + lr + dx) of the halo, chunk cq*4 + lg of its CIN/8 chunks
        //   B: weight row nt*16 + lr, chunk ks*4 + lg
        constexpr int NV = CH / 4;          // swizzle variants of the chunk index (CIN / 32)
        int offA[MTW][3][NV];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int mt = wave * MTW + m;
            const int col = (mt % MPR) * 16 + lr;
            const int base = ((mt / MPR) * AW + col) * CH * 16;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int v = 0; v < NV; ++v) offA[m][dx][v] = base + swz<CH>(swz_key<CH>(col + dx), v * 4 + lg) * 16;
        }
        int offB[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) offB[v] = lr * WCH * 16 + swz<WCH>(swz_key<WCH>(lr), v * 4 + lg) * 16;
        auto lda = [&](int ks, bf16x8(&x)[MTW]) {
            const int tap = (ks * 32) / CIN, cq = ((ks * 32) % CIN) / 32;
            const int imm = ((tap / 3) * AW + (tap % 3)) * CH * 16;
#pragma unroll
            for (int m = 0; m < MTW; ++m) x[m] = *(const bf16x8*)(hsm + offA[m][tap % 3][cq] + imm);
        };
        auto ldb = [&](int ks, bf16x8(&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *(const bf16x8*)(wsm + offB[ks % NV] + nt * 16 * WCH * 16 + (ks / NV) * NV * 64);
        };
        bf16x8 aq[2][MTW], bq[2][NT];
        lda(0, aq[0]);
        ldb(0, bq[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
                lda(ks + 1, aq[(ks + 1) & 1]);
                ldb(ks + 1, bq[(ks + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ks & 1][m], bq[ks & 1][nt], acc[m][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();              // halo consumed: its LDS becomes the per-wave epilogue transpose buffers
        float* epi = (float*)hsm + wave * EpiLds<NT>::FLOATS;
#pragma unroll
        for (int half = 0; half < MTW / 2; ++half) {
            const int mt = wave * MTW + 2 * half;          // two m-tiles: 32 consecutive pixels of one tile row when MPR == 2,
            auto pix = [&](int row, long& m, int& nn, int& h, int& w) -> bool {     // one m-tile of each of two rows when MPR == 1
                const int mtt = mt + (row >> 4);
                nn = n;
                h = h0 + mtt / MPR;
                w = w0 + (mtt % MPR) * 16 + (row & 15);
                m = ((long)n * H + h) * W + w;
                return FULL ? true : (h < H && w < W);
            };
            const f32x4(&sub)[2][NT] = *reinterpret_cast<const f32x4(*)[2][NT]>(&acc[2 * half]);
            conv_epilogue<BNB, NT, 2, false>(a, sub, epi, n_base, pix, s1, s2);
        }
    }
    if (a.stats != nullptr) {
        __syncthreads();
        stats_flush<NT, NW>(a, s1, s2, n_base, red, (float*)hsm, bid, event);
    }
}

template <bool AFF, bool RELU, int RS, bool BNB>
static int lds_launch(const ConvArgs& a, hipStream_t st) {
    const int n_events = (a.stats != nullptr && a.n_per_event > 0) ? a.N / a.n_per_event : 1;
#define LDS_LAUNCH(CINV, NTV, THV, TWV, NWV)                                                                                       \
    {                                                                                                                              \
        const int tiles_w = (a.W + TWV - 1) / TWV, tiles_h = (a.H + THV - 1) / THV;                                                \
        const int ntiles = a.N * tiles_w * tiles_h;                                                                                \
        const int tpe = ntiles / n_events;                                                                                         \
        const int gy = (a.Cout + 16 * NTV - 1) / (16 * NTV);                                                                       \
        /* ONE round of persistent blocks (one or two per CU, by the LDS image): a block loads its weight slice once and keeps the \
           next tile's halo in flight while it computes */                                                                         \
        const int slots = 256 * (lds_image_bytes<CINV, NTV, THV, TWV, NWV>() > 80 * 1024 ? 1 : 2);                \
        int tpb = (ntiles * gy + slots - 1) / slots;                                                                               \
        if (tpb < 1) tpb = 1;                                                                                                      \
        if (tpb > tpe) tpb = tpe;                                                                                                  \
        const int bpe = (tpe + tpb - 1) / tpb;                                                                                     \
        const int nblk = bpe * n_events;                                                                                           \
        CONV_PLAN_POINT(bpe, 1)                                                                                                    \
        const size_t lds = (size_t)lds_image_bytes<CINV, NTV, THV, TWV, NWV>();                             \
        const bool full = a.H % THV == 0 && a.W % TWV == 0;                                                                        \
        auto kern = full ? conv3x3_lds_kernel<AFF, RELU, RS, CINV, NTV, THV, TWV, NWV, BNB, true>                                  \
                         : conv3x3_lds_kernel<AFF, RELU, RS, CINV, NTV, THV, TWV, NWV, BNB, false>;                                \
        static bool attr_set[2] = {false, false};                                                                                  \
        if (!attr_set[full]) {                                                                                                     \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {      \
                ieagan_set_error("conv3x3_lds: cannot reserve %zu bytes of LDS", lds);                                             \
                return IEAGAN_ELAUNCH;                                                                                             \
            }                                                                                                                      \
            attr_set[full] = true;                                                                                                 \
        }                                                                                                                          \
        hipLaunchKernelGGL(kern, dim3(nblk * gy), dim3(NWV * 64), lds, st, a, tiles_w, tiles_h, tpe, tpb, nblk, bpe);               \
        return 1;                                                                                                                  \
    }
    // C = 64 on the smallest maps (16x48: 80 tile pairs): the 8-wave blocks would cover a third of the CUs -- the 4-wave 8x16 form with
    // 32 output channels per block (60 KB of LDS: two blocks per CU) gives 8x as many blocks (16x48: 9.4 vs 11.9 us; 32x96 is better off with
    // the 8-wave form since it prefetches: 20.3 -> 19.2 us)
    if (a.Cin == 64 && a.Cout % 64 == 0 && a.H >= 8 && a.W >= 16 &&
        (long)a.N * ((a.H + 15) / 16) * ((a.W + 31) / 32) * (a.Cout / 64) < 100) LDS_LAUNCH(64, 2, 8, 16, 4)
    if (a.Cin == 64 && a.Cout % 64 == 0 && a.H >= 16 && a.W >= 32) LDS_LAUNCH(64, 4, 8, 32, 8)
    if (a.Cin == 128 && a.Cout % 32 == 0 && a.H >= 8 && a.W >= 16) LDS_LAUNCH(128, 2, 8, 16, 4)
#undef LDS_LAUNCH
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// fp8 variant (BASELINE configs[4]): the same layers with OCP e4m3 MFMA operands (v_mfma_f32_16x16x32_fp8_fp8), fp32 accumulate.
//   * scales: one per BLOCK weight slice (amax over its NT*16 x 9*CIN weights) and one per TILE of activations (amax over the
//     transformed halo), both mapped to 224 = half of e4m3's largest finite value; the accumulators are rescaled by
//     1 / (s_w * s_a) before the epilogue.  Finer than per-tensor scales and needs no extra pass over memory.
//   * LDS images hold 1 byte per element: weights 36.9 KB + halo (10 x 34 pixels, 8x32 tiles) 21.8 KB -> TWO 4-wave blocks per CU,
//     so one block's staging / epilogue runs under the other's MFMAs (the bf16 image of the same tile shape does not fit twice).
//   * a lane's 16-byte chunk = 16 consecutive k of its row = the operands of TWO MFMAs (k order inside a 64-wide pair is a free
//     choice as long as both operands agree).
//   * MX = true (default): the block-scaled form v_mfma_scale_f32_16x16x128_f8f6f4 with all E8M0 block scales = 1 (the per-slice /
//     per-tile fp32 scales above carry the range): one instruction consumes 128 k = the two 16-byte chunks a lane holds for two
//     consecutive 64-k pairs, at TWICE the bf16 MFMA rate (the non-scaled K = 32 form runs at the bf16 rate on gfx950).  K = 9 * 64
//     leaves one 64-k pair over, which takes the non-scaled instruction.
//   * forward AND dgrad launches (ReLU-mask and BatchNorm-backward epilogues included); the weight gradients stay bf16.
// HBM tensors stay bf16: only the operand precision of the MFMAs changes.
// ------------------------------------------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef long i64;

__device__ __forceinline__ i64 pack_fp8x8(const float (&v)[8], float scale) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * scale, v[1] * scale, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * scale, v[3] * scale, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * scale, v[5] * scale, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * scale, v[7] * scale, hi, true);
    return (i64)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

// max over the block of a non-negative per-thread value (NW waves)
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch /*[NW]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();                 // scratch free
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = scratch[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, scratch[w]);
    return m;
}

template <bool AFF, bool RELU, int RS, int CIN, int NT, int TH, int TW, int NW, bool BNB, bool MX>
__global__ __launch_bounds__(NW * 64, 2) void conv3x3_lds_fp8_kernel(ConvArgs a, int tiles_w, int tiles_h, int tpe, int tpb, int nblk, int bpe) {
    constexpr int AW = TW + 2, AH = TH + 2;
    constexpr int K = 9 * CIN;
    constexpr int KP = K / 64;                  // k pairs: 64 k = two MFMA k-steps per 16-byte chunk
    constexpr int CH = CIN / 16;                // 16-byte (16 x fp8) chunks per halo pixel
    constexpr int WCH = K / 16;                 // 16-byte chunks per weight row
    constexpr int MTW = (TH * TW / 16) / NW;
    constexpr int MPR = TW / 16;
    constexpr int NTHR = NW * 64;
    static_assert(MTW >= 2 && MTW % 2 == 0 && (TH * TW / 16) % NW == 0, "each wave owns an even number of m-tiles");
    constexpr int W_BYTES = NT * 16 * K;
    constexpr int HALO_BYTES = AH * AW * CIN;
    constexpr int EPI_BYTES = NW * EpiLds<NT>::FLOATS * 4;
    constexpr int H_REGION = HALO_BYTES > EPI_BYTES ? HALO_BYTES : EPI_BYTES;
    static_assert(NW * STATS_SX_FLOATS * 4 <= H_REGION, "fold scratch reuses the halo region");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    char* wsm = smem_all;
    char* hsm = smem_all + W_BYTES;
    __shared__ float red[NW * NT * 16 * 2];
    __shared__ float mx[NW];
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * AFF_MAXC : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W;
    const int gy = (a.Cout + NT * 16 - 1) / (NT * 16);         // n-tile column = fast index of the XCD-aware order (see the bf16 kernel)
    int bid = blockIdx.x;
    {
        const int tot = nblk * gy, q = tot / 8, r = tot % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n_base = (bid % gy) * NT * 16;
    bid /= gy;
    const int event = bid / bpe;
    const int t0 = event * tpe + (bid - event * bpe) * tpb;
    const int t1 = min(t0 + tpb, (event + 1) * tpe);
    if (t0 >= t1) return;

    // ---- weights: bf16 pack -> amax of the block's slice -> e4m3 image in LDS
    float inv_sw;
    {
        constexpr int TOT = NT * 16 * (K / 8), PER = (TOT + NTHR - 1) / NTHR, SB = 6;      // 8-value source chunks
        // two passes over the (L2-resident) pack instead of holding PER chunks in registers across the block reduction
        auto wload = [&](int j) -> bf16x8 {
            const int idx = threadIdx.x + j * NTHR;
            if (idx >= TOT) return zero8();
            const int row = idx / (K / 8), kc = idx - row * (K / 8);
            if (n_base + row >= a.Cout) return zero8();
            return *(const bf16x8*)((const bf16*)a.w + (long)(n_base + row) * a.Kpad + kc * 8);
        };
        float am = 0.f;
        for (int j0 = 0; j0 < PER; j0 += SB) {
            bf16x8 wr[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) wr[j] = (j0 + j < PER) ? wload(j0 + j) : zero8();
#pragma unroll
            for (int j = 0; j < SB; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(bf2f(wr[j][i])));
        }
        am = block_max<NW>(am, mx);
        const float sw = 224.f / fmaxf(am, 1e-20f);
        inv_sw = 1.f / sw;
        for (int j0 = 0; j0 < PER; j0 += SB) {
            bf16x8 wr[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) wr[j] = (j0 + j < PER) ? wload(j0 + j) : zero8();
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int idx = threadIdx.x + (j0 + j) * NTHR;
                if (j0 + j >= PER || idx >= TOT) continue;
                const int row = idx / (K / 8), kc = idx - row * (K / 8);
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = bf2f(wr[j][i]);
                *(i64*)(wsm + (row * WCH + swz<WCH>(swz_key<WCH>(row), kc >> 1)) * 16 + (kc & 1) * 8) = pack_fp8x8(v, sw);
            }
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    int aff_n = -1;
    for (int t = t0; t < t1; ++t) {
        const int n = t / (tiles_w * tiles_h);
        const int trem = t - n * tiles_w * tiles_h;
        const int h0 = (trem / tiles_w) * TH, w0 = (trem % tiles_w) * TW;
        if (AFF && n != aff_n) {
            __syncthreads();
            stage_aff(aff_s, a.src, n, CIN);
            aff_n = n;
        }
        __syncthreads();
        // ---- halo: raw bf16 -> prologue -> amax of the tile -> e4m3 image in LDS
        float inv_sa;
        {
            constexpr int TOT = AH * AW * (CIN / 8), PER = (TOT + NTHR - 1) / NTHR, SB = 6;
            // two passes over the halo (second one from L1 / L2): nothing is held in registers across the block reduction
            auto hload = [&](int j, bool& ok) -> bf16x8 {
                const int idx = threadIdx.x + j * NTHR;
                ok = false;
                if (idx >= TOT) return zero8();
                const int hp = idx / (CIN / 8), cc = idx - hp * (CIN / 8);
                const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
                if (!(hh >= 0 && hh < H && ww >= 0 && ww < W)) return zero8();
                ok = true;
                const int sh_ = (RS == 1) ? (hh >> 1) : hh, sw_ = (RS == 1) ? (ww >> 1) : ww;
                return *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
            };
            auto xf = [&](int j, const bf16x8& raw, bool ok, float (&v)[8]) {
                const int idx = threadIdx.x + j * NTHR;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = bf2f(raw[i]);
                if (ok) xform8<AFF, RELU>(v, a.src, n, (idx % (CIN / 8)) * 8, aff_s);
            };
            float am = 0.f;
            for (int j0 = 0; j0 < PER; j0 += SB) {
                bf16x8 rawb[SB];
                bool okb[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) rawb[j] = hload(j0 + j, okb[j]);
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    float v[8];
                    xf(j0 + j, rawb[j], okb[j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(v[i]));
                }
            }
            am = block_max<NW>(am, mx);
            const float sa = 224.f / fmaxf(am, 1e-20f);
            inv_sa = 1.f / sa;
            for (int j0 = 0; j0 < PER; j0 += SB) {
                bf16x8 rawb[SB];
                bool okb[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) rawb[j] = hload(j0 + j, okb[j]);
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int idx = threadIdx.x + (j0 + j) * NTHR;
                    if (idx >= TOT) continue;
                    const int hp = idx / (CIN / 8), cc = idx - hp * (CIN / 8);
                    float v[8];
                    xf(j0 + j, rawb[j], okb[j], v);
                    *(i64*)(hsm + (hp * CH + swz<CH>(swz_key<CH>(hp % AW), cc >> 1)) * 16 + (cc & 1) * 8) = pack_fp8x8(v, sa);
                }
            }
        }
        __syncthreads();
        f32x4 acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // fragment addresses: per-lane swizzled chunk offset + immediates (see the bf16 kernel)
        constexpr int NVA = (CH >= 4) ? CH / 4 : 1;           // swizzle variants of the A chunk index (CIN / 64)
        constexpr int NVB = (WCH % 16 == 4) ? 1 : ((WCH % 16 == 8) ? 2 : 4);
        int offA[MTW][3][NVA];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int mt = wave * MTW + m;
            const int col = (mt % MPR) * 16 + lr;
            const int base = ((mt / MPR) * AW + col) * CH * 16;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int v = 0; v < NVA; ++v) offA[m][dx][v] = base + swz<CH>(swz_key<CH>(col + dx), v * 4 + lg) * 16;
        }
        int offB[NVB];
#pragma unroll
        for (int v = 0; v < NVB; ++v) offB[v] = lr * WCH * 16 + swz<WCH>(swz_key<WCH>(lr), v * 4 + lg) * 16;
        struct F2 { i64 lo, hi; };
        auto lda = [&](int kp, F2(&x)[MTW]) {
            const int tap = (kp * 64) / CIN, cq = ((kp * 64) % CIN) / 64;
            const int imm = ((tap / 3) * AW + (tap % 3)) * CH * 16;
#pragma unroll
            for (int m = 0; m < MTW; ++m) x[m] = *(const F2*)(hsm + offA[m][tap % 3][cq] + imm);
        };
        auto ldb = [&](int kp, F2(&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *(const F2*)(wsm + offB[kp % NVB] + nt * 16 * WCH * 16 + (kp / NVB) * NVB * 64);
        };
        if constexpr (MX) {
            // block-scaled form: one instruction per 128 k = the two 16-byte chunks a lane holds for two consecutive k pairs.
            // DB (small register tiles, C = 128): fragments one step ahead in software; otherwise (C = 64: 16 accumulator tiles) the
            // two resident blocks per CU cover each other's LDS latency -- a second fragment buffer would spill.
            constexpr int KQ = KP / 2;
            constexpr int ONE = 0x7F7F7F7F;           // E8M0 1.0 in every scale byte
            constexpr bool DB = MTW * NT <= 4;
            constexpr int NB = DB ? 2 : 1;
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            auto lda2 = [&](int q, i32x8(&x)[MTW]) {
                const int k0 = 2 * q, k1 = 2 * q + 1;
                const int tap0 = (k0 * 64) / CIN, cq0 = ((k0 * 64) % CIN) / 64, tap1 = (k1 * 64) / CIN, cq1 = ((k1 * 64) % CIN) / 64;
                const int imm0 = ((tap0 / 3) * AW + (tap0 % 3)) * CH * 16, imm1 = ((tap1 / 3) * AW + (tap1 % 3)) * CH * 16;
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const i32x4 lo = *(const i32x4*)(hsm + offA[m][tap0 % 3][cq0] + imm0);
                    const i32x4 hi = *(const i32x4*)(hsm + offA[m][tap1 % 3][cq1] + imm1);
                    x[m] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            auto ldb2 = [&](int q, i32x8(&b)[NT]) {
                const int k0 = 2 * q, k1 = 2 * q + 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const i32x4 lo = *(const i32x4*)(wsm + offB[k0 % NVB] + nt * 16 * WCH * 16 + (k0 / NVB) * NVB * 64);
                    const i32x4 hi = *(const i32x4*)(wsm + offB[k1 % NVB] + nt * 16 * WCH * 16 + (k1 / NVB) * NVB * 64);
                    b[nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            i32x8 aq[NB][MTW], bq[NB][NT];
            if (DB) {
                lda2(0, aq[0]);
                ldb2(0, bq[0]);
            }
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (DB) {
                    if (q + 1 < KQ) {
                        lda2(q + 1, aq[(q + 1) & 1]);
                        ldb2(q + 1, bq[(q + 1) & 1]);
                    }
                } else {
                    lda2(q, aq[0]);
                    ldb2(q, bq[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[m][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aq[DB ? (q & 1) : 0][m], bq[DB ? (q & 1) : 0][nt], acc[m][nt], 0, 0, 0, ONE,
                                                                                      0, ONE);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KP & 1) {                             // the left-over 64-k pair (K = 9 * 64): the non-scaled instruction
                F2 ta[MTW], tb[NT];
                lda(KP - 1, ta);
                ldb(KP - 1, tb);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ta[m].lo, tb[nt].lo, acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ta[m].hi, tb[nt].hi, acc[m][nt], 0, 0, 0);
                    }
            }
        } else {
        F2 aq[2][MTW], bq[2][NT];
        lda(0, aq[0]);
        ldb(0, bq[0]);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            if (kp + 1 < KP) {
                lda(kp + 1, aq[(kp + 1) & 1]);
                ldb(kp + 1, bq[(kp + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq[kp & 1][m].lo, bq[kp & 1][nt].lo, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq[kp & 1][m].hi, bq[kp & 1][nt].hi, acc[m][nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        const float inv = inv_sw * inv_sa;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][nt][r] *= inv;
        __syncthreads();
        float* epi = (float*)hsm + wave * EpiLds<NT>::FLOATS;
#pragma unroll
        for (int half = 0; half < MTW / 2; ++half) {
            const int mt = wave * MTW + 2 * half;
            auto pix = [&](int row, long& m, int& nn, int& h, int& w) -> bool {
                const int mtt = mt + (row >> 4);
                nn = n;
                h = h0 + mtt / MPR;
                w = w0 + (mtt % MPR) * 16 + (row & 15);
                m = ((long)n * H + h) * W + w;
                return h < H && w < W;
            };
            const f32x4(&sub)[2][NT] = *reinterpret_cast<const f32x4(*)[2][NT]>(&acc[2 * half]);
            conv_epilogue<BNB, NT, 2, false>(a, sub, epi, n_base, pix, s1, s2);
        }
    }
    if (a.stats != nullptr) {
        __syncthreads();
        stats_flush<NT, NW>(a, s1, s2, n_base, red, (float*)hsm, bid, event);
    }
}

template <bool AFF, bool RELU, int RS, bool BNB = false>
static int lds_fp8_launch(const ConvArgs& a, hipStream_t st) {
    const int n_events = (a.stats != nullptr && a.n_per_event > 0) ? a.N / a.n_per_event : 1;
#define F8_LAUNCH(CINV, NTV, THV, TWV, NWV)                                                                                        \
    {                                                                                                                              \
        const int tiles_w = (a.W + TWV - 1) / TWV, tiles_h = (a.H + THV - 1) / THV;                                                \
        const int ntiles = a.N * tiles_w * tiles_h;                                                                                \
        const int tpe = ntiles / n_events;                                                                                         \
        const int gy = (a.Cout + 16 * NTV - 1) / (16 * NTV);                                                                       \
        int tpb = (ntiles * gy + 1023) / 1024;          /* two resident blocks per CU */                                           \
        if (tpb < 1) tpb = 1;                                                                                                      \
        if (tpb > tpe) tpb = tpe;                                                                                                  \
        const int bpe = (tpe + tpb - 1) / tpb;                                                                                     \
        const int nblk = bpe * n_events;                                                                                           \
        CONV_PLAN_POINT(bpe, 1)                                                                                                    \
        const size_t halo = (size_t)(THV + 2) * (TWV + 2) * CINV, epi = (size_t)NWV * EpiLds<NTV>::FLOATS * 4;                     \
        const size_t lds = (size_t)NTV * 16 * 9 * CINV + (halo > epi ? halo : epi);                                                \
        if (a.flags & IEAGAN_CONV_FP8_NOSCALE)                                                                                     \
            hipLaunchKernelGGL((conv3x3_lds_fp8_kernel<AFF, RELU, RS, CINV, NTV, THV, TWV, NWV, BNB, false>), dim3(nblk * gy), dim3(NWV * 64), lds, \
                               st, a, tiles_w, tiles_h, tpe, tpb, nblk, bpe);                                                      \
        else                                                                                                                       \
            hipLaunchKernelGGL((conv3x3_lds_fp8_kernel<AFF, RELU, RS, CINV, NTV, THV, TWV, NWV, BNB, true>), dim3(nblk * gy), dim3(NWV * 64), lds, \
                               st, a, tiles_w, tiles_h, tpe, tpb, nblk, bpe);                                                      \
        return 1;                                                                                                                  \
    }
    // (C = 64 on the small maps -- fewer than 1000 tile-blocks -- stays with the bf16 4-wave 8x16 form: more, smaller blocks; measured)
    if (a.Cin == 64 && a.Cout % 64 == 0 && a.H >= 8 && a.W >= 32 &&
        (long)a.N * ((a.H + 7) / 8) * ((a.W + 31) / 32) * (a.Cout / 64) >= 1000) F8_LAUNCH(64, 4, 8, 32, 4)
    if (a.Cin == 128 && a.Cout % 32 == 0 && a.H >= 8 && a.W >= 16) F8_LAUNCH(128, 2, 8, 16, 4)
#undef F8_LAUNCH
    return 0;
}

// 1 = launched, 0 = not applicable (caller falls back to conv3x3_halo), < 0 = error
int conv3x3_lds_launch(const ConvArgs& a, hipStream_t st) {
    if (a.taps != 9 || (a.src.rs != 0 && a.src.rs != 1) || a.Kpad != 9 * a.Cin || (a.Cin != 64 && a.Cin != 128)) return 0;
    if (a.src.scale != nullptr && a.Cin > AFF_MAXC) return 0;
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (a.flags & IEAGAN_CONV_FP8) {          // forward launches and dgrad launches (plain prologue + ReLU-mask / BatchNorm-backward epilogue)
        int r = 0;
        if (a.bnb_scale != nullptr) r = (a.src.rs == 0 && !aff && !relu) ? lds_fp8_launch<false, false, 0, true>(a, st) : 0;
        else if (a.src.rs == 0) r = aff ? (relu ? lds_fp8_launch<true, true, 0>(a, st) : lds_fp8_launch<true, false, 0>(a, st))
                                   : (relu ? lds_fp8_launch<false, true, 0>(a, st) : lds_fp8_launch<false, false, 0>(a, st));
        else r = aff ? (relu ? lds_fp8_launch<true, true, 1>(a, st) : lds_fp8_launch<true, false, 1>(a, st))
                     : (relu ? lds_fp8_launch<false, true, 1>(a, st) : lds_fp8_launch<false, false, 1>(a, st));
        if (r != 0) return r;
    }
    if (a.src.rs == 0) {
        if (aff && relu) return lds_launch<true, true, 0, false>(a, st);
        if (aff) return lds_launch<true, false, 0, false>(a, st);
        if (relu) return lds_launch<false, true, 0, false>(a, st);
        if (a.bnb_scale != nullptr) return lds_launch<false, false, 0, true>(a, st);
        return lds_launch<false, false, 0, false>(a, st);
    }
    if (aff && relu) return lds_launch<true, true, 1, false>(a, st);
    if (aff) return lds_launch<true, false, 1, false>(a, st);
    if (relu) return lds_launch<false, true, 1, false>(a, st);
    return lds_launch<false, false, 1, false>(a, st);
}
