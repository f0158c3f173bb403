// conv3x3_ws: the HBM-bound 3x3 layers (Cin = Cout = 16 on 128x384 / 256x768 maps; layers.py:197-206 with the fused prologue /
// epilogue of conv_igemm.hip) as a wave-specialised kernel.
//
// conv3x3_halo runs the phases of a tile one after the other in every wave (wait for the halo -> transform + LDS write -> barrier ->
// MFMA loop -> epilogue -> stores): measured on the C = 16 layer at 256x768 they add up to 1.3x the kernel time, i.e. they overlap
// only through the other resident blocks, and HBM requests are in flight for ~30 % of a block's life (2.4 TB/s).
// Here a block is 8 waves: waves 4..7 (producers) only move data -- two tiles of raw halo chunks in flight in registers at any time,
// prologue transform (BatchNorm apply / ReLU / zero padding) and the LDS write of tile t+1 while waves 0..3 (consumers) run the MFMA
// loop and the epilogue of tile t from the other halo buffer.  One block barrier per tile; persistent blocks (2 per CU) walk
// consecutive tiles inside one statistics group, weights stay in LDS, the statistics are flushed once per block.
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"

#define WS_H 8
#define WS_W 32

// FULL: H % WS_H == 0 and W % WS_W == 0 -- every output pixel of every tile exists, so the epilogue's stores and the mask requests are
// unconditional and the compiler can COUNT them: a consumer then waits for the mask chunks of tile r only, not (vmcnt(0)) for the
// stores of tile r-1 that were issued behind them.
template <bool AFF, bool RELU, int RS, int CIN, bool BNB, bool MPF, int NCW, bool FULL>
__global__ __launch_bounds__((NCW + 4) * 64, (NCW == 4 ? 4 : 3)) void conv3x3_ws_kernel(ConvArgs a, int tiles_w, int tiles_h, int tpe, int tpb, int nblk, int bpe) {
    constexpr int NT = CIN / 16;
    constexpr int AW = WS_W + 2, AH = WS_H + 2;
    constexpr int PS = CIN * 2 + 16;                         // bytes per halo pixel (16 consecutive pixels -> 16 distinct 16-byte slots)
    constexpr int KP = ((9 * CIN + 31) / 32) * 32;
    constexpr int WSB = KP * 2 + 16;                         // bytes per weight row in LDS
    constexpr int CH = CIN / 8;                              // 16-byte chunks per pixel
    constexpr int TOT = AH * AW * CH;
    constexpr int PF = (TOT + 255) / 256;                    // chunks per producer thread and tile (3 / 6)
    constexpr int HALO_BYTES = AH * AW * PS;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    constexpr int RPW = WS_H / NCW, MTW = 2 * RPW;          // tile rows / m-tiles per consumer wave
    __shared__ float red[NCW * NT * 16 * 2];
    __shared__ __attribute__((aligned(32))) float aff_all[4 * 64];                        // per producer wave: [0, 32) scale, [32, 64) shift of the current image
    char* wlds = smem_all;                                   // [NT*16][WSB]
    char* halo0 = smem_all + NT * 16 * WSB;                  // two halo buffers
    float* epi_all = (float*)(halo0 + 2 * HALO_BYTES);       // NCW consumer waves x EpiLds<NT>::FLOATS (later: the statistics fold scratch)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool consumer = wave < NCW;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W;
    int bid = blockIdx.x;
    {   // XCD-aware order: each XCD (blocks b, b+8, ...) walks a contiguous run of tiles
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int event = bid / bpe;
    const int t0 = event * tpe + (bid - event * bpe) * tpb;
    const int t1 = min(t0 + tpb, (event + 1) * tpe);
    const int ntl = t1 - t0;

    // weights -> LDS once per block (all 8 waves)
    for (int idx = threadIdx.x; idx < NT * 16 * (KP / 8); idx += (NCW + 4) * 64) {
        const int row = idx / (KP / 8), kc = idx - row * (KP / 8);
        *(bf16x8*)(wlds + row * WSB + kc * 16) = *(const bf16x8*)((const bf16*)a.w + (long)row * KP + kc * 8);
    }

    auto tile_coords = [&](int t, int& n, int& h0, int& w0) {
        n = t / (tiles_w * tiles_h);
        const int trem = t - n * tiles_w * tiles_h;
        h0 = (trem / tiles_w) * WS_H;
        w0 = (trem % tiles_w) * WS_W;
    };

    // The two roles are separate loops (not one loop with a role branch inside): their register live ranges then do not overlap
    // and the kernel is allocated max(producer, consumer) registers instead of the sum.  Both execute the same barriers:
    // one after the prologue, one per tile, one inside / beside the statistics flush.
    if (!consumer) {
        // -------------------------------------------------------------------------------------- producers
        const int ptid = threadIdx.x - NCW * 64;
        bf16x8 rawA[PF], rawB[PF];
        unsigned okA = 0, okB = 0;
        float* aff_w = aff_all + (wave & 3) * 64;
        int aff_n = -1;
        auto load_tile = [&](int t, bf16x8(&raw)[PF], unsigned& okmask) {
            int n, h0, w0;
            tile_coords(t, n, h0, w0);
            okmask = 0;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                // UNCONDITIONAL loads (coordinates clamped into the image, the padding ring zeroed by okmask in store_tile): under a
                // per-lane condition the compiler cannot count the younger loads of the other register set and drains the counter
                // -- vmcnt(0) -- before every store_tile, which left ONE tile in flight instead of two
                const int idx = min(ptid + j * 256, TOT - 1);
                const int hp = idx / CH, cc = idx - hp * CH;
                const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
                const int hc = min(max(hh, 0), H - 1), wc = min(max(ww, 0), W - 1);
                const int sh_ = (RS == 1) ? (hc >> 1) : hc, sw_ = (RS == 1) ? (wc >> 1) : wc;
                raw[j] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
                if (ptid + j * 256 < TOT && hh >= 0 && hh < H && ww >= 0 && ww < W) okmask |= 1u << j;
            }
        };
        auto store_tile = [&](int t, const bf16x8(&raw)[PF], unsigned okmask, char* dst) {
            int n, h0, w0;
            tile_coords(t, n, h0, w0);
            if (AFF && n != aff_n) {       // this wave's private copy of image n's scale / shift rows (wave-ordered LDS: no block barrier)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int c = lane & 31;
                if (c < CIN) aff_w[lane] = (lane < 32 ? a.src.scale : a.src.shift)[(long)n * a.src.aff_nstride + c];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                aff_n = n;
            }
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int idx = ptid + j * 256;
                if (idx >= TOT) continue;
                const int hp = idx / CH, cc = idx - hp * CH;
                bf16x8 o = zero8();
                if (okmask & (1u << j)) {
                    if (!AFF && RELU) {
                        o = relu8(raw[j]);
                    } else if (AFF) {
                        const f32x8 sc = *(const f32x8*)(aff_w + cc * 8), sh = *(const f32x8*)(aff_w + 32 + cc * 8);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            float v = bf2f(raw[j][i]) * sc[i] + sh[i];
                            if (RELU) v = fmaxf(v, 0.f);
                            o[i] = f2bf(v);
                        }
                    } else {
                        o = raw[j];
                    }
                }
                *(bf16x8*)(dst + hp * PS + cc * 16) = o;
            }
        };
        // produce step r (runs next to the consumers' tile r): tile r+1 -> the other halo buffer, then request tile r+3 into the
        // register set that has just been drained
        // (requests past the block's last tile re-read that tile and are never stored: the request stays unconditional, and the loop is
        //  unrolled by the two register sets instead of branching on the parity of r -- a branch or a join in front of a use makes
        //  the compiler assume the youngest possible request and wait for everything in flight)
        auto produce = [&](int r, bf16x8(&raw)[PF], unsigned& okmask) {
            if (r + 1 < ntl) store_tile(t0 + r + 1, raw, okmask, halo0 + ((r + 1) & 1) * HALO_BYTES);
            load_tile(t0 + min(r + 3, ntl - 1), raw, okmask);
        };
        if (ntl <= 0) {                                       // (its own exit: no join in front of the loop below)
            __syncthreads();
            return;
        }
        load_tile(t0, rawA, okA);
        store_tile(t0, rawA, okA, halo0);
        load_tile(t0 + min(1, ntl - 1), rawB, okB);           // set B: odd tiles
        load_tile(t0 + min(2, ntl - 1), rawA, okA);           // set A: even tiles
        __syncthreads();
        for (int r = 0; r < ntl; r += 2) {
            produce(r, rawB, okB);                  // tile r+1 is odd -> set B
            __syncthreads();
            if (r + 1 >= ntl) break;
            produce(r + 1, rawA, okA);              // tile r+2 is even -> set A
            __syncthreads();
        }
        if (a.stats != nullptr) __syncthreads();              // matches the block barrier inside stats_flush
        return;
    }

    // ------------------------------------------------------------------------------------------ consumers
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    int pbase[MTW];   // byte offset of this lane's pixel in m-tile mt at tap (0,0): row RPW*wave + (mt>>1), col (mt&1)*16 + lr
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) pbase[mt] = ((RPW * wave + (mt >> 1)) * AW + (mt & 1) * 16 + lr) * PS;
    float* epi = epi_all + wave * EpiLds<NT>::FLOATS;
    // per-lane epilogue constants: the lane owns channel chunk lane % CPP of pixel row (it*64 + lane) / CPP of each half tile
    constexpr int CPP = NT * 2, EIT = (32 * CPP) / 64;
    const int ecc = lane % CPP;
    float bias_r[8];
    load_bias8<NT>(a, 0, bias_r);
    // ReLU mask / BatchNorm input of the epilogue: the chunks of tile r+1 are requested before the epilogue of tile r
    constexpr bool has_mask = MPF;                  // launcher: MPF == (a.mask != nullptr)
    bf16x8 mk_cur[RPW * EIT], mk_nxt[RPW * EIT];
    auto mask_request = [&](int t, bf16x8(&mk)[RPW * EIT]) {
        int n, h0, w0;
        tile_coords(t, n, h0, w0);
#pragma unroll
        for (int half = 0; half < RPW; ++half)
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int hh = h0 + RPW * wave + half, ww = w0 + (it * 64 + lane) / CPP;
                if (FULL) {
                    mk[half * EIT + it] = *(const bf16x8*)((const bf16*)a.mask + (((long)n * H + hh) * W + ww) * CIN + ecc * 8);
                } else {
                    mk[half * EIT + it] = zero8();
                    if (hh < H && ww < W) mk[half * EIT + it] = *(const bf16x8*)((const bf16*)a.mask + (((long)n * H + hh) * W + ww) * CIN + ecc * 8);
                }
            }
    };
    if (has_mask && ntl > 0) mask_request(t0, mk_cur);
    __syncthreads();
    for (int r = 0; r < ntl; ++r) {
        const char* smem = halo0 + (r & 1) * HALO_BYTES;
        int n, h0, w0;
        tile_coords(t0 + r, n, h0, w0);
        f32x4 acc[MTW][NT];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // one-step-ahead software pipeline over the K steps: fragments of step ks+1 requested before the MFMAs of step ks
        constexpr int KS = KP / 32;
        bf16x8 bq[CIN >= 32 ? 2 : 1][NT], aq[CIN >= 32 ? 2 : 1][MTW];
        auto ldb = [&](int ks, bf16x8(&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *(const bf16x8*)(wlds + (nt * 16 + lr) * WSB + (ks * 32 + lg * 8) * 2);
        };
        auto lda = [&](int ks, bf16x8(&x)[MTW]) {
            const int k = ks * 32 + lg * 8;               // lane-group dependent when CIN == 16 (two taps per K step)
            int tap = k / CIN;
            const int c = k - tap * CIN;
            const bool kval = tap < 9;                    // K padding (C = 16: 144 -> 160)
            if (!kval) tap = 0;
            const int toff = ((tap / 3) * AW + (tap - (tap / 3) * 3)) * PS + c * 2;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                x[mt] = *(const bf16x8*)(smem + pbase[mt] + toff);
                if (!kval) x[mt] = zero8();
            }
        };
        if constexpr (CIN >= 32) {
            ldb(0, bq[0]);
            lda(0, aq[0]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                    ldb(ks + 1, bq[(ks + 1) & 1]);
                    lda(ks + 1, aq[(ks + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ks & 1][mt], bq[ks & 1][nt], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {        // C = 16: 20 MFMAs per tile and wave -- the loop is not what the tile waits for; keep the registers
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ldb(ks, bq[0]);
                lda(ks, aq[0]);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[0][mt], bq[0][nt], acc[mt][nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int half = 0; half < RPW; ++half) {           // tile row RPW*wave + half: 32 pixels
            const int hh = h0 + RPW * wave + half;
            auto pix = [&](int row, long& m, int& nn, int& h, int& w) -> bool {
                nn = n;
                h = hh;
                w = w0 + row;
                m = ((long)n * H + h) * W + w;
                return FULL ? true : (h < H && w < W);
            };
            const f32x4(&sub)[2][NT] = *reinterpret_cast<const f32x4(*)[2][NT]>(&acc[2 * half]);
            if (half == 0 && has_mask) mask_request(t0 + min(r + 1, ntl - 1), mk_nxt);      // (last tile: re-requests itself, unused)
            if (has_mask) conv_epilogue<BNB, NT>(a, sub, epi, 0, pix, s1, s2, bias_r, &mk_cur[half * EIT]);      // (two calls: a selected
            else conv_epilogue<BNB, NT>(a, sub, epi, 0, pix, s1, s2, bias_r, nullptr);                           //  pointer would pin the arrays in scratch)
        }
        if (has_mask) {
#pragma unroll
            for (int i = 0; i < RPW * EIT; ++i) mk_cur[i] = mk_nxt[i];
        }
        __syncthreads();
    }
    if (a.stats != nullptr && ntl > 0) stats_flush<NT, NCW>(a, s1, s2, 0, red, epi_all, bid, event);
}

template <bool AFF, bool RELU, int RS, bool BNB, bool MPF, bool FULL>
static int ws_launch_f(const ConvArgs& a, hipStream_t st) {
    const int tiles_w = (a.W + WS_W - 1) / WS_W, tiles_h = (a.H + WS_H - 1) / WS_H;
    const int ntiles = a.N * tiles_w * tiles_h;
    const int n_events = (a.stats != nullptr && a.n_per_event > 0) ? a.N / a.n_per_event : 1;
    const int tpe = ntiles / n_events;
    int bpe = (a.Cin == 16 ? 512 : 256) / n_events;      // persistent blocks: two (C = 16) / one (C = 32) per CU
    if (bpe < 1) bpe = 1;
    int tpb = (tpe + bpe - 1) / bpe;
    if (tpb < 1) tpb = 1;
    bpe = (tpe + tpb - 1) / tpb;
    const int nblk = bpe * n_events;
#define WS_GO(CINV, NCWV)                                                                                                    \
    {                                                                                                                        \
        constexpr int NTV = CINV / 16;                                                                                       \
        constexpr int KPV = ((9 * CINV + 31) / 32) * 32;                                                                     \
        size_t tail = (size_t)NCWV * EpiLds<NTV>::FLOATS * 4;                                                                \
        if (tail < (size_t)NCWV * STATS_SX_FLOATS * 4) tail = (size_t)NCWV * STATS_SX_FLOATS * 4;                             \
        const size_t lds = (size_t)NTV * 16 * (KPV * 2 + 16) + 2 * (size_t)(WS_H + 2) * (WS_W + 2) * (CINV * 2 + 16) + tail;  \
        CONV_PLAN_POINT(bpe, 1)                                                                                              \
        auto kern = conv3x3_ws_kernel<AFF, RELU, RS, CINV, BNB, MPF, NCWV, FULL>;                                            \
        static bool attr_set = false;                                                                                        \
        if (!attr_set) {                                                                                                     \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
                ieagan_set_error("conv3x3_ws: cannot reserve %zu bytes of LDS", lds);                                        \
                return IEAGAN_ELAUNCH;                                                                                       \
            }                                                                                                                \
            attr_set = true;                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, dim3(nblk), dim3((NCWV + 4) * 64), lds, st, a, tiles_w, tiles_h, tpe, tpb, nblk, bpe);      \
    }
    // C = 16: 4 consumer + 4 producer waves, two blocks per CU
    WS_GO(16, 4)
#undef WS_GO
    return 1;
}

template <bool AFF, bool RELU, int RS, bool BNB, bool MPF>
static int ws_launch_c(const ConvArgs& a, hipStream_t st) {
    if (a.H % WS_H == 0 && a.W % WS_W == 0) return ws_launch_f<AFF, RELU, RS, BNB, MPF, true>(a, st);
    return ws_launch_f<AFF, RELU, RS, BNB, MPF, false>(a, st);
}

template <int RS>
static int ws_launch_pro(const ConvArgs& a, hipStream_t st) {
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (aff && !relu) return 0;                                              // no layer of the path has an affine prologue without ReLU
    if (a.mask != nullptr && (aff || relu)) return 0;                        // a masked epilogue belongs to a dgrad launch (plain prologue)
    if (aff) return ws_launch_c<true, true, RS, false, false>(a, st);
    if (relu) return ws_launch_c<false, true, RS, false, false>(a, st);
    if (RS == 0 && a.bnb_scale != nullptr) return ws_launch_c<false, false, (RS == 0 ? 0 : RS), (RS == 0), (RS == 0)>(a, st);
    if (RS == 0 && a.mask != nullptr) return ws_launch_c<false, false, (RS == 0 ? 0 : RS), false, (RS == 0)>(a, st);
    if (a.mask != nullptr) return 0;
    return ws_launch_c<false, false, RS, false, false>(a, st);
}

// 1 = launched, 0 = not applicable (the caller falls back to conv3x3_halo)
int conv3x3_ws_launch(const ConvArgs& a, hipStream_t st) {
    // C = 32 needs 92-110 KB of LDS, i.e. one block per CU: measured slower than conv3x3_halo's three 4-wave blocks with 4 consumer
    // waves (94 vs 77 us at 128x384) and with 8 (97 vs 73 us) -- nothing runs while the one block waits at its barrier.  Only the
    // C = 16 layers come here (the kernel stays generic in CIN / NCW).
    if (a.taps != 9 || a.Cin != a.Cout || a.Cin != 16 || a.src.rs == 2) return 0;
    if (a.Kpad != ((9 * a.Cin + 31) / 32) * 32 || a.src.Cx % 8 != 0) return 0;
    const int tiles = a.N * ((a.W + WS_W - 1) / WS_W) * ((a.H + WS_H - 1) / WS_H);
    if (tiles < 1024) return 0;                   // < 2 tiles per persistent block: nothing to pipeline
    return a.src.rs == 0 ? ws_launch_pro<0>(a, st) : ws_launch_pro<1>(a, st);
}
