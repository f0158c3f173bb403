// conv3x3_bwd: the WHOLE backward of a 3x3 convolution with Cin = Cout = 16 / 32 on a large feature map in ONE launch -- the
// batch-statistics term of the following BatchNorm applied on load (effgrad), the data gradient with the backward of the fused
// prologue (ReLU mask / BatchNorm apply [/ nearest x2 up-sample: 2x2 sum]) in its store phase, the weight gradient and the bias
// column sums.  Replaces, per layer and backward pass, the chain
//     effgrad (2R + 1W) -> conv_forward as dgrad (R g, R x, W dx) [-> prologue_bwd (R da, R x, W dx)] -> conv_wgrad (R x, R g) -> wgrad_reduce
// of autograd(F.conv2d(relu(bn(x)) | relu(x) | up(relu(bn(x))), W / sigma, b)) (reference layers.py:197-206, 656-689;
// model.py:54-71 GBlock, 541-557 DBlock).  These layers are HBM-bound (72-144 FLOP/B): g (and y) are read once with their halo,
// x once, dx is written once.
//
// Structure: persistent blocks of 4 waves walk consecutive 8 x 32-pixel tiles of ONE image.  Per tile
//   1. the raw 16-byte chunks requested one tile ago are retired: g_eff = g + dsum[c] + 2 y dsumsq[c] (zero outside the image) goes to
//      the (8+2) x (32+2) halo image in LDS, x (raw) to the tile image; both with a pixel stride that is an odd multiple of 32 bytes,
//      which serves the 16-byte row reads of the dgrad AND the transposed 8-byte reads of the wgrad without bank conflicts;
//   2. the next tile's chunks are requested (unconditional, clamped addresses: the compiler can count them);
//   3. dgrad: da[q] = sum_tap' g_eff[q + tap'] Wt[tap']  (A = halo rows, B = the flipped / transposed pack in LDS);
//   4. wgrad with the SAME halo image: dW[co][tap, ci] += sum_q g_eff[q - off(tap)][co] a[q][ci] -- the activated input a is only needed
//      at the tile's own pixels when the out-gradient carries the halo; both operands are read K(pixel)-major with
//      ds_read_b64_tr_b16, the prologue is applied to the fragment (a lane holds 8 pixels of ONE channel); accumulators stay in
//      registers over all tiles of the block (C = 16: every wave owns two tile rows and all nine taps; C = 32: wave = one
//      (cout tile, cin tile) pair over the whole tile); an all-ones B tile yields the bias column sums;
//   5. epilogue: accumulators transposed through LDS so that a lane owns 8 channels of one pixel; ReLU mask / BatchNorm backward
//      with x from the LDS tile (up-sampled source: 2x2 sum in fp32 first), 16-byte stores of dx.
// Block end: the partial dW of the four waves are folded in wave order (bit-reproducible) and leave as ONE slab; wgrad_reduce
// folds the slabs.  The per-image BatchNorm accumulators (sum d, sum d x) go to replica slot bid % 8 of the image.
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"
#include <stdio.h>

typedef ieagan_conv3x3_bwd_desc Bwd3Args;

#define B3_TH 8
#define B3_TW 32

template <int C>
struct B3Cfg {
    static constexpr int NT = C / 16, CH = C / 8, CPP = CH;
    static constexpr int PS = (C == 16) ? 32 : 96;                 // bytes per LDS pixel: an odd multiple of 32
    static constexpr int PSE = PS / 2;
    static constexpr int AW = B3_TW + 2, AH = B3_TH + 2;
    static constexpr int KP = ((9 * C + 31) / 32) * 32;            // 160 / 288
    static constexpr int KS = KP / 32;
    static constexpr int WSB = KP * 2 + 16;                        // bytes per weight row in LDS
    static constexpr int W_BYTES = C * WSB;
    static constexpr int G_BYTES = AH * AW * PS;
    static constexpr int X_BYTES = B3_TH * B3_TW * PS;
    static constexpr int XS_BYTES = 4 * 16 * PS;                   // up-sampled source: the compact 4 x 16 source tile
    static constexpr int LDW = NT * 16 + 4;
    static constexpr int E_BYTES = 4 * 32 * LDW * 4;               // four waves x 32 pixel rows (aliases the halo image)
    static constexpr int T_FLOATS = C * KP + C;                    // block-end fold: dW [C][KP] + column sums
    static constexpr int GTOT = AH * AW * CH, GCHN = (GTOT + 255) / 256;
    static_assert(E_BYTES <= G_BYTES, "the epilogue buffer aliases the halo image");
    static_assert(T_FLOATS * 4 <= G_BYTES + XS_BYTES, "the block-end fold aliases the halo + tile images");
    static_assert(4 * STATS_SX_FLOATS * 4 <= G_BYTES, "the statistics fold aliases the halo image");
};

// K(pixel)-major fragment (see conv_wgrad.hip: frag_T): rows pix0 .. pix0+3 and pix0+16 .. pix0+19 of column (col0 + lr).
__device__ __forceinline__ bf16x8 b3_frag_T(const char* lds, int stride_bytes, int pix0, int col0, int lr) {
    const int q = lr >> 2, p = lr & 3;
    const char* p0 = lds + (pix0 + q) * stride_bytes + (col0 + 4 * p) * 2;
    const char* p1 = p0 + 16 * stride_bytes;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hh[0]; f[5] = hh[1]; f[6] = hh[2]; f[7] = hh[3];
    return f;
}

// the same for an x2 nearest-up-sampled operand kept as its SOURCE tile [4][16] pixels: conv columns col0c + q and col0c + 16 + q of conv
// row 2*srow (+1) live at source pixel (srow, col >> 1)
__device__ __forceinline__ bf16x8 b3_frag_T_up(const char* lds, int stride_bytes, int srow, int col0c, int col0, int lr) {
    const int q = lr >> 2, p = lr & 3;
    const char* p0 = lds + (srow * 16 + ((col0c + q) >> 1)) * stride_bytes + (col0 + 4 * p) * 2;
    const char* p1 = p0 + 8 * stride_bytes;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hh[0]; f[5] = hh[1]; f[6] = hh[2]; f[7] = hh[3];
    return f;
}

// blocks per CU (= waves per SIMD): C = 16 -> 3 (168 registers), but 2 for the generator's variants, which carry the BatchNorm prologue AND the
// effgrad operand (their prefetch registers spill at 168); C = 32 -> 2 (76 KB of LDS)
template <int C, bool AFF, int RS, bool EFF>
__host__ __device__ constexpr int b3_occ() { return C == 16 ? ((AFF && EFF) ? 2 : 3) : 2; }

template <int C, bool AFF, int RS, bool EFF>
__global__ __launch_bounds__(256, (b3_occ<C, AFF, RS, EFF>())) void conv3x3_bwd_kernel(Bwd3Args a, int tiles_w, int tpi, int tpb, int bpi, int nblk) {
    typedef B3Cfg<C> K;
    constexpr int NT = K::NT, CH = K::CH, CPP = K::CPP, PS = K::PS, AW = K::AW, KP = K::KP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                   // [C][WSB] dgrad pack
    char* gl = smem + K::W_BYTES;                      // [AH*AW][PS] g_eff halo image (later: epilogue buffer / fold scratch)
    char* xl = gl + K::G_BYTES;                        // raw x tile: [8*32][PS] at conv resolution, or (RS == 1) the 4 x 16 SOURCE tile [64][PS]
    __shared__ float red[4 * C * 2];
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * C : 8];
    __shared__ __attribute__((aligned(32))) float eff_s[EFF ? 2 * C : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W;
    const int Hs = a.src.Hs, Ws = a.src.Ws;
    int bid = blockIdx.x;
    {   // XCD-aware order: blocks b and b + 8 share an XCD (and its L2): every XCD walks one contiguous run of tiles
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n = bid / bpi;                           // a block stays inside ONE image (per-image BatchNorm accumulators)
    const int t0 = (bid - n * bpi) * tpb;
    const int t1 = min(t0 + tpb, tpi);
    const bool want_cs = a.colsum != nullptr;

    // ---- block prologue: dgrad pack -> LDS, BatchNorm rows of this image, effgrad terms of this image's event
    for (int idx = tid; idx < C * (KP / 8); idx += 256) {
        const int row = idx / (KP / 8), kc = idx - row * (KP / 8);
        *(bf16x8*)(wl + row * K::WSB + kc * 16) = *(const bf16x8*)((const bf16*)a.w_bwd + (long)row * KP + kc * 8);
    }
    if (AFF) {
        for (int i = tid; i < C; i += 256) {
            aff_s[i] = a.src.scale[(long)n * a.src.aff_nstride + i];
            aff_s[C + i] = a.src.shift[(long)n * a.src.aff_nstride + i];
        }
    }
    if (EFF) {
        const int ev = (a.n_per_event > 0) ? n / a.n_per_event : 0;
        for (int i = tid; i < C; i += 256) {
            eff_s[i] = a.dstat[(long)ev * 2 * C + i];
            eff_s[C + i] = 2.f * a.dstat[(long)ev * 2 * C + C + i];
        }
    }
    typedef const __attribute__((address_space(3))) float* lds_cf;
    typedef const __attribute__((address_space(3))) f32x4* lds_cf4;

    auto tile_coords = [&](int t, int& h0, int& w0) {
        h0 = (t / tiles_w) * B3_TH;
        w0 = (t - (t / tiles_w) * tiles_w) * B3_TW;
    };

    // ---- raw operand requests of one tile: unconditional loads at clamped addresses (a per-lane condition in front of a load makes the
    // compiler drain the memory counter before the first use of ANY prefetched register); what lies outside the image / past the end
    // of the chunk list is zeroed or skipped in stage()
    constexpr int XCHN = (RS == 0) ? CH : 1;           // x chunks per thread: 256 px x CH chunks, or 64 source px x CH chunks
    bf16x8 rg[K::GCHN], ry[EFF ? K::GCHN : 1], rx[XCHN];
    auto request = [&](int t) {
        int h0, w0;
        tile_coords(t, h0, w0);
#pragma unroll
        for (int j = 0; j < K::GCHN; ++j) {
            const int idx = min(tid + j * 256, K::GTOT - 1);
            const int hp = idx / CH, cc = idx - hp * CH;
            const int hh = min(max(h0 - 1 + hp / AW, 0), H - 1), ww = min(max(w0 - 1 + hp % AW, 0), W - 1);
            const long m = ((long)n * H + hh) * W + ww;
            rg[j] = *(const bf16x8*)((const bf16*)a.g + m * a.Cg + cc * 8);
            if (EFF) ry[j] = *(const bf16x8*)((const bf16*)a.y + m * C + cc * 8);
        }
        if (RS == 0) {
#pragma unroll
            for (int j = 0; j < XCHN; ++j) {
                const int idx = tid + j * 256;
                const int px = idx / CH, cc = idx - px * CH;
                rx[j] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * H + h0 + (px >> 5)) * W + w0 + (px & 31)) * a.src.Cx + cc * 8);
            }
        } else {
            const int idx = min(tid, 64 * CH - 1);
            const int sp = idx / CH, cc = idx - sp * CH;
            rx[0] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * Hs + (h0 >> 1) + (sp >> 4)) * Ws + (w0 >> 1) + (sp & 15)) * a.src.Cx + cc * 8);
        }
    };
    auto stage = [&](int t, int zo) {
        int h0, w0;
        tile_coords(t, h0, w0);
#pragma unroll
        for (int j = 0; j < K::GCHN; ++j) {
            const int idx = tid + j * 256;
            if (idx >= K::GTOT) continue;
            const int hp = idx / CH, cc = idx - hp * CH;
            const int hh = h0 - 1 + hp / AW, ww = w0 - 1 + hp % AW;
            bf16x8 o = zero8();
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                if (EFF) {
                    const f32x4 d0 = *(lds_cf4)(eff_s + zo + cc * 8), d1 = *(lds_cf4)(eff_s + zo + cc * 8 + 4);
                    const f32x4 q0 = *(lds_cf4)(eff_s + zo + C + cc * 8), q1 = *(lds_cf4)(eff_s + zo + C + cc * 8 + 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        o[i] = f2bf(bf2f(rg[j][i]) + (i < 4 ? d0[i & 3] : d1[i & 3]) + bf2f(ry[j][i]) * (i < 4 ? q0[i & 3] : q1[i & 3]));
                } else {
                    o = rg[j];
                }
            }
            *(bf16x8*)(gl + hp * PS + cc * 16) = o;
        }
        if (RS == 0) {
#pragma unroll
            for (int j = 0; j < XCHN; ++j) {
                const int idx = tid + j * 256;
                const int px = idx / CH, cc = idx - px * CH;
                *(bf16x8*)(xl + px * PS + cc * 16) = rx[j];
            }
        } else if (tid < 64 * CH) {
            const int sp = tid / CH, cc = tid - sp * CH;
            *(bf16x8*)(xl + sp * PS + cc * 16) = rx[0];
        }
    };

    // ---- persistent state
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    constexpr int NTAP = (C == 16) ? 3 : 9;            // dW accumulator tiles per wave
    f32x4 accw[NTAP], accc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < NTAP; ++tp) accw[tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
    const int mtw = (C == 32) ? (wave >> 1) : 0, njw = (C == 32) ? (wave & 1) : 0;      // this wave's (cout tile, cin tile) of dW
    int pbase[4];      // byte offset of this lane's halo pixel in m-tile mt at tap (0,0): tile row 2*wave + (mt >> 1), column (mt & 1)*16 + lr
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) pbase[mt] = ((2 * wave + (mt >> 1)) * AW + (mt & 1) * 16 + lr) * PS;
    float* wbuf = (float*)gl + wave * 32 * K::LDW;     // this wave's transpose buffer (valid between the two barriers behind the MFMAs)
    const int ecc = lane % CPP;

    if (t0 >= t1) return;                              // (the launcher never makes empty blocks; all waves leave together)
    request(t0);
    __syncthreads();                                   // weights / tables staged
    for (int t = t0; t < t1; ++t) {
        int h0, w0;
        tile_coords(t, h0, w0);
        // an offset of zero the compiler cannot see through: the per-lane rows of the BatchNorm / effgrad tables are loop-invariant, and
        // hoisted out of the tile loop they pin ~34 VGPRs for the whole block (spills); re-reading 16-byte LDS rows per tile is free
        int zo = 0;
        asm volatile("" : "+v"(zo));
        stage(t, zo);
        __syncthreads();
        request(min(t + 1, t1 - 1));                   // in flight during the MFMAs and the epilogue (last tile: re-requests itself, unused)

        // ---- dgrad: 4 m-tiles (two tile rows) x NT n-tiles per wave
        f32x4 acc[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks) {
            const int k = ks * 32 + lg * 8;                // lane-group dependent for C = 16 (two taps per K step)
            int tap = k / C;
            const int c = k - tap * C;
            const bool kval = tap < 9;                     // K padding (C = 16: 144 -> 160)
            if (!kval) tap = 0;
            const int toff = ((tap / 3) * AW + (tap - (tap / 3) * 3)) * PS + c * 2;
            bf16x8 bq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[nt] = *(const bf16x8*)(wl + (nt * 16 + lr) * K::WSB + k * 2);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                bf16x8 af = *(const bf16x8*)(gl + pbase[mt] + toff);
                if (!kval) af = zero8();
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bq[nt], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);             // one K step's fragments live at a time (the unrolled loop otherwise hoists every read)
        }
        // ---- wgrad (+ bias column sums): K step = one tile row of 32 pixels.  C = 32: wave = (cout tile, cin tile), all nine taps;
        // C = 16: wave w owns taps w, w + 4 (, w + 8) over the whole tile (three accumulator tiles instead of nine per wave)
        {
#pragma unroll
            for (int r = 0; r < B3_TH; ++r) {
                // (up-sampled source: conv pixel (r, col) reads source pixel (r >> 1, col >> 1) of the compact tile -- every lane of a transposed
                //  read supplies its own row address, pairs of lanes share one)
                bf16x8 b = (RS == 0) ? b3_frag_T(xl, PS, r * B3_TW + lg * 4, njw * 16, lr) : b3_frag_T_up(xl, PS, r >> 1, lg * 4, njw * 16, lr);
                if (AFF) {                                  // forward prologue on the fragment: 8 pixels of channel njw*16 + lr
                    const float sc = *(lds_cf)(aff_s + zo + njw * 16 + lr), sh = *(lds_cf)(aff_s + zo + C + njw * 16 + lr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] = f2bf(fmaxf(bf2f(b[i]) * sc + sh, 0.f));
                } else {
                    b = relu8(b);
                }
#pragma unroll
                for (int j = 0; j < NTAP; ++j) {
                    const int tp = (C == 16) ? min(wave + 4 * j, 8) : j;      // (C = 16, waves 1..3: slot 2 repeats tap 8 and is dropped at the end)
                    const int dy = tp / 3, dx = tp - dy * 3;
                    const bf16x8 af = b3_frag_T(gl, PS, (r + 2 - dy) * AW + (2 - dx) + lg * 4, mtw * 16, lr);
                    accw[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b, accw[j], 0, 0, 0);
                    if (tp == 4 && want_cs && njw == 0 && (C == 32 || j == 1)) accc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, accc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                   // halo consumed by every wave: its LDS becomes the transpose buffers

        // ---- epilogue
        if (RS == 0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) wbuf[(m2 * 16 + lg * 4 + r) * K::LDW + nt * 16 + lr] = acc[2 * half + m2][nt][r];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                constexpr int EIT = 32 * CPP / 64;
#pragma unroll
                for (int it = 0; it < EIT; ++it) {
                    const int row = (it * 64 + lane) / CPP;
                    const f32x4 lo = *(const f32x4*)(wbuf + row * K::LDW + ecc * 8);
                    const f32x4 hi = *(const f32x4*)(wbuf + row * K::LDW + ecc * 8 + 4);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const int px = (2 * wave + half) * B3_TW + row;
                    const bf16x8 xv = *(const bf16x8*)(xl + px * PS + ecc * 16);
                    if (AFF) {
                        const f32x4 c0 = *(lds_cf4)(aff_s + zo + ecc * 8), c1 = *(lds_cf4)(aff_s + zo + ecc * 8 + 4);
                        const f32x4 e0 = *(lds_cf4)(aff_s + zo + C + ecc * 8), e1 = *(lds_cf4)(aff_s + zo + C + ecc * 8 + 4);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float xf = bf2f(xv[i]);
                            const float sc = i < 4 ? c0[i & 3] : c1[i & 3], sh = i < 4 ? e0[i & 3] : e1[i & 3];
                            const float d = !(xf * sc + sh > 0.f) ? 0.f : v[i];
                            s1[i] += d;
                            s2[i] += d * xf;
                            v[i] = d * sc;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = (bf2f(xv[i]) > 0.f) ? v[i] : 0.f;
                    }
                    bf16x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                    *(bf16x8*)((bf16*)a.dx + (((long)n * H + h0 + 2 * wave + half) * W + w0 + row) * C + ecc * 8) = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // up-sampled source: tile rows 2w, 2w+1 are source row w of the 4 x 16 source tile; item = (source pixel j, chunk) per lane
            constexpr int ITEMS = 16 * CPP;            // 32 (C = 16) / 64 (C = 32) per wave
            const bool act = lane < ITEMS;
            const int j = lane / CPP;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) wbuf[(m2 * 16 + lg * 4 + r) * K::LDW + nt * 16 + lr] = acc[2 * half + m2][nt][r];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (act) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 lo = *(const f32x4*)(wbuf + (2 * j + q) * K::LDW + ecc * 8);
                        const f32x4 hi = *(const f32x4*)(wbuf + (2 * j + q) * K::LDW + ecc * 8 + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] += lo[i];
                            v[4 + i] += hi[i];
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (act) {
                const bf16x8 xv = *(const bf16x8*)(xl + (wave * 16 + j) * PS + ecc * 16);
                if (AFF) {
                    const f32x4 c0 = *(lds_cf4)(aff_s + zo + ecc * 8), c1 = *(lds_cf4)(aff_s + zo + ecc * 8 + 4);
                    const f32x4 e0 = *(lds_cf4)(aff_s + zo + C + ecc * 8), e1 = *(lds_cf4)(aff_s + zo + C + ecc * 8 + 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float xf = bf2f(xv[i]);
                        const float sc = i < 4 ? c0[i & 3] : c1[i & 3], sh = i < 4 ? e0[i & 3] : e1[i & 3];
                        const float d = !(xf * sc + sh > 0.f) ? 0.f : v[i];
                        s1[i] += d;
                        s2[i] += d * xf;
                        v[i] = d * sc;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (bf2f(xv[i]) > 0.f) ? v[i] : 0.f;
                }
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                *(bf16x8*)((bf16*)a.dx + (((long)n * Hs + (h0 >> 1) + wave) * Ws + (w0 >> 1) + j) * C + ecc * 8) = o;
            }
        }
        __syncthreads();                                   // transpose buffers / x tile free: the next tile is staged over them
    }

    // ---- block end 1: per-image BatchNorm accumulators (sum d, sum d*x): the lanes that share a chunk are folded through LDS, the four
    // waves in a fixed order, one add per channel and block into replica slot bid % BNB_REPL of the image
    if (AFF && a.bn_acc != nullptr) {
        float* sx = (float*)gl + wave * STATS_SX_FLOATS;
        constexpr int SH = 64 / CPP;
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
                red[(wave * C + cc * 8 + i) * 2 + w] = tt;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        if (tid < C) {
            float x1 = 0.f, x2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                x1 += red[(wv * C + tid) * 2 + 0];
                x2 += red[(wv * C + tid) * 2 + 1];
            }
            // slot = this block's index inside its image when bn_slots == blocks per image (one adder per address: bit-reproducible)
            const int R = a.bn_slots > 0 ? a.bn_slots : BNB_REPL;
            float* st = a.bn_acc + ((long)n * R + bid % R) * 2 * C;
            atomicAdd(st + tid, x1);
            atomicAdd(st + C + tid, x2);
        }
        __syncthreads();
    }
    // ---- block end 2: dW slab [C][KP] (+ column sums): every wave owns disjoint entries (C = 32: its (cout tile, cin tile) pair; C = 16: its
    // taps), so the slab is bit-reproducible (no LDS float atomics)
    {
        float* T = (float*)gl;
        for (int i = tid; i < K::T_FLOATS; i += 256) T[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            const int tp = (C == 16) ? wave + 4 * j : j;
            if (tp < 9) {
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(mtw * 16 + lg * 4 + r) * KP + tp * C + njw * 16 + lr] = accw[j][r];
            }
        }
        // column sums ride on tap 4 (the centre): C = 16 -> wave 0, slot 1; C = 32 -> the waves with cin tile 0
        if (want_cs && njw == 0 && lr == 0 && (C == 32 || wave == 0)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) T[C * KP + mtw * 16 + lg * 4 + r] = accc[r];
        }
        __syncthreads();
        float* slab = a.partials + (long)blockIdx.x * C * KP;
        for (int i = tid; i < C * KP / 4; i += 256) *(f32x4*)(slab + i * 4) = *(const f32x4*)(T + i * 4);
        if (want_cs && tid < C) atomicAdd(a.colsum + (long)(blockIdx.x % STAT_REPL) * C + tid, T[C * KP + tid]);
    }
}

// ------------------------------------------------------------------------------------------------
struct B3Plan {
    int tiles_w, tpi, tpb, bpi, nblk;
    size_t lds;
    long ws_elems;
};

static bool b3_shape_ok(int C, int rs, bool aff, bool relu) { return (C == 16 || C == 32) && (rs == 0 || rs == 1) && relu; }
// C = 32 with BatchNorm prologue + effgrad at the same resolution: 16 prefetched chunks per thread next to 76 accumulator registers spill at
// the 256-register cap of two blocks per CU and the launch then loses to the separate ones (322 vs 297 us at 128x384, N = 40): not offered
// Only the combinations the networks have are instantiated: the discriminator's layers (bare ReLU, no BatchNorm behind them) and the
// generator's (BatchNorm prologue + effgrad).
static bool b3_variant_ok(int C, int rs, bool aff, bool eff) { return aff == eff && !(C == 32 && rs == 0 && aff && eff); }

static int b3_plan(const Bwd3Args& a, B3Plan& p) {
    CHECK_ARG(b3_shape_ok(a.C, a.src.rs, a.src.scale != nullptr, a.src.relu != 0), "conv3x3_bwd: C = %d, rs %d, relu %d is not instantiated", a.C, a.src.rs, a.src.relu);
    CHECK_ARG(b3_variant_ok(a.C, a.src.rs, a.src.scale != nullptr, a.y != nullptr), "conv3x3_bwd: this (C, rs, affine prologue, effgrad) combination is not instantiated");
    CHECK_ARG(a.H % B3_TH == 0 && a.W % B3_TW == 0, "conv3x3_bwd: H %% 8 == 0 and W %% 32 == 0 required (%d x %d)", a.H, a.W);
    if (a.src.rs == 0) CHECK_ARG(a.H == a.src.Hs && a.W == a.src.Ws, "conv3x3_bwd: geometry mismatch");
    if (a.src.rs == 1) CHECK_ARG(a.H == 2 * a.src.Hs && a.W == 2 * a.src.Ws, "conv3x3_bwd: upsample geometry mismatch");
    CHECK_ARG(a.Kpad == ((9 * a.C + 31) / 32) * 32, "conv3x3_bwd: Kpad must be the canonical pack row length");
    CHECK_ARG(a.Cg >= a.C && a.Cg % 8 == 0 && a.src.Cx >= a.C && a.src.Cx % 8 == 0, "conv3x3_bwd: bad channel strides");
    CHECK_ARG((a.src.scale == nullptr) == (a.src.shift == nullptr), "conv3x3_bwd: scale / shift must come together");
    CHECK_ARG((a.y == nullptr) == (a.dstat == nullptr), "conv3x3_bwd: the effgrad operands y / dstat must come together");
    CHECK_ARG(a.n_per_event >= 0 && (a.n_per_event == 0 || a.N % a.n_per_event == 0), "conv3x3_bwd: N is not a whole number of events");
    CHECK_ARG((a.src.scale != nullptr) == (a.bn_acc != nullptr), "conv3x3_bwd: the affine prologue and bn_acc come together");
    CHECK_ARG(a.dx != nullptr && a.dw != nullptr && a.partials != nullptr, "conv3x3_bwd: dx, dw and the slab workspace are required");
    CHECK_ARG((a.flags & ~IEAGAN_BWD_NO_REDUCE) == 0 && a.bn_slots >= 0, "conv3x3_bwd: unknown flag bits 0x%x / bad bn_slots %d", a.flags, a.bn_slots);
    p.tiles_w = a.W / B3_TW;
    p.tpi = p.tiles_w * (a.H / B3_TH);
    // ONE round of persistent blocks: whole blocks per image
    const int occ = a.C == 16 ? ((a.src.scale != nullptr && a.y != nullptr) ? 2 : 3) : 2;      // == b3_occ of the variant
    const int slots = 256 * occ;
    int bpi = slots / a.N;
    if (bpi < 1) bpi = 1;
    int tpb = (p.tpi + bpi - 1) / bpi;
    if (tpb < 2 && p.tpi >= 2) tpb = 2;
    bpi = (p.tpi + tpb - 1) / tpb;
    p.tpb = tpb;
    p.bpi = bpi;
    p.nblk = bpi * a.N;
    const int PS = a.C == 16 ? 32 : 96;
    const int KP = a.Kpad;
    p.lds = (size_t)a.C * (KP * 2 + 16) + (size_t)(B3_TH + 2) * (B3_TW + 2) * PS + (a.src.rs == 1 ? (size_t)64 * PS : (size_t)B3_TH * B3_TW * PS);
    p.ws_elems = (long)p.nblk * a.C * KP;
    return 0;
}

extern "C" long ieagan_conv3x3_bwd_workspace(const ieagan_conv3x3_bwd_desc* d) {
    B3Plan p;
    if (d == nullptr) return 0;
    Bwd3Args a = *d;
    if (a.partials == nullptr) a.partials = (float*)16;      // (the query is made before the workspace exists)
    if (b3_plan(a, p) != 0) return 0;
    return p.ws_elems;
}

extern "C" int ieagan_conv3x3_bwd_slots(const ieagan_conv3x3_bwd_desc* d) {
    B3Plan p;
    if (d == nullptr) return IEAGAN_EINVAL;
    Bwd3Args a = *d;
    if (a.partials == nullptr) a.partials = (float*)16;      // (the query is made before the workspace exists)
    const int rc = b3_plan(a, p);
    return rc != 0 ? rc : p.bpi;
}

extern "C" int ieagan_conv3x3_bwd_supported(int C, int rs, int affine, int relu, int eff, int H, int W) {
    return (b3_shape_ok(C, rs, affine != 0, relu != 0) && b3_variant_ok(C, rs, affine != 0, eff != 0) && H % B3_TH == 0 && W % B3_TW == 0) ? 1 : 0;
}

int wgrad_reduce_launch(const float* part, float* dw, int S, int Cout, int Kpad, int K, hipStream_t st);      // conv_wgrad.hip

template <int C, bool AFF, int RS, bool EFF>
static int b3_go(const Bwd3Args& a, const B3Plan& p, hipStream_t st) {
    auto kern = conv3x3_bwd_kernel<C, AFF, RS, EFF>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
            ieagan_set_error("conv3x3_bwd: cannot reserve dynamic LDS");
            return IEAGAN_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.nblk), dim3(256), p.lds, st, a, p.tiles_w, p.tpi, p.tpb, p.bpi, p.nblk);
    return 0;
}

template <int C, int RS>
static int b3_dispatch(const Bwd3Args& a, const B3Plan& p, hipStream_t st) {
    const bool aff = a.src.scale != nullptr;            // == (a.y != nullptr): b3_variant_ok
    if constexpr (C == 32 && RS == 0) {
        return b3_go<C, false, RS, false>(a, p, st);    // (the BatchNorm + effgrad form of this shape is not offered)
    } else {
        if (aff) return b3_go<C, true, RS, true>(a, p, st);
        return b3_go<C, false, RS, false>(a, p, st);
    }
}

extern "C" int ieagan_conv3x3_bwd(const ieagan_conv3x3_bwd_desc* d, void* stream) {
    CHECK_ARG(d != nullptr && d->g != nullptr && d->src.x != nullptr && d->w_bwd != nullptr, "conv3x3_bwd: null pointer");
    const Bwd3Args& a = *d;
    B3Plan p;
    const int prc = b3_plan(a, p);
    if (prc != 0) return prc;
    hipStream_t st = (hipStream_t)stream;
    const double P = (double)a.N * a.H * a.W, Ps = (double)a.N * a.src.Hs * a.src.Ws;
    const double flops = 2.0 * P * 9.0 * a.C * a.C * 2.0;
    // algorithmic bytes (layer-granular): dgrad R g + W da at the layer's own resolution, wgrad R x + R g -- what the replaced launches
    // were charged (tools/arch_calc.py "as launched")
    const double bytes_min = 2.0 * (P * a.C + P * a.C + Ps * a.C + P * a.C);
    double bytes = 2.0 * (P * a.C + Ps * a.C + Ps * a.C);          // what this launch moves: g, x, dx (+ y)
    if (a.y) bytes += 2.0 * P * a.C;
    char tag[64] = "";
    if (prof_tags_on())
        snprintf(tag, sizeof(tag), "c%d %dx%d rs%d a%d eff%d", a.C, a.H, a.W, a.src.rs, a.src.scale != nullptr, a.y != nullptr);
    ProfScope prof("conv3x3_bwd", flops, bytes, st, tag, bytes_min);
    int rc;
    if (a.C == 16) rc = (a.src.rs == 0) ? b3_dispatch<16, 0>(a, p, st) : b3_dispatch<16, 1>(a, p, st);
    else rc = (a.src.rs == 0) ? b3_dispatch<32, 0>(a, p, st) : b3_dispatch<32, 1>(a, p, st);
    if (rc != 0) return rc;
    CHECK_LAUNCH("conv3x3_bwd");
    if (a.flags & IEAGAN_BWD_NO_REDUCE) return 0;
    return wgrad_reduce_launch(a.partials, a.dw, p.nblk, a.C, a.Kpad, 9 * a.C, st);
}
