// 3x3 convolutions with a single-channel image on one side:
//   conv_1toC   image[N,H,W] fp32 -> NHWC bf16 [N,H,W,C]    D.input_conv forward (model.py:905) and the
//                                                            dgrad of G's output conv (with tanh'); K = 9 taps on MFMA
//                                                            with the image split into bf16 high + low parts
//   conv_Cto1   NHWC bf16 [N,H,W,C] -> image fp32            G.output_layer: BN apply + ReLU + conv + tanh (+ the
//                                                            detector-unit export) (model.py:379-387,487,1139-1147) and
//                                                            dgrad of D.input_conv; per-tap dot products on MFMA
//   wgrad_c1    dW[tap][c] = sum_p img[p +/- d(tap)] * t[p,c]   taps x channels over K = pixels on MFMA
#include "common.h"

// conv_1toC on the matrix cores.  out[p][c] = bias[c] + sum_tap img(p + d(tap)) * w[tap][c] is a GEMM with K = 9; one
// 16x16x32 MFMA step carries it twice: k slots 0..8 hold the bf16 HIGH parts of the nine image values, k slots 16..24 their
// bf16 LOW parts (img - hi), against the same weights in both halves -- the fp32 image is thus represented to ~16 mantissa
// bits while the weights are bf16 operands as in every other conv.  Block = 8 x 32 output pixels, image halo in LDS,
// accumulators transposed through LDS so that every lane stores 8 channels (16 bytes) of one pixel.
// w: fp32 [9][C].  tanh_y (optional): the image is multiplied by (1 - y*y) first (tanh backward).
#define I1_TH 8
#define I1_TW 32
// BNB (the dgrad of G.output_layer): the BatchNorm-apply + ReLU backward of the layer's prologue is done in the store phase -- the
// value stored is d * scale[c] with d = ReLU'(x * scale + shift) * conv, and sum d / sum d * x go to dshift / dscale (row n * nstride) --
// so the [N, H, W, C] gradient w.r.t. the activated tensor (0.5 GB at 256x768) is never written or read back.
struct C1Bnb {
    const bf16* x;            // the BatchNorm input, bf16 [N, H, W, C]
    const float* scale;       // rows n * nstride
    const float* shift;
    int nstride, relu;
    float* dscale;            // accumulated (atomics), rows n * nstride
    float* dshift;
    // slots > 0: dscale = acc [rows][slots][2][C] ({sum d, sum d x}; rows = N, or 1 when nstride == 0), caller-zeroed; a block adds into slot
    // (its index - first block of the image) of the image's row -- one adder per address, bit-reproducible sums; dshift unused
    int slots, tiles_img;
};
template <int C, bool BNB = false>
__global__ __launch_bounds__(256) void conv_1toC_kernel(const float* __restrict__ img, const float* __restrict__ tanh_y,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        bf16* __restrict__ out, int N, int H, int W, int flip, int tiles_w,
                                                        int tiles_h, C1Bnb bnb = C1Bnb{}) {
    constexpr int NT = C / 16;
    constexpr int AW = I1_TW + 2, AH = I1_TH + 2;
    constexpr int LDO = C + 4;                                   // padded fp32 row of the transposed output tile
    __shared__ float halo[AH][AW + 2];
    __shared__ __attribute__((aligned(16))) float ot[4][16 * LDO];   // per wave: one m-tile (16 pixels) x C channels
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    // B fragments: B[k][cout lr]; k slots 8*lg + j: lg 0 -> taps 0..7, lg 1 -> tap 8 (j = 0), lg 2 / 3 the same again (low parts)
    bf16x8 bfrag[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = (lg & 1) * 8 + j;
            float v = 0.f;
            if (tap < 9) v = w[(flip ? 8 - tap : tap) * C + nt * 16 + lr];
            bfrag[nt][j] = f2bf(v);
        }
    float bv[8];
    const int cc = lane % (C / 8);                               // this lane's 8-channel chunk in the store phase
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = bias ? bias[cc * 8 + i] : 0.f;
    // persistent blocks: the weight fragments and the bias are fetched once, a block walks consecutive tiles
    const int ntiles = N * tiles_w * tiles_h;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tile_end = min((int)(blockIdx.x + 1) * per, ntiles);
    constexpr int SIT = (16 * (C / 8) + 63) / 64;                // store-phase items per lane and m-tile
    __shared__ float bred[BNB ? 256 * 8 : 8];
    float bsc[8], bsh[8], p_ds[8], p_dt[8];
    int bn_n = -1;
#pragma unroll
    for (int i = 0; i < 8; ++i) bsc[i] = bsh[i] = p_ds[i] = p_dt[i] = 0.f;
    // sum over the lanes of the block that own channel chunk cc, one atomic per channel (cf. bn_elem.hip reduce_groups_atomic)
    auto bnb_flush = [&](int n_row) {
        constexpr int G = C / 8;
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) bred[threadIdx.x * 8 + i] = w2 ? p_dt[i] : p_ds[i];
            __syncthreads();
            for (int o2 = threadIdx.x; o2 < C; o2 += 256) {
                const int g = o2 >> 3, i = o2 & 7;
                float sacc = 0.f;
                for (int u = g; u < 256; u += G) sacc += bred[u * 8 + i];
                if (bnb.slots > 0) {
                    const int per_ = (gridDim.x + N * bnb.tiles_img - 1) / gridDim.x;               // tiles per block (== `per` below)
                    const long row = bnb.nstride != 0 ? n_row : 0;
                    const int slot = bnb.nstride != 0 ? (int)blockIdx.x - (int)(((long)n_row * bnb.tiles_img) / per_) : (int)blockIdx.x;
                    atomicAdd(bnb.dscale + ((row * bnb.slots + slot) * 2 + (w2 ? 0 : 1)) * C + o2, sacc);
                } else {
                    atomicAdd((w2 ? bnb.dshift : bnb.dscale) + (long)n_row * bnb.nstride + o2, sacc);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) p_ds[i] = p_dt[i] = 0.f;
    };
    for (int tile = blockIdx.x * per; tile < tile_end; ++tile) {
    const int n = tile / (tiles_w * tiles_h);
    if (BNB && n != bn_n) {                                      // block-uniform: this image's scale / shift row; flush the sums of the last one
        if (bn_n >= 0 && bnb.nstride != 0) bnb_flush(bn_n);
        bn_n = n;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bsc[i] = bnb.scale[(long)n * bnb.nstride + cc * 8 + i];
            bsh[i] = bnb.shift[(long)n * bnb.nstride + cc * 8 + i];
        }
    }
    const int trem = tile - n * tiles_w * tiles_h;
    const int y0 = (trem / tiles_w) * I1_TH, x0 = (trem % tiles_w) * I1_TW;
    const float* im = img + (size_t)n * H * W;
    const float* ty_ = tanh_y ? tanh_y + (size_t)n * H * W : nullptr;
    __syncthreads();                                             // the previous tile's halo has been consumed
    for (int i = threadIdx.x; i < AH * AW; i += 256) {
        const int qy = i / AW, qx = i - qy * AW;
        const int yy = y0 - 1 + qy, xx = x0 - 1 + qx;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            v = im[yy * W + xx];
            if (ty_) {
                const float t = ty_[yy * W + xx];
                v *= 1.f - t * t;
            }
        }
        halo[qy][qx] = v;
    }
    __syncthreads();
    bf16x8 xk[BNB ? 4 : 1][BNB ? SIT : 1];                       // BNB: this lane's BatchNorm-input chunks of the four m-tiles, requested now
    if (BNB) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                const int item = min(it * 64 + lane, 16 * (C / 8) - 1);
                const int y = min(y0 + 2 * wave + (mi >> 1), H - 1), x = min(x0 + (mi & 1) * 16 + item / (C / 8), W - 1);
                xk[mi][it] = *(const bf16x8*)(bnb.x + (((size_t)n * H + y) * W + x) * C + cc * 8);
            }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {                             // wave: tile rows 2*wave, 2*wave+1 = 4 m-tiles of 16 pixels
        const int ty = 2 * wave + (mi >> 1), tx0 = (mi & 1) * 16;
        // A fragment: row = pixel lr of the m-tile; k slots of this lane group: image values at taps 8*(lg&1) + j
        bf16x8 af;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = (lg & 1) * 8 + j;
            float v = 0.f;
            if (tap < 9) v = halo[ty + tap / 3][tx0 + lr + tap % 3];
            const bf16 hi = f2bf(v);
            af[j] = (lg < 2) ? hi : f2bf(v - bf2f(hi));
        }
        float* o = ot[wave];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[nt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * lg + r) * LDO + nt * 16 + lr] = d[r];      // D[pixel 4lg+r][cout lr]
        }
        // wave-private buffer: LDS operations of one wave complete in order, the fences keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < (16 * (C / 8) + 63) / 64; ++it) {
            const int item = it * 64 + lane;
            if (item >= 16 * (C / 8)) break;
            const int px = item / (C / 8);
            const f32x4 lo = *(const f32x4*)(o + px * LDO + cc * 8);
            const f32x4 hi = *(const f32x4*)(o + px * LDO + cc * 8 + 4);
            const int y = y0 + ty, x = x0 + tx0 + px;
            if (y < H && x < W) {
                float v[8] = {lo[0] + bv[0], lo[1] + bv[1], lo[2] + bv[2], lo[3] + bv[3], hi[0] + bv[4], hi[1] + bv[5], hi[2] + bv[6], hi[3] + bv[7]};
                if (BNB) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float xf = bf2f(xk[BNB ? mi : 0][BNB ? it : 0][i]);
                        const float d = (bnb.relu && !(xf * bsc[i] + bsh[i] > 0.f)) ? 0.f : v[i];
                        p_dt[i] += d;
                        p_ds[i] += d * xf;
                        v[i] = d * bsc[i];
                    }
                }
                bf16x8 ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = f2bf(v[i]);
                *(bf16x8*)(out + (((size_t)n * H + y) * W + x) * C + cc * 8) = ov;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the next m-tile overwrites the buffer
        __builtin_amdgcn_wave_barrier();
    }
    }
    if (BNB && bn_n >= 0) bnb_flush(bn_n);
}

// conv_1toC with the BatchNorm-apply + ReLU backward folded into its store phase (see C1Bnb): the dgrad of G.output_layer
// (model.py:379-387).  dx [N,H,W,C] bf16; dscale / dshift fp32, caller-zeroed, rows n * nstride (nstride 0: one row for the batch).
// grid of the launch: blocks, tiles per image, slots per accumulator row (see C1Bnb)
static void c1_bnb_plan(int N, int H, int W, int nstride, long& blocks, int& tiles_img, int& slots) {
    const int tiles_w = (W + I1_TW - 1) / I1_TW, tiles_h = (H + I1_TH - 1) / I1_TH;
    tiles_img = tiles_w * tiles_h;
    const long ntl = (long)N * tiles_img;
    blocks = ntl < 2048 ? ntl : 2048;
    if (blocks < 1) blocks = 1;
    const long per = (ntl + blocks - 1) / blocks;
    slots = nstride != 0 ? (int)((tiles_img + per - 1) / per) + 1 : (int)blocks;
}

extern "C" int ieagan_conv_1toC_bnb_slots(int N, int H, int W, int nstride) {
    long blocks;
    int tiles_img, slots;
    c1_bnb_plan(N, H, W, nstride, blocks, tiles_img, slots);
    return slots;
}

extern "C" int ieagan_conv_1toC_bnb(const float* img, const float* tanh_y, const float* w, const void* x, const float* scale,
                                    const float* shift, int nstride, int relu, void* dx, float* dscale, float* dshift, int N, int H,
                                    int W, int C, int flip, int slots, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(x && scale && shift && dx && dscale && (dshift || slots > 0), "conv_1toC_bnb: null operand");
    {
        long pb;
        int pt, ps;
        c1_bnb_plan(N, H, W, nstride, pb, pt, ps);
        CHECK_ARG(slots == 0 || slots >= ps, "conv_1toC_bnb: %d slots, the launch needs %d (ieagan_conv_1toC_bnb_slots)", slots, ps);
    }
    ProfScope prof("conv_1toC", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 4.0 * C), st);
    const int tiles_w = (W + I1_TW - 1) / I1_TW, tiles_h = (H + I1_TH - 1) / I1_TH;
    const long ntl = (long)N * tiles_w * tiles_h;
    CHECK_ARG(ntl > 0 && ntl < (1L << 31), "conv_1toC_bnb: bad geometry");
    const long blocks = ntl < 2048 ? ntl : 2048;
    C1Bnb b{(const bf16*)x, scale, shift, nstride, relu, dscale, dshift, slots, tiles_w * tiles_h};
#define L(CC) hipLaunchKernelGGL((conv_1toC_kernel<CC, true>), dim3((unsigned)blocks), dim3(256), 0, st, img, tanh_y, w, (const float*)nullptr, (bf16*)dx, N, H, W, flip, tiles_w, tiles_h, b)
    if (C == 16) L(16);
    else if (C == 32) L(32);
    else if (C == 64) L(64);
    else { ieagan_set_error("conv_1toC_bnb: C=%d not instantiated (16/32/64)", C); return IEAGAN_EINVAL; }
#undef L
    CHECK_LAUNCH("conv_1toC_bnb");
    return 0;
}

extern "C" int ieagan_conv_1toC(const float* img, const float* tanh_y, const float* w, const float* bias, void* out, int N,
                                int H, int W, int C, int flip, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("conv_1toC", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 2.0 * C), st);
    const int tiles_w = (W + I1_TW - 1) / I1_TW, tiles_h = (H + I1_TH - 1) / I1_TH;
    const long ntl = (long)N * tiles_w * tiles_h;
    CHECK_ARG(ntl > 0 && ntl < (1L << 31), "conv_1toC: bad geometry");
    const long blocks = ntl < 4096 ? ntl : 4096;                 // persistent: ~16 blocks per CU, several tiles each
#define L(CC) hipLaunchKernelGGL((conv_1toC_kernel<CC, false>), dim3((unsigned)blocks), dim3(256), 0, st, img, tanh_y, w, bias, (bf16*)out, N, H, W, flip, tiles_w, tiles_h, C1Bnb{})
    if (C == 16) L(16);
    else if (C == 32) L(32);
    else if (C == 64) L(64);
    else { ieagan_set_error("conv_1toC: C=%d not instantiated (16/32/64)", C); return IEAGAN_EINVAL; }
#undef L
    CHECK_LAUNCH("conv_1toC");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// conv_Cto1 on the matrix cores.  For every pixel q of the (8+2) x (64+2) halo of an 8 x 64 output tile one MFMA row
// gives the nine per-tap dot products  s[tap][q] = sum_c T(x[q, c]) * w[tap][c]  (A = 16 pixels x C channels with the
// BatchNorm apply + ReLU done once per element, B = C x 16 "taps", 9 used), written to LDS; an output pixel is then
// bias + sum_tap s[tap][p + d(tap)], nine conflict-free LDS reads.  The VALU form of this kernel transformed every
// input element nine times and ran at 1.0-1.3 TB/s.
#define C1_TH 8
#define C1_TW 64
template <int C, bool AFF, bool RELU>
__global__ __launch_bounds__(256) void conv_Cto1_kernel(const bf16* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int nstride, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int tanh_out,
                                                        int N, int H, int W, int flip, int tiles_w, int tiles_h) {
    constexpr int AW = C1_TW + 2, AH = C1_TH + 2, NPIX = AH * AW;      // 660 halo pixels
    constexpr int MTILES = (NPIX + 15) / 16;                            // 42
    constexpr int LDQ = MTILES * 16;                                    // 672
    constexpr int KS = (C + 31) / 32;
    __shared__ __attribute__((aligned(16))) float s[9][LDQ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x;
    const int n = tile / (tiles_w * tiles_h);
    const int trem = tile - n * tiles_w * tiles_h;
    const int y0 = (trem / tiles_w) * C1_TH, x0 = (trem % tiles_w) * C1_TW;

    // B fragments: B[k = channel][col = tap]; lane (lg, lr) holds channels ks*32 + lg*8 .. +7 of tap lr
    bf16x8 bfrag[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c0 = ks * 32 + lg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (lr < 9 && c0 + j < C) v = w[(flip ? 8 - lr : lr) * C + c0 + j];
            bfrag[ks][j] = f2bf(v);
        }
    }
    float sc[KS][8], sh[KS][8];
    if (AFF) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = ks * 32 + lg * 8 + j;
                sc[ks][j] = (c < C) ? scale[(long)n * nstride + c] : 0.f;
                sh[ks][j] = (c < C) ? shift[(long)n * nstride + c] : 0.f;
            }
    }
    // all of a wave's halo fragments (11 m-tiles) are requested before the first one is used: unconditional loads (coordinates and
    // channel group clamped, zeroed below) -- one tile per block, so nothing else overlaps a wave's load latencies
    constexpr int MPW = (MTILES + 3) / 4;
    bf16x8 raw[MPW][KS];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int q = min((wave + 4 * i) * 16 + lr, NPIX - 1);
        const int qy = q / AW, qx = q - qy * AW;
        const int yc = min(max(y0 - 1 + qy, 0), H - 1), xc = min(max(x0 - 1 + qx, 0), W - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) raw[i][ks] = *(const bf16x8*)(x + (((long)n * H + yc) * W + xc) * C + min(ks * 32 + lg * 8, C - 8));
    }
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int mt = wave + 4 * i;
        if (mt >= MTILES) break;
        const int q = mt * 16 + lr;
        const int qy = q / AW, qx = q - qy * AW;
        const int yy = y0 - 1 + qy, xx = x0 - 1 + qx;
        const bool ok = q < NPIX && yy >= 0 && yy < H && xx >= 0 && xx < W;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = ks * 32 + lg * 8;
            bf16x8 af = zero8();
            if (ok && c0 < C) {
                af = raw[i][ks];
                if (AFF || RELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float f = bf2f(af[j]);
                        if (AFF) f = fmaf(f, sc[ks][j], sh[ks][j]);
                        if (RELU) f = fmaxf(f, 0.f);
                        af[j] = f2bf(f);
                    }
                }
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[ks], acc, 0, 0, 0);
        }
        if (lr < 9) *(f32x4*)&s[lr][mt * 16 + lg * 4] = acc;      // D[row = pixel 4*lg + r][col = tap lr]
    }
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
    for (int i = 0; i < (C1_TH * C1_TW) / 256; ++i) {
        const int p = threadIdx.x + i * 256;
        const int ty = p / C1_TW, tx = p - ty * C1_TW;
        const int y = y0 + ty, xw = x0 + tx;
        if (y >= H || xw >= W) continue;
        float r = b0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) r += s[tap][(ty + tap / 3) * AW + tx + tap % 3];
        if (tanh_out) r = tanhf(r);
        if (tanh_out == 2) {        // detector-unit export (model.py:1140-1148): threshold, 256^((r+1)/2) - 1, clamp, crop 3 rows
            if (y >= 3 && y < H - 3) {
                r = (r > -0.26f) ? r : -1.f;
                const float adu = fminf(fmaxf(exp2f(8.f * (r * 0.5f + 0.5f)) - 1.f, 0.f), 255.f);
                out[((long)n * (H - 6) + (y - 3)) * W + xw] = adu;
            }
        } else {
            out[((long)n * H + y) * W + xw] = r;
        }
    }
}

extern "C" int ieagan_conv_Cto1(const void* x, const float* scale, const float* shift, int nstride, int relu, const float* w,
                                const float* bias, float* out, int tanh_out, int N, int H, int W, int C, int flip,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("conv_Cto1", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 2.0 * C), st);
    const int tiles_w = (W + C1_TW - 1) / C1_TW, tiles_h = (H + C1_TH - 1) / C1_TH;
    const long blocks = (long)N * tiles_w * tiles_h;
    CHECK_ARG(blocks > 0 && blocks < (1L << 31), "conv_Cto1: bad geometry");
    const bool aff = scale != nullptr;
#define L(CC, A, R) hipLaunchKernelGGL((conv_Cto1_kernel<CC, A, R>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, scale, \
                                       shift, nstride, w, bias, out, tanh_out, N, H, W, flip, tiles_w, tiles_h)
#define LC(CC)                        \
    if (aff && relu) L(CC, true, true); \
    else if (!aff && !relu) L(CC, false, false); \
    else if (!aff) L(CC, false, true);  \
    else L(CC, true, false)
    if (C == 16) { LC(16); }
    else if (C == 32) { LC(32); }
    else if (C == 64) { LC(64); }
    else { ieagan_set_error("conv_Cto1: C=%d not instantiated (16/32/64)", C); return IEAGAN_EINVAL; }
#undef LC
#undef L
    CHECK_LAUNCH("conv_Cto1");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// wgrad_c1 on the matrix cores: dw[tap][c] += sum_p imgval(p +/- d(tap)) * T(t[p, c]);  dw fp32 [9][C] (caller zeroes).
// GEMM view: D[tap (16 rows, 9 used)][c] over K = pixels.  A = image patches (row = tap, k = 8 consecutive pixels of a tile
// row, taken from an fp32 halo in LDS and split into bf16 high + low parts -> two MFMAs per n-tile), B = the transformed
// t tile staged in its natural [pixel][C] layout and read pixel-major with ds_read_b64_tr_b16.  Block = persistent over
// 8 x 32 pixel tiles; wave w owns tile rows 2w, 2w+1.  One atomic per (tap, c) per block at the end.
// ------------------------------------------------------------------------------------------------
#define W1_TH 8
#define W1_TW 32
__device__ __forceinline__ bf16x8 tr8(const bf16* lds, int stride_elems, int pix0, int col0, int lr) {
    const int q = lr >> 2, p = lr & 3;
    const bf16* p0 = lds + (pix0 + q) * stride_elems + col0 + 4 * p;
    const bf16* p1 = p0 + 4 * stride_elems;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int C, bool AFF, bool RELU>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const float* __restrict__ img, const float* __restrict__ tanh_y,
                                                       const bf16* __restrict__ t, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int nstride, float* __restrict__ dw,
                                                       int N, int H, int W, int flip, int tiles_w, int tiles_h, int ntiles, int tpb) {
    constexpr int NT = C / 16, CH = C / 8;
    constexpr int AW = W1_TW + 2, AH = W1_TH + 2;
    __shared__ float halo[AH][AW + 2];
    __shared__ __attribute__((aligned(16))) bf16 tl[W1_TH * W1_TW * C];          // [pixel][C]
    __shared__ float red[4][NT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int sgn = flip ? -1 : 1;
    const int dy = sgn * (lr / 3 - 1), dx = sgn * (lr % 3 - 1);                   // tap lr (rows 9..15 of A are zero)
    const int cc = threadIdx.x % CH;                                             // staging: this thread's 8-channel chunk (256 % CH == 0)
    float sc[8], sh[8];
    int n_cur = -1;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int t0 = blockIdx.x * tpb, t1 = min(t0 + tpb, ntiles);
    for (int tile = t0; tile < t1; ++tile) {
        const int n = tile / (tiles_w * tiles_h);
        const int trem = tile - n * tiles_w * tiles_h;
        const int y0 = (trem / tiles_w) * W1_TH, x0 = (trem % tiles_w) * W1_TW;
        if (AFF && n != n_cur) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sc[i] = scale[(size_t)n * nstride + cc * 8 + i];
                sh[i] = shift[(size_t)n * nstride + cc * 8 + i];
            }
            n_cur = n;
        }
        __syncthreads();                                         // previous tile's fragments consumed
        const float* im = img + (size_t)n * H * W;
        const float* ty_ = tanh_y ? tanh_y + (size_t)n * H * W : nullptr;
        for (int i = threadIdx.x; i < AH * AW; i += 256) {
            const int qy = i / AW, qx = i - qy * AW;
            const int yy = y0 - 1 + qy, xx = x0 - 1 + qx;
            float v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                v = im[yy * W + xx];
                if (ty_) {
                    const float q = ty_[yy * W + xx];
                    v *= 1.f - q * q;
                }
            }
            halo[qy][qx] = v;
        }
#pragma unroll
        for (int j = 0; j < (W1_TH * W1_TW * CH) / 256; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int px = idx / CH;                             // chunk index of this thread stays cc
            const int yy = y0 + px / W1_TW, xx = x0 + px % W1_TW;
            bf16x8 v = zero8();
            if (yy < H && xx < W) {
                v = *(const bf16x8*)(t + (((size_t)n * H + yy) * W + xx) * C + cc * 8);
                if (AFF || RELU) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float f = bf2f(v[i]);
                        if (AFF) f = fmaf(f, sc[i], sh[i]);
                        if (RELU) f = fmaxf(f, 0.f);
                        v[i] = f2bf(f);
                    }
                }
            }
            *(bf16x8*)(tl + px * C + cc * 8) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
            bf16x8 ahi = zero8(), alo = zero8();
            if (lr < 9) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = halo[row + 1 + dy][8 * lg + j + 1 + dx];
                    const bf16 h = f2bf(v);
                    ahi[j] = h;
                    alo[j] = f2bf(v - bf2f(h));
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 b = tr8(tl, C, row * W1_TW + 8 * lg, nt * 16, lr);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, b, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, b, acc[nt], 0, 0, 0);
            }
        }
    }
    // D[tap 4lg + r][channel nt*16 + lr]: fold the four waves, one atomic per (tap, c) per block
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][nt * 256 + (4 * lg + r) * 16 + lr] = acc[nt][r];
    __syncthreads();
    for (int i = threadIdx.x; i < NT * 256; i += 256) {
        const int nt = i / 256, tap = (i % 256) / 16, c = i % 16;
        if (tap < 9) atomicAdd(dw + tap * C + nt * 16 + c, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}

extern "C" int ieagan_wgrad_c1(const float* img, const float* tanh_y, const void* t, const float* scale, const float* shift,
                               int nstride, int relu, float* dw, int N, int H, int W, int C, int flip, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("wgrad_c1", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 2.0 * C), st);
    const int tiles_w = (W + W1_TW - 1) / W1_TW, tiles_h = (H + W1_TH - 1) / W1_TH;
    const long ntl = (long)N * tiles_w * tiles_h;
    CHECK_ARG(ntl > 0 && ntl < (1L << 31), "wgrad_c1: bad geometry");
    const int ntiles = (int)ntl;
    int tpb = (ntiles + 1023) / 1024;                 // <= 1024 blocks: every block ends with 9*C atomics on the same addresses
    if (tpb < 1) tpb = 1;
    const int blocks = (ntiles + tpb - 1) / tpb;
    const bool aff = scale != nullptr;
#define L(CC, A, R) hipLaunchKernelGGL((wgrad_c1_kernel<CC, A, R>), dim3((unsigned)blocks), dim3(256), 0, st, img, tanh_y, (const bf16*)t, \
                                       scale, shift, nstride, dw, N, H, W, flip, tiles_w, tiles_h, ntiles, tpb)
#define LC(CC)                        \
    if (aff && relu) L(CC, true, true); \
    else if (!aff && !relu) L(CC, false, false); \
    else if (!aff) L(CC, false, true);  \
    else L(CC, true, false)
    if (C == 16) { LC(16); }
    else if (C == 32) { LC(32); }
    else if (C == 64) { LC(64); }
    else { ieagan_set_error("wgrad_c1: C=%d not instantiated (16/32/64)", C); return IEAGAN_EINVAL; }
#undef LC
#undef L
    CHECK_LAUNCH("wgrad_c1");
    return 0;
}
