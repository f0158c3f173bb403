// 3x3 convolutions with a single-channel image on one side:
//   conv_1toC   image[N,H,W] fp32 -> NHWC bf16 [N,H,W,C]    D.input_conv forward (model.py:905) and the
//                                                            dgrad of G's output conv (with tanh'); coalesced VALU,
//                                                            thread = (pixel, 8-channel group), weights in registers
//   conv_Cto1   NHWC bf16 [N,H,W,C] -> image fp32            G.output_layer: BN apply + ReLU + conv + tanh (+ the
//                                                            detector-unit export) (model.py:379-387,487,1139-1147) and
//                                                            dgrad of D.input_conv; per-tap dot products on MFMA
//   wgrad_c1    dW[tap][c] = sum_p img[p +/- d(tap)] * t[p,c]   (coalesced VALU)
#include "common.h"

// w: fp32 [9][C].  img_mul (optional): the image is multiplied by (1 - y*y) first (tanh backward).
template <int C>
__global__ __launch_bounds__(256) void conv_1toC_kernel(const float* __restrict__ img, const float* __restrict__ tanh_y,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        bf16* __restrict__ out, int N, int H, int W, int flip) {
    constexpr int G = C / 8;
    // 256 % G == 0 and the grid stride is a multiple of 256: a thread keeps its channel group for every chunk it
    // visits, so its 9 x 8 weights and 8 biases live in registers (no LDS / L1 traffic inside the loop)
    const int cg = threadIdx.x % G;
    float wr[9][8], br[8];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[tap][i] = w[(flip ? 8 - tap : tap) * C + cg * 8 + i];     // flip: correlate with the rotated kernel
#pragma unroll
    for (int i = 0; i < 8; ++i) br[i] = bias ? bias[cg * 8 + i] : 0.f;
    const unsigned HW = (unsigned)H * W;
    const unsigned pixels = (unsigned)N * HW;                    // < 2^31 (checked by the launcher)
    const unsigned pstride = gridDim.x * (256 / G);
    for (unsigned p = blockIdx.x * (256 / G) + threadIdx.x / G; p < pixels; p += pstride) {
        const unsigned n = p / HW, rem = p - n * HW;
        const int y = (int)(rem / W), x = (int)(rem - (unsigned)y * W);
        const float* im = img + (size_t)n * HW;
        const float* ty_ = tanh_y ? tanh_y + (size_t)n * HW : nullptr;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = br[i];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            float v = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                v = im[yy * W + xx];
                if (ty_) {
                    const float ty = ty_[yy * W + xx];
                    v *= 1.f - ty * ty;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(v, wr[tap][i], acc[i]);
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(acc[i]);
        *(bf16x8*)(out + ((size_t)p * G + cg) * 8) = o;
    }
}

extern "C" int ieagan_conv_1toC(const float* img, const float* tanh_y, const float* w, const float* bias, void* out, int N,
                                int H, int W, int C, int flip, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG((long)N * H * W < (1L << 31), "conv_1toC: more than 2^31 pixels");
    ProfScope prof("conv_1toC", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 2.0 * C), st);
    long blocks = ((long)N * H * W * (C / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
#define L(CC) hipLaunchKernelGGL((conv_1toC_kernel<CC>), dim3((unsigned)blocks), dim3(256), 0, st, img, tanh_y, w, bias, (bf16*)out, N, H, W, flip)
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
    for (int mt = wave; mt < MTILES; mt += 4) {
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
                af = *(const bf16x8*)(x + (((long)n * H + yy) * W + xx) * C + c0);
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
// wgrad_c1: dw[tap][c] += sum_p imgval(p +/- d(tap)) * T(t[p,c]);  dw fp32 [9][C] (caller zeroes).
// ------------------------------------------------------------------------------------------------
template <int C, bool AFF, bool RELU>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const float* __restrict__ img, const float* __restrict__ tanh_y,
                                                       const bf16* __restrict__ t, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int nstride, float* __restrict__ dw,
                                                       int N, int H, int W, int flip) {
    constexpr int G = C / 8;
    __shared__ float red[256 * 8];
    float acc[9][8];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[a][i] = 0.f;
    const int cg = threadIdx.x % G;
    const long chunks = (long)N * H * W * G;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < chunks; idx += (long)gridDim.x * 256) {
        const long p = idx / G;
        const int x = (int)(p % W);
        const long tt = p / W;
        const int y = (int)(tt % H);
        const int n = (int)(tt / H);
        const long nb = (long)n * H * W;
        const bf16x8 v = *(const bf16x8*)(t + idx * 8);
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f[i] = bf2f(v[i]);
            if (AFF) f[i] = f[i] * scale[(long)n * nstride + cg * 8 + i] + shift[(long)n * nstride + cg * 8 + i];
            if (RELU) f[i] = fmaxf(f[i], 0.f);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int yy = flip ? y - dy : y + dy, xx = flip ? x - dx : x + dx;
            float iv = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                iv = img[nb + (long)yy * W + xx];
                if (tanh_y) {
                    const float ty = tanh_y[nb + (long)yy * W + xx];
                    iv *= 1.f - ty * ty;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[tap][i] += iv * f[i];
        }
    }
    // block reduce per tap by channel group, one atomic per (tap, c) per block
    const int tid = threadIdx.x;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = acc[tap][i];
        __syncthreads();
        for (int o = tid; o < C; o += 256) {
            const int g = o >> 3, i = o & 7;
            float s = 0.f;
            for (int u = g; u < 256; u += G) s += red[u * 8 + i];
            atomicAdd(dw + tap * C + o, s);
        }
        __syncthreads();
    }
}

extern "C" int ieagan_wgrad_c1(const float* img, const float* tanh_y, const void* t, const float* scale, const float* shift,
                               int nstride, int relu, float* dw, int N, int H, int W, int C, int flip, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("wgrad_c1", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 2.0 * C), st);
    long blocks = ((long)N * H * W * (C / 8) + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const bool aff = scale != nullptr;
#define L(CC, A, R) hipLaunchKernelGGL((wgrad_c1_kernel<CC, A, R>), dim3((unsigned)blocks), dim3(256), 0, st, img, tanh_y, (const bf16*)t, \
                                       scale, shift, nstride, dw, N, H, W, flip)
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
