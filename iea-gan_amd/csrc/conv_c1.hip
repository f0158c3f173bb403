// 3x3 convolutions with a single-channel image on one side (coalesced VALU kernels; K or N of the
// implicit GEMM is 1, so MFMA has nothing to do here):
//   conv_1toC   image[N,H,W] fp32 -> NHWC bf16 [N,H,W,C]    D.input_conv forward (model.py:905) and the
//                                                            dgrad of G's output conv (with tanh')
//   conv_Cto1   NHWC bf16 [N,H,W,C] -> image fp32            G.output_layer: BN apply + ReLU + conv + tanh
//                                                            (model.py:379-387,487) and dgrad of D.input_conv
//   wgrad_c1    dW[tap][c] = sum_p img[p +/- d(tap)] * t[p,c]
// Thread = (pixel, 8-channel group): 16-byte accesses, lanes of one pixel adjacent.
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
template <int C, bool AFF, bool RELU>
__global__ __launch_bounds__(256) void conv_Cto1_kernel(const bf16* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int nstride, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int tanh_out,
                                                        int N, int H, int W, int flip) {
    constexpr int G = C / 8;
    __shared__ float ws[9 * C];
    for (int i = threadIdx.x; i < 9 * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const long chunks = (long)N * H * W * G;
    const long chunks_pad = (chunks + 255) / 256 * 256;     // keep whole waves alive for the shuffles
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < chunks_pad; idx += (long)gridDim.x * 256) {
        const bool live = idx < chunks;
        const long id2 = live ? idx : 0;
        const int cg = (int)(id2 % G);
        const long p = id2 / G;
        const int xw = (int)(p % W);
        const long t = p / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float sc[8], sh[8];
        if (AFF) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sc[i] = scale[(long)n * nstride + cg * 8 + i];
                sh[i] = shift[(long)n * nstride + cg * 8 + i];
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int yy = flip ? y - dy : y + dy, xx = flip ? xw - dx : xw + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const bf16x8 v = *(const bf16x8*)(x + (((long)n * H + yy) * W + xx) * C + cg * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float f = bf2f(v[i]);
                    if (AFF) f = f * sc[i] + sh[i];
                    if (RELU) f = fmaxf(f, 0.f);
                    acc += f * ws[tap * C + cg * 8 + i];
                }
            }
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (live && cg == 0) {
            float r = acc + (bias ? bias[0] : 0.f);
            if (tanh_out) r = tanhf(r);
            if (tanh_out == 2) {        // detector-unit export (model.py:1140-1148): threshold, 256^((r+1)/2) - 1, clamp, crop 3 rows
                if (y >= 3 && y < H - 3) {
                    r = (r > -0.26f) ? r : -1.f;
                    const float adu = fminf(fmaxf(exp2f(8.f * (r * 0.5f + 0.5f)) - 1.f, 0.f), 255.f);
                    out[((long)n * (H - 6) + (y - 3)) * W + xw] = adu;
                }
            } else {
                out[p] = r;
            }
        }
    }
}

extern "C" int ieagan_conv_Cto1(const void* x, const float* scale, const float* shift, int nstride, int relu, const float* w,
                                const float* bias, float* out, int tanh_out, int N, int H, int W, int C, int flip,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("conv_Cto1", 18.0 * N * H * W * (double)C, (double)N * H * W * (4.0 + 2.0 * C), st);
    long blocks = ((long)N * H * W * (C / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    const bool aff = scale != nullptr;
#define L(CC, A, R) hipLaunchKernelGGL((conv_Cto1_kernel<CC, A, R>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, scale, \
                                       shift, nstride, w, bias, out, tanh_out, N, H, W, flip)
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
