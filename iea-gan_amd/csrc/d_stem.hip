// d_stem: the first discriminator block's input side as ONE forward and ONE backward kernel.
//
// Reference (model.py:902-909, 534-557 with the shipped config): h0 = input_conv(x) (3x3, 1 -> 32 channels at 256x768), then the first
// DBlock (no pre-activation) reads h0 three times -- conv1 (1x1, 32 -> 16), conv_sc on AvgPool2d(h0) (1x1, 32 -> 32) and the pooled
// identity shortcut -- 503 MB per read at N = 40, 25 % of the discriminator's forward traffic.  h0 is a K = 9 convolution of a
// 31 MB single-channel image: cheaper to recompute than to move.  So
//   d_stem_fwd:  img -> { h1 = conv1(h0), p0 = AvgPool2d(h0), sc = conv_sc(p0) }     h0 never reaches HBM (writes 504 MB instead of
//                2.4 GB of reads + writes); conv4 of the block then takes p0 as a same-resolution shortcut operand.
//   d_stem_bwd:  (img, dh1, dp0) -> dW_in, db_in, dW1, db1     with h0 recomputed and dh0 = dh1 W1 + 0.25 expand(dp0) formed in LDS only
//                (replaces conv1's dgrad + wgrad and input_conv's wgrad: 1.9 GB -> 0.41 GB per backward pass).  D-phase passes only:
//                the G-phase pass needs d img and no weight gradients and keeps the generic launches.
// Numerics follow the separate kernels: the image enters the matrix cores as bf16 high + low parts against bf16 weights
// (conv_c1.hip), h0 / p0 are rounded to bf16 where the separate path stored them, every accumulation is fp32.
// Tile = 8 x 32 pixels per block, wave w owns rows 2w, 2w+1 (= one pooled row); everything after the halo load is wave-private.
#include "common.h"
#include "conv_args.h"

#define ST_TH 8
#define ST_TW 32
#define ST_C0 32          // input_conv output channels
#define ST_C1 16          // conv1 output channels (hidden)
#define ST_CS 32          // conv_sc output channels
#define ST_XS 48          // bf16 row stride of the 32-channel LDS tiles: 96 bytes = an odd multiple of 32 (conflict-free transposed reads)
#define ST_LDO 36         // fp32 row stride of the accumulator transpose buffer

struct DStemArgs {
    const float* img;     // fp32 [N, H, W]
    int N, H, W;
    const float* w_in;    // fp32 [9][32]   input_conv weight / sigma (tap-major)
    const float* b_in;    // [32]
    const void* w1;       // bf16 [16][32]  conv1 forward pack (k = cin)
    const float* b1;      // [16]
    const void* wsc;      // bf16 [32][32]  conv_sc forward pack
    const float* bsc;     // [32]
    void* h1;             // bf16 [N, H, W, 16]
    void* p0;             // bf16 [N, H/2, W/2, 32]
    void* sc;             // bf16 [N, H/2, W/2, 32]
    // backward
    const void* dh1;      // bf16 [N, H, W, 16]
    const void* dp0;      // bf16 [N, H/2, W/2, 32]
    const void* w1_bwd;   // bf16 [32][32]  conv1 transposed pack ([cin][k = cout], 16 used)
    float* dw_in;         // fp32 [9][32]        accumulated
    float* db_in;         // fp32 [32 repl][32]  accumulated (replica = block % 32)
    float* dw1;           // fp32 [16][32]       accumulated
    float* db1;           // fp32 [32 repl][16]  accumulated
};

// K(pixel)-major fragment: 8 consecutive rows pix0 .. pix0+7 of column (col0 + lr) of a [pixel][channel] 16-bit LDS image
__device__ __forceinline__ bf16x8 st_tr8(const bf16* lds, int stride_elems, int pix0, int col0, int lr) {
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

#define ST_WAVE_FENCE()                                     \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
    __builtin_amdgcn_wave_barrier()

// h0 of this wave's two tile rows (64 pixels x 32 channels, + bias, rounded to bf16) into h0s[(rr*32 + col)][ST_XS]
__device__ __forceinline__ void st_h0_tile(const float (*halo)[ST_TW + 4], int wave, int lane, const bf16x8 (&bfrag)[2], const float (&bv)[8], float* o,
                                           bf16* h0s) {
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int ty = 2 * wave + (mi >> 1), tx0 = (mi & 1) * 16;
        bf16x8 af;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = (lg & 1) * 8 + j;
            float v = 0.f;
            if (tap < 9) v = halo[ty + tap / 3][tx0 + lr + tap % 3];
            const bf16 hi = f2bf(v);
            af[j] = (lg < 2) ? hi : f2bf(v - bf2f(hi));
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[nt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * lg + r) * ST_LDO + nt * 16 + lr] = d[r];
        }
        ST_WAVE_FENCE();
        {
            const int px = lane >> 2, cc = lane & 3;               // 16 pixels x 4 chunks = 64 items
            const f32x4 lo = *(const f32x4*)(o + px * ST_LDO + cc * 8);
            const f32x4 hi = *(const f32x4*)(o + px * ST_LDO + cc * 8 + 4);
            bf16x8 ov;
            ov[0] = f2bf(lo[0] + bv[0]); ov[1] = f2bf(lo[1] + bv[1]); ov[2] = f2bf(lo[2] + bv[2]); ov[3] = f2bf(lo[3] + bv[3]);
            ov[4] = f2bf(hi[0] + bv[4]); ov[5] = f2bf(hi[1] + bv[5]); ov[6] = f2bf(hi[2] + bv[6]); ov[7] = f2bf(hi[3] + bv[7]);
            *(bf16x8*)(h0s + ((mi >> 1) * 32 + tx0 + px) * ST_XS + cc * 8) = ov;
        }
        ST_WAVE_FENCE();
    }
}

__device__ __forceinline__ void st_load_halo(float (*halo)[ST_TW + 4], const float* im, int H, int W, int y0, int x0) {
    constexpr int AW = ST_TW + 2, AH = ST_TH + 2;
    for (int i = threadIdx.x; i < AH * AW; i += 256) {
        const int qy = i / AW, qx = i - qy * AW;
        const int yy = y0 - 1 + qy, xx = x0 - 1 + qx;
        halo[qy][qx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? im[(long)yy * W + xx] : 0.f;
    }
}

__device__ __forceinline__ void st_in_frags(const float* w_in, int lane, bf16x8 (&bfrag)[2]) {
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = (lg & 1) * 8 + j;
            bfrag[nt][j] = f2bf(tap < 9 ? w_in[tap * ST_C0 + nt * 16 + lr] : 0.f);
        }
}

__global__ __launch_bounds__(256, 3) void d_stem_fwd_kernel(DStemArgs a, int tiles_w, int tiles_h) {
    __shared__ float halo[ST_TH + 2][ST_TW + 4];
    __shared__ __attribute__((aligned(16))) float ot[4][16 * ST_LDO];
    __shared__ __attribute__((aligned(16))) bf16 h0t[4][64 * ST_XS];
    __shared__ __attribute__((aligned(16))) bf16 p0t[4][16 * ST_XS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1;
    bf16x8 bin[2];
    st_in_frags(a.w_in, lane, bin);
    const int cc4 = lane & 3;
    float bv[8], bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bv[i] = a.b_in[cc4 * 8 + i];
        bs[i] = a.bsc[cc4 * 8 + i];
    }
    // conv1: B[k = cin][col = cout lr] = w1[lr][lg*8 ..];  conv_sc: two n-tiles of the [32][32] pack
    const bf16x8 w1f = *(const bf16x8*)((const bf16*)a.w1 + lr * 32 + lg * 8);
    bf16x8 wsf[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wsf[nt] = *(const bf16x8*)((const bf16*)a.wsc + (nt * 16 + lr) * 32 + lg * 8);
    float b1v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b1v[i] = a.b1[(lane & 1) * 8 + i];
    float* o = ot[wave];
    bf16* h0s = h0t[wave];
    bf16* p0s = p0t[wave];
    const int ntiles = a.N * tiles_w * tiles_h;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tile_end = min((int)(blockIdx.x + 1) * per, ntiles);
    // The image halo of tile t+1 (340 floats) is requested into two registers per thread right after tile t's halo went to LDS, and
    // every global access of the loop is unconditional (coordinates clamped, the padding ring zeroed through a mask; all 64 lanes
    // store): this kernel writes 0.5 GB per launch, and a halo load that follows the stores in program order -- or that the
    // compiler cannot count past -- waits for every one of them (stores count on vmcnt on gfx9).
    constexpr int AWH = ST_TW + 2, AHH = ST_TH + 2, HN = (AHH * AWH + 255) / 256;
    float hn[HN];
    unsigned hok = 0;
    auto request = [&](int tile) {
        const int n = tile / (tiles_w * tiles_h);
        const int trem = tile - n * tiles_w * tiles_h;
        const int y0 = (trem / tiles_w) * ST_TH, x0 = (trem % tiles_w) * ST_TW;
        const float* im = a.img + (long)n * H * W;
        hok = 0;
#pragma unroll
        for (int j = 0; j < HN; ++j) {
            const int i = min((int)threadIdx.x + j * 256, AHH * AWH - 1);
            const int qy = i / AWH, qx = i - qy * AWH;
            const int yy = y0 - 1 + qy, xx = x0 - 1 + qx;
            hn[j] = im[(long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) hok |= 1u << j;
        }
    };
    const int tile_begin = blockIdx.x * per;
    if (tile_begin < tile_end) request(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int n = tile / (tiles_w * tiles_h);
        const int trem = tile - n * tiles_w * tiles_h;
        const int y0 = (trem / tiles_w) * ST_TH, x0 = (trem % tiles_w) * ST_TW;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < HN; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < AHH * AWH) halo[i / AWH][i % AWH] = (hok & (1u << j)) ? hn[j] : 0.f;
        }
        __syncthreads();
        request(min(tile + 1, tile_end - 1));                      // (last tile: re-requests itself, unused)
        st_h0_tile(halo, wave, lane, bin, bv, o, h0s);
        // ---- conv1: h1[px][16] = h0[px][32] W1^T + b1, one K step; 16 pixels per MFMA
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int prow = (mi >> 1) * 32 + (mi & 1) * 16;
            const bf16x8 af = *(const bf16x8*)(h0s + (prow + lr) * ST_XS + lg * 8);
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, w1f, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(4 * lg + r) * ST_LDO + lr] = d[r];
            ST_WAVE_FENCE();
            {                                                      // 16 pixels x 2 chunks: lanes 32..63 repeat lanes 0..31 (the same
                const int px = (lane & 31) >> 1, cc = lane & 1;    // bytes to the same address: an unconditional store instruction)
                const f32x4 lo = *(const f32x4*)(o + px * ST_LDO + cc * 8);
                const f32x4 hi = *(const f32x4*)(o + px * ST_LDO + cc * 8 + 4);
                bf16x8 ov;
                ov[0] = f2bf(lo[0] + b1v[0]); ov[1] = f2bf(lo[1] + b1v[1]); ov[2] = f2bf(lo[2] + b1v[2]); ov[3] = f2bf(lo[3] + b1v[3]);
                ov[4] = f2bf(hi[0] + b1v[4]); ov[5] = f2bf(hi[1] + b1v[5]); ov[6] = f2bf(hi[2] + b1v[6]); ov[7] = f2bf(hi[3] + b1v[7]);
                const int y = y0 + 2 * wave + (mi >> 1), x = x0 + (mi & 1) * 16 + px;
                *(bf16x8*)((bf16*)a.h1 + (((long)n * H + y) * W + x) * ST_C1 + cc * 8) = ov;
            }
            ST_WAVE_FENCE();
        }
        // ---- p0 = 2x2 average of the bf16 h0 (as the pooled prologue of the separate kernels: fp32 sum, one rounding)
        {
            const int ppx = lane >> 2;                             // 16 pooled pixels x 4 chunks
            const bf16* b0 = h0s + (2 * ppx) * ST_XS + cc4 * 8;
            const bf16x8 t0 = *(const bf16x8*)b0, t1 = *(const bf16x8*)(b0 + ST_XS), t2 = *(const bf16x8*)(b0 + 32 * ST_XS),
                         t3 = *(const bf16x8*)(b0 + 33 * ST_XS);
            bf16x8 pv;
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = f2bf(0.25f * (bf2f(t0[i]) + bf2f(t1[i]) + bf2f(t2[i]) + bf2f(t3[i])));
            *(bf16x8*)(p0s + ppx * ST_XS + cc4 * 8) = pv;
            const int yp = (y0 >> 1) + wave, xp = (x0 >> 1) + ppx;
            *(bf16x8*)((bf16*)a.p0 + (((long)n * Hp + yp) * Wp + xp) * ST_C0 + cc4 * 8) = pv;
        }
        ST_WAVE_FENCE();
        // ---- conv_sc: sc[ppx][32] = p0[ppx][32] Wsc^T + bsc
        {
            const bf16x8 af = *(const bf16x8*)(p0s + lr * ST_XS + lg * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wsf[nt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(4 * lg + r) * ST_LDO + nt * 16 + lr] = d[r];
            }
            ST_WAVE_FENCE();
            const int ppx = lane >> 2;
            const f32x4 lo = *(const f32x4*)(o + ppx * ST_LDO + cc4 * 8);
            const f32x4 hi = *(const f32x4*)(o + ppx * ST_LDO + cc4 * 8 + 4);
            bf16x8 ov;
            ov[0] = f2bf(lo[0] + bs[0]); ov[1] = f2bf(lo[1] + bs[1]); ov[2] = f2bf(lo[2] + bs[2]); ov[3] = f2bf(lo[3] + bs[3]);
            ov[4] = f2bf(hi[0] + bs[4]); ov[5] = f2bf(hi[1] + bs[5]); ov[6] = f2bf(hi[2] + bs[6]); ov[7] = f2bf(hi[3] + bs[7]);
            const int yp = (y0 >> 1) + wave, xp = (x0 >> 1) + ppx;
            *(bf16x8*)((bf16*)a.sc + (((long)n * Hp + yp) * Wp + xp) * ST_CS + cc4 * 8) = ov;
            ST_WAVE_FENCE();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward (weight gradients only)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void d_stem_bwd_kernel(DStemArgs a, int tiles_w, int tiles_h) {
    __shared__ float halo[ST_TH + 2][ST_TW + 4];
    __shared__ __attribute__((aligned(16))) float ot[4][16 * ST_LDO];
    __shared__ __attribute__((aligned(16))) bf16 h0t[4][64 * ST_XS];
    __shared__ __attribute__((aligned(16))) bf16 d0t[4][64 * ST_XS];       // dh0
    __shared__ __attribute__((aligned(16))) bf16 g1t[4][64 * ST_C1];       // dh1 (32-byte rows)
    __shared__ __attribute__((aligned(16))) bf16 dpt[4][16 * ST_XS];       // dp0 of the wave's pooled row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1;
    bf16x8 bin[2];
    st_in_frags(a.w_in, lane, bin);
    const int cc4 = lane & 3;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = a.b_in[cc4 * 8 + i];
    // dgrad of conv1: dh0[px][cin] = sum_cout dh1[px][cout] W1[cout][cin]: B[k = cout][col = cin lr] = w1_bwd[nt*16 + lr][lg*8 ..], 16 k used
    bf16x8 w1b[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w1b[nt] = (lg < 2) ? *(const bf16x8*)((const bf16*)a.w1_bwd + (nt * 16 + lr) * 32 + lg * 8) : zero8();
    float* o = ot[wave];
    bf16* h0s = h0t[wave];
    bf16* d0s = d0t[wave];
    bf16* g1s = g1t[wave];
    bf16* dps = dpt[wave];
    f32x4 aw1[2], awin[2], ab1;                                  // dW1 [16 couts][32 cin], dW_in [16 "taps" (9 + the bias row)][32], db1 [16]
    ab1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) aw1[nj] = awin[nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (bf16)1.0f;
    const int ntiles = a.N * tiles_w * tiles_h;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tile_end = min((int)(blockIdx.x + 1) * per, ntiles);
    // Every operand of tile t+1 -- dh1 (64 pixels x 2 chunks per wave), dp0 (16 pooled pixels x 4 chunks), the image halo -- is
    // requested into registers (unconditionally, halo coordinates clamped) once tile t's copies sit in LDS: the 60 tiles of a block
    // were a chain of exposed load latencies.
    constexpr int AWH = ST_TW + 2, AHH = ST_TH + 2, HN = (AHH * AWH + 255) / 256;
    bf16x8 rg[2], rp;
    float hn[HN];
    unsigned hok = 0;
    auto request = [&](int tile) {
        const int n = tile / (tiles_w * tiles_h);
        const int trem = tile - n * tiles_w * tiles_h;
        const int y0 = (trem / tiles_w) * ST_TH, x0 = (trem % tiles_w) * ST_TW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int item = j * 64 + lane, p = item >> 1, cc = item & 1;
            const int y = y0 + 2 * wave + (p >> 5), x = x0 + (p & 31);
            rg[j] = *(const bf16x8*)((const bf16*)a.dh1 + (((long)n * H + y) * W + x) * ST_C1 + cc * 8);
        }
        {
            const int ppx = lane >> 2;
            rp = *(const bf16x8*)((const bf16*)a.dp0 + (((long)n * Hp + (y0 >> 1) + wave) * Wp + (x0 >> 1) + ppx) * ST_C0 + cc4 * 8);
        }
        const float* im = a.img + (long)n * H * W;
        hok = 0;
#pragma unroll
        for (int j = 0; j < HN; ++j) {
            const int i = min((int)threadIdx.x + j * 256, AHH * AWH - 1);
            const int qy = i / AWH, qx = i - qy * AWH;
            const int yy = y0 - 1 + qy, xx = x0 - 1 + qx;
            hn[j] = im[(long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) hok |= 1u << j;
        }
    };
    const int tile_begin = blockIdx.x * per;
    if (tile_begin < tile_end) request(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int n = tile / (tiles_w * tiles_h);
        const int trem = tile - n * tiles_w * tiles_h;
        const int y0 = (trem / tiles_w) * ST_TH, x0 = (trem % tiles_w) * ST_TW;
        (void)n;
        __syncthreads();                                           // the previous tile is done with halo / g1s / dps
#pragma unroll
        for (int j = 0; j < HN; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < AHH * AWH) halo[i / AWH][i % AWH] = (hok & (1u << j)) ? hn[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int item = j * 64 + lane, p = item >> 1, cc = item & 1;
            *(bf16x8*)(g1s + p * ST_C1 + cc * 8) = rg[j];
        }
        *(bf16x8*)(dps + (lane >> 2) * ST_XS + cc4 * 8) = rp;
        __syncthreads();
        request(min(tile + 1, tile_end - 1));                      // (last tile: re-requests itself, unused)
        st_h0_tile(halo, wave, lane, bin, bv, o, h0s);
        ST_WAVE_FENCE();
        // ---- dh0 = dh1 W1 + 0.25 expand(dp0), rounded to bf16 (what the separate path stored), LDS only
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int prow = (mi >> 1) * 32 + (mi & 1) * 16;
            const bf16x8 af = (lg < 2) ? *(const bf16x8*)(g1s + (prow + lr) * ST_C1 + lg * 8) : zero8();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, w1b[nt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(4 * lg + r) * ST_LDO + nt * 16 + lr] = d[r];
            }
            ST_WAVE_FENCE();
            {
                const int px = lane >> 2;
                const f32x4 lo = *(const f32x4*)(o + px * ST_LDO + cc4 * 8);
                const f32x4 hi = *(const f32x4*)(o + px * ST_LDO + cc4 * 8 + 4);
                const bf16x8 dp = *(const bf16x8*)(dps + (((mi & 1) * 16 + px) >> 1) * ST_XS + cc4 * 8);
                bf16x8 ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = f2bf((i < 4 ? lo[i & 3] : hi[i & 3]) + 0.25f * bf2f(dp[i]));
                *(bf16x8*)(d0s + (prow + px) * ST_XS + cc4 * 8) = ov;
            }
            ST_WAVE_FENCE();
        }
        // ---- weight gradients: K = the 64 pixels of the wave's two rows, 8 consecutive pixels of a row per lane group
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int pix0 = rr * 32 + 8 * lg;
            // dW1[cout lr][cin] += dh1^T h0;  db1 via an all-ones B tile
            const bf16x8 ag = st_tr8(g1s, ST_C1, pix0, 0, lr);
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) aw1[nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag, st_tr8(h0s, ST_XS, pix0, nj * 16, lr), aw1[nj], 0, 0, 0);
            ab1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ag, ones, ab1, 0, 0, 0);
            // dW_in[tap lr][c] += patch^T dh0 (image values as bf16 high + low parts); row 9 = ones -> db_in
            bf16x8 ahi = zero8(), alo = zero8();
            if (lr < 9) {
                const int row = 2 * wave + rr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = halo[row + lr / 3][8 * lg + j + lr % 3];
                    const bf16 h = f2bf(v);
                    ahi[j] = h;
                    alo[j] = f2bf(v - bf2f(h));
                }
            } else if (lr == 9) {
                ahi = ones;
            }
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const bf16x8 b = st_tr8(d0s, ST_XS, pix0, nj * 16, lr);
                awin[nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, b, awin[nj], 0, 0, 0);
                awin[nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, b, awin[nj], 0, 0, 0);
            }
        }
        ST_WAVE_FENCE();
    }
    // ---- fold the four waves through LDS, one round of atomics per block.  D[row 4lg + r][col nj*16 + lr]
    __syncthreads();
    float* T = (float*)h0t;                                      // [16][32] dW1 | [16][32] dW_in (+ bias row) | [16] db1
    for (int i = threadIdx.x; i < 16 * 32 * 2 + 16; i += 256) T[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            atomicAdd(&T[(4 * lg + r) * 32 + nj * 16 + lr], aw1[nj][r]);
            atomicAdd(&T[512 + (4 * lg + r) * 32 + nj * 16 + lr], awin[nj][r]);
        }
    if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&T[1024 + 4 * lg + r], ab1[r]);
    }
    __syncthreads();
    const int rep = blockIdx.x % STAT_REPL;
    for (int i = threadIdx.x; i < 512; i += 256) atomicAdd(a.dw1 + i, T[i]);                       // [16][32] (Kpad = 32)
    for (int i = threadIdx.x; i < 9 * 32; i += 256) atomicAdd(a.dw_in + i, T[512 + i]);            // taps 0..8
    if (threadIdx.x < 32) atomicAdd(a.db_in + rep * 32 + threadIdx.x, T[512 + 9 * 32 + threadIdx.x]);
    if (threadIdx.x < 16) atomicAdd(a.db1 + rep * 16 + threadIdx.x, T[1024 + threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
static int stem_check(const ieagan_d_stem_desc* d) {
    CHECK_ARG(d != nullptr && d->img != nullptr && d->w_in != nullptr && d->b_in != nullptr, "d_stem: null pointer");
    CHECK_ARG(d->N >= 1 && d->H % ST_TH == 0 && d->W % ST_TW == 0, "d_stem: H %% 8 and W %% 32 required (%d x %d)", d->H, d->W);
    return 0;
}

static DStemArgs stem_args(const ieagan_d_stem_desc* d) {
    DStemArgs a{};
    a.img = d->img; a.N = d->N; a.H = d->H; a.W = d->W;
    a.w_in = d->w_in; a.b_in = d->b_in; a.w1 = d->w1; a.b1 = d->b1; a.wsc = d->wsc; a.bsc = d->bsc;
    a.h1 = d->h1; a.p0 = d->p0; a.sc = d->sc;
    a.dh1 = d->dh1; a.dp0 = d->dp0; a.w1_bwd = d->w1_bwd; a.dw_in = d->dw_in; a.db_in = d->db_in; a.dw1 = d->dw1; a.db1 = d->db1;
    return a;
}

extern "C" int ieagan_d_stem_fwd(const ieagan_d_stem_desc* d, void* stream) {
    if (int rc = stem_check(d)) return rc;
    CHECK_ARG(d->w1 != nullptr && d->b1 != nullptr && d->wsc != nullptr && d->bsc != nullptr && d->h1 != nullptr && d->p0 != nullptr && d->sc != nullptr,
              "d_stem_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const double P = (double)d->N * d->H * d->W;
    ProfScope prof("d_stem_fwd", 2.0 * P * (9 * 32 + 32 * 16 + 0.25 * 32 * 32), P * (4.0 + 2.0 * (16 + 0.25 * 64)), st);
    const int tiles_w = d->W / ST_TW, tiles_h = d->H / ST_TH;
    const long ntl = (long)d->N * tiles_w * tiles_h;
    const long blocks = ntl < 768 ? ntl : 768;                    // one round of persistent blocks (41 KB of LDS: three per CU)
    hipLaunchKernelGGL(d_stem_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, stem_args(d), tiles_w, tiles_h);
    CHECK_LAUNCH("d_stem_fwd");
    return 0;
}

extern "C" int ieagan_d_stem_bwd(const ieagan_d_stem_desc* d, void* stream) {
    if (int rc = stem_check(d)) return rc;
    CHECK_ARG(d->dh1 != nullptr && d->dp0 != nullptr && d->w1_bwd != nullptr && d->dw_in != nullptr && d->db_in != nullptr && d->dw1 != nullptr && d->db1 != nullptr,
              "d_stem_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const double P = (double)d->N * d->H * d->W;
    ProfScope prof("d_stem_bwd", 2.0 * P * (9 * 32 + 3 * 32 * 16 + 2 * 9 * 32), P * (4.0 + 2.0 * (16 + 0.25 * 32)), st);
    const int tiles_w = d->W / ST_TW, tiles_h = d->H / ST_TH;
    const long ntl = (long)d->N * tiles_w * tiles_h;
    const long blocks = ntl < 512 ? ntl : 512;                    // one round of persistent blocks (two per CU)
    hipLaunchKernelGGL(d_stem_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, stem_args(d), tiles_w, tiles_h);
    CHECK_LAUNCH("d_stem_bwd");
    return 0;
}
