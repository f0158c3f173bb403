// Coalesced element-wise kernels on fp32 single-channel events and on the flat parameter arena:
//   DiffAugment 'color,translation,cutout' forward / backward  (diff_aug.py:10-109)
//   CR_DiffAug flip + reflect translation                       (cr_diff_aug.py:11-63)
//   Adam (no amsgrad / weight decay) on the flat arena          (model.py:410-416, 858-864)
//   EMA over the whole flat state                               (utils/__init__.py:825-837)
#include "common.h"

// per-image sums in AUG_SLOTS slots per image: block b of image n STORES its partial into out[n][b] (one writer per slot; the grid never
// has more than AUG_SLOTS blocks per image), the consumers fold the slots in a fixed order (aug_fold) -- bit-reproducible, unlike one
// float atomic per block on out[n]
//   MODE 0: f = x                                   (mean for rand_contrast)
//   MODE 1: f = gout at positions that survive translation + cutout (mean term of the backward)
struct AugDraws {
    const float* bright;   // [N] uniform
    const float* contrast; // [N] uniform
    const long* tx;        // [N]
    const long* ty;
    const long* ox;
    const long* oy;
};

__device__ __forceinline__ bool in_cut(int h, int w, int H, int W, int ox, int oy) {
    const int ch = (int)(H * 0.5f + 0.5f), cw = (int)(W * 0.5f + 0.5f);
    int r0 = ox - ch / 2, r1 = ox - ch / 2 + ch - 1;
    int c0 = oy - cw / 2, c1 = oy - cw / 2 + cw - 1;
    r0 = max(0, min(r0, H - 1));
    r1 = max(0, min(r1, H - 1));
    c0 = max(0, min(c0, W - 1));
    c1 = max(0, min(c1, W - 1));
    return h >= r0 && h <= r1 && w >= c0 && w <= c1;
}

// Row-based mapping (round 3): a block walks rows h = blockIdx.x, + gridDim.x, ..., a thread four consecutive columns -- 16-byte
// accesses on the aligned side, no per-pixel 64-bit division (the flat one-float-per-thread form with p / W per pixel took 120 us
// for a 31 MB image batch, 0.5 TB/s).  Source indices are clamped (unconditional loads), validity is a select.
typedef float af4 __attribute__((ext_vector_type(4)));

#define AUG_SLOTS 128          // partial-sum slots per image (== the largest row grid); include/ieagan_hip.h: IEAGAN_AUG_SLOTS
template <int MODE>
__global__ __launch_bounds__(256) void aug_sum_kernel(const float* __restrict__ x, AugDraws d, float* __restrict__ out, int H, int W) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const long HW = (long)H * W;
    const float* xi = x + (long)n * HW;
    int tx = 0, ty = 0, ox = 0, oy = 0;
    if (MODE == 1) {
        tx = (int)d.tx[n]; ty = (int)d.ty[n]; ox = (int)d.ox[n]; oy = (int)d.oy[n];
    }
    const bool w4ok = (W & 3) == 0;
    float s = 0.f;
    for (int h = blockIdx.x; h < H; h += gridDim.x) {
        const float* row = xi + (long)h * W;
        const bool hok = MODE == 0 || (h + tx >= 0 && h + tx < H);
        for (int w0 = threadIdx.x * 4; w0 < W; w0 += 1024) {
            float v[4];
            if (w4ok) {
                const af4 t = *(const af4*)(row + w0);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (w0 + j < W) ? row[w0 + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = w0 + j;
                bool keep = w < W;
                if (MODE == 1) keep = keep && hok && w + ty >= 0 && w + ty < W && !in_cut(h, w, H, W, ox, oy);
                s += keep ? v[j] : 0.f;
            }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)n * AUG_SLOTS + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// sum of the AUG_SLOTS slots of image n, by every thread in the same fixed order (unused slots hold the caller's zeros)
__device__ __forceinline__ float aug_fold(const float* __restrict__ sums, int n) {
    float s = 0.f;
    for (int k = 0; k < AUG_SLOTS; ++k) s += sums[(long)n * AUG_SLOTS + k];
    return s;
}

__global__ __launch_bounds__(256) void diffaug_fwd_kernel(const float* __restrict__ x, AugDraws d, const float* __restrict__ sums,
                                                          float* __restrict__ out, int H, int W) {
    const int n = blockIdx.y;
    const long HW = (long)H * W;
    const float b = d.bright[n] - 0.5f, c = d.contrast[n] + 0.5f;
    const float m = aug_fold(sums, n) / (float)HW + b;          // mean of the brightened image
    const int tx = (int)d.tx[n], ty = (int)d.ty[n], ox = (int)d.ox[n], oy = (int)d.oy[n];
    const float* xi = x + (long)n * HW;
    float* oi = out + (long)n * HW;
    const bool w4ok = (W & 3) == 0;
    for (int h = blockIdx.x; h < H; h += gridDim.x) {
        const int hs = h + tx;
        const bool hok = hs >= 0 && hs < H;
        const float* srow = xi + (long)min(max(hs, 0), H - 1) * W;
        float* orow = oi + (long)h * W;
        for (int w0 = threadIdx.x * 4; w0 < W; w0 += 1024) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = w0 + j, ws = w + ty;
                const float xb = srow[min(max(ws, 0), W - 1)] + b;          // (the source column may be unaligned: four scalar loads)
                const bool keep = hok && ws >= 0 && ws < W && !in_cut(h, w, H, W, ox, oy);
                v[j] = keep ? (xb - m) * c + m : 0.f;
            }
            if (w4ok) {
                *(af4*)(orow + w0) = (af4){v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (w0 + j < W) orow[w0 + j] = v[j];
            }
        }
    }
}

// gx[q] = c * g'(q) + (1-c) * mean(g'),  g'(q) = gout[q - t] if that output pixel exists and is kept
__global__ __launch_bounds__(256) void diffaug_bwd_kernel(const float* __restrict__ gout, AugDraws d, const float* __restrict__ gsums,
                                                          float* __restrict__ gx, int H, int W) {
    const int n = blockIdx.y;
    const long HW = (long)H * W;
    const float c = d.contrast[n] + 0.5f;
    const float mterm = (1.f - c) * aug_fold(gsums, n) / (float)HW;
    const int tx = (int)d.tx[n], ty = (int)d.ty[n], ox = (int)d.ox[n], oy = (int)d.oy[n];
    const float* gi = gout + (long)n * HW;
    float* oi = gx + (long)n * HW;
    const bool w4ok = (W & 3) == 0;
    for (int hs = blockIdx.x; hs < H; hs += gridDim.x) {
        const int h = hs - tx;
        const bool hok = h >= 0 && h < H;
        const float* grow = gi + (long)min(max(h, 0), H - 1) * W;
        float* orow = oi + (long)hs * W;
        for (int w0 = threadIdx.x * 4; w0 < W; w0 += 1024) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ws = w0 + j, w = ws - ty;
                const float g = grow[min(max(w, 0), W - 1)];
                const bool keep = hok && w >= 0 && w < W && !in_cut(h, w, H, W, ox, oy);
                v[j] = c * (keep ? g : 0.f) + mterm;
            }
            if (w4ok) {
                *(af4*)(orow + w0) = (af4){v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (w0 + j < W) orow[w0 + j] = v[j];
            }
        }
    }
}

static inline dim3 img_grid(int N, long HW) {
    long b = (HW + 255) / 256;
    if (b > 256) b = 256;
    return dim3((unsigned)b, N);
}
static inline dim3 row_grid(int N, int H) {        // the row-based kernels: one block per 1..4 rows of an image
    int b = H < AUG_SLOTS ? H : AUG_SLOTS;
    return dim3((unsigned)b, N);
}

extern "C" int ieagan_diffaug_fwd(const float* x, const float* bright, const float* contrast, const long* tx, const long* ty,
                                  const long* ox, const long* oy, float* sums /*[N][128] zeroed*/, float* out, int N, int H, int W,
                                  void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AugDraws d{bright, contrast, tx, ty, ox, oy};
    ProfScope prof("diffaug_fwd", 0.0, 12.0 * N * H * W, st);
    hipLaunchKernelGGL((aug_sum_kernel<0>), row_grid(N, H), dim3(256), 0, st, x, d, sums, H, W);
    hipLaunchKernelGGL(diffaug_fwd_kernel, row_grid(N, H), dim3(256), 0, st, x, d, (const float*)sums, out, H, W);
    CHECK_LAUNCH("diffaug_fwd");
    return 0;
}

extern "C" int ieagan_diffaug_bwd(const float* gout, const float* contrast, const long* tx, const long* ty, const long* ox,
                                  const long* oy, float* gsums /*[N][128] zeroed*/, float* gx, int N, int H, int W, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AugDraws d{nullptr, contrast, tx, ty, ox, oy};
    ProfScope prof("diffaug_bwd", 0.0, 12.0 * N * H * W, st);
    hipLaunchKernelGGL((aug_sum_kernel<1>), row_grid(N, H), dim3(256), 0, st, gout, d, gsums, H, W);
    hipLaunchKernelGGL(diffaug_bwd_kernel, row_grid(N, H), dim3(256), 0, st, gout, d, (const float*)gsums, gx, H, W);
    CHECK_LAUNCH("diffaug_bwd");
    return 0;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(256) void cr_diffaug_kernel(const float* __restrict__ x, const float* __restrict__ flip_u,
                                                         const long* __restrict__ tx, const long* __restrict__ ty,
                                                         float* __restrict__ out, int H, int W) {
    const int n = blockIdx.y;
    const long HW = (long)H * W;
    const bool flip = flip_u[n] < 0.5f;
    const int sx = (int)tx[n], sy = (int)ty[n];
    const float* xi = x + (long)n * HW;
    float* oi = out + (long)n * HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        const int h = (int)(p / W), w = (int)(p - (long)h * W);
        const int hs = reflect_idx(h + sx, H);
        int ws = reflect_idx(w + sy, W);
        if (flip) ws = W - 1 - ws;
        oi[p] = xi[(long)hs * W + ws];
    }
}

extern "C" int ieagan_cr_diffaug(const float* x, const float* flip_u, const long* tx, const long* ty, float* out, int N, int H,
                                 int W, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("cr_diffaug", 0.0, 8.0 * N * H * W, st);
    hipLaunchKernelGGL(cr_diffaug_kernel, img_grid(N, (long)H * W), dim3(256), 0, st, x, flip_u, tx, ty, out, H, W);
    CHECK_LAUNCH("cr_diffaug");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// flat-arena optimiser kernels (float4 per lane)
// ------------------------------------------------------------------------------------------------
// Hyper-parameters live in device memory so that a captured HIP graph of the train step stays valid
// while the step counter / learning rate / EMA decay change:
//   hp[0] lr, hp[1] beta1, hp[2] beta2, hp[3] eps, hp[4] step (as float), hp[5] grad scale,
//   hp[6] lr / (1 - beta1^step), hp[7] 1 / sqrt(1 - beta2^step)      (written by adam_tick)
__global__ void adam_tick_kernel(float* __restrict__ hp) {
    const float step = hp[4] + 1.f;
    hp[4] = step;
    hp[6] = hp[0] / (1.f - powf(hp[1], step));
    hp[7] = rsqrtf(1.f - powf(hp[2], step));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ hp) {
    const float b1 = hp[1], b2 = hp[2], eps = hp[3], gscale = hp[5], lr_bc1 = hp[6], inv_sqrt_bc2 = hp[7];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

extern "C" int ieagan_adam_step(float* p, const float* g, float* m, float* v, long n, float* hp, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(hp != nullptr, "adam: hyper-parameter block missing");
    ProfScope prof("adam_step", 0.0, 28.0 * n, st);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, hp);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, (const float*)hp);
    CHECK_LAUNCH("adam_step");
    return 0;
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tgt, const float* __restrict__ src, long n,
                                                  const float* __restrict__ decay_ptr) {
    const float decay = decay_ptr[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        tgt[i] = tgt[i] * decay + src[i] * (1.f - decay);
}

extern "C" int ieagan_ema_update(float* tgt, const float* src, long n, const float* decay, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("ema_update", 0.0, 12.0 * n, st);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tgt, src, n, decay);
    CHECK_LAUNCH("ema_update");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Event ingestion (reference utils/dataloader.py:66-77): uint8 sensor images [N, Hin, W] ->
// Pad((0, pad, 0, pad)) -> ToTensor (/255) -> fn_lognorm255 = log(255 t + 1) / log 256 (utils/norm.py:9-20) ->
// UniformNoise(scale) = + scale * u, u ~ U[0,1) given explicitly (utils/noise.py:29-32) -> Normalize(0.5, 0.5)
// = fp32 [N, 1, Hin + 2 pad, W] in [-1, 1].  7.7 MB of uint8 cross PCIe per event instead of 31.5 MB of fp32.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void event_ingest_kernel(const uint8_t* __restrict__ ev, const float* __restrict__ noise,
                                                           float* __restrict__ out, long total, int Hin, int W, int pad, float scale) {
    const int Ho = Hin + 2 * pad;
    const float inv_log256 = 0.18033688011112042f;      // 1 / ln(256)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int y = (int)(t % Ho);
        const long n = t / Ho;
        const int ys = y - pad;
        float v = 0.f;
        if (ys >= 0 && ys < Hin) v = logf((float)ev[(n * Hin + ys) * W + x] + 1.f) * inv_log256;     // 255 * (u/255) + 1 = u + 1
        if (noise) v += scale * noise[i];
        out[i] = (v - 0.5f) / 0.5f;
    }
}

extern "C" int ieagan_event_ingest(const void* ev_u8, const float* noise, float* out, int N, int Hin, int W, int pad, float scale,
                                   void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(ev_u8 && out && N > 0 && Hin > 0 && W > 0 && pad >= 0, "event_ingest: bad arguments");
    const long total = (long)N * (Hin + 2 * pad) * W;
    ProfScope prof("event_ingest", 0.0, (double)N * Hin * W + total * (noise ? 8.0 : 4.0), st);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(event_ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t*)ev_u8, noise, out, total, Hin, W, pad,
                       scale);
    CHECK_LAUNCH("event_ingest");
    return 0;
}
