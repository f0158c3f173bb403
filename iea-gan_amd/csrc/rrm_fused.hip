// Stage-wise fused kernels of the Relational Reasoning Module (reference RRM.py:66-133: pre-LN encoder block over the 40 sensor tokens
// of an event) and of the other small fp32 linear layers around it.  The block runs as
//     [LayerNorm + qkv projection] -> [attention core, small_ops.hip] -> [o projection + residual] -> [LayerNorm + FFN1 + ReLU]
//     -> [FFN2 + residual] -> [final LayerNorm]
// i.e. six launches instead of ~13 library kernels, and backward as eight (slin_bwd: data gradient, weight gradient and bias
// gradient of one linear layer in ONE launch; ln_bwd: LayerNorm backward + residual-path add + d gamma / d beta).
// M = B * S rows (40 ... 160), K, N <= 1536: latency-bound work, so the point is launch count, not FLOP/s -- but the arithmetic is
// exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32 == an ordered fmaf chain), which keeps the 1e-4 parity with the reference.
//
// MFMA 16x16x4 f32 lane map: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[row = 4 * (l >> 4) + r][col = l & 15].
// "float4 trick": a lane loads 4 consecutive k of its row (16 bytes) and feeds element t of both operands to MFMA number t -- slot
// q of MFMA t then stands for k = 16 s + 4 q + t on both sides, so four MFMAs consume one 16-byte load per operand.
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct SlinFwdArgs {
    const float* X;       // [M, K]
    const float* W;       // [N, K]   (nn.Linear layout; for an SN layer the normalised weight of this pass)
    const float* b;       // [N] or NULL
    const float* R;       // [M, N] residual added after bias / ReLU, or NULL
    float* Y;             // [M, N]
    const float* ln_g;    // LayerNorm prologue on the rows of X: gamma / beta [K], or NULL
    const float* ln_b;
    float* xhat;          // [M, K] normalised rows (saved for the backward), with ln_g
    float* rstd;          // [M]
    int M, K, N, relu;
    float eps;
};

__global__ __launch_bounds__(256) void slin_fwd_kernel(SlinFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];       // [16][K16 + 4], then (LayerNorm prologue) gamma [K16] and beta [K16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int M = a.M, K = a.K, N = a.N;
    const int K16 = (K + 15) & ~15, KS = K16 + 4;                    // rows zero-padded to whole 16-wide k groups
    const int m0 = blockIdx.x * 16;
    const int n0 = blockIdx.y * 64 + wave * 16;
    {
        // the 16 x K16 input rows -> LDS in batches of 8 unconditional 16-byte loads per thread (row / column clamped, zeroed by a select
        // afterwards): as a rolled load -> store loop this was one memory round trip per 4 KB, 24 of them in a row at K = 1536
        const int c4n = K16 >> 2, tot = 16 * c4n;
        for (int b0 = 0; b0 < tot; b0 += 8 * 256) {
            f32x4v v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = min(b0 + u * 256 + (int)threadIdx.x, tot - 1);
                const int r = idx / c4n, c4 = idx - r * c4n;
                v[u] = *(const f32x4v*)(a.X + (long)min(m0 + r, M - 1) * K + min(c4 * 4, K - 4));
                if (m0 + r >= M || c4 * 4 >= K) v[u] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = b0 + u * 256 + (int)threadIdx.x;
                if (idx < tot) {
                    const int r = idx / c4n, c4 = idx - r * c4n;
                    *(f32x4v*)(xs + r * KS + c4 * 4) = v[u];
                }
            }
        }
    }
    float* lgs = xs + 16 * KS;                   // gamma / beta of the LayerNorm prologue: staged once (read per row and column below -- as
    float* lbs = lgs + K16;                      // global loads inside the row loop they were a memory round trip per 64 columns and row)
    if (a.ln_g != nullptr) {
        constexpr int LB = 6;                    // K <= 1536: 256 threads x 6
        float rg[LB], rb[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int c = min(u * 256 + (int)threadIdx.x, K - 1);
            rg[u] = a.ln_g[c];
            rb[u] = a.ln_b[c];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int c = u * 256 + (int)threadIdx.x;
            if (c < K) { lgs[c] = rg[u]; lbs[c] = rb[u]; }
        }
        for (int c = LB * 256 + threadIdx.x; c < K; c += 256) { lgs[c] = a.ln_g[c]; lbs[c] = a.ln_b[c]; }      // (K > 1536: not in this network)
    }
    __syncthreads();
    if (a.ln_g != nullptr) {                     // LayerNorm (biased variance, eps inside the root) on rows wave*4 .. wave*4+3
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int row = wave * 4 + rr;
            float s = 0.f;
            for (int c = lane; c < K; c += 64) s += xs[row * KS + c];
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
            for (int c = lane; c < K; c += 64) {
                const float d = xs[row * KS + c] - mean;
                q += d * d;
            }
            const float rstd = rsqrtf(wave_sum(q) / (float)K + a.eps);
            const bool save = blockIdx.y == 0 && m0 + row < M;
            for (int c = lane; c < K; c += 64) {
                const float xh = (xs[row * KS + c] - mean) * rstd;
                if (save) a.xhat[(long)(m0 + row) * K + c] = xh;
                xs[row * KS + c] = xh * lgs[c] + lbs[c];
            }
            if (save && lane == 0) a.rstd[m0 + row] = rstd;
        }
        __syncthreads();
    }
    if (n0 >= N) return;
    const bool nok = n0 + i < N;                 // (a partly filled n-tile: N = 1 for the discriminator's logit layer)
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    // The weight loads are UNCONDITIONAL: a column beyond N reads row 0 (its accumulator column is never stored), a k group beyond K
    // re-reads the row's last group (its x operand is the zero padding of xs).  Under the old per-lane conditions the compiler
    // waited for every load before the next one was issued: K / 16 dependent L2 round trips per wave, 19 us for a 40 x 1536 x 1536
    // layer whose weights stream in 1 us.
    const float* wbase = a.W + (long)(nok ? n0 + i : 0) * K;
    const float* xrow = xs + i * KS + 4 * kq;
    const int nsteps = K16 >> 4;
    auto mac = [&](int s, const f32x4v& wb) {
        const f32x4v xa = *(const f32x4v*)(xrow + 16 * s);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], wb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1], wb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[2], wb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[3], wb[3], acc, 0, 0, 0);
    };
    int s = 0;
    for (; s + 8 <= nsteps; s += 8) {            // eight weight groups requested together (the pragma alone left the loop rolled)
        f32x4v wb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wb[u] = *(const f32x4v*)(wbase + min(16 * (s + u) + 4 * kq, K - 4));
        __builtin_amdgcn_sched_barrier(0);       // all eight requests first (the scheduler otherwise pairs each load with its MFMAs)
#pragma unroll
        for (int u = 0; u < 8; ++u) mac(s + u, wb[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; s < nsteps; ++s) mac(s, *(const f32x4v*)(wbase + min(16 * s + 4 * kq, K - 4)));
    const int n = n0 + i;
    if (!nok) return;
    const float bias = a.b != nullptr ? a.b[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * kq + r;
        if (m >= M) continue;
        float v = acc[r] + bias;
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.R != nullptr) v += a.R[(long)m * N + n];
        a.Y[(long)m * N + n] = v;
    }
}

struct SlinBwdArgs {
    const float* dY;      // [M, N]
    const float* Ymask;   // [M, N] post-ReLU forward output (dY is masked where it is <= 0) or NULL
    const float* Xn;      // [M, K] the GEMM input; NULL: xhat * ln_g + ln_b
    const float* xhat;
    const float* ln_g;
    const float* ln_b;
    const float* W;       // [N, K]
    float* dX;            // [M, K] gradient w.r.t. the GEMM input (= w.r.t. the LayerNorm OUTPUT when the prologue was a LayerNorm), or NULL
    float* dW;            // [N, K] (assigned) or NULL
    float* db;            // [N] (assigned) or NULL
    int M, K, N;
    int nbx;              // blocks [0, nbx) compute dX tiles, the others dW (+ db) tiles
    int ktiles;           // 64-wide k tiles of a dX row block
    int wktiles;          // 64-wide k tiles of a dW row block (1 when only db is wanted)
    int nsplit;           // > 1: the reduction over n of a dX tile is cut into nsplit ranges of nchunk (multiple of 16) columns, partial
    int nchunk;           //      tiles are ADDED to the caller-zeroed dX with float atomics (N = 24576 of G.linear: 48 x 512)
};

__device__ __forceinline__ float slin_dy(const SlinBwdArgs& a, int m, int n) {
    if (m >= a.M) return 0.f;
    const float g = a.dY[(long)m * a.N + n];
    return (a.Ymask != nullptr && !(a.Ymask[(long)m * a.N + n] > 0.f)) ? 0.f : g;
}

__device__ __forceinline__ float slin_xn(const SlinBwdArgs& a, int m, int k) {
    if (m >= a.M) return 0.f;
    if (a.Xn != nullptr) return a.Xn[(long)m * a.K + k];
    return a.xhat[(long)m * a.K + k] * a.ln_g[k] + a.ln_b[k];
}

__global__ __launch_bounds__(256) void slin_bwd_kernel(SlinBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int M = a.M, K = a.K, N = a.N;
    if ((int)blockIdx.x < a.nbx) {
        // ---- dX[m0 .. m0+16, k0 .. k0+16) = dY W: reduction over n.  A: lane (row i, slot q) holds dY[m0+i][16s + 4q + t];
        //      B: lane (col i, slot q) holds W[16s + 4q + t][k0 + i]
        const int tile = blockIdx.x / a.nsplit, ns = blockIdx.x - tile * a.nsplit;
        const int mt = tile / a.ktiles, kt = tile - mt * a.ktiles;
        const int m0 = mt * 16, k0 = kt * 64 + wave * 16;
        if (k0 >= K) return;
        const bool mok = m0 + i < M, kok = k0 + i < K;
        const int nb = ns * a.nchunk, ne = min(nb + a.nchunk, N);
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        if ((N & 15) == 0) {
            const float* dyrow = a.dY + (long)(mok ? m0 + i : 0) * N + 4 * q;
            const float* ymrow = a.Ymask != nullptr ? a.Ymask + (long)(mok ? m0 + i : 0) * N + 4 * q : nullptr;
            // every load of a step is unconditional (rows / columns beyond the matrix read row 0 / column 0 and are zeroed by a
            // select), the ReLU-mask branch is taken outside the loop and the operands of UB steps are requested before the first
            // MFMA: N / 16 dependent round trips per wave otherwise
            constexpr int UB = 4;
            auto batch = [&](int s0, int cnt, bool masked) {       // cnt <= UB steps starting at s0
                f32x4v ga[UB], ym[UB];
                float wv[UB][4];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int s = s0 + (u < cnt ? u : 0);
                    ga[u] = *(const f32x4v*)(dyrow + 16 * s);
                    if (masked) ym[u] = *(const f32x4v*)(ymrow + 16 * s);
                    const float* wp = a.W + (long)(16 * s + 4 * q) * K + (kok ? k0 + i : 0);
                    wv[u][0] = wp[0]; wv[u][1] = wp[K]; wv[u][2] = wp[2 * (long)K]; wv[u][3] = wp[3 * (long)K];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const bool live = mok && u < cnt;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float g = ga[u][t];
                        if (masked) g = ym[u][t] > 0.f ? g : 0.f;
                        if (!live) g = 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(g, kok ? wv[u][t] : 0.f, acc, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            const int s1 = ne >> 4;
            int s = nb >> 4;
            if (ymrow != nullptr) {
                for (; s + UB <= s1; s += UB) batch(s, UB, true);
                if (s < s1) batch(s, s1 - s, true);
            } else {
                for (; s + UB <= s1; s += UB) batch(s, UB, false);
                if (s < s1) batch(s, s1 - s, false);
            }
        } else {                                  // ragged N (N = 1: the logit layer): one guarded column per MFMA slot
            for (int s = 0; s < ((N + 3) >> 2); ++s) {
                const int nn = 4 * s + q;
                const float ga = slin_dy(a, mok ? m0 + i : M, nn < N ? nn : 0) * (nn < N ? 1.f : 0.f);
                const float wv = (nn < N && kok) ? a.W[(long)nn * K + k0 + i] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, wv, acc, 0, 0, 0);
            }
        }
        if (!kok) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * q + r;
            if (m >= M) continue;
            if (a.nsplit > 1) atomicAdd(a.dX + (long)m * K + k0 + i, acc[r]);
            else a.dX[(long)m * K + k0 + i] = acc[r];
        }
        return;
    }
    // ---- dW[n0 .. n0+16, k0 .. k0+64) = dY^T Xn: reduction over m.  A: lane (row i = n, slot q) holds dY[4t + q][n0 + i];
    //      B: lane (col i = k, slot q) holds Xn[4t + q][k0 + 16 jj + i].  db[n] = sum_m dY[m][n] rides on the k0 == 0 tiles.
    const int bw = blockIdx.x - a.nbx;
    const int nt = bw / a.wktiles, kt = bw - nt * a.wktiles;
    const int n0 = nt * 64 + wave * 16, k0 = kt * 64;
    if (n0 >= N) return;
    const bool nok = n0 + i < N;
    f32x4v acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[jj] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    {
        // TB reduction steps (4 rows each) per batch, every load unconditional (row / column clamped, zeroed by selects): the rolled loop was one
        // memory round trip per step, M / 4 in a row
        constexpr int TB = 4;
        const int nsteps = (M + 3) >> 2;
        const bool masked = a.Ymask != nullptr, plain = a.Xn != nullptr, want_w = a.dW != nullptr;
        const float* xsrc = plain ? a.Xn : a.xhat;
        const int ncol = nok ? n0 + i : 0;
        int kc[4];
        float lg[4], lb[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            kc[jj] = min(k0 + 16 * jj + i, K - 1);
            lg[jj] = (!plain && want_w) ? a.ln_g[kc[jj]] : 1.f;
            lb[jj] = (!plain && want_w) ? a.ln_b[kc[jj]] : 0.f;
        }
        for (int t0 = 0; t0 < nsteps; t0 += TB) {
            float gv[TB], yv[TB], xv[TB][4];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const int mc = min(4 * (t0 + u) + q, M - 1);
                gv[u] = a.dY[(long)mc * N + ncol];
                yv[u] = masked ? a.Ymask[(long)mc * N + ncol] : 1.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) xv[u][jj] = want_w ? xsrc[(long)mc * K + kc[jj]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                if (t0 + u >= nsteps) break;            // wave-uniform
                const int m = 4 * (t0 + u) + q;
                float ga = (nok && m < M) ? gv[u] : 0.f;
                if (masked && !(yv[u] > 0.f)) ga = 0.f;
                bsum += ga;
                if (!want_w) continue;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = k0 + 16 * jj + i;
                    float xb = plain ? xv[u][jj] : xv[u][jj] * lg[jj] + lb[jj];
                    if (k >= K || m >= M) xb = 0.f;
                    acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, xb, acc[jj], 0, 0, 0);
                }
            }
        }
    }
    if (a.dW != nullptr) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 4 * q + r, k = k0 + 16 * jj + i;
                if (k < K && n < N) a.dW[(long)n * K + k] = acc[jj][r];
            }
    }
    if (a.db != nullptr && k0 == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (q == 0 && nok) a.db[n0 + i] = bsum;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward (stand-alone: the final norm of the module) and backward; one block per row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float blk_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = red[0] + red[1] + red[2] + red[3];
    return s;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ X, const float* __restrict__ g, const float* __restrict__ b,
                                                      float* __restrict__ Y, float* __restrict__ xhat, float* __restrict__ rstd_out, int K, float eps,
                                                      int l2norm) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const float* x = X + (long)m * K;
    float s = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) s += x[c];
    const float mean = blk_sum(s, red) / (float)K;
    float q = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) {
        const float d = x[c] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(blk_sum(q, red) / (float)K + eps);
    float nn = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) {
        const float xh = (x[c] - mean) * rstd;
        const float y = xh * g[c] + b[c];
        xhat[(long)m * K + c] = xh;
        if (!l2norm) Y[(long)m * K + c] = y;
        nn += y * y;
    }
    if (threadIdx.x == 0) rstd_out[m] = rstd;
    if (l2norm) {                                 // F.normalize(., dim = 1) of the LayerNorm output (model.py:920-935), eps 1e-12
        const float inv = 1.f / fmaxf(sqrtf(blk_sum(nn, red)), 1e-12f);
        for (int c = threadIdx.x; c < K; c += 256) Y[(long)m * K + c] = (xhat[(long)m * K + c] * g[c] + b[c]) * inv;
    }
}

// dX = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)) + dRes,  dxh = dY * gamma;   dgamma += dY * xhat, dbeta += dY  (atomics over rows)
// beta != NULL: the forward ended with y = u / max(|u|, 1e-12), u = xhat * gamma + beta (F.normalize of the LayerNorm output): dY is then the
// gradient w.r.t. y and is first mapped to du = (dY - y <dY, y>) / |u|.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                      const float* __restrict__ g, const float* __restrict__ beta, const float* __restrict__ dRes,
                                                      float* __restrict__ dX, float* __restrict__ dg, float* __restrict__ dbeta, int K) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const float* dy = dY + (long)m * K;
    const float* xh = xhat + (long)m * K;
    float inv = 1.f, proj = 0.f;                  // du = (dy - u * proj) * inv
    if (beta != nullptr) {
        float uu = 0.f, du = 0.f;
        for (int c = threadIdx.x; c < K; c += 256) {
            const float u = xh[c] * g[c] + beta[c];
            uu += u * u;
            du += dy[c] * u;
        }
        uu = blk_sum(uu, red);
        du = blk_sum(du, red);
        const float nrm = fmaxf(sqrtf(uu), 1e-12f);
        inv = 1.f / nrm;
        proj = du / (nrm * nrm);
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) {
        const float dyc = beta != nullptr ? (dy[c] - (xh[c] * g[c] + beta[c]) * proj) * inv : dy[c];
        const float d = dyc * g[c];
        s1 += d;
        s2 += d * xh[c];
    }
    s1 = blk_sum(s1, red) / (float)K;
    s2 = blk_sum(s2, red) / (float)K;
    const float r = rstd[m];
    for (int c = threadIdx.x; c < K; c += 256) {
        const float dyc = beta != nullptr ? (dy[c] - (xh[c] * g[c] + beta[c]) * proj) * inv : dy[c];
        const float d = dyc * g[c];
        float v = r * (d - s1 - xh[c] * s2);
        if (dRes != nullptr) v += dRes[(long)m * K + c];
        dX[(long)m * K + c] = v;
        if (dg != nullptr) {
            atomicAdd(dg + c, dyc * xh[c]);
            atomicAdd(dbeta + c, dyc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
extern "C" int ieagan_slin_fwd(const float* X, const float* W, const float* b, const float* R, float* Y, const float* ln_g, const float* ln_b,
                               float* xhat, float* rstd, int M, int K, int N, int relu, float eps, void* stream) {
    CHECK_ARG(X != nullptr && W != nullptr && Y != nullptr, "slin_fwd: null pointer");
    CHECK_ARG(M >= 1 && K >= 4 && K % 4 == 0 && N >= 1 && K <= 2048, "slin_fwd: M=%d K=%d N=%d (K a multiple of 4, <= 2048)", M, K, N);
    CHECK_ARG((ln_g == nullptr) == (ln_b == nullptr) && (ln_g == nullptr || (xhat != nullptr && rstd != nullptr)), "slin_fwd: LayerNorm prologue needs gamma, beta, xhat, rstd");
    SlinFwdArgs a{X, W, b, R, Y, ln_g, ln_b, xhat, rstd, M, K, N, relu, eps};
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("slin_fwd", 2.0 * M * (double)K * N, 4.0 * ((double)M * K + (double)N * K + (double)M * N), st);
    const size_t lds = (size_t)16 * (((K + 15) & ~15) + 4) * 4 + (ln_g != nullptr ? (size_t)2 * ((K + 15) & ~15) * 4 : 0);
    hipLaunchKernelGGL(slin_fwd_kernel, dim3((M + 15) / 16, (N + 63) / 64), dim3(256), lds, st, a);
    CHECK_LAUNCH("slin_fwd");
    return 0;
}

extern "C" int ieagan_slin_bwd(const float* dY, const float* Ymask, const float* Xn, const float* xhat, const float* ln_g, const float* ln_b,
                               const float* W, float* dX, float* dW, float* db, int M, int K, int N, int dx_zeroed, void* stream) {
    CHECK_ARG(dY != nullptr && W != nullptr, "slin_bwd: null pointer");
    CHECK_ARG(M >= 1 && K >= 1 && N >= 1, "slin_bwd: M=%d K=%d N=%d", M, K, N);
    CHECK_ARG(Xn != nullptr || (xhat != nullptr && ln_g != nullptr && ln_b != nullptr) || (dW == nullptr), "slin_bwd: the weight gradient needs the GEMM input");
    CHECK_ARG(dX != nullptr || dW != nullptr || db != nullptr, "slin_bwd: nothing to compute");
    const int ktiles = (K + 63) / 64, mtiles = (M + 15) / 16;
    const int wktiles = dW != nullptr ? ktiles : 1;
    // long reductions over n (G.linear: N = 24576; the stacked ccbn linears: 12096) are cut into ranges of 512 columns whose partial
    // dX tiles are added with float atomics -- the caller passes a ZEROED dX then (dx_zeroed)
    int nsplit = 1, nchunk = (N + 15) & ~15;
    if (dX != nullptr && dx_zeroed && N % 16 == 0 && N >= 2048) {
        nchunk = 512;
        nsplit = (N + nchunk - 1) / nchunk;
    }
    SlinBwdArgs a{dY, Ymask, Xn, xhat, ln_g, ln_b, W, dX, dW, db, M, K, N, dX != nullptr ? mtiles * ktiles * nsplit : 0, ktiles, wktiles, nsplit, nchunk};
    const int nbw = (dW != nullptr || db != nullptr) ? ((N + 63) / 64) * wktiles : 0;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("slin_bwd", 2.0 * M * (double)K * N * ((dX ? 1 : 0) + (dW ? 1 : 0)), 4.0 * ((double)M * K + 2.0 * N * K + (double)M * N), st);
    hipLaunchKernelGGL(slin_bwd_kernel, dim3(a.nbx + nbw), dim3(256), 0, st, a);
    CHECK_LAUNCH("slin_bwd");
    return 0;
}

extern "C" int ieagan_ln_fwd(const float* X, const float* g, const float* b, float* Y, float* xhat, float* rstd, int M, int K, float eps, int l2norm,
                             void* stream) {
    CHECK_ARG(X != nullptr && g != nullptr && b != nullptr && Y != nullptr && xhat != nullptr && rstd != nullptr && M >= 1 && K >= 1, "ln_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("ln_fwd", 0.0, 12.0 * M * K, st);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(M), dim3(256), 0, st, X, g, b, Y, xhat, rstd, K, eps, l2norm);
    CHECK_LAUNCH("ln_fwd");
    return 0;
}

extern "C" int ieagan_ln_bwd(const float* dY, const float* xhat, const float* rstd, const float* g, const float* l2_beta, const float* dRes, float* dX,
                             float* dg, float* dbeta, int M, int K, void* stream) {
    CHECK_ARG(dY != nullptr && xhat != nullptr && rstd != nullptr && g != nullptr && dX != nullptr && M >= 1 && K >= 1, "ln_bwd: bad arguments");
    CHECK_ARG((dg == nullptr) == (dbeta == nullptr), "ln_bwd: d gamma / d beta come together");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("ln_bwd", 0.0, 16.0 * M * K, st);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(M), dim3(256), 0, st, dY, xhat, rstd, g, l2_beta, dRes, dX, dg, dbeta, K);
    CHECK_LAUNCH("ln_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Class proxies of the discriminator head: F.normalize(F.embedding(y, W / sigma), dim = 1) (reference model.py:916, 933); one block per row.
// Backward: dW[y[m]] += (dp - p <dp, p>) / |w|, scattered with float atomics into the caller-zeroed [classes, D] gradient.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_norm_fwd_kernel(const long* __restrict__ y, const float* __restrict__ Wn, float* __restrict__ out,
                                                              float* __restrict__ inv_out, int D) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    const float* row = Wn + y[m] * (long)D;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) s += row[c] * row[c];
    const float inv = 1.f / fmaxf(sqrtf(blk_sum(s, red)), 1e-12f);
    for (int c = threadIdx.x; c < D; c += 256) out[(long)m * D + c] = row[c] * inv;
    if (threadIdx.x == 0) inv_out[m] = inv;
}

__global__ __launch_bounds__(256) void embed_norm_bwd_kernel(const long* __restrict__ y, const float* __restrict__ p, const float* __restrict__ inv,
                                                              const float* __restrict__ dp, float* __restrict__ dW, int D) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) s += dp[(long)m * D + c] * p[(long)m * D + c];
    const float dot = blk_sum(s, red);
    float* row = dW + y[m] * (long)D;
    for (int c = threadIdx.x; c < D; c += 256) atomicAdd(row + c, (dp[(long)m * D + c] - p[(long)m * D + c] * dot) * inv[m]);
}

extern "C" int ieagan_embed_norm_fwd(const long* y, const float* Wn, float* out, float* inv, int M, int D, void* stream) {
    CHECK_ARG(y != nullptr && Wn != nullptr && out != nullptr && inv != nullptr && M >= 1 && D >= 1, "embed_norm_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("embed_norm_fwd", 0.0, 8.0 * M * D, st);
    hipLaunchKernelGGL(embed_norm_fwd_kernel, dim3(M), dim3(256), 0, st, y, Wn, out, inv, D);
    CHECK_LAUNCH("embed_norm_fwd");
    return 0;
}

extern "C" int ieagan_embed_norm_bwd(const long* y, const float* p, const float* inv, const float* dp, float* dW, int M, int D, void* stream) {
    CHECK_ARG(y != nullptr && p != nullptr && inv != nullptr && dp != nullptr && dW != nullptr && M >= 1 && D >= 1, "embed_norm_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("embed_norm_bwd", 0.0, 12.0 * M * D, st);
    hipLaunchKernelGGL(embed_norm_bwd_kernel, dim3(M), dim3(256), 0, st, y, p, inv, dp, dW, D);
    CHECK_LAUNCH("embed_norm_bwd");
    return 0;
}
