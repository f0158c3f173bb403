// Small single-workgroup kernels of the train step (fp32; no MFMA -- 40 tokens are too few):
//   rrm_attention   per (batch, head): softmax(q k^T / sqrt(hd)) v with the S x S affinity held in LDS
//                   (RRM.py:10-16, 46-58).  Consumes the packed qkv projection [B,S,H,3*hd] directly and
//                   writes [B,S,H*hd] for o_proj; backward returns d(qkv).
//   loss_block      every loss of one phase on the [n] logits and [n, d] unit-sphere embeddings in ONE launch,
//                   value AND gradient: hinge (dis / gen), 2C contrastive, uniformity, IEA (loss.py:8-44, 79-132).
//                   The three embedding losses share one n x n Gram matrix, kept in LDS.
//   relu_sum_pool   D head: sum_{h,w} relu(x) of a bf16 NHWC map -> [N, C] fp32, and its backward (model.py:912)
#include "common.h"

#define SMAX 64          // max tokens / samples handled by these kernels

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {   // blockDim.x <= 1024
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) s += red[i];
    __syncthreads();
    return s;
}

// ------------------------------------------------------------------------------------------------
// RRM attention core.  grid (H, B), 256 threads.  LDS: q,k,v [S][hd] + att [S][S].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rrm_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ att_out,
                                                           int S, int Hh, int hd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = hd + 1;                          // padded LDS row: consecutive tokens hit different banks
    float* q = sm;
    float* k = q + S * LD;
    float* v = k + S * LD;
    float* att = v + S * LD;                       // [S][S]
    const int h = blockIdx.x, b = blockIdx.y;
    const int E3 = Hh * 3 * hd;
    const float* base = qkv + (long)b * S * E3 + h * 3 * hd;
    // q, k, v of this head -> LDS.  (hd % 4 == 0: 16-byte loads, four per operand and thread in flight -- as a rolled scalar loop the 4 blocks
    // of a launch spent 20 serial memory round trips here, two thirds of the kernel)
    if ((hd & 3) == 0) {
        const int hd4 = hd >> 2, tot = S * hd4;
        for (int b0 = 0; b0 < tot; b0 += 4 * 256) {
            f32x4 rq[4], rk[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(b0 + u * 256 + (int)threadIdx.x, tot - 1);
                const int s = idx / hd4, d4 = idx - s * hd4;
                const float* p = base + (long)s * E3 + d4 * 4;
                rq[u] = *(const f32x4*)p;
                rk[u] = *(const f32x4*)(p + hd);
                rv[u] = *(const f32x4*)(p + 2 * hd);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = b0 + u * 256 + (int)threadIdx.x;
                if (idx >= tot) continue;
                const int s = idx / hd4, d4 = idx - s * hd4;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    q[s * LD + d4 * 4 + c] = rq[u][c];
                    k[s * LD + d4 * 4 + c] = rk[u][c];
                    v[s * LD + d4 * 4 + c] = rv[u][c];
                }
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < S * hd; idx += 256) {
            const int s = idx / hd, d = idx - s * hd;
            q[s * LD + d] = base[(long)s * E3 + d];
            k[s * LD + d] = base[(long)s * E3 + hd + d];
            v[s * LD + d] = base[(long)s * E3 + 2 * hd + d];
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    for (int idx = threadIdx.x; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx - i * S;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a += q[i * LD + d] * k[j * LD + d];
        att[idx] = a * scale;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += 256) {   // row softmax (S <= 64 rows)
        float m = -1e30f;
        for (int j = 0; j < S; ++j) m = fmaxf(m, att[i * S + j]);
        float l = 0.f;
        for (int j = 0; j < S; ++j) {
            const float e = __expf(att[i * S + j] - m);
            att[i * S + j] = e;
            l += e;
        }
        const float il = 1.f / l;
        for (int j = 0; j < S; ++j) att[i * S + j] *= il;
    }
    __syncthreads();
    float* ao = att_out + ((long)b * Hh + h) * S * S;
    for (int idx = threadIdx.x; idx < S * S; idx += 256) ao[idx] = att[idx];
    const int E = Hh * hd;
    for (int idx = threadIdx.x; idx < S * hd; idx += 256) {
        const int i = idx / hd, d = idx - i * hd;
        float a = 0.f;
        for (int j = 0; j < S; ++j) a += att[i * S + j] * v[j * LD + d];
        out[((long)b * S + i) * E + h * hd + d] = a;
    }
}

__global__ __launch_bounds__(256) void rrm_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ att_in,
                                                           const float* __restrict__ dout, float* __restrict__ dqkv, int S, int Hh, int hd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = hd + 1;
    float* q = sm;
    float* k = q + S * LD;
    float* v = k + S * LD;
    float* go = v + S * LD;                        // dout of this head [S][hd]
    float* att = go + S * LD;                      // [S][S]
    float* ds = att + S * S;                       // d score [S][S]
    const int h = blockIdx.x, b = blockIdx.y;
    const int E3 = Hh * 3 * hd, E = Hh * hd;
    const float* base = qkv + (long)b * S * E3 + h * 3 * hd;
    if ((hd & 3) == 0) {                            // 16-byte loads, a batch of them in flight (see the forward kernel)
        const int hd4 = hd >> 2, tot = S * hd4;
        for (int b0 = 0; b0 < tot; b0 += 4 * 256) {
            f32x4 rq[4], rk[4], rv[4], rg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(b0 + u * 256 + (int)threadIdx.x, tot - 1);
                const int s = idx / hd4, d4 = idx - s * hd4;
                const float* p = base + (long)s * E3 + d4 * 4;
                rq[u] = *(const f32x4*)p;
                rk[u] = *(const f32x4*)(p + hd);
                rv[u] = *(const f32x4*)(p + 2 * hd);
                rg[u] = *(const f32x4*)(dout + ((long)b * S + s) * E + h * hd + d4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = b0 + u * 256 + (int)threadIdx.x;
                if (idx >= tot) continue;
                const int s = idx / hd4, d4 = idx - s * hd4;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    q[s * LD + d4 * 4 + c] = rq[u][c];
                    k[s * LD + d4 * 4 + c] = rk[u][c];
                    v[s * LD + d4 * 4 + c] = rv[u][c];
                    go[s * LD + d4 * 4 + c] = rg[u][c];
                }
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < S * hd; idx += 256) {
            const int s = idx / hd, d = idx - s * hd;
            q[s * LD + d] = base[(long)s * E3 + d];
            k[s * LD + d] = base[(long)s * E3 + hd + d];
            v[s * LD + d] = base[(long)s * E3 + 2 * hd + d];
            go[s * LD + d] = dout[((long)b * S + s) * E + h * hd + d];
        }
    }
    const float* ai = att_in + ((long)b * Hh + h) * S * S;
    {
        const int tot = S * S;                      // (S <= 64: at most 16 values per thread, all requested before the first store)
        float ra[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) ra[u] = ai[min(u * 256 + (int)threadIdx.x, tot - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u * 256 + (int)threadIdx.x < tot) att[u * 256 + threadIdx.x] = ra[u];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < S * S; idx += 256) {    // d att_ij = dout_i . v_j
        const int i = idx / S, j = idx - i * S;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a += go[i * LD + d] * v[j * LD + d];
        ds[idx] = a;
    }
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    for (int i = threadIdx.x; i < S; i += 256) {               // softmax backward per row, folded with 1/sqrt(hd)
        float dot = 0.f;
        for (int j = 0; j < S; ++j) dot += att[i * S + j] * ds[i * S + j];
        for (int j = 0; j < S; ++j) ds[i * S + j] = att[i * S + j] * (ds[i * S + j] - dot) * scale;
    }
    __syncthreads();
    float* ob = dqkv + (long)b * S * E3 + h * 3 * hd;
    for (int idx = threadIdx.x; idx < S * hd; idx += 256) {
        const int i = idx / hd, d = idx - i * hd;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < S; ++j) {
            dq += ds[i * S + j] * k[j * LD + d];
            dk += ds[j * S + i] * q[j * LD + d];
            dv += att[j * S + i] * go[j * LD + d];
        }
        ob[(long)i * E3 + d] = dq;
        ob[(long)i * E3 + hd + d] = dk;
        ob[(long)i * E3 + 2 * hd + d] = dv;
    }
}

// ------------------------------------------------------------------------------------------------
// The same two kernels for hd % 4 == 0 (every head of the path: 64 / 128): LDS rows padded to hd + 4 floats, so a row is 16-byte aligned and
// 16 consecutive rows start on 16 different 4-bank groups -- every inner-product step is two ds_read_b128 per four FMAs instead of eight
// ds_read_b32; the row softmax runs one wave per row (lanes = columns).  The scalar kernels above spent their time in LDS reads: 26.6 us a
// launch for 40 tokens (forward), five launches per step on the critical path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
// inner product of two [hd] LDS rows: eight 16-byte reads requested per step, four independent partial sums (one running sum made every
// step wait for its own LDS read: ~130 cycles x hd / 4 steps x 7 outputs per thread)
__device__ __forceinline__ float row_dot(const float* a, const float* b, int hd4) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int d4 = 0;
    for (; d4 + 4 <= hd4; d4 += 4) {
        const f32x4 a0 = *(const f32x4*)(a + 4 * d4), a1 = *(const f32x4*)(a + 4 * d4 + 4), a2 = *(const f32x4*)(a + 4 * d4 + 8),
                    a3 = *(const f32x4*)(a + 4 * d4 + 12);
        const f32x4 b0 = *(const f32x4*)(b + 4 * d4), b1 = *(const f32x4*)(b + 4 * d4 + 4), b2 = *(const f32x4*)(b + 4 * d4 + 8),
                    b3 = *(const f32x4*)(b + 4 * d4 + 12);
        s0 += dot4(a0, b0);
        s1 += dot4(a1, b1);
        s2 += dot4(a2, b2);
        s3 += dot4(a3, b3);
    }
    for (; d4 < hd4; ++d4) s0 += dot4(*(const f32x4*)(a + 4 * d4), *(const f32x4*)(b + 4 * d4));
    return (s0 + s1) + (s2 + s3);
}

// rows of [S][hd] operands of this head -> LDS (NOPS operands, 16-byte loads, all of a batch in flight)
template <int NOPS>
__device__ __forceinline__ void rrm_stage(const float* const (&src)[NOPS], const long (&rstride)[NOPS], float* const (&dst)[NOPS], int S, int hd, int LD) {
    const int hd4 = hd >> 2, tot = S * hd4;
    for (int b0 = 0; b0 < tot; b0 += 4 * 256) {
        f32x4 r[4][NOPS];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = min(b0 + u * 256 + (int)threadIdx.x, tot - 1);
            const int s = idx / hd4, d4 = idx - s * hd4;
#pragma unroll
            for (int o = 0; o < NOPS; ++o) r[u][o] = *(const f32x4*)(src[o] + (long)s * rstride[o] + d4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = b0 + u * 256 + (int)threadIdx.x;
            if (idx >= tot) continue;
            const int s = idx / hd4, d4 = idx - s * hd4;
#pragma unroll
            for (int o = 0; o < NOPS; ++o) *(f32x4*)(dst[o] + s * LD + d4 * 4) = r[u][o];
        }
    }
}

__global__ __launch_bounds__(256) void rrm_attn_fwd4_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ att_out,
                                                            int S, int Hh, int hd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = hd + 4, hd4 = hd >> 2;
    float* q = sm;
    float* k = q + S * LD;
    float* v = k + S * LD;
    float* att = v + S * LD;                       // [S][S]
    const int h = blockIdx.x, b = blockIdx.y;
    const int E3 = Hh * 3 * hd, E = Hh * hd;
    const float* base = qkv + (long)b * S * E3 + h * 3 * hd;
    {
        const float* const src[3] = {base, base + hd, base + 2 * hd};
        const long rs[3] = {E3, E3, E3};
        float* const dst[3] = {q, k, v};
        rrm_stage<3>(src, rs, dst, S, hd, LD);
    }
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    for (int idx = threadIdx.x; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx - i * S;
        att[idx] = row_dot(q + i * LD, k + j * LD, hd4) * scale;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* ao = att_out + ((long)b * Hh + h) * S * S;
    for (int i = wave; i < S; i += 4) {            // row softmax: one wave per row, lane = column (S <= 64)
        const float x = lane < S ? att[i * S + lane] : -1e30f;
        const float m = wave_max(x);
        const float e = lane < S ? __expf(x - m) : 0.f;
        const float p = e / wave_sum(e);
        if (lane < S) {
            att[i * S + lane] = p;
            ao[i * S + lane] = p;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < S * hd4; idx += 256) {
        const int i = idx / hd4, d4 = idx - i * hd4;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        int j = 0;
        for (; j + 4 <= S; j += 4) {                // four rows requested per step, two partial sums
            const f32x4 v0 = *(const f32x4*)(v + j * LD + 4 * d4), v1 = *(const f32x4*)(v + (j + 1) * LD + 4 * d4),
                        v2 = *(const f32x4*)(v + (j + 2) * LD + 4 * d4), v3 = *(const f32x4*)(v + (j + 3) * LD + 4 * d4);
            const float p0 = att[i * S + j], p1 = att[i * S + j + 1], p2 = att[i * S + j + 2], p3 = att[i * S + j + 3];
            a0 += p0 * v0 + p2 * v2;
            a1 += p1 * v1 + p3 * v3;
        }
        for (; j < S; ++j) a0 += att[i * S + j] * *(const f32x4*)(v + j * LD + 4 * d4);
        *(f32x4*)(out + ((long)b * S + i) * E + h * hd + 4 * d4) = a0 + a1;
    }
}

__global__ __launch_bounds__(256) void rrm_attn_bwd4_kernel(const float* __restrict__ qkv, const float* __restrict__ att_in,
                                                            const float* __restrict__ dout, float* __restrict__ dqkv, int S, int Hh, int hd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = hd + 4, hd4 = hd >> 2;
    float* q = sm;
    float* k = q + S * LD;
    float* v = k + S * LD;
    float* go = v + S * LD;                        // dout of this head [S][hd]
    float* att = go + S * LD;                      // [S][S]
    float* ds = att + S * S;                       // d score [S][S]
    const int h = blockIdx.x, b = blockIdx.y;
    const int E3 = Hh * 3 * hd, E = Hh * hd;
    const float* base = qkv + (long)b * S * E3 + h * 3 * hd;
    {
        const float* const src[4] = {base, base + hd, base + 2 * hd, dout + (long)b * S * E + h * hd};
        const long rs[4] = {E3, E3, E3, E};
        float* const dst[4] = {q, k, v, go};
        rrm_stage<4>(src, rs, dst, S, hd, LD);
    }
    const float* ai = att_in + ((long)b * Hh + h) * S * S;
    {
        const int tot = S * S;
        float ra[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) ra[u] = ai[min(u * 256 + (int)threadIdx.x, tot - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u * 256 + (int)threadIdx.x < tot) att[u * 256 + threadIdx.x] = ra[u];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < S * S; idx += 256) {    // d att_ij = dout_i . v_j
        const int i = idx / S, j = idx - i * S;
        ds[idx] = row_dot(go + i * LD, v + j * LD, hd4);
    }
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < S; i += 4) {                       // softmax backward per row (one wave per row), folded with 1/sqrt(hd)
        const float a = lane < S ? att[i * S + lane] : 0.f, d = lane < S ? ds[i * S + lane] : 0.f;
        const float dot = wave_sum(a * d);
        if (lane < S) ds[i * S + lane] = a * (d - dot) * scale;
    }
    __syncthreads();
    float* ob = dqkv + (long)b * S * E3 + h * 3 * hd;
    for (int idx = threadIdx.x; idx < S * hd4; idx += 256) {
        const int i = idx / hd4, d4 = idx - i * hd4;
        f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dk = dq, dv = dq;
#pragma unroll 4
        for (int j = 0; j < S; ++j) {               // (three independent sums per step; unrolled: four steps of reads in flight)
            dq += ds[i * S + j] * *(const f32x4*)(k + j * LD + 4 * d4);
            dk += ds[j * S + i] * *(const f32x4*)(q + j * LD + 4 * d4);
            dv += att[j * S + i] * *(const f32x4*)(go + j * LD + 4 * d4);
        }
        *(f32x4*)(ob + (long)i * E3 + 4 * d4) = dq;
        *(f32x4*)(ob + (long)i * E3 + hd + 4 * d4) = dk;
        *(f32x4*)(ob + (long)i * E3 + 2 * hd + 4 * d4) = dv;
    }
}

extern "C" int ieagan_rrm_attention_fwd(const float* qkv, float* out, float* att, int B, int S, int H, int hd, void* stream) {
    CHECK_ARG(S >= 1 && S <= SMAX && hd >= 1, "rrm_attention: S must be <= %d", SMAX);
    const bool v4 = (hd & 3) == 0;
    const size_t lds = (size_t)(3 * S * (hd + (v4 ? 4 : 1)) + S * S) * 4;
    CHECK_ARG(lds <= 150 * 1024, "rrm_attention: head does not fit LDS");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("rrm_attention_fwd", 4.0 * B * H * S * S * hd, 0.0, st);
    if (v4) hipLaunchKernelGGL(rrm_attn_fwd4_kernel, dim3(H, B), dim3(256), lds, st, qkv, out, att, S, H, hd);
    else hipLaunchKernelGGL(rrm_attn_fwd_kernel, dim3(H, B), dim3(256), lds, st, qkv, out, att, S, H, hd);
    CHECK_LAUNCH("rrm_attention_fwd");
    return 0;
}

extern "C" int ieagan_rrm_attention_bwd(const float* qkv, const float* att, const float* dout, float* dqkv, int B, int S, int H, int hd,
                                        void* stream) {
    CHECK_ARG(S >= 1 && S <= SMAX && hd >= 1, "rrm_attention: S must be <= %d", SMAX);
    const bool v4 = (hd & 3) == 0;
    const size_t lds = (size_t)(4 * S * (hd + (v4 ? 4 : 1)) + 2 * S * S) * 4;
    CHECK_ARG(lds <= 150 * 1024, "rrm_attention: head does not fit LDS");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("rrm_attention_bwd", 8.0 * B * H * S * S * hd, 0.0, st);
    if (v4) hipLaunchKernelGGL(rrm_attn_bwd4_kernel, dim3(H, B), dim3(256), lds, st, qkv, att, dout, dqkv, S, H, hd);
    else hipLaunchKernelGGL(rrm_attn_bwd_kernel, dim3(H, B), dim3(256), lds, st, qkv, att, dout, dqkv, S, H, hd);
    CHECK_LAUNCH("rrm_attention_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// loss_block: one workgroup of 1024 threads.
//   inputs (any may be null): dfake[n], dreal[n] logits; e[n,d] embeddings, p[n,d] proxies, er[n,d] real
//   embeddings (IEA target, no gradient).  w[6] = weights of {hinge_real, hinge_fake, hinge_gen, contra,
//   unif, iea} in the total (0 = term off; a term is evaluated iff its weight != 0 and its inputs exist).
//   outputs: vals[8] = {total, hinge_real, hinge_fake, hinge_gen, contra, unif, iea, 0} (unweighted terms),
//            g_dfake[n], g_dreal[n], g_e[n,d], g_p[n,d] = d total / d input.
// ------------------------------------------------------------------------------------------------
struct LossArgs {
    const float* dfake;
    const float* dreal;
    const float* e;
    const float* p;
    const float* er;
    float w[6];
    float temperature;
    float* vals;
    float* g_dfake;
    float* g_dreal;
    float* g_e;
    float* g_p;
    int n, d;
};

// G[i][j] = sum_k A[i][k] B[j][k]; staged in k-chunks of 64 through LDS (A and B may alias)
__device__ void gram(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ G, float* la, float* lb, int n, int d) {
    const int tid = threadIdx.x, nt = blockDim.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                        // up to 4 (i,j) pairs per thread (n*n <= 4096)
    for (int k0 = 0; k0 < d; k0 += 64) {
        __syncthreads();
        for (int idx = tid; idx < n * 64; idx += nt) {
            const int i = idx >> 6, kk = idx & 63;
            la[i * 65 + kk] = (k0 + kk < d) ? A[(long)i * d + k0 + kk] : 0.f;
            lb[i * 65 + kk] = (k0 + kk < d) ? B[(long)i * d + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pr = tid + s * nt;
            if (pr < n * n) {
                const int i = pr / n, j = pr - i * n;
                float a = 0.f;
                for (int kk = 0; kk < 64; ++kk) a += la[i * 65 + kk] * lb[j * 65 + kk];
                acc[s] += a;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pr = tid + s * nt;
        if (pr < n * n) G[pr] = acc[s];
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void loss_block_kernel(LossArgs a) {
    // one block per event: the batch holds gridDim.x events of n rows each (the Grams are intra-event)
    {
        const long ev = blockIdx.x;
        if (a.dfake) a.dfake += ev * a.n;
        if (a.dreal) a.dreal += ev * a.n;
        if (a.e) a.e += ev * a.n * a.d;
        if (a.p) a.p += ev * a.n * a.d;
        if (a.er) a.er += ev * a.n * a.d;
        if (a.g_dfake) a.g_dfake += ev * a.n;
        if (a.g_dreal) a.g_dreal += ev * a.n;
        if (a.g_e) a.g_e += ev * a.n * a.d;
        if (a.g_p) a.g_p += ev * a.n * a.d;
        a.vals += ev * 8;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[16];
    __shared__ float sc[8];
    const int n = a.n, d = a.d, tid = threadIdx.x;
    float* G = sm;                     // e e^T
    float* R = G + n * n;              // er er^T (IEA target logits)
    float* DG = R + n * n;             // d total / d G_ij  (G_ij and G_ji treated as separate variables)
    float* la = DG + n * n;            // staging [n][65] x 2
    float* lb = la + n * 65;
    float* ep = lb + n * 65;           // e_i . p_i
    float* pp = ep + n;                // p_i . p_i
    float* dep = pp + n;               // d total / d (e.p)_i
    float* dpp = dep + n;              // d total / d (p.p)_i
    float* rowa = dpp + n;             // scratch [n]
    float* rowb = rowa + n;
    const float w_hr = a.w[0], w_hf = a.w[1], w_hg = a.w[2];
    float w_c = a.w[3], w_u = a.w[4], w_i = a.w[5];
    if (a.e == nullptr) w_c = w_u = w_i = 0.f;
    if (a.p == nullptr) w_c = 0.f;
    if (a.er == nullptr) w_i = 0.f;
    if (tid < 8) sc[tid] = 0.f;
    __syncthreads();

    // ---- hinge terms on the logits (loss.py:30-38)
    if (a.dfake != nullptr || a.dreal != nullptr) {
        float hr = 0.f, hf = 0.f, hg = 0.f;
        for (int i = tid; i < n; i += blockDim.x) {
            float gf = 0.f;
            if (a.dreal != nullptr) {
                const float x = 1.f - a.dreal[i];
                hr += fmaxf(x, 0.f);
                if (a.g_dreal) a.g_dreal[i] = (x > 0.f) ? -w_hr / n : 0.f;
            }
            if (a.dfake != nullptr) {
                const float x = 1.f + a.dfake[i];
                hf += fmaxf(x, 0.f);
                hg += -a.dfake[i];
                gf = ((x > 0.f) ? w_hf / n : 0.f) - w_hg / n;
                if (a.g_dfake) a.g_dfake[i] = gf;
            }
        }
        hr = block_reduce_sum(hr, red);
        hf = block_reduce_sum(hf, red);
        hg = block_reduce_sum(hg, red);
        if (tid == 0) {
            sc[1] = hr / n;
            sc[2] = hf / n;
            sc[3] = hg / n;
        }
    }
    const bool emb = (w_c != 0.f || w_u != 0.f || w_i != 0.f);
    if (emb) {
        gram(a.e, a.e, G, la, lb, n, d);
        if (w_i != 0.f) gram(a.er, a.er, R, la, lb, n, d);
        for (int idx = tid; idx < n * n; idx += blockDim.x) DG[idx] = 0.f;
        if (w_c != 0.f) {      // e_i . p_i and p_i . p_i
            for (int i = tid >> 4; i < n; i += blockDim.x >> 4) {       // 16 lanes per row
                float s1 = 0.f, s2 = 0.f;
                for (int k = tid & 15; k < d; k += 16) {
                    const float pv = a.p[(long)i * d + k];
                    s1 += a.e[(long)i * d + k] * pv;
                    s2 += pv * pv;
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if ((tid & 15) == 0) {
                    ep[i] = s1;
                    pp[i] = s2;
                }
            }
        }
        for (int i = tid; i < n; i += blockDim.x) dep[i] = dpp[i] = 0.f;
        __syncthreads();

        // ---- 2C contrastive (loss.py:103-132, pos_collected_numerator = False, margin 0)
        if (w_c != 0.f) {
            const float t = a.temperature;
            float part = 0.f;
            for (int i = tid; i < n; i += blockDim.x) {
                const float ri = 1.f / fmaxf(sqrtf(G[i * n + i]), 1e-8f), si = 1.f / fmaxf(sqrtf(pp[i]), 1e-8f);
                const float ci = ep[i] * ri * si;
                const float pos = __expf(ci / t);
                float den = pos;
                for (int j = 0; j < n; ++j)
                    if (j != i) den += __expf(G[i * n + j] * ri / fmaxf(sqrtf(G[j * n + j]), 1e-8f) / t);
                part += -__logf(t * pos / den);
                // L_i = -log t - c_i/t + log den ;  dL_i/dc_i = (-1 + pos/den)/t ;  dL_i/dcos_ij = exp(cos_ij/t)/den/t
                const float cw = w_c / n;
                const float B = cw * (-1.f + pos / den) / t;
                dep[i] = B * ri * si;
                dpp[i] = B * ci * (-0.5f / fmaxf(pp[i], 1e-16f));
                // Every DG element has ONE writer per section (the sections are separated by block barriers), so the sums are formed in
                // a fixed order -- bit-reproducible, unlike LDS float atomics: row i belongs to thread i; the column-wise diagonal terms
                // (row i's share of d total / d G_jj) are parked in the staging matrix `la` and summed by thread j after a barrier.
                float dgii = B * ci * (-0.5f / fmaxf(G[i * n + i], 1e-16f));
                for (int j = 0; j < n; ++j) {
                    if (j == i) {
                        la[i * 65 + j] = 0.f;
                        continue;
                    }
                    const float rj = 1.f / fmaxf(sqrtf(G[j * n + j]), 1e-8f);
                    const float cij = G[i * n + j] * ri * rj;
                    const float A = cw * __expf(cij / t) / den / t;           // d total / d cos_ij (row i)
                    DG[i * n + j] += A * ri * rj;
                    dgii += A * cij * (-0.5f / fmaxf(G[i * n + i], 1e-16f));
                    la[i * 65 + j] = A * cij * (-0.5f / fmaxf(G[j * n + j], 1e-16f));
                }
                DG[i * n + i] += dgii;
            }
            part = block_reduce_sum(part, red);
            if (tid == 0) sc[4] = part / n;
            __syncthreads();
            for (int j = tid; j < n; j += blockDim.x) {
                float s = 0.f;
                for (int i = 0; i < n; ++i) s += la[i * 65 + j];
                DG[j * n + j] += s;
            }
        }
        __syncthreads();

        // ---- uniformity: log mean_{i<j} exp(-2 |e_i - e_j|^2)   (loss.py:8-9)
        if (w_u != 0.f) {
            float part = 0.f;
            for (int idx = tid; idx < n * n; idx += blockDim.x) {
                const int i = idx / n, j = idx - i * n;
                if (i < j) part += __expf(-2.f * fmaxf(G[i * n + i] + G[j * n + j] - 2.f * G[idx], 0.f));
            }
            const float tot = block_reduce_sum(part, red);
            const float npairs = 0.5f * n * (n - 1);
            if (tid == 0) sc[5] = __logf(tot / npairs);
            for (int idx = tid; idx < n * n; idx += blockDim.x) {
                const int i = idx / n, j = idx - i * n;
                if (i < j) {
                    const float wgt = w_u * __expf(-2.f * fmaxf(G[i * n + i] + G[j * n + j] - 2.f * G[idx], 0.f)) / tot;  // d/d(-2 d2)
                    DG[idx] += 4.f * wgt;                       // (one writer per element)
                }
            }
            // diagonal: DG_kk -= 2 sum_{j != k} wgt_kj, by thread k in a fixed order (was: two float atomics per pair)
            for (int k = tid; k < n; k += blockDim.x) {
                float s = 0.f;
                for (int j = 0; j < n; ++j) {
                    if (j == k) continue;
                    const int lo = min(k, j), hi = max(k, j);
                    s += w_u * __expf(-2.f * fmaxf(G[lo * n + lo] + G[hi * n + hi] - 2.f * G[lo * n + hi], 0.f)) / tot;
                }
                DG[k * n + k] += -2.f * s;
            }
        }
        __syncthreads();

        // ---- IEA: KL_batchmean(softmax(er er^T) || softmax(e e^T))   (loss.py:14-27)
        if (w_i != 0.f) {
            float part = 0.f;
            for (int i = tid; i < n; i += blockDim.x) {
                float mf = -1e30f, mr = -1e30f;
                for (int j = 0; j < n; ++j) {
                    mf = fmaxf(mf, G[i * n + j]);
                    mr = fmaxf(mr, R[i * n + j]);
                }
                float lf = 0.f, lr_ = 0.f;
                for (int j = 0; j < n; ++j) {
                    lf += __expf(G[i * n + j] - mf);
                    lr_ += __expf(R[i * n + j] - mr);
                }
                const float lsef = mf + __logf(lf), lser = mr + __logf(lr_);
                for (int j = 0; j < n; ++j) {
                    const float logp = G[i * n + j] - lsef, logt = R[i * n + j] - lser;
                    const float T = __expf(logt);
                    part += T * (logt - logp);
                    DG[i * n + j] += w_i * (__expf(logp) - T) / n;              // row i belongs to thread i
                }
            }
            part = block_reduce_sum(part, red);
            if (tid == 0) sc[6] = part / n;
        }
        __syncthreads();

        // ---- d total / d e_i = sum_j (DG_ij + DG_ji) e_j + dep_i p_i ;  d/dp_i = dep_i e_i + 2 dpp_i p_i
        // (the diagonal DG_ii already counts once per appearance of G_ii; d G_ii / d e_i = 2 e_i)
        for (int idx = tid; idx < n * d; idx += blockDim.x) {
            const int i = idx / d, k = idx - i * d;
            float g = 0.f;
            for (int j = 0; j < n; ++j) {
                const float c = (j == i) ? 2.f * DG[i * n + i] : DG[i * n + j] + DG[j * n + i];
                g += c * a.e[(long)j * d + k];
            }
            if (w_c != 0.f) {
                g += dep[i] * a.p[idx];
                if (a.g_p) a.g_p[idx] = dep[i] * a.e[idx] + 2.f * dpp[i] * a.p[idx];
            }
            if (a.g_e) a.g_e[idx] = g;
        }
    }
    __syncthreads();
    if (tid == 0) {
        sc[0] = w_hr * sc[1] + w_hf * sc[2] + w_hg * sc[3] + w_c * sc[4] + w_u * sc[5] + w_i * sc[6];
        for (int i = 0; i < 8; ++i) a.vals[i] = sc[i];
    }
    (void)rowa; (void)rowb;
}

extern "C" int ieagan_loss_block_events(const float* dfake, const float* dreal, const float* e, const float* p, const float* er,
                                        const float* weights6, float temperature, float* vals8, float* g_dfake, float* g_dreal, float* g_e,
                                        float* g_p, int n, int d, int events, void* stream);

extern "C" int ieagan_loss_block(const float* dfake, const float* dreal, const float* e, const float* p, const float* er,
                                 const float* weights6, float temperature, float* vals8, float* g_dfake, float* g_dreal, float* g_e,
                                 float* g_p, int n, int d, void* stream) {
    return ieagan_loss_block_events(dfake, dreal, e, p, er, weights6, temperature, vals8, g_dfake, g_dreal, g_e, g_p, n, d, 1, stream);
}

extern "C" int ieagan_loss_block_events(const float* dfake, const float* dreal, const float* e, const float* p, const float* er,
                                        const float* weights6, float temperature, float* vals8, float* g_dfake, float* g_dreal, float* g_e,
                                        float* g_p, int n, int d, int events, void* stream) {
    CHECK_ARG(n >= 2 && n <= SMAX, "loss_block: n must be in [2, %d]", SMAX);
    CHECK_ARG(events >= 1, "loss_block: events must be >= 1 (got %d)", events);
    LossArgs a;
    a.dfake = dfake; a.dreal = dreal; a.e = e; a.p = p; a.er = er;
    for (int i = 0; i < 6; ++i) a.w[i] = weights6[i];
    a.temperature = temperature;
    a.vals = vals8; a.g_dfake = g_dfake; a.g_dreal = g_dreal; a.g_e = g_e; a.g_p = g_p;
    a.n = n; a.d = d;
    const size_t lds = (size_t)(3 * n * n + 2 * n * 65 + 6 * n) * 4;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("loss_block", 0.0, 0.0, st);
    hipLaunchKernelGGL(loss_block_kernel, dim3(events), dim3(1024), lds, st, a);
    CHECK_LAUNCH("loss_block");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// D head: h[n][c] = sum_{hw} relu(x[n,hw,c]); backward dx = (x > 0) * dh[n][c].  x bf16 NHWC.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_sum_pool_kernel(const bf16* __restrict__ x, float* __restrict__ out, int HW, int C) {
    const int n = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += fmaxf(bf2f(x[((long)n * HW + p) * C + c]), 0.f);
    out[(long)n * C + c] = s;
}

__global__ __launch_bounds__(256) void relu_sum_pool_bwd_kernel(const bf16* __restrict__ x, const float* __restrict__ dh, bf16* __restrict__ dx,
                                                                long total, int HW, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long n = i / ((long)HW * C);
        dx[i] = f2bf(bf2f(x[i]) > 0.f ? dh[n * C + c] : 0.f);
    }
}

extern "C" int ieagan_relu_sum_pool(const void* x, float* out, int N, int HW, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof("relu_sum_pool", 0.0, 2.0 * N * HW * (double)C, st);
    hipLaunchKernelGGL(relu_sum_pool_kernel, dim3(cdiv(C, 256), N), dim3(256), 0, st, (const bf16*)x, out, HW, C);
    CHECK_LAUNCH("relu_sum_pool");
    return 0;
}

extern "C" int ieagan_relu_sum_pool_bwd(const void* x, const float* dh, void* dx, int N, int HW, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * HW * C;
    ProfScope prof("relu_sum_pool_bwd", 0.0, 4.0 * total, st);
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(relu_sum_pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, dh, (bf16*)dx, total, HW, C);
    CHECK_LAUNCH("relu_sum_pool_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Non-local block glue (layers.py:276-300): 2x2 max-pool of phi / g on bf16 NHWC maps (argmax kept as one byte per
// output element, first maximum in window scan order like F.max_pool2d) and out = gamma * o + x with a device scalar
// gamma; backward d_o = gamma * d, dgamma = sum(d * o) (dx is d itself).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ out, uint64_t* __restrict__ idx,
                                                           long total, int Ho, int Wo, int C8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        const long W = 2L * Wo;
        const long base = ((n * 2 * Ho + 2 * ho) * W + 2 * wo) * C8 + c;
        const bf16x8 v0 = x[base], v1 = x[base + C8], v2 = x[base + W * C8], v3 = x[base + W * C8 + C8];
        bf16x8 m;
        uint64_t sel = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float best = bf2f(v0[j]);
            unsigned k = 0;
            float a = bf2f(v1[j]);
            if (a > best || a != a) { best = a; k = 1; }
            a = bf2f(v2[j]);
            if (a > best || a != a) { best = a; k = 2; }
            a = bf2f(v3[j]);
            if (a > best || a != a) { best = a; k = 3; }
            m[j] = f2bf(best);
            sel |= (uint64_t)k << (8 * j);
        }
        out[i] = m;
        idx[i] = sel;
    }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16x8* __restrict__ dout, const uint64_t* __restrict__ idx,
                                                           bf16x8* __restrict__ dx, long total, int Ho, int Wo, int C8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        const long W = 2L * Wo;
        const long base = ((n * 2 * Ho + 2 * ho) * W + 2 * wo) * C8 + c;
        const bf16x8 d = dout[i];
        const uint64_t sel = idx[i];
        bf16x8 o[4] = {zero8(), zero8(), zero8(), zero8()};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned k = (unsigned)(sel >> (8 * j)) & 3u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k == (unsigned)q) o[q][j] = d[j];
        }
        dx[base] = o[0];
        dx[base + C8] = o[1];
        dx[base + W * C8] = o[2];
        dx[base + W * C8 + C8] = o[3];
    }
}

__global__ __launch_bounds__(256) void gamma_residual_fwd_kernel(const bf16x8* __restrict__ o, const bf16x8* __restrict__ x,
                                                                 const float* __restrict__ gamma, bf16x8* __restrict__ out, long total8) {
    const float g = gamma[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const bf16x8 a = o[i], b = x[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = f2bf(fmaf(g, bf2f(a[j]), bf2f(b[j])));
        out[i] = r;
    }
}

__global__ __launch_bounds__(256) void gamma_residual_bwd_kernel(const bf16x8* __restrict__ d, const bf16x8* __restrict__ o,
                                                                 const float* __restrict__ gamma, bf16x8* __restrict__ d_o,
                                                                 float* __restrict__ dgamma, long total8) {
    __shared__ float red[4];
    const float g = gamma[0];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const bf16x8 dv = d[i], ov = o[i];
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float df = bf2f(dv[j]);
            acc = fmaf(df, bf2f(ov[j]), acc);
            r[j] = f2bf(g * df);
        }
        d_o[i] = r;
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&dgamma[blockIdx.x % STAT_REPL], acc);
}

static inline unsigned ew_blocks(long total) {
    long b = (total + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int ieagan_maxpool2_fwd(const void* x, void* out, void* idx, int N, int H, int W, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(C % 8 == 0 && H % 2 == 0 && W % 2 == 0 && N > 0, "maxpool2: need C %% 8 == 0 and even H, W (got C=%d H=%d W=%d)", C, H, W);
    const long total = (long)N * (H / 2) * (W / 2) * (C / 8);
    ProfScope prof("maxpool2_fwd", 0.0, total * (64.0 + 16.0 + 8.0), st);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, (const bf16x8*)x, (bf16x8*)out, (uint64_t*)idx, total,
                       H / 2, W / 2, C / 8);
    CHECK_LAUNCH("maxpool2_fwd");
    return 0;
}

extern "C" int ieagan_maxpool2_bwd(const void* dout, const void* idx, void* dx, int N, int H, int W, int C, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(C % 8 == 0 && H % 2 == 0 && W % 2 == 0 && N > 0, "maxpool2_bwd: need C %% 8 == 0 and even H, W (got C=%d H=%d W=%d)", C, H, W);
    const long total = (long)N * (H / 2) * (W / 2) * (C / 8);
    ProfScope prof("maxpool2_bwd", 0.0, total * (64.0 + 16.0 + 8.0), st);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, (const bf16x8*)dout, (const uint64_t*)idx, (bf16x8*)dx,
                       total, H / 2, W / 2, C / 8);
    CHECK_LAUNCH("maxpool2_bwd");
    return 0;
}

extern "C" int ieagan_gamma_residual_fwd(const void* o, const void* x, const float* gamma, void* out, long n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(n % 8 == 0 && n > 0 && gamma, "gamma_residual: element count must be a positive multiple of 8 (got %ld)", n);
    ProfScope prof("gamma_residual_fwd", 2.0 * n, 6.0 * n, st);
    hipLaunchKernelGGL(gamma_residual_fwd_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, (const bf16x8*)o, (const bf16x8*)x, gamma,
                       (bf16x8*)out, n / 8);
    CHECK_LAUNCH("gamma_residual_fwd");
    return 0;
}

/* dgamma: STAT_REPL floats, zeroed by the caller; the caller folds them. */
extern "C" int ieagan_gamma_residual_bwd(const void* d, const void* o, const float* gamma, void* d_o, float* dgamma, long n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(n % 8 == 0 && n > 0 && gamma && dgamma, "gamma_residual_bwd: element count must be a positive multiple of 8 (got %ld)", n);
    ProfScope prof("gamma_residual_bwd", 3.0 * n, 6.0 * n, st);
    hipLaunchKernelGGL(gamma_residual_bwd_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, (const bf16x8*)d, (const bf16x8*)o, gamma,
                       (bf16x8*)d_o, dgamma, n / 8);
    CHECK_LAUNCH("gamma_residual_bwd");
    return 0;
}
