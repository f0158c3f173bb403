// Non-local (SAGAN-style) self-attention core of the discriminator, streaming softmax on MFMA:
//   beta = softmax_k(theta_q . phi_k),  o_q = sum_k beta_qk g_k          (layers.py:283-300)
// with Lq = H*W queries (3072), Lk = H*W/4 max-pooled keys (768), d_qk = C/8 (32), d_v = C/2 (128), no
// 1/sqrt(d) scaling.  The reference materialises beta [40, 3072, 768] (377 MB fp32); here a workgroup of
// 4 waves owns 64 queries (forward, dQ) or 64 keys (dK/dV) and streams the other side in chunks of 32, so
// beta only ever exists as MFMA fragments.
//
// Layout trick used by all three kernels: the score tile is computed TRANSPOSED (keys on the MFMA rows for
// forward/dQ, queries on the rows for dK/dV), so that the accumulator a lane holds -- 4 consecutive "k"
// indices for ONE column -- is, after bf16 packing, directly the A fragment of the next MFMA (whose k slots
// are assigned to those indices in the same permuted order), and the matching B fragment is fetched from an
// LDS-staged natural-layout tile with ds_read_b64_tr_b16.  No shuffles, no LDS round trip for P / dS.
// k-slot (g, j) of a 32-deep MFMA step  <->  chunk row  pi(g, j) = 4g + j (j < 4) | 16 + 4g + (j - 4).
#include "common.h"

#define AT_CH 32        // chunk of the streamed side

__device__ __forceinline__ bf16x8 pack2(const f32x4& a, const f32x4& b) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = f2bf(a[j]);
        o[4 + j] = f2bf(b[j]);
    }
    return o;
}

// B fragment for k-slots pi(g, .) and columns [c0, c0+16) of an LDS tile [rows][stride] (bf16, natural layout)
__device__ __forceinline__ bf16x8 trfrag(const bf16* tile, int stride, int c0, int lr, int lg) {
    const int q = lr >> 2, p = lr & 3;
    const bf16* p0 = tile + (4 * lg + q) * stride + c0 + 4 * p;
    const bf16* p1 = p0 + 16 * stride;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// Swizzled LDS tiles (backward kernels): the 16-byte chunk c of row r of a [32][W] tile lives at chunk position c ^ key(r), with
// key(r) = r & 15 for W = 128 (16 chunks per row) and (r >> 2) & 3 for W = 32 (4 chunks per row).  A-fragment reads (lane lr reads
// row lr, one chunk: rows 256 / 64 bytes apart would all hit the same banks) and the transposed B-fragment reads then spread over
// all banks; staging writes apply the same key.
template <int W>
__device__ __forceinline__ int swz_key(int r) { return W == 128 ? (r & 15) : W == 64 ? (r & 7) : ((r >> 2) & 3); }
template <int W>
__device__ __forceinline__ int swz_off(int r, int c) { return r * W + ((c ^ swz_key<W>(r)) << 3); }      // bf16 elements
// B fragment (as trfrag) from a swizzled tile
template <int W>
__device__ __forceinline__ bf16x8 trfrag_swz(const bf16* tile, int c0, int lr, int lg) {
    const int q = lr >> 2, p = lr & 3;
    const int row = 4 * lg + q;                                   // rows row and row + 16 share the key
    const bf16* p0 = tile + swz_off<W>(row, (c0 >> 3) + (p >> 1)) + 4 * (p & 1);
    const bf16* p1 = p0 + 16 * W;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// 8 consecutive features [8g, 8g+8) of row `row` of a [rows][width] bf16 matrix (zero beyond rows / width)
__device__ __forceinline__ bf16x8 rowfrag(const bf16* base, long row, long rows, int width, int lg) {
    if (row >= rows || 8 * lg >= width) return zero8();
    return *(const bf16x8*)(base + row * width + 8 * lg);
}

// stage `nrows` (<= 32) rows x width of a row-major bf16 matrix into LDS (zero fill beyond `rows`)
__device__ __forceinline__ void stage_rows(bf16* lds, const bf16* base, long row0, long rows, int width) {
    const int chunks = width >> 3;
    for (int idx = threadIdx.x; idx < AT_CH * chunks; idx += 256) {
        const int r = idx / chunks, c = idx - r * chunks;
        bf16x8 v = zero8();
        if (row0 + r < rows) v = *(const bf16x8*)(base + (row0 + r) * width + c * 8);
        *(bf16x8*)(lds + r * width + c * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// forward: O[q] = softmax(Q K^T) V, LSE[q] = log sum exp.  grid (ceil(Lq/64), N), 4 waves x 16 queries.
// ------------------------------------------------------------------------------------------------
template <int DV>
__global__ __launch_bounds__(256) void nl_attn_fwd_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                          const bf16* __restrict__ V, bf16* __restrict__ O, float* __restrict__ LSE,
                                                          int Lq, int Lk, int dqk) {
    __shared__ __attribute__((aligned(16))) bf16 vlds[AT_CH * DV];
    __shared__ __attribute__((aligned(16))) bf16 klds[AT_CH * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long n = blockIdx.y;
    const long q0 = (long)blockIdx.x * 64 + wave * 16;
    const bf16* Qn = Q + n * Lq * dqk;
    const bf16* Kn = K + n * Lk * dqk;
    const bf16* Vn = V + n * Lk * DV;
    const bf16x8 qf = rowfrag(Qn, q0 + lr, Lq, dqk, lg);          // B operand: [d 8g+j][query lr]
    float m = -1e30f, l = 0.f;
    f32x4 o[DV / 16];
#pragma unroll
    for (int nt = 0; nt < DV / 16; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // These kernels are bound by VALU ISSUE (~45 VALU instructions per MFMA), and close to half of those instructions were register
    // copies of the software pipeline (`cur = next`) and, here, AGPR <-> VGPR moves of the output accumulators for the per-chunk
    // softmax rescale.  So: two register sets used in turn by a loop unrolled twice (no copies); loads unconditional with rows clamped
    // to the last key (keys >= Lk are masked out of the softmax, a clamped row only ever meets p = 0) -- under per-lane conditions or
    // behind a branch the compiler cannot count the younger loads and drains the counter; and the rescale of o / l only when some
    // row's maximum has grown by more than RESCALE_T since its reference was set (wave-uniform branch): p = exp(s - m_ref) stays
    // below e^RESCALE_T, exact in fp32 and harmless in the bf16 pack.
    constexpr int CPR = DV / 8;                                    // 16-byte chunks per V row
    constexpr int SV = (AT_CH * CPR + 255) / 256;                  // V staging chunks per thread
    constexpr float RESCALE_T = 8.f;
    struct Pre { bf16x8 sk, sv[SV]; };
    Pre pa_, pb_;
    const int sr = (threadIdx.x >> 2) & (AT_CH - 1), sc = threadIdx.x & 3;
    const int scol = min(sc * 8, dqk - 8);
    auto prefetch = [&](int kc, Pre& P) {                          // a key chunk crosses L2 -> LDS once, in full rows
        P.sk = *(const bf16x8*)(Kn + min((long)kc + sr, (long)Lk - 1) * dqk + scol);
#pragma unroll
        for (int j = 0; j < SV; ++j) {
            const int idx = min((int)threadIdx.x + j * 256, AT_CH * CPR - 1);
            const int r = idx / CPR, c = idx - r * CPR;
            P.sv[j] = *(const bf16x8*)(Vn + min((long)kc + r, (long)Lk - 1) * DV + c * 8);
        }
    };
    auto chunk = [&](int kc, Pre& P) {
        __syncthreads();
        if (threadIdx.x < AT_CH * 4) *(bf16x8*)(klds + swz_off<32>(sr, sc)) = (sc * 8 < dqk) ? P.sk : zero8();
#pragma unroll
        for (int j = 0; j < SV; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int r = idx / CPR, c = idx - r * CPR;
            if (idx < AT_CH * CPR) *(bf16x8*)(vlds + swz_off<DV>(r, c)) = P.sv[j];
        }
        __syncthreads();
        prefetch(kc + 2 * AT_CH, P);                               // (past the end: clamped, never consumed)
        f32x4 s[2];                                                // S^T tile t: rows = keys kc+16t.., cols = queries
        s[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(klds + swz_off<32>(lr, lg)), qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(klds + swz_off<32>(16 + lr, lg)), qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        float cm = -1e30f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kc + 16 * t + 4 * lg + r >= Lk) s[t][r] = -1e30f;
                cm = fmaxf(cm, s[t][r]);
            }
        cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));                    // chunk maximum of query lr
        if (__any(cm > m + RESCALE_T)) {                           // rare after the first chunk
            const float mn = fmaxf(m, cm);
            const float alpha = __expf(m - mn);
            l *= alpha;
            m = mn;
            float ar[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * lg + r, 64);   // output rows of this lane are queries 4g+r
#pragma unroll
            for (int nt = 0; nt < DV / 16; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[nt][r] *= ar[r];
        }
        float cs = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = (kc + 16 * t + 4 * lg + r >= Lk) ? 0.f : __expf(s[t][r] - m);
                cs += s[t][r];
            }
        cs += __shfl_xor(cs, 16, 64);
        cs += __shfl_xor(cs, 32, 64);
        l += cs;
        const bf16x8 pa = pack2(s[0], s[1]);                       // A operand: [query lr][k-slot (g,j) = key pi(g,j)]
#pragma unroll
        for (int nt = 0; nt < DV / 16; ++nt)
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, trfrag_swz<DV>(vlds, nt * 16, lr, lg), o[nt], 0, 0, 0);
    };
    prefetch(0, pa_);
    prefetch(AT_CH, pb_);
    for (int kc = 0; kc < Lk; kc += 2 * AT_CH) {
        chunk(kc, pa_);
        chunk(kc + AT_CH, pb_);     // unconditional: an odd chunk count runs one fully masked chunk (p = 0 against clamped rows)
    }
    if (lg == 0 && q0 + lr < Lq) LSE[n * Lq + q0 + lr] = m + __logf(l);
    float il[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) il[r] = 1.f / __shfl(l, 4 * lg + r, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long q = q0 + 4 * lg + r;
        if (q >= Lq) continue;
#pragma unroll
        for (int nt = 0; nt < DV / 16; ++nt) O[(n * Lq + q) * DV + nt * 16 + lr] = f2bf(o[nt][r] * il[r]);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, query side: delta[q] = dO[q].O[q];  dQ[q] = sum_k dS[q,k] K[k],  dS = P (dP - delta), dP = dO V^T.
// ------------------------------------------------------------------------------------------------
template <int DV>
__global__ __launch_bounds__(256) void nl_attn_bwd_q_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                            const bf16* __restrict__ V, const bf16* __restrict__ O,
                                                            const bf16* __restrict__ dO, const float* __restrict__ LSE,
                                                            float* __restrict__ delta, bf16* __restrict__ dQ, int Lq, int Lk, int dqk) {
    __shared__ __attribute__((aligned(16))) bf16 klds[AT_CH * 32];
    __shared__ __attribute__((aligned(16))) bf16 vlds[AT_CH * DV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long n = blockIdx.y;
    const long q0 = (long)blockIdx.x * 64 + wave * 16;
    const bf16* Qn = Q + n * Lq * dqk;
    const bf16* Kn = K + n * Lk * dqk;
    const bf16* Vn = V + n * Lk * DV;
    const bf16x8 qf = rowfrag(Qn, q0 + lr, Lq, dqk, lg);
    bf16x8 dof[DV / 32];                                           // dO^T as B operand: [dv 32s+8g+j][query lr]
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < DV / 32; ++s) {
        dof[s] = zero8();
        if (q0 + lr < Lq) {
            dof[s] = *(const bf16x8*)(dO + (n * Lq + q0 + lr) * DV + 32 * s + 8 * lg);
            const bf16x8 ov = *(const bf16x8*)(O + (n * Lq + q0 + lr) * DV + 32 * s + 8 * lg);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(dof[s][j]) * bf2f(ov[j]);
        }
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);                                  // delta of query lr
    const float lse = (q0 + lr < Lq) ? LSE[n * Lq + q0 + lr] : 0.f;
    if (lg == 0 && q0 + lr < Lq) delta[n * Lq + q0 + lr] = dl;
    f32x4 dq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    // A key chunk crosses L2 -> LDS once in full rows (K 2 KB, V 8 KB) and every fragment -- the A operands of S and dP with
    // ds_read_b128, the B operand of dQ with transposed reads -- comes from the swizzled LDS tiles (see the key-side kernel: fragment-
    // shaped global loads, repeated by all four waves, were what bound these kernels).  Two register sets used in turn,
    // unconditional clamped loads (see the forward kernel).
    constexpr int CPR = DV / 8;                                    // 16-byte chunks per V row
    constexpr int SV = (AT_CH * CPR + 255) / 256;                  // V staging chunks per thread
    struct Pre { bf16x8 sk, sv[SV]; };
    Pre pa_, pb_;
    const int sr = (threadIdx.x >> 2) & (AT_CH - 1), sc = threadIdx.x & 3;
    const int scol = min(sc * 8, dqk - 8);
    auto prefetch = [&](int kc, Pre& P) {
        P.sk = *(const bf16x8*)(Kn + min((long)kc + sr, (long)Lk - 1) * dqk + scol);
#pragma unroll
        for (int j = 0; j < SV; ++j) {
            const int idx = min((int)threadIdx.x + j * 256, AT_CH * CPR - 1);
            const int r = idx / CPR, c = idx - r * CPR;
            P.sv[j] = *(const bf16x8*)(Vn + min((long)kc + r, (long)Lk - 1) * DV + c * 8);
        }
    };
    auto chunk = [&](int kc, Pre& P) {
        __syncthreads();
        // K chunk [32 keys][32] (zero beyond Lk / dqk) and V chunk [32 keys][DV], swizzled
        if (threadIdx.x < AT_CH * 4) *(bf16x8*)(klds + swz_off<32>(sr, sc)) = (kc + sr < Lk && sc * 8 < dqk) ? P.sk : zero8();
#pragma unroll
        for (int j = 0; j < SV; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int r = idx / CPR, c = idx - r * CPR;
            if (idx < AT_CH * CPR) *(bf16x8*)(vlds + swz_off<DV>(r, c)) = P.sv[j];      // (rows beyond Lk: clamped copies, masked below)
        }
        __syncthreads();
        prefetch(kc + 2 * AT_CH, P);                               // the set is free again (past the end: clamped, never consumed)
        f32x4 p[2], dp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 16 * t + lr;
            p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(klds + swz_off<32>(row, lg)), qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DV / 32; ++s)                      // dP^T tile: rows = keys, cols = queries
                dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(vlds + swz_off<DV>(row, 4 * s + lg)), dof[s], dp[t], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = kc + 16 * t + 4 * lg + r < Lk;
                const float pr = ok ? __expf(p[t][r] - lse) : 0.f;
                p[t][r] = ok ? pr * (dp[t][r] - dl) : 0.f;         // dS[query lr][key]
            }
        }
        const bf16x8 dsa = pack2(p[0], p[1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            dq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsa, trfrag_swz<32>(klds, nt * 16, lr, lg), dq[nt], 0, 0, 0);
    };
    prefetch(0, pa_);
    prefetch(AT_CH, pb_);
    for (int kc = 0; kc < Lk; kc += 2 * AT_CH) {
        chunk(kc, pa_);
        chunk(kc + AT_CH, pb_);     // unconditional (fully masked when past the end)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long q = q0 + 4 * lg + r;
        if (q >= Lq) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (nt * 16 + lr < dqk) dQ[(n * Lq + q) * dqk + nt * 16 + lr] = f2bf(dq[nt][r]);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, key side: dV[k] = sum_q P[q,k] dO[q],  dK[k] = sum_q dS[q,k] Q[q].  4 waves x 16 keys, queries streamed.
// ------------------------------------------------------------------------------------------------
template <int DV, bool LQ4>        // LQ4: Lq % 4 == 0 -- LSE / delta of the 4 rows of a lane come as one 16-byte load
__global__ __launch_bounds__(256, 2) void nl_attn_bwd_k_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                            const bf16* __restrict__ V, const bf16* __restrict__ dO,
                                                            const float* __restrict__ LSE, const float* __restrict__ delta,
                                                            bf16* __restrict__ dK, bf16* __restrict__ dV, int Lq, int Lk, int dqk) {
    __shared__ __attribute__((aligned(16))) bf16 dolds[AT_CH * DV];
    __shared__ __attribute__((aligned(16))) bf16 qlds[AT_CH * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long n = blockIdx.y;
    const long k0 = (long)blockIdx.x * 64 + wave * 16;
    const bf16* Qn = Q + n * Lq * dqk;
    const bf16* Kn = K + n * Lk * dqk;
    const bf16* Vn = V + n * Lk * DV;
    const bf16* dOn = dO + n * Lq * DV;
    const bf16x8 kf = rowfrag(Kn, k0 + lr, Lk, dqk, lg);          // B operand of S: [d 8g+j][key lr]
    bf16x8 vf[DV / 32];                                            // V^T as B operand of dP: [dv][key lr]
#pragma unroll
    for (int s = 0; s < DV / 32; ++s) {
        vf[s] = zero8();
        if (k0 + lr < Lk) vf[s] = *(const bf16x8*)(Vn + (k0 + lr) * DV + 32 * s + 8 * lg);
    }
    f32x4 dv[DV / 16], dk[2];
#pragma unroll
    for (int nt = 0; nt < DV / 16; ++nt) dv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dk[0] = dk[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // This kernel was bound by the L2 -> CU path, not by arithmetic or latency: every wave fetched the A fragments of Q and dO of
    // every query chunk straight from global memory (16 rows x 64 bytes per instruction, the same fragments in all four waves, on
    // top of the rows staged for the transposed reads: 68 KB per block and chunk, 3.1 GB per launch; switching the in-loop loads off
    // took it from 235 to 160 us, nothing else moved it).  Now a chunk crosses L2 -> LDS ONCE in full rows (10 KB) and all fragments
    // -- A operands with ds_read_b128, B operands with transposed reads -- come from swizzled LDS tiles.  Two register sets used in
    // turn by a loop unrolled twice, unconditional clamped loads (see the forward kernel).
    constexpr int SD = (AT_CH * DV / 8) / 256;                     // dO staging chunks per thread (1 / 1 / 2 for DV 32 / 64 / 128)
    static_assert(SD >= 1 || DV == 32, "staging split");
    constexpr int SDN = SD > 0 ? SD : 1;
    constexpr int CPR = DV / 8;                                    // 16-byte chunks per dO row
    struct Pre { bf16x8 sdo[SDN], sq; f32x4 lse[2], dlt[2]; };
    Pre pa_, pb_;
    const int sr = (threadIdx.x >> 2) & (AT_CH - 1), sc = threadIdx.x & 3;
    const int scol = min(sc * 8, dqk - 8);
    const float* LSEn = LSE + n * Lq;
    const float* DLTn = delta + n * Lq;
    auto prefetch = [&](int qc, Pre& P) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long qb = (long)qc + 16 * t + 4 * lg;
            if (LQ4) {
                const long q4 = min(qb, (long)Lq - 4);
                P.lse[t] = *(const f32x4*)(LSEn + q4);
                P.dlt[t] = *(const f32x4*)(DLTn + q4);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long q = min(qb + r, (long)Lq - 1);
                    P.lse[t][r] = LSEn[q];
                    P.dlt[t][r] = DLTn[q];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SDN; ++j) {
            const int idx = min((int)threadIdx.x + j * 256, AT_CH * CPR - 1);
            const int r = idx / CPR, c = idx - r * CPR;
            P.sdo[j] = *(const bf16x8*)(dOn + min((long)qc + r, (long)Lq - 1) * DV + c * 8);
        }
        P.sq = *(const bf16x8*)(Qn + min((long)qc + sr, (long)Lq - 1) * dqk + scol);
    };
    auto chunk = [&](int qc, Pre& P) {
        __syncthreads();                                           // the previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < SDN; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int r = idx / CPR, c = idx - r * CPR;
            if (idx < AT_CH * CPR) *(bf16x8*)(dolds + swz_off<DV>(r, c)) = (qc + r < Lq) ? P.sdo[j] : zero8();
        }
        if (threadIdx.x < AT_CH * 4) *(bf16x8*)(qlds + swz_off<32>(sr, sc)) = (qc + sr < Lq && sc * 8 < dqk) ? P.sq : zero8();
        __syncthreads();
        const f32x4 lse0 = P.lse[0], lse1 = P.lse[1], dlt0 = P.dlt[0], dlt1 = P.dlt[1];
        prefetch(qc + 2 * AT_CH, P);                               // the set is free again (past the end: clamped, never consumed)
        f32x4 p[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {                              // tile t: rows = queries qc+16t.., cols = keys
            const int row = 16 * t + lr;
            const bf16x8 qa = *(const bf16x8*)(qlds + swz_off<32>(row, lg));            // [query lr][d 8g..] (zero beyond dqk)
            p[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            ds[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DV / 32; ++s)                      // dP tile: rows = queries, cols = keys
                ds[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(dolds + swz_off<DV>(row, 4 * s + lg)), vf[s], ds[t], 0, 0, 0);
            const f32x4 lse = t ? lse1 : lse0, dlt = t ? dlt1 : dlt0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long q = qc + 16 * t + 4 * lg + r;           // query (row) of this accumulator element
                const bool ok = q < Lq && k0 + lr < Lk;
                const float pr = ok ? __expf(p[t][r] - lse[r]) : 0.f;
                p[t][r] = pr;                                      // P[q][key lr]
                ds[t][r] = ok ? pr * (ds[t][r] - dlt[r]) : 0.f;    // dS[q][key lr]
            }
        }
        const bf16x8 pa = pack2(p[0], p[1]);                       // A: [key lr][k-slot (g,j) = query pi(g,j)]
        const bf16x8 dsa = pack2(ds[0], ds[1]);
#pragma unroll
        for (int nt = 0; nt < DV / 16; ++nt)
            dv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, trfrag_swz<DV>(dolds, nt * 16, lr, lg), dv[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            dk[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsa, trfrag_swz<32>(qlds, nt * 16, lr, lg), dk[nt], 0, 0, 0);
    };
    prefetch(0, pa_);
    prefetch(AT_CH, pb_);
    for (int qc = 0; qc < Lq; qc += 2 * AT_CH) {
        chunk(qc, pa_);
        chunk(qc + AT_CH, pb_);     // unconditional (fully masked when past the end)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long k = k0 + 4 * lg + r;
        if (k >= Lk) continue;
#pragma unroll
        for (int nt = 0; nt < DV / 16; ++nt) dV[(n * Lk + k) * DV + nt * 16 + lr] = f2bf(dv[nt][r]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (nt * 16 + lr < dqk) dK[(n * Lk + k) * dqk + nt * 16 + lr] = f2bf(dk[nt][r]);
    }
}

// ------------------------------------------------------------------------------------------------
static bool attn_args_ok(int dqk, int dv) { return dqk >= 8 && dqk <= 32 && dqk % 8 == 0 && (dv == 32 || dv == 64 || dv == 128); }

extern "C" int ieagan_nl_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int N, int Lq, int Lk,
                                       int dqk, int dv, void* stream) {
    CHECK_ARG(attn_args_ok(dqk, dv), "nl_attention: d_qk must be 8..32 (multiple of 8) and d_v in {32,64,128} (got %d, %d)", dqk, dv);
    hipStream_t st = (hipStream_t)stream;
    const double flops = 2.0 * N * (double)Lq * Lk * (dqk + dv);
    ProfScope prof("nl_attention_fwd", flops, 2.0 * N * ((double)Lq * (dqk + dv) + (double)Lk * (dqk + dv)), st);
    dim3 grid((Lq + 63) / 64, N);
#define L(D) hipLaunchKernelGGL((nl_attn_fwd_kernel<D>), grid, dim3(256), 0, st, (const bf16*)Q, (const bf16*)K, (const bf16*)V, (bf16*)O, LSE, Lq, Lk, dqk)
    if (dv == 32) L(32); else if (dv == 64) L(64); else L(128);
#undef L
    CHECK_LAUNCH("nl_attention_fwd");
    return 0;
}

extern "C" int ieagan_nl_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                       float* delta, void* dQ, void* dK, void* dV, int N, int Lq, int Lk, int dqk, int dv,
                                       void* stream) {
    CHECK_ARG(attn_args_ok(dqk, dv), "nl_attention: d_qk must be 8..32 (multiple of 8) and d_v in {32,64,128} (got %d, %d)", dqk, dv);
    hipStream_t st = (hipStream_t)stream;
    const double flops = 2.0 * N * (double)Lq * Lk * (3.0 * dqk + 2.0 * dv) + 2.0 * N * (double)Lq * Lk * (dqk + dv);
    ProfScope prof("nl_attention_bwd", flops, 4.0 * N * ((double)Lq * (dqk + dv) + (double)Lk * (dqk + dv)), st);
    dim3 gq((Lq + 63) / 64, N), gk((Lk + 63) / 64, N);
#define L(D)                                                                                                                   \
    hipLaunchKernelGGL((nl_attn_bwd_q_kernel<D>), gq, dim3(256), 0, st, (const bf16*)Q, (const bf16*)K, (const bf16*)V,       \
                       (const bf16*)O, (const bf16*)dO, LSE, delta, (bf16*)dQ, Lq, Lk, dqk);                                   \
    if ((Lq & 3) == 0)                                                                                                         \
        hipLaunchKernelGGL((nl_attn_bwd_k_kernel<D, true>), gk, dim3(256), 0, st, (const bf16*)Q, (const bf16*)K, (const bf16*)V, \
                           (const bf16*)dO, LSE, (const float*)delta, (bf16*)dK, (bf16*)dV, Lq, Lk, dqk);                        \
    else                                                                                                                       \
        hipLaunchKernelGGL((nl_attn_bwd_k_kernel<D, false>), gk, dim3(256), 0, st, (const bf16*)Q, (const bf16*)K, (const bf16*)V, \
                           (const bf16*)dO, LSE, (const float*)delta, (bf16*)dK, (bf16*)dV, Lq, Lk, dqk)
    if (dv == 32) { L(32); } else if (dv == 64) { L(64); } else { L(128); }
#undef L
    CHECK_LAUNCH("nl_attention_bwd");
    return 0;
}
