// Device helpers shared by the convolution kernels (conv_igemm.hip, conv1x1_stream.hip).
#pragma once
#include "common.h"
#include "conv_args.h"

// Per-(n, c) scale / shift of the fused BatchNorm apply.  `aff` (optional) is a block-resident LDS copy of image n's rows,
// [0, AFF_MAXC) scale and [AFF_MAXC, 2*AFF_MAXC) shift: the table is 64 bytes per 16 bytes of activation when fetched
// through the vector-memory path, which made the affine variants address-unit bound.
#define AFF_MAXC 512
__device__ __forceinline__ void stage_aff(float* aff, const SrcDesc& s, int n, int Cin) {
    for (int i = threadIdx.x; i < Cin; i += 256) {
        aff[i] = s.scale[(long)n * s.aff_nstride + i];
        aff[AFF_MAXC + i] = s.shift[(long)n * s.aff_nstride + i];
    }
}

template <bool AFF, bool RELU>
__device__ __forceinline__ void xform8(float (&v)[8], const SrcDesc& s, int n, int c, const float* aff = nullptr) {
    if (AFF) {
        f32x8 sc, sh;
        if (aff != nullptr) {
            // `aff` is an LDS table: read it through an LDS-qualified pointer.  As a generic pointer these were FLAT loads, which
            // count on the vector-memory counter as well -- every use then waited for ALL outstanding global prefetches.
            typedef const __attribute__((address_space(3))) f32x8* lds_f32x8;
            sc = *(lds_f32x8)(aff + c);
            sh = *(lds_f32x8)(aff + AFF_MAXC + c);
        } else {
            sc = *(const f32x8*)(s.scale + (long)n * s.aff_nstride + c);
            sh = *(const f32x8*)(s.shift + (long)n * s.aff_nstride + c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sh[i];
    }
    if (RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
}


// fold the per-lane statistics partials (lane owns channel chunk lane % (2*NT)) and add them to the
// replicated statistics buffer: one atomic per channel per block.
// The fold over the 64 / CPP lanes that share a chunk goes through a wave-private LDS matrix sx[16 values][64 lanes]
// (XOR-swizzled columns): 16 stores + 64 / CPP loads per lane instead of a 16-value x log2(64 / CPP)-step shuffle
// butterfly -- the flush was ~20 % of the instruction stream of the generator's 1x1 convolutions.
#define STATS_SX_FLOATS (8 * 64)           // per wave (sums and sums of squares take turns)
template <int NT, int NW = 4>
__device__ __forceinline__ void stats_flush(const ConvArgs& a, float (&s1)[8], float (&s2)[8], int n_base, float* red /*[NW][NT*16][2]*/,
                                            float* sx_all /* NW * STATS_SX_FLOATS, free at this point */, int replica, int event) {
    constexpr int CPP = NT * 2;
    constexpr int SH = 64 / CPP;                      // lanes sharing a chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sx = sx_all + wave * STATS_SX_FLOATS;
    // wave-private matrix: LDS operations of one wave execute in order, no barrier needed between the phases
#pragma unroll
    for (int w = 0; w < 2; ++w) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sx[i * 64 + (lane ^ i)] = w ? s2[i] : s1[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // make the wave's stores visible to its other lanes
        __builtin_amdgcn_wave_barrier();
        if (lane < 8 * CPP) {                         // output (chunk cc, channel i of the chunk): 8 * CPP <= 64 per wave
            const int cc = lane % CPP, i = lane / CPP;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < SH; ++k) t += sx[i * 64 + ((cc + CPP * k) ^ i)];
            red[(wave * NT * 16 + cc * 8 + i) * 2 + w] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // phase 1 overwrites what other lanes of this wave just read
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NT * 16 && n_base + t < a.Cout) {
        float x1 = 0.f, x2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) {
            x1 += red[(wv * NT * 16 + t) * 2 + 0];
            x2 += red[(wv * NT * 16 + t) * 2 + 1];
        }
        // slots per group (ieagan_conv_desc.stats_slots): == the launch's blocks per group -> one adder per address, bit-reproducible sums;
        // 0: the legacy replica counts (several adders per slot, order-dependent rounding)
        const int R = a.stats_slots > 0 ? a.stats_slots : ((a.bnb_scale != nullptr) ? BNB_REPL : STAT_REPL);
        float* st = a.stats + ((long)event * R + replica % R) * 2 * a.Cout;
        atomicAdd(st + n_base + t, x1);
        atomicAdd(st + a.Cout + n_base + t, x2);
    }
}


// ------------------------------------------------------------------------------------------------
// Shared epilogue.  The MFMA accumulators of one wave (2 m-tiles x NT n-tiles = 32 pixels x 16*NT
// channels) are transposed through a wave-private LDS buffer so that every lane then owns 8 consecutive
// channels of one pixel: bias, ReLU-mask, residual and the output store are all 16-byte accesses, and
// the per-channel (sum, sumsq) partials stay in registers of a fixed channel group per lane.
//   pix(row, m, n, h, w) -> bool : pixel of wave-local row (0..31); m = linear NHW index
// ------------------------------------------------------------------------------------------------
template <int NT>
struct EpiLds {
    static constexpr int LDW = NT * 16 + 4;             // padded row (floats): 4 rows apart -> 16 banks apart
    static constexpr int FLOATS = 32 * LDW;
};

// BNBOK: the BatchNorm-backward epilogue mode (ieagan_conv_desc.bnb_*) is compiled in -- a separate instantiation of the
// plain-prologue kernels (a dgrad launch has no prologue), so that every other variant keeps its register budget.
// This lane's 8 bias values in the epilogue (channel chunk lane % (2*NT) of the block's n-tile group): requested at kernel start --
// as eight scalar loads inside the epilogue they were a global-memory round trip per tile in every consumer wave.
template <int NT>
__device__ __forceinline__ void load_bias8(const ConvArgs& a, int n_base, float (&bv)[8]) {
    const int co0 = n_base + ((threadIdx.x & 63) % (NT * 2)) * 8;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr && co0 < a.Cout) {          // Cout % 8 == 0 and the bias vector is 32-byte aligned (launcher)
        lo = *(const f32x4*)(a.bias + co0);
        hi = *(const f32x4*)(a.bias + co0 + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bv[i] = lo[i];
        bv[4 + i] = hi[i];
    }
}

template <bool BNBOK, int NT, int MTS = 2, bool EARLY = true, typename PixFn>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x4 (&acc)[MTS][NT], float* wlds, int n_base, PixFn pix,
                                              float (&s1)[8], float (&s2)[8], const float* bias_pre = nullptr,
                                              const bf16x8* mask_pre = nullptr) {
    // bias_pre: this lane's 8 bias values (channel chunk lane % CPP), loaded once by a kernel that walks several tiles;
    // mask_pre: the ReLU-mask / BatchNorm-input chunks of this call's (16*MTS*CPP)/64 iterations, requested by the caller ahead of
    // time (same lane -> (row, chunk) mapping as below) so that their latency is not paid inside the epilogue.
    constexpr int LDW = EpiLds<NT>::LDW;
    constexpr int CPP = NT * 2;                         // 8-channel chunks per pixel
    static_assert((16 * MTS * CPP) % 64 == 0, "epilogue rows x chunks must fill whole waves");
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) wlds[(mt * 16 + lg * 4 + r) * LDW + nt * 16 + lr] = acc[mt][nt][r];
    // the transpose buffer is private to this wave: its LDS operations complete in order, the fence only pins the compiler
    // (a block-wide barrier here made every wave wait for the slowest one four times per tile)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cc = lane % CPP;
    const int co0 = n_base + cc * 8;
    const bool ch_ok = co0 < a.Cout;
    const bool bnb = BNBOK && a.bnb_scale != nullptr;
    const int H = a.H, W = a.W;
    constexpr int EIT = (16 * MTS * CPP) / 64;
    // ---- phase 1: every global operand of this call is requested before anything is computed -- bias (two 16-byte loads),
    // ReLU mask / BatchNorm input, same- or half-resolution residual chunk.  In program order "load, use, load, use" each of
    // them was a memory round trip of its own inside the epilogue (the latency-bound small-map launches felt it most).
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
    if (bias_pre == nullptr && a.bias != nullptr && ch_ok) {        // Cout % 8 == 0; the bias vector is 16-byte aligned (launcher)
        b_lo = *(const f32x4*)(a.bias + co0);
        b_hi = *(const f32x4*)(a.bias + co0 + 4);
    }
    // EARLY = false (the 3x3 kernels, whose register budget is spent on accumulators and prefetched halos): the mask / residual
    // chunks are loaded where they are used, as before
    const bool want_mask = EARLY && a.mask != nullptr && mask_pre == nullptr;
    const bool ra_here = a.ra != nullptr && co0 < a.Ca;
    const bool rb_here = !ra_here && a.rb != nullptr && co0 >= a.Ca;
    bf16x8 mk_r[EIT], rs_r[EIT];
    long m_r[EIT];
    bool ok_r[EIT];
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
        const int row = (it * 64 + lane) / CPP;
        int n, h, w;
        ok_r[it] = pix(row, m_r[it], n, h, w) && ch_ok;
        mk_r[it] = zero8();
        rs_r[it] = zero8();
        if (!ok_r[it]) continue;
        const long m = m_r[it];
        if (want_mask) mk_r[it] = *(const bf16x8*)((const bf16*)a.mask + m * a.Cout + co0);
        if (!EARLY) continue;
        if (ra_here && a.ra_rs == 0) rs_r[it] = *(const bf16x8*)((const bf16*)a.ra + m * a.Cra + co0);
        else if (ra_here && a.ra_rs == 1)      // operand lives at half resolution: nearest x2 upsample
            rs_r[it] = *(const bf16x8*)((const bf16*)a.ra + (((long)n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * a.Cra + co0);
        else if (rb_here) rs_r[it] = *(const bf16x8*)((const bf16*)a.rb + m * a.Crb + (co0 - a.Ca));
    }
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = bias_pre ? bias_pre[i] : (i < 4 ? b_lo[i & 3] : b_hi[i & 3]);
    // ---- phase 2
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
        const int row = (it * 64 + lane) / CPP;
        const long m = m_r[it];
        const bool ok = ok_r[it];
        const f32x4 lo = *(const f32x4*)(wlds + row * LDW + cc * 8);
        const f32x4 hi = *(const f32x4*)(wlds + row * LDW + cc * 8 + 4);
        if (!ok) continue;
        float v[8] = {lo[0] + bv[0], lo[1] + bv[1], lo[2] + bv[2], lo[3] + bv[3], hi[0] + bv[4], hi[1] + bv[5], hi[2] + bv[6], hi[3] + bv[7]};
        if (bnb) {                      // v = d(conv input); the conv input was relu(x*scale + shift): fold that apply's backward
            const bf16x8 xv = mask_pre ? mask_pre[it] : (EARLY ? mk_r[it] : *(const bf16x8*)((const bf16*)a.mask + m * a.Cout + co0));
            int n, h, w;
            long mm;
            pix(row, mm, n, h, w);
            const long so = (long)n * a.bnb_nstride + co0;
            const f32x4 sc0 = *(const f32x4*)(a.bnb_scale + so), sc1 = *(const f32x4*)(a.bnb_scale + so + 4);
            const f32x4 sh0 = *(const f32x4*)(a.bnb_shift + so), sh1 = *(const f32x4*)(a.bnb_shift + so + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xf = bf2f(xv[i]);
                const float sc = i < 4 ? sc0[i & 3] : sc1[i & 3], sh = i < 4 ? sh0[i & 3] : sh1[i & 3];
                const float d = (a.bnb_relu && !(xf * sc + sh > 0.f)) ? 0.f : v[i];
                s1[i] += d;                 // -> d shift (statistics slot 0)
                s2[i] += d * xf;            // -> d scale (statistics slot 1)
                v[i] = d * sc;
            }
        } else if (a.mask != nullptr) {        // fused ReLU backward of the main path (residual is added after it)
            const bf16x8 mk = mask_pre ? mask_pre[it] : (EARLY ? mk_r[it] : *(const bf16x8*)((const bf16*)a.mask + m * a.Cout + co0));
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (bf2f(mk[i]) > 0.f) ? v[i] : 0.f;
        }
        if (ra_here) {
            float rv[8];
            if (a.ra_rs != 2) {
                bf16x8 t = rs_r[it];
                if (!EARLY) {
                    int n, h, w;
                    long mm;
                    pix(row, mm, n, h, w);
                    t = (a.ra_rs == 0) ? *(const bf16x8*)((const bf16*)a.ra + m * a.Cra + co0)
                                       : *(const bf16x8*)((const bf16*)a.ra + (((long)n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * a.Cra + co0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) rv[i] = bf2f(t[i]);
            } else {                        // operand lives at double resolution: 2x2 average (four loads: not requested ahead)
                int n, h, w;
                long mm;
                pix(row, mm, n, h, w);
                const bf16* p = (const bf16*)a.ra + (((long)n * (2 * H) + 2 * h) * (2 * W) + 2 * w) * a.Cra + co0;
                const long rs_ = (long)2 * W * a.Cra;
                const bf16x8 t0 = *(const bf16x8*)p, t1 = *(const bf16x8*)(p + a.Cra), t2 = *(const bf16x8*)(p + rs_),
                             t3 = *(const bf16x8*)(p + rs_ + a.Cra);
#pragma unroll
                for (int i = 0; i < 8; ++i) rv[i] = 0.25f * (bf2f(t0[i]) + bf2f(t1[i]) + bf2f(t2[i]) + bf2f(t3[i]));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += a.ra_scale * rv[i];
        } else if (rb_here) {
            const bf16x8 t = EARLY ? rs_r[it] : *(const bf16x8*)((const bf16*)a.rb + m * a.Crb + (co0 - a.Ca));
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bf2f(t[i]);
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i] = f2bf(v[i]);
            if (!bnb) {
                s1[i] += v[i];
                s2[i] += v[i] * v[i];
            }
        }
        *(bf16x8*)((bf16*)a.out + m * a.Cout + co0) = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the caller may overwrite the buffer (next half / next phase);
    __builtin_amdgcn_wave_barrier();                             // anything that touches ANOTHER wave's region needs a block barrier
}


// ------------------------------------------------------------------------------------------------
// A-operand gather with the fused prologue: 8 consecutive channels [c, c+8) of the (virtual) conv-input pixel (n, hh, ww) at
// conv resolution; zero outside the image (conv padding is applied AFTER the activation, as in the reference).
// ------------------------------------------------------------------------------------------------
template <bool AFF, bool RELU, int RS>
__device__ __forceinline__ bf16x8 gather8(const SrcDesc& s, int H, int W, int n, int hh, int ww, int c, bool ok,
                                          const float* aff = nullptr) {
    ok = ok && (hh >= 0) && (hh < H) && (ww >= 0) && (ww < W);
    if (!ok) return zero8();
    if (RS == 2) {  // conv pixel = mean of the 2x2 source block (AvgPool2d(2) of the activated source)
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16x8 raw = *(const bf16x8*)((const bf16*)s.x + (((long)n * s.Hs + 2 * hh + (q >> 1)) * s.Ws + 2 * ww + (q & 1)) * s.Cx + c);
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = bf2f(raw[i]);
            xform8<AFF, RELU>(v, s, n, c, aff);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(0.25f * acc[i]);
        return o;
    }
    const int sh_ = (RS == 1) ? (hh >> 1) : hh, sw_ = (RS == 1) ? (ww >> 1) : ww;
    const bf16x8 raw = *(const bf16x8*)((const bf16*)s.x + (((long)n * s.Hs + sh_) * s.Ws + sw_) * s.Cx + c);
    if (!AFF && !RELU) return raw;
    if (!AFF && RELU) return relu8(raw);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(raw[i]);
    xform8<AFF, RELU>(v, s, n, c, aff);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
    return o;
}

