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
            sc = *(const f32x8*)(aff + c);
            sh = *(const f32x8*)(aff + AFF_MAXC + c);
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
template <int NT>
__device__ __forceinline__ void stats_flush(const ConvArgs& a, float (&s1)[8], float (&s2)[8], int n_base, float* red /*[4][NT*16][2]*/,
                                            float* sx_all /* 4 * STATS_SX_FLOATS, free at this point */, int replica, int event) {
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
        for (int wv = 0; wv < 4; ++wv) {
            x1 += red[(wv * NT * 16 + t) * 2 + 0];
            x2 += red[(wv * NT * 16 + t) * 2 + 1];
        }
        float* st = a.stats + ((long)event * STAT_REPL + replica % STAT_REPL) * 2 * a.Cout;
        atomicAdd(st + n_base + t, x1);
        atomicAdd(st + a.Cout + n_base + t, x2);
    }
}

