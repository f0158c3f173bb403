// Batched spectral normalisation (one power iteration, one singular value) for ALL spectrally
// normalised layers of a network in three launches, instead of ~5 tiny ATen ops per layer
// (the reference issues 147 (G) + 64 (D) power iterations per forward: layers.py:89-111, 151-165):
//   phase 1: v_raw = u W                       (per-row-chunk partial column sums, folded in order)
//   phase 2: t = W v,  v = v_raw / max(|v_raw|, eps);  tt += t^2
//   phase 3: sigma = tt / max(sqrt(tt), eps), u' = t / max(sqrt(tt), eps);  u <- u' and sv <- sigma when
//            training;  write the normalised weight W / sigma in the layout its consumer wants
// and the backward through sigma (sigma = u' W v^T with u', v constant):
//   dW = dWsn / sigma - (<dWsn, W> / sigma^2) u'^T v.
//
// Layer table: int64 [L][SN_FIELDS], offsets in elements relative to the base pointers.
#include "common.h"

#define SN_FIELDS 16
enum { F_W = 0, F_U, F_SV, F_OUT, F_IN, F_TAPS, F_CIN, F_KIND, F_CTX, F_PACK, F_PACK2, F_KPAD, F_KPAD2, F_PART };
// kinds: 0 fp32 [out][in] copy (linear / embedding)
//        1 conv pack: bf16 fwd [out][kpad] (k = tap*cin + c) at F_PACK and bf16 dgrad [cin][kpad2]
//          (k' = (taps-1-tap)*out + o) at F_PACK2 (byte offsets)
//        2 single-channel INPUT conv (weight [C][1][3][3]) -> fp32 [9][C] at F_PACK
//        3 single-channel OUTPUT conv (weight [1][C][3][3]) -> fp32 [9][C] at F_PACK
// ctx layout per layer (fp32, at F_CTX): [0] sigma, [8 .. 8+out) u', [8+out .. 8+out+in) v_raw,
//   [8+out+in .. 8+out+2*in) v, [8+out+2*in .. 8+2*out+2*in) t = W v
#define SN_ROWS 32
#define SN_P3SPLIT 4

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// All reductions are ordered (per-chunk partial sums folded in a fixed order, no float atomics), so the
// normalised weights -- and with them every replica in a data-parallel run -- are bit-reproducible.
__global__ __launch_bounds__(256) void sn_phase1_kernel(const long* __restrict__ tab, const int* __restrict__ blocks,
                                                        const float* __restrict__ params, float* __restrict__ part) {
    const long* L = tab + (long)blocks[2 * blockIdx.x] * SN_FIELDS;
    const int row0 = blocks[2 * blockIdx.x + 1];
    const int out = (int)L[F_OUT], in = (int)L[F_IN];
    const float* W = params + L[F_W];
    const float* u = params + L[F_U];
    float* dst = part + L[F_PART] + (long)(row0 / SN_ROWS) * in;
    const int r1 = min(row0 + SN_ROWS, out);
    for (int i = threadIdx.x; i < in; i += 256) {
        float s = 0.f;
        for (int o0 = row0; o0 < r1; o0 += 8) {      // eight rows in flight (clamped, masked): same order of additions as the plain loop
            float w[8], uu[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int o = min(o0 + k, r1 - 1);
                w[k] = W[(long)o * in + i];
                uu[k] = u[o];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (o0 + k < r1) s += uu[k] * w[k];
        }
        dst[i] = s;
    }
}

// fold the row-chunk partials: one block per (layer, 32-column tile); 8 threads share a column (fixed split, ordered
// combine -> still bit-reproducible), so G.linear's 768 partials per column are 96-long chains instead of 768-long
#define SN_CB 32
__global__ __launch_bounds__(256) void sn_phase1b_kernel(const long* __restrict__ tab, const int* __restrict__ cblocks,
                                                         const float* __restrict__ part, float* __restrict__ ctx) {
    __shared__ float sub[8][SN_CB];
    const long* L = tab + (long)cblocks[2 * blockIdx.x] * SN_FIELDS;
    const int col = threadIdx.x % SN_CB, part_id = threadIdx.x / SN_CB;
    const int i = cblocks[2 * blockIdx.x + 1] + col;
    const int out = (int)L[F_OUT], in = (int)L[F_IN];
    const int chunks = (out + SN_ROWS - 1) / SN_ROWS;
    const int per = (chunks + 7) / 8;
    float s = 0.f;
    if (i < in) {
        const float* src = part + L[F_PART] + i;
        const int c1 = min(chunks, (part_id + 1) * per);
        for (int c = part_id * per; c < c1; ++c) s += src[(long)c * in];
    }
    sub[part_id][col] = s;
    __syncthreads();
    if (part_id == 0 && i < in) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sub[k][col];
        ctx[L[F_CTX] + 8 + out + i] = t;
    }
}

__global__ __launch_bounds__(256) void sn_phase2_kernel(const long* __restrict__ tab, const int* __restrict__ blocks,
                                                        const float* __restrict__ params, float* __restrict__ ctx, float* __restrict__ part,
                                                        float eps) {
    __shared__ float red[4];
    __shared__ float tsq[4];
    const long* L = tab + (long)blocks[2 * blockIdx.x] * SN_FIELDS;
    const int row0 = blocks[2 * blockIdx.x + 1];
    const int out = (int)L[F_OUT], in = (int)L[F_IN];
    const float* W = params + L[F_W];
    float* c = ctx + L[F_CTX];
    const float* vraw = c + 8 + out;
    float* v = c + 8 + out + in;
    float ss = 0.f;
    for (int i = threadIdx.x; i < in; i += 256) ss += vraw[i] * vraw[i];
    ss = block_sum(ss, red);
    const float inv = 1.f / fmaxf(sqrtf(ss), eps);
    if (row0 == 0)
        for (int i = threadIdx.x; i < in; i += 256) v[i] = vraw[i] * inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r1 = min(row0 + SN_ROWS, out);
    float sq = 0.f;
    for (int o = row0 + wave; o < r1; o += 4) {
        float s = 0.f;
        for (int i0 = lane; i0 < in; i0 += 6 * 64) {   // six columns per lane in flight (clamped, masked), same order of additions
            float w[6], vv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i = min(i0 + k * 64, in - 1);
                w[k] = W[(long)o * in + i];
                vv[k] = vraw[i];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (i0 + k * 64 < in) s += w[k] * (vv[k] * inv);
        }
        s = wave_sum(s);
        sq += s * s;
        if (lane == 0) c[8 + out + 2 * in + o] = s;            // t[o]; phase 3 turns it into u'
    }
    // |t|^2 of this block's rows (fixed order): phase 3 then folds out/32 partials instead of re-reading all of t in every
    // block (24576 floats x 768 blocks for G.linear).  The phase-1 partial buffer is free again at this point.
    if (lane == 0) tsq[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[L[F_PART] + row0 / SN_ROWS] = (tsq[0] + tsq[1]) + (tsq[2] + tsq[3]);
}

__global__ __launch_bounds__(256) void sn_phase3_kernel(const long* __restrict__ tab, const int* __restrict__ blocks,
                                                        float* __restrict__ params, float* __restrict__ ctx, const float* __restrict__ part,
                                                        char* __restrict__ pack, float eps, int training) {
    // SN_P3SPLIT blocks share one (layer, 32-row chunk) of the block table: the packing loops below are the long part of this phase (a C = 128
    // 3x3 chunk is 74 K elements), every block takes a contiguous 1 / SN_P3SPLIT of each loop; sigma is recomputed by each (a 24-long fold)
    const int bix = blockIdx.x / SN_P3SPLIT, part_ix = blockIdx.x % SN_P3SPLIT;
    const long* L = tab + (long)blocks[2 * bix] * SN_FIELDS;
    const int row0 = blocks[2 * bix + 1];
    const int out = (int)L[F_OUT], in = (int)L[F_IN], taps = (int)L[F_TAPS], cin = (int)L[F_CIN], kind = (int)L[F_KIND];
    const int kpad = (int)L[F_KPAD], kpad2 = (int)L[F_KPAD2];
    const float* W = params + L[F_W];
    float* c = ctx + L[F_CTX];
    __shared__ float red3[4];
    const float* tvec = c + 8 + out + 2 * in;
    float tt = 0.f;
    const int chunks = (out + SN_ROWS - 1) / SN_ROWS;
    for (int ch = threadIdx.x; ch < chunks; ch += 256) tt += part[L[F_PART] + ch];     // same order in every block
    tt = block_sum(tt, red3);
    const float un = fmaxf(sqrtf(tt), eps);
    const float sigma = tt / un;
    const float isg = 1.f / sigma;
    const int r1 = min(row0 + SN_ROWS, out);
    if (part_ix == 0) {
        for (int o = row0 + threadIdx.x; o < r1; o += 256) {
            const float un_o = tvec[o] / un;
            c[8 + o] = un_o;
            if (training) params[L[F_U] + o] = un_o;
        }
    }
    if (row0 == 0 && part_ix == 0 && threadIdx.x == 0) {
        c[0] = sigma;
        if (training) params[L[F_SV]] = sigma;
    }
    const int rows = r1 - row0;
    // The packing loops below: 8 elements per thread per batch, every weight load (index clamped to a valid element) issued before the first
    // store.  Rolled, each element was a dependent load -> convert -> store round trip: 144 in a row for a C = 128 3x3 layer, 110 us for the
    // phase although it moves 93 MB.  (A block's share of a layer has < 2^31 elements: 32-bit index arithmetic.)
    auto batched = [&](int total_all, auto&& src_index, auto&& put) {
        const int per = (total_all + SN_P3SPLIT - 1) / SN_P3SPLIT;
        const int total = min(total_all, (part_ix + 1) * per);              // this block's share: [part_ix * per, total)
        for (int e0 = part_ix * per + threadIdx.x; e0 < total; e0 += 8 * 256) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = W[src_index(min(e0 + u * 256, total - 1))];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * 256 < total) put(e0 + u * 256, w[u]);
        }
    };
    if (kind == 0) {
        float* dst = (float*)(pack + L[F_PACK]);
        batched(rows * in, [&](int e) { return (long)row0 * in + e; }, [&](int e, float w) { dst[(long)row0 * in + e] = w * isg; });
    } else if (kind == 1) {
        bf16* p1 = (bf16*)(pack + L[F_PACK]);
        bf16* p2 = (bf16*)(pack + L[F_PACK2]);
        const int kreal = taps * cin;
        batched(rows * kpad,                                                 // forward pack (incl. zero padding)
                [&](int e) {
                    const int r = e / kpad, k = min(e - r * kpad, kreal - 1);
                    const int tap = k / cin, ci = k - tap * cin;
                    return (long)(row0 + r) * in + ci * taps + tap;
                },
                [&](int e, float w) {
                    const int r = e / kpad, k = e - r * kpad;
                    p1[(long)(row0 + r) * kpad + k] = f2bf(k < kreal ? w * isg : 0.f);
                });
        batched(rows * in,                                                   // dgrad pack: o fastest -> the 2-byte stores of a wave are
                [&](int e) {                                                 // contiguous runs (the reads hit in cache)
                    const int i = e / rows, o = row0 + (e - i * rows);
                    return (long)o * in + i;
                },
                [&](int e, float w) {
                    const int i = e / rows, o = row0 + (e - i * rows);
                    const int ci = i / taps, tap = i - ci * taps;
                    p2[(long)ci * kpad2 + (taps - 1 - tap) * out + o] = f2bf(w * isg);
                });
        // zero padding columns of the dgrad pack: k' in [taps*out, kpad2), written by the block that owns row 0
        if (row0 == 0 && part_ix == 0 && kpad2 > taps * out) {
            const int padw = kpad2 - taps * out;
            for (long e = threadIdx.x; e < (long)cin * padw; e += 256)
                p2[(e / padw) * kpad2 + taps * out + (e % padw)] = f2bf(0.f);
        }
    } else if (part_ix == 0) {  // 2: weight [C][1][9] -> [9][C];   3: weight [1][C][9] -> [9][C]   (tiny: one of the split blocks)
        float* dst = (float*)(pack + L[F_PACK]);
        for (long e = threadIdx.x; e < (long)rows * in; e += 256) {
            const int o = row0 + (int)(e / in), i = (int)(e % in);
            if (kind == 2) dst[i * out + o] = W[(long)o * in + i] * isg;                 // i = tap, o = channel
            else dst[(i % 9) * cin + (i / 9)] = W[(long)o * in + i] * isg;              // i = c*9 + tap
        }
    }
}

extern "C" int ieagan_sn_forward(const long* tab, const int* blocks, int nblocks, const int* cblocks, int ncblocks,
                                 float* params, float* ctx, float* part, void* pack, float eps, int training, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nblocks <= 0) return 0;
    ProfScope prof("sn_forward", 0.0, 0.0, st);
    hipLaunchKernelGGL(sn_phase1_kernel, dim3(nblocks), dim3(256), 0, st, tab, blocks, (const float*)params, part);
    hipLaunchKernelGGL(sn_phase1b_kernel, dim3(ncblocks), dim3(256), 0, st, tab, cblocks, (const float*)part, ctx);
    hipLaunchKernelGGL(sn_phase2_kernel, dim3(nblocks), dim3(256), 0, st, tab, blocks, (const float*)params, ctx, part, eps);
    hipLaunchKernelGGL(sn_phase3_kernel, dim3(nblocks * SN_P3SPLIT), dim3(256), 0, st, tab, blocks, params, ctx, (const float*)part, (char*)pack, eps, training);
    CHECK_LAUNCH("sn_forward");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward through W / sigma for one layer.
//   gsn : gradient w.r.t. the normalised weight in the consumer's layout
//         kind 0: fp32 [out][in];  kind 1: fp32 [out][kpad] (k = tap*cin + c);  kind 2/3: fp32 [9][C]
//   dW  : fp32, the parameter's own layout ([out][in] row-major == OIHW)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gsn_at(const float* gsn, int kind, int o, int i, int out, int in, int taps, int cin, int kpad) {
    if (kind == 0) return gsn[(long)o * in + i];
    if (kind == 1) {
        const int ci = i / taps, tap = i - ci * taps;
        return gsn[(long)o * kpad + tap * cin + ci];
    }
    if (kind == 2) return gsn[i * out + o];
    return gsn[(i % 9) * cin + (i / 9)];
}

// The two passes of the backward over elements e = start, start + stride, ... < end of one layer, UB elements per thread in flight: every
// load of a batch is issued (index clamped to a valid element) before the first use.  As rolled loops these passes were chains of dependent
// memory round trips -- 74 per pass and block in sn_bwd_stack (0.19 ms for 6 MB), 8 per block in the batched kernels.  Same summation order
// per thread as the rolled form.
template <int UB>
__device__ __forceinline__ float sn_dot(const float* __restrict__ gsn, const float* __restrict__ W, int kind, int out, int in, int taps, int cin,
                                        int kpad, long start, long end, long stride) {
    float s = 0.f;
    for (long e0 = start; e0 < end; e0 += UB * stride) {
        float gv[UB], wv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned e = (unsigned)min(e0 + u * stride, end - 1);      // (a layer has < 2^31 elements: 32-bit division)
            const int o = (int)(e / (unsigned)in), i = (int)(e - (unsigned)o * (unsigned)in);
            gv[u] = gsn_at(gsn, kind, o, i, out, in, taps, cin, kpad);
            wv[u] = W[e];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (e0 + u * stride < end) s += gv[u] * wv[u];
    }
    return s;
}

template <int UB>
__device__ __forceinline__ void sn_apply(const float* __restrict__ gsn, int kind, int out, int in, int taps, int cin, int kpad,
                                         const float* __restrict__ u_, const float* __restrict__ v_, float isg, float coef,
                                         float* __restrict__ dW, bool accumulate, long start, long end, long stride) {
    for (long e0 = start; e0 < end; e0 += UB * stride) {
        float gv[UB], uv[UB], vv[UB], dv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const unsigned e = (unsigned)min(e0 + u * stride, end - 1);
            const int o = (int)(e / (unsigned)in), i = (int)(e - (unsigned)o * (unsigned)in);
            gv[u] = gsn_at(gsn, kind, o, i, out, in, taps, cin, kpad);
            uv[u] = u_[o];
            vv[u] = v_[i];
            dv[u] = accumulate ? dW[e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const long e = e0 + u * stride;
            if (e < end) {
                const float d = gv[u] * isg - coef * uv[u] * vv[u];
                dW[e] = accumulate ? dv[u] + d : d;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sn_bwd_inner_kernel(const float* __restrict__ gsn, const float* __restrict__ W, int kind,
                                                           int out, int in, int taps, int cin, int kpad, float* __restrict__ inner) {
    __shared__ float red[4];
    const long total = (long)out * in;
    float s = sn_dot<4>(gsn, W, kind, out, in, taps, cin, kpad, (long)blockIdx.x * 256 + threadIdx.x, total, (long)gridDim.x * 256);
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(inner, s);
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ gsn, int kind, int out, int in, int taps, int cin,
                                                           int kpad, const float* __restrict__ ctx, const float* __restrict__ inner,
                                                           float* __restrict__ dW, int accumulate) {
    const float sigma = ctx[0];
    const float isg = 1.f / sigma;
    const float coef = inner[0] * isg * isg;
    const float* u = ctx + 8;
    const float* v = ctx + 8 + out + in;
    const long total = (long)out * in;
    sn_apply<4>(gsn, kind, out, in, taps, cin, kpad, u, v, isg, coef, dW, accumulate != 0, (long)blockIdx.x * 256 + threadIdx.x, total,
                (long)gridDim.x * 256);
}

// one launch for a whole (small) layer: ordered block reduction of <gsn, W>, then the apply pass, plus
// the fold of the replicated bias column sums -- replaces memset + 2 kernels + sum(0) + 2 autograd adds
__global__ __launch_bounds__(1024) void sn_bwd_fused_kernel(const float* __restrict__ gsn, const float* __restrict__ W, int kind, int out,
                                                            int in, int taps, int cin, int kpad, const float* __restrict__ ctx,
                                                            float* __restrict__ dW, int accumulate, const float* __restrict__ colsum,
                                                            float* __restrict__ dbias, int bias_accumulate) {
    __shared__ float red[16];
    const long total = (long)out * in;
    const float s = sn_dot<8>(gsn, W, kind, out, in, taps, cin, kpad, threadIdx.x, total, 1024);
    const float inner = block_sum(s, red);
    const float sigma = ctx[0];
    const float isg = 1.f / sigma;
    const float coef = inner * isg * isg;
    const float* u = ctx + 8;
    const float* v = ctx + 8 + out + in;
    sn_apply<8>(gsn, kind, out, in, taps, cin, kpad, u, v, isg, coef, dW, accumulate != 0, threadIdx.x, total, 1024);
    if (colsum != nullptr) {
        const int nb = (kind == 3) ? 1 : out;
        for (int c = threadIdx.x; c < nb; c += 1024) {
            float b = 0.f;
            for (int r = 0; r < STAT_REPL; ++r) b += colsum[r * nb + c];
            dbias[c] = bias_accumulate ? dbias[c] + b : b;
        }
    }
}

__global__ void bias_fold_kernel(const float* __restrict__ colsum, float* __restrict__ dbias, int nb, int accumulate) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nb; c += gridDim.x * blockDim.x) {
        float b = 0.f;
        for (int r = 0; r < STAT_REPL; ++r) b += colsum[r * nb + c];
        dbias[c] = accumulate ? dbias[c] + b : b;
    }
}

extern "C" int ieagan_sn_backward(const float* gsn, const float* W, int kind, int out, int in, int taps, int cin, int kpad,
                                  const float* ctx, float* inner_scratch, float* dW, int accumulate, const float* colsum,
                                  float* dbias, int bias_accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(kind >= 0 && kind <= 3, "sn_backward: bad kind");
    CHECK_ARG(colsum == nullptr || dbias != nullptr, "sn_backward: colsum needs dbias");
    ProfScope prof("sn_backward", 0.0, 0.0, st);
    if ((long)out * in <= 20000) {
        hipLaunchKernelGGL(sn_bwd_fused_kernel, dim3(1), dim3(1024), 0, st, gsn, W, kind, out, in, taps, cin, kpad, ctx, dW, accumulate,
                           colsum, dbias, bias_accumulate);
        CHECK_LAUNCH("sn_backward");
        return 0;
    }
    CHECK_ARG(inner_scratch != nullptr, "sn_backward: large layer needs a zeroed scratch float");
    long blocks = ((long)out * in + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(sn_bwd_inner_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gsn, W, kind, out, in, taps, cin, kpad, inner_scratch);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gsn, kind, out, in, taps, cin, kpad, ctx,
                       (const float*)inner_scratch, dW, accumulate);
    if (colsum != nullptr)
        hipLaunchKernelGGL(bias_fold_kernel, dim3(cdiv(out, 256)), dim3(256), 0, st, colsum, dbias, kind == 3 ? 1 : out, bias_accumulate);
    CHECK_LAUNCH("sn_backward");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Every conv layer of a network in two launches, at the end of a backward pass.  The wgrad kernels of the pass have
// accumulated their gradients w.r.t. the normalised weights (and the replicated bias column sums) into ONE scratch
// arena; this maps them through d(W/sigma)/dW and adds the result to the flat gradient arena.
//   btab  int64[12] per layer: {weight offset (params == grad arena), out, in, taps, cin, kind, kpad, ctx offset,
//                               gsn offset in scratch, colsum offset in scratch or -1, bias offset in grad or -1, bias length}
//   work  int32[2] per block: {layer, chunk of SNB_CHUNK elements};  scratch[0 .. nlayers) = <gsn, W> accumulators (zeroed)
// ------------------------------------------------------------------------------------------------
#define SNB_FIELDS 12
#define SNB_CHUNK 2048

__global__ __launch_bounds__(256) void sn_bwd_batch_inner_kernel(const long* __restrict__ btab, const int* __restrict__ work,
                                                                 const float* __restrict__ params, float* __restrict__ scratch) {
    __shared__ float red[4];
    const int layer = work[2 * blockIdx.x], chunk = work[2 * blockIdx.x + 1];
    const long* L = btab + (long)layer * SNB_FIELDS;
    const float* W = params + L[0];
    const int out = (int)L[1], in = (int)L[2], taps = (int)L[3], cin = (int)L[4], kind = (int)L[5], kpad = (int)L[6];
    const float* gsn = scratch + L[8];
    const long total = (long)out * in;
    const long e1 = min(total, (long)(chunk + 1) * SNB_CHUNK);
    float s = sn_dot<SNB_CHUNK / 256>(gsn, W, kind, out, in, taps, cin, kpad, (long)chunk * SNB_CHUNK + threadIdx.x, e1, 256);
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(scratch + layer, s);
}

__global__ __launch_bounds__(256) void sn_bwd_batch_apply_kernel(const long* __restrict__ btab, const int* __restrict__ work,
                                                                 const float* __restrict__ ctx_all, const float* __restrict__ scratch,
                                                                 float* __restrict__ grad) {
    const int layer = work[2 * blockIdx.x], chunk = work[2 * blockIdx.x + 1];
    const long* L = btab + (long)layer * SNB_FIELDS;
    const int out = (int)L[1], in = (int)L[2], taps = (int)L[3], cin = (int)L[4], kind = (int)L[5], kpad = (int)L[6];
    const float* ctx = ctx_all + L[7];
    const float* gsn = scratch + L[8];
    float* dW = grad + L[0];
    const float isg = 1.f / ctx[0];
    const float coef = scratch[layer] * isg * isg;
    const float* u = ctx + 8;
    const float* v = ctx + 8 + out + in;
    const long total = (long)out * in;
    const long e1 = min(total, (long)(chunk + 1) * SNB_CHUNK);
    sn_apply<SNB_CHUNK / 256>(gsn, kind, out, in, taps, cin, kpad, u, v, isg, coef, dW, true, (long)chunk * SNB_CHUNK + threadIdx.x, e1, 256);
    if (chunk == 0 && L[9] >= 0) {
        const float* colsum = scratch + L[9];
        float* dbias = grad + L[10];
        const int nb = (int)L[11];
        for (int c = threadIdx.x; c < nb; c += 256) {
            float b = 0.f;
            for (int r = 0; r < STAT_REPL; ++r) b += colsum[r * nb + c];
            dbias[c] += b;
        }
    }
}

extern "C" int ieagan_sn_backward_batched(const long* btab, const int* work, int nwork, const float* params, const float* ctx,
                                          float* scratch, float* grad, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nwork <= 0) return 0;
    CHECK_ARG(btab && work && params && ctx && scratch && grad, "sn_backward_batched: null pointer");
    ProfScope prof("sn_backward_batched", 0.0, 0.0, st);
    hipLaunchKernelGGL(sn_bwd_batch_inner_kernel, dim3(nwork), dim3(256), 0, st, btab, work, params, scratch);
    hipLaunchKernelGGL(sn_bwd_batch_apply_kernel, dim3(nwork), dim3(256), 0, st, btab, work, ctx, (const float*)scratch, grad);
    CHECK_LAUNCH("sn_backward_batched");
    return 0;
}

// every ccbn gain / bias SNLinear of a generator at once: block b handles stack layer b.
//   gst [sum out][in] gradient w.r.t. the stacked normalised rows; layers[b] = table row of layer b;
//   dst[b] = element offset of that layer's weight gradient inside grad_base
__global__ __launch_bounds__(1024) void sn_bwd_stack_kernel(const long* __restrict__ tab, const int* __restrict__ layers,
                                                            const long* __restrict__ row0, const long* __restrict__ dst,
                                                            const float* __restrict__ gst, const float* __restrict__ params,
                                                            const float* __restrict__ ctx, float* __restrict__ grad_base, int accumulate) {
    __shared__ float red[16];
    const long* L = tab + (long)layers[blockIdx.x] * SN_FIELDS;
    const int out = (int)L[F_OUT], in = (int)L[F_IN];
    const float* W = params + L[F_W];
    const float* c = ctx + L[F_CTX];
    const float* g = gst + row0[blockIdx.x] * in;
    float* dW = grad_base + dst[blockIdx.x];
    const long total = (long)out * in;
    const float s = sn_dot<8>(g, W, 0, out, in, 1, in, in, threadIdx.x, total, 1024);           // kind 0: g is [out][in]
    const float inner = block_sum(s, red);
    const float isg = 1.f / c[0];
    const float coef = inner * isg * isg;
    const float* u = c + 8;
    const float* v = c + 8 + out + in;
    sn_apply<8>(g, 0, out, in, 1, in, in, u, v, isg, coef, dW, accumulate != 0, threadIdx.x, total, 1024);
}

extern "C" int ieagan_sn_backward_stack(const long* tab, const int* layers, const long* row0, const long* dst, int nlayers,
                                        const float* gst, const float* params, const float* ctx, float* grad_base, int accumulate,
                                        void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nlayers <= 0) return 0;
    ProfScope prof("sn_backward_stack", 0.0, 0.0, st);
    hipLaunchKernelGGL(sn_bwd_stack_kernel, dim3(nlayers), dim3(1024), 0, st, tab, layers, row0, dst, gst, params, ctx, grad_base, accumulate);
    CHECK_LAUNCH("sn_backward_stack");
    return 0;
}
