// conv_wgrad: dWp[Cout, taps*Cin] += G^T * A over pixel tiles (autograd of F.conv2d w.r.t. the weight, layers.py:197-206); both
// operands staged in LDS in their natural NHWC layout and read K(pixel)-major with ds_read_b64_tr_b16.
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"
#include <stdio.h>

// ------------------------------------------------------------------------------------------------
// conv_wgrad
// ------------------------------------------------------------------------------------------------
#define WG_TH 8
#define WG_TW 16

// K(pixel)-major fragment of one column block from a [pixel rows][channels] 16-bit LDS image: 8 values = rows pix0 .. pix0+3 and
// pix0+hi .. pix0+hi+3 of column (col0 + lr).
// TR: two ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 and receives column (lane&15) of
// the 4 rows).  !TR: eight scalar reads (reference path for the test).
// Bank layout: one half-wave instruction (lane groups lg = 0, 1) reads 8 image rows x 32 bytes.  The callers assign the rows so
// that these are 8 CONSECUTIVE rows (lg 0 -> rows +0..3 / +8..11, lg 1 -> rows +4..7 / +12..15) and pad the row stride to an odd
// multiple of 32 bytes (wg_stride): 8 consecutive rows then cover the 64 banks exactly once.  (With the natural strides the same
// reads were 2-way (C = 16 / 32), 4-way (C = 64) and 8-way (C = 128) bank conflicts.)
template <bool TR>
__device__ __forceinline__ bf16x8 frag_T(const bf16* lds, int stride_elems, int pix0, int hi, int col0, int lr) {
    bf16x8 f;
    if (TR) {
        const int q = lr >> 2, p = lr & 3;
        const bf16* p0 = lds + (pix0 + q) * stride_elems + col0 + 4 * p;
        const bf16* p1 = p0 + hi * stride_elems;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p0);
        const bf16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p1);
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hh[0]; f[5] = hh[1]; f[6] = hh[2]; f[7] = hh[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = lds[(pix0 + (j < 4 ? j : hi + j - 4)) * stride_elems + col0 + lr];
    }
    return f;
}

// row stride (elements) of a C-channel LDS image: an odd multiple of 32 bytes
__host__ __device__ constexpr int wg_stride(int C) { return ((C / 16) % 2 == 0) ? C + 16 : C; }

// Block = 4 waves: MT m-tiles (16 couts each) x 4*NJ n-tile slots ((tap, 16 cins) pairs, dealt round-robin to the waves) over
// `tiles_per_block` pixel tiles of 8 x 16; fp32 atomics into dW at the end (split-K over the blocks).
//   NJ    n-tiles per wave: 4*NJ >= taps*Cin/16 makes ONE block column (gridDim.y == 1) cover the whole weight -- the g / a tiles are
//         then staged once instead of once per n-tile group (C = 32: 5, C = 64: 9; accumulators = MT*NJ*4 VGPRs).
//   CINV  > 0: Cin is a compile-time constant and the RAW 16-byte chunks of the next pixel tile are requested into registers before
//         the MFMAs of the current one (RS != 2).  These launches run few, long blocks (the larger dW is, the fewer pixel splits pay
//         off against the atomic tail), so global-memory latency must be hidden inside the block, not by occupancy.
template <int TAPS, bool AFF, bool RELU, int RS, int MT, bool TR, int NJ, int CINV>
__global__ __launch_bounds__(256, (CINV > 0 ? (CINV == 16 ? 4 : (CINV == 32 ? 3 : 1)) : (TR ? (MT >= 4 ? 2 : 1) : 1))) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int AW = WG_TW + 2 * HALO, AH = WG_TH + 2 * HALO;
    constexpr bool PFW = CINV > 0 && RS != 2;
    constexpr int GC = MT * 16;                   // cout columns of the g tile
    constexpr int GCH = (WG_TH * WG_TW * (GC / 8) + 255) / 256;                     // g chunks per thread
    constexpr int ACH = PFW ? (AH * AW * (CINV / 8) + 255) / 256 : 1;               // a chunks per thread
    const int GS = a.pad_rows ? wg_stride(GC) : GC;      // padded row strides (elements): see frag_T
    bf16* lds_g = (bf16*)smem;                    // [WG_TH*WG_TW][GS]
    bf16* lds_a = lds_g + WG_TH * WG_TW * GS;     // [AH*AW][AS]
    __shared__ __attribute__((aligned(32))) float aff_s[(AFF && PFW) ? 2 * AFF_MAXC : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, Cin = (CINV > 0) ? CINV : a.Cin;
    const int AS = a.pad_rows ? wg_stride(Cin) : Cin;
    const int tiles_w = (W + WG_TW - 1) / WG_TW, tiles_h = (H + WG_TH - 1) / WG_TH;
    const int tiles_img = tiles_w * tiles_h;
    const long tiles_total = (long)a.N * tiles_img;
    const int cout0 = blockIdx.z * GC;
    const int cin_tiles = Cin >> 4;
    const int nt_total = TAPS * cin_tiles;

    // this wave's NJ n-tiles: (tap, cin0); invalid ones are clamped for addressing and skipped in the final accumulation
    int t_dy[NJ], t_dx[NJ], t_c0[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ntg = blockIdx.y * (4 * NJ) + j * 4 + wave;      // round-robin: 9 n-tiles (3x3, C=16) -> 3/2/2/2 per wave
        const int q = (ntg < nt_total) ? ntg : 0;
        const int tap = q / cin_tiles;
        t_c0[j] = (q - tap * cin_tiles) * 16;
        t_dy[j] = (TAPS == 9) ? tap / 3 : 0;          // already offset by +HALO-1 (dy-1+1)
        t_dx[j] = (TAPS == 9) ? tap - (tap / 3) * 3 : 0;
    }
    f32x4 acc[MT][NJ];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool do_colsum = a.colsum != nullptr && blockIdx.y == 0;      // one n-tile group per (pixel, cout) tile does it
    float csum = 0.f;
    const long tile_begin = (long)blockIdx.x * a.tiles_per_block;
    long tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > tiles_total) tile_end = tiles_total;

    auto coords = [&](long tile, int& n, int& h0, int& w0) {
        n = (int)(tile / tiles_img);
        const int tr_ = (int)(tile - (long)n * tiles_img);
        h0 = (tr_ / tiles_w) * WG_TH;
        w0 = (tr_ % tiles_w) * WG_TW;
    };
    // ---- raw requests of one tile (PFW): no transform yet, validity bits for the zero padding
    bf16x8 rg[GCH], ra[ACH];
    unsigned okg = 0, oka = 0;
    auto request = [&](long tile) {
        int n, h0, w0;
        coords(tile, n, h0, w0);
        okg = oka = 0;
#pragma unroll
        for (int j = 0; j < GCH; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int px = idx / (GC / 8), cc = idx - px * (GC / 8);
            const int hh = h0 + px / WG_TW, ww = w0 + px % WG_TW;
            if (idx < WG_TH * WG_TW * (GC / 8) && hh < H && ww < W && cout0 + cc * 8 < a.Cout) {
                rg[j] = *(const bf16x8*)((const bf16*)a.g + (((long)n * H + hh) * W + ww) * a.Cg + cout0 + cc * 8);
                okg |= 1u << j;
            }
        }
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int idx = threadIdx.x + j * 256;
            const int hp = idx / (CINV > 0 ? CINV / 8 : 1), cc = idx - hp * (CINV > 0 ? CINV / 8 : 1);
            const int hh = h0 - HALO + hp / AW, ww = w0 - HALO + hp % AW;
            if (idx < AH * AW * (CINV / 8) && hh >= 0 && hh < H && ww >= 0 && ww < W) {
                const int sh_ = (RS == 1) ? (hh >> 1) : hh, sw_ = (RS == 1) ? (ww >> 1) : ww;
                ra[j] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
                oka |= 1u << j;
            }
        }
    };
    int aff_n = -1;
    if (PFW && tile_begin < tile_end) request(tile_begin);
    for (long tile = tile_begin; tile < tile_end; ++tile) {
        int n, h0, w0;
        coords(tile, n, h0, w0);
        __syncthreads();   // previous tile's fragments consumed
        if (PFW) {
            if (AFF && n != aff_n) {          // block-uniform: this image's scale / shift rows
                stage_aff(aff_s, a.src, n, Cin);
                aff_n = n;
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < GCH; ++j) {
                const int idx = threadIdx.x + j * 256;
                if (idx >= WG_TH * WG_TW * (GC / 8)) continue;
                *(bf16x8*)(lds_g + (idx / (GC / 8)) * GS + (idx % (GC / 8)) * 8) = (okg & (1u << j)) ? rg[j] : zero8();
            }
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const int idx = threadIdx.x + j * 256;
                if (idx >= AH * AW * (CINV / 8)) continue;
                bf16x8 o = zero8();
                if (oka & (1u << j)) {
                    if (!AFF && RELU) {
                        o = relu8(ra[j]);
                    } else if (AFF || RELU) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = bf2f(ra[j][i]);
                        xform8<AFF, RELU>(v, a.src, n, (idx % (CINV > 0 ? CINV / 8 : 1)) * 8, aff_s);
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                    } else {
                        o = ra[j];
                    }
                }
                *(bf16x8*)(lds_a + (idx / (CINV > 0 ? CINV / 8 : 1)) * AS + (idx % (CINV > 0 ? CINV / 8 : 1)) * 8) = o;
            }
        } else {
            // ---- stage the g tile (128 pixels x GC couts) and the a tile (+halo, prologue fused) in batches of independent 16-byte
            // loads per thread: coordinates clamped into the image (every address valid, no branch around a load), the padding /
            // ragged edge zeroed through a mask afterwards.  (As rolled load -> wait -> ds_write loops these kept ONE 16-byte load per
            // thread in flight: 20 serial memory round trips per 82 KB tile at Cin = 256, 16-18 us per tile.)
            constexpr int GTOT = WG_TH * WG_TW * (GC / 8);
            // opaque per tile: the chunk -> (pixel, channel) index math below is tile-invariant, and hoisted out of the tile loop it
            // takes ~50 registers next to the accumulators (spills whose reloads share the counter of the very loads batched here)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            bf16x8 vg[GCH];
            unsigned okb = 0;
#pragma unroll
            for (int j = 0; j < GCH; ++j) {
                const int idx = tid + j * 256;
                const int px = idx / (GC / 8), cc = idx - px * (GC / 8);
                const int hh = h0 + px / WG_TW, ww = w0 + px % WG_TW;
                const bool ok = idx < GTOT && hh < H && ww < W && cout0 + cc * 8 < a.Cout;
                vg[j] = *(const bf16x8*)((const bf16*)a.g + (((long)n * H + min(hh, H - 1)) * W + min(ww, W - 1)) * a.Cg + cout0 + (ok ? cc * 8 : 0));
                okb |= (unsigned)ok << j;
            }
            constexpr int SB = (RS == 2) ? (MT == 8 ? 1 : 4) : (MT == 8 ? 4 : 8);      // chunks per thread and batch (RS == 2: four source pixels per chunk;
                                                                              // MT == 8: 128 accumulator registers stay live across the staging)
            constexpr int NQ = (RS == 2) ? 4 : 1;
            const int cpp = Cin >> 3, atot = AH * AW * cpp;
            for (int b0 = 0; b0 < atot; b0 += SB * 256) {
                bf16x8 raw[SB][NQ];
                unsigned oka2 = 0;
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int idx = b0 + j * 256 + tid;
                    const int idc = min(idx, atot - 1);
                    const int hp = idc / cpp, cc = idc - hp * cpp;
                    const int hh = h0 - HALO + hp / AW, ww = w0 - HALO + hp % AW;
                    if (idx < atot && hh >= 0 && hh < H && ww >= 0 && ww < W) oka2 |= 1u << j;
                    const int hc = min(max(hh, 0), H - 1), wc = min(max(ww, 0), W - 1);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int sh_ = (RS == 2) ? 2 * hc + (q >> 1) : ((RS == 1) ? (hc >> 1) : hc);
                        const int sw_ = (RS == 2) ? 2 * wc + (q & 1) : ((RS == 1) ? (wc >> 1) : wc);
                        raw[j][q] = *(const bf16x8*)((const bf16*)a.src.x + (((long)n * a.src.Hs + sh_) * a.src.Ws + sw_) * a.src.Cx + cc * 8);
                    }
                }
                if (b0 == 0) {                            // the g chunks went out first: store them behind the a requests
#pragma unroll
                    for (int j = 0; j < GCH; ++j) {
                        const int idx = tid + j * 256;
                        if (idx < GTOT) *(bf16x8*)(lds_g + (idx / (GC / 8)) * GS + (idx % (GC / 8)) * 8) = (okb & (1u << j)) ? vg[j] : zero8();
                    }
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int idx = b0 + j * 256 + tid;
                    if (idx >= atot) continue;
                    const int idc = min(idx, atot - 1);       // (== idx here: the same expression as above, one division for both)
                    const int hp = idc / cpp, cc = idc - hp * cpp;
                    bf16x8 o = zero8();
                    if (oka2 & (1u << j)) {
                        if (RS != 2 && !AFF && !RELU) {
                            o = raw[j][0];
                        } else if (RS != 2 && !AFF && RELU) {
                            o = relu8(raw[j][0]);
                        } else {
                            float acc8[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc8[i] = 0.f;
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                float v[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) v[i] = bf2f(raw[j][q][i]);
                                xform8<AFF, RELU>(v, a.src, n, cc * 8);
#pragma unroll
                                for (int i = 0; i < 8; ++i) acc8[i] += v[i];
                            }
#pragma unroll
                            for (int i = 0; i < 8; ++i) o[i] = f2bf((RS == 2 ? 0.25f : 1.f) * acc8[i]);
                        }
                    }
                    *(bf16x8*)(lds_a + hp * AS + cc * 8) = o;
                }
            }
        }
        __syncthreads();
        if (PFW && tile + 1 < tile_end) request(tile + 1);        // in flight during the MFMAs below
        if (do_colsum) {      // bias gradient: column sums of the staged g tile (thread = column t % GC, pixel phase t / GC)
            // (unrolled: as a rolled loop every one of the GC/2 reads was an LDS round trip of its own -- longer than the MFMA loop)
            const bf16* colp = lds_g + (threadIdx.x / GC) * GS + (threadIdx.x % GC);
#pragma unroll 8
            for (int i = 0; i < GC / 2; ++i) csum += bf2f(colp[i * (256 / GC) * GS]);
        }
        // ---- 4 k-steps of 32 pixels (two tile rows each) x NJ n-tiles, as ONE software pipeline over the 4*NJ (k-step, n-tile)
        // pairs: the B fragment of pair s+2 (and, at the start of a k-step, the A fragments of the NEXT k-step) is requested
        // before the MT MFMAs of pair s issue.  These launches run 1-2 waves per SIMD (the accumulators take the register file),
        // so LDS latency has to be hidden inside the wave: with the plain read -> wait -> MFMA order every wave spent ~80 % of
        // its cycles waiting (rocprofv3 SQ_WAIT_ANY + SQ_WAIT_INST_ANY on the C = 64 layers).
        // K index (lg, i) of a k-step <-> pixel (tile row 2*ks + (lg >> 1), column (lg & 1)*4 + (i < 4 ? i : i + 4)): any bijection
        // works as long as both operands use it; this one makes a half-wave read 8 consecutive image rows (see frag_T).
        {
            constexpr int STEPS = 4 * NJ;
            const int colk = (lg & 1) * 4;
            auto ldA = [&](int ks, bf16x8(&x)[MT]) {
                const int row = 2 * ks + (lg >> 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x[mt] = frag_T<TR>(lds_g, GS, row * WG_TW + colk, 8, mt * 16, lr);
            };
            auto ldB = [&](int ks, int j) -> bf16x8 {
                const int row = 2 * ks + (lg >> 1);
                return frag_T<TR>(lds_a, AS, (row + t_dy[j]) * AW + colk + t_dx[j], 8, t_c0[j], lr);
            };
            bf16x8 afr[2][MT], bfr[3];
            ldA(0, afr[0]);
            bfr[0] = ldB(0, 0);
            if (STEPS > 1) bfr[1] = ldB(1 / NJ, 1 % NJ);
#pragma unroll
            for (int s2 = 0; s2 < STEPS; ++s2) {
                const int ks = s2 / NJ, j = s2 % NJ;
                if (s2 + 2 < STEPS) bfr[(s2 + 2) % 3] = ldB((s2 + 2) / NJ, (s2 + 2) % NJ);
                if (j == 0 && ks + 1 < 4) ldA(ks + 1, afr[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks & 1][mt], bfr[s2 % 3], acc[mt][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (do_colsum) {          // fold the 256 / GC pixel phases of a column through LDS, one atomic per column per block
        __syncthreads();
        float* red = (float*)smem;
        red[threadIdx.x] = csum;
        __syncthreads();
        if (threadIdx.x < GC && cout0 + threadIdx.x < a.Cout) {
            float t = 0.f;
            for (int ph = 0; ph < 256 / GC; ++ph) t += red[ph * GC + threadIdx.x];
            atomicAdd(a.colsum + (long)(blockIdx.x % STAT_REPL) * a.Cout + cout0 + threadIdx.x, t);
        }
    }
    // ---- accumulate into dWp[cout][k].  The four waves own the n-tiles 4j .. 4j+3 of slot group j, i.e. the 64 CONSECUTIVE k
    // columns [64j, 64j+64) (k = 16 * n-tile index): transposed through LDS, every atomic wave-instruction adds 256 contiguous
    // bytes of one dW row -- the access shape float atomics retire at full rate (four 64-byte pieces in four rows, what the
    // accumulator layout gives directly, are several times slower: MI355X_MICROARCH.md, Global float atomics).
    __syncthreads();
    float* T = (float*)smem;          // [GC rows][64 k]   (GC*256 bytes = the g-tile region)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(mt * 16 + lg * 4 + r) * 64 + wave * 16 + lr] = acc[mt][j][r];
        __syncthreads();
        const int ntg = blockIdx.y * (4 * NJ) + j * 4 + (lane >> 4);       // n-tile of this lane's column block
        const int kcol = ntg * 16 + (lane & 15);
        if (ntg < nt_total) {
            if (a.partials != nullptr) {          // two-stage accumulation: this pixel split's slab, plain coalesced stores
                float* slab = a.partials + (long)blockIdx.x * a.Cout * a.Kpad;
#pragma unroll 4
                for (int row = wave; row < GC; row += 4)
                    if (cout0 + row < a.Cout) slab[(long)(cout0 + row) * a.Kpad + kcol] = T[row * 64 + lane];
            } else {
#pragma unroll 4
                for (int row = wave; row < GC; row += 4)
                    if (cout0 + row < a.Cout) atomicAdd(a.dw + (long)(cout0 + row) * a.Kpad + kcol, T[row * 64 + lane]);
            }
        }
        __syncthreads();
    }
}

// n-tiles per wave and compile-time Cin of the specialised 3x3 variants: whole weight in one block column + next-tile prefetch
template <int TAPS, bool AFF, bool RELU, int RS, bool TR>
static bool launch_wgrad_special(const WgradArgs& a, hipStream_t st, int mt, dim3 grid, size_t lds) {
    if (!(TAPS == 9 && TR && RS != 2)) return false;
#define WG_S(MTV, NJV, CINV)                                                                                                    \
    {                                                                                                                           \
        hipLaunchKernelGGL((conv_wgrad_kernel<TAPS, AFF, RELU, RS, MTV, TR, NJV, CINV>), grid, dim3(256), lds, st, a);          \
        return true;                                                                                                            \
    }
    if constexpr (TAPS == 9 && TR && RS != 2 && ((!AFF && RELU) || (AFF && RELU))) {      // the prologues the 3x3 layers of D / G have
        if (a.Cin == 16 && mt == 1) WG_S(1, 3, 16)
        if (a.Cin == 32 && mt == 2) WG_S(2, 5, 32)
        if (a.Cin == 64 && mt == 4) WG_S(4, 9, 64)
        if (a.Cin == 128 && mt == 8) WG_S(8, 4, 128)
    }
#undef WG_S
    return false;
}

template <int TAPS, bool AFF, bool RELU, int RS, bool TR>
static void launch_wgrad_mt(const WgradArgs& a, hipStream_t st, int mt, dim3 grid, size_t lds, bool special) {
    if (special && launch_wgrad_special<TAPS, AFF, RELU, RS, TR>(a, st, mt, grid, lds)) return;
    const bool one = TAPS == 1 && TR && a.Cin <= 64;       // <= 4 n-tiles in total: every wave owns at most one
#define WG_L(MTV)                                                                                                       \
    {                                                                                                                   \
        if (TAPS == 1 && TR && one) hipLaunchKernelGGL((conv_wgrad_kernel<TAPS, AFF, RELU, RS, MTV, TR, (TAPS == 1 && TR) ? 1 : 4, 0>), grid, dim3(256), lds, st, a); \
        else hipLaunchKernelGGL((conv_wgrad_kernel<TAPS, AFF, RELU, RS, MTV, TR, 4, 0>), grid, dim3(256), lds, st, a);  \
    }
    switch (mt) {
        case 1: WG_L(1) break;
        case 2: WG_L(2) break;
        case 4: WG_L(4) break;
        default: WG_L(8) break;
    }
#undef WG_L
}

template <int TAPS, int RS, bool TR>
static void launch_wgrad_pro(const WgradArgs& a, hipStream_t st, int mt, dim3 grid, size_t lds, bool special) {
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (aff && relu) launch_wgrad_mt<TAPS, true, true, RS, TR>(a, st, mt, grid, lds, special);
    else if (aff) launch_wgrad_mt<TAPS, true, false, RS, TR>(a, st, mt, grid, lds, special);
    else if (relu) launch_wgrad_mt<TAPS, false, true, RS, TR>(a, st, mt, grid, lds, special);
    else launch_wgrad_mt<TAPS, false, false, RS, TR>(a, st, mt, grid, lds, special);
}

// Second stage of the two-stage accumulation: dw[row][k] += sum over the S pixel-split slabs, k < K.  A block is 64 float4 columns
// of the FLAT [Cout][Kpad] slab x 4 slab lanes (the first form mapped a 256-thread block onto the K/4 float4 columns of ONE row:
// 4..36 live threads per block for the 1x1 / C = 16 layers, each walking 32 slabs serially); the four lanes meet in LDS, slab
// chunks (blockIdx.y, at most 32 so that no address sees more than 32 atomics) in float atomics.
#define WG_TWO_STAGE_MIN_BYTES (8L << 20)
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw, int S, int chunk, int n4, int kp4, int K,
                                                  int bx, int by, int ny) {
    __shared__ f32x4 red[3][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = bx * 64 + tx;
    const bool live = i < n4 && (i % kp4) * 4 < K;            // Kpad padding columns are not written by the first stage
    const int s0 = by * chunk, s1 = min(s0 + chunk, S);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const f32x4* p = (const f32x4*)part + i;
#pragma unroll 4
        for (int s = s0 + ty; s < s1; s += 4) acc += p[(long)s * n4];
    }
    if (ty > 0) red[ty - 1][tx] = acc;
    __syncthreads();
    if (ty > 0 || !live) return;
    acc += red[0][tx] + red[1][tx] + red[2][tx];
    float* d = dw + (long)i * 4;
    if (ny == 1) {
        f32x4 v = *(f32x4*)d;
        *(f32x4*)d = v + acc;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(d + j, acc[j]);
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int chunk, int n4, int kp4, int K) {
    wgrad_reduce_body(part, dw, S, chunk, n4, kp4, K, blockIdx.x, blockIdx.y, gridDim.y);
}

// slab-chunk split of one fold: ~2 blocks per CU, at least 16 slabs per block, at most 32 adders per address
static void wgrad_reduce_geometry(int S, int Cout, int Kpad, int& bx, int& zc, int& chunk) {
    const int n4 = Cout * (Kpad / 4);
    bx = (n4 + 63) / 64;
    zc = (512 + bx - 1) / bx;
    if (zc > 32) zc = 32;
    if (zc > (S + 15) / 16) zc = (S + 15) / 16;
    if (zc < 1) zc = 1;
    chunk = (S + zc - 1) / zc;
    zc = (S + chunk - 1) / chunk;
}

// second stage, also for other producers of partial slabs (conv1x1_bwd.hip)
int wgrad_reduce_launch(const float* part, float* dw, int S, int Cout, int Kpad, int K, hipStream_t st) {
    int bx, zc, chunk;
    wgrad_reduce_geometry(S, Cout, Kpad, bx, zc, chunk);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, zc), dim3(256), 0, st, part, dw, S, chunk, Cout * (Kpad / 4), Kpad / 4, K);
    CHECK_LAUNCH("wgrad_reduce");
    return 0;
}

// Several folds in ONE launch (ieagan_wgrad_reduce_batched): the whole-backward kernels of a pass leave their slabs (IEAGAN_BWD_NO_REDUCE) and
// the pass folds them all at its end, right before the spectral-norm backward that first reads dW -- instead of one ~8 us launch (and a
// drained chip) behind each of the 36 kernels.  The item table travels by value in the kernel argument.
#define WGR_MAX_ITEMS 32
struct WgrBatch {
    const float* part[WGR_MAX_ITEMS];
    float* dw[WGR_MAX_ITEMS];
    int S[WGR_MAX_ITEMS], chunk[WGR_MAX_ITEMS], n4[WGR_MAX_ITEMS], kp4[WGR_MAX_ITEMS], K[WGR_MAX_ITEMS], bx[WGR_MAX_ITEMS], zc[WGR_MAX_ITEMS];
    int blk0[WGR_MAX_ITEMS + 1];
    int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(WgrBatch b) {
    int it = 0;
    while (it + 1 < b.n && (int)blockIdx.x >= b.blk0[it + 1]) ++it;          // block-uniform
    const int local = blockIdx.x - b.blk0[it];
    wgrad_reduce_body(b.part[it], b.dw[it], b.S[it], b.chunk[it], b.n4[it], b.kp4[it], b.K[it], local % b.bx[it], local / b.bx[it], b.zc[it]);
}

extern "C" int ieagan_wgrad_reduce_batched(const ieagan_reduce_item* items, int n, void* stream) {
    CHECK_ARG(items != nullptr && n >= 1, "wgrad_reduce_batched: empty item list");
    hipStream_t st = (hipStream_t)stream;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const ieagan_reduce_item& q = items[i];
        CHECK_ARG(q.partials != nullptr && q.dw != nullptr && q.S >= 1 && q.Cout >= 1 && q.Kpad % 4 == 0 && q.K <= q.Kpad, "wgrad_reduce_batched: bad item %d", i);
        bytes += 4.0 * ((double)q.S + 2.0) * q.Cout * q.Kpad;
    }
    ProfScope prof("wgrad_reduce", 0.0, bytes, st);
    for (int i0 = 0; i0 < n; i0 += WGR_MAX_ITEMS) {
        WgrBatch b;
        b.n = n - i0 < WGR_MAX_ITEMS ? n - i0 : WGR_MAX_ITEMS;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const ieagan_reduce_item& q = items[i0 + i];
            int bx, zc, chunk;
            wgrad_reduce_geometry(q.S, q.Cout, q.Kpad, bx, zc, chunk);
            b.part[i] = q.partials; b.dw[i] = q.dw; b.S[i] = q.S; b.chunk[i] = chunk; b.n4[i] = q.Cout * (q.Kpad / 4); b.kp4[i] = q.Kpad / 4;
            b.K[i] = q.K; b.bx[i] = bx; b.zc[i] = zc; b.blk0[i] = blocks;
            blocks += bx * zc;
        }
        b.blk0[b.n] = blocks;
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blocks), dim3(256), 0, st, b);
    }
    CHECK_LAUNCH("wgrad_reduce_batched");
    return 0;
}

extern "C" int ieagan_wgrad_reduce(const float* partials, float* dw, int S, int Cout, int Kpad, int K, void* stream) {
    CHECK_ARG(partials != nullptr && dw != nullptr && S >= 1 && Cout >= 1 && Kpad % 4 == 0 && K <= Kpad, "wgrad_reduce: bad arguments");
    ProfScope prof("wgrad_reduce", 0.0, 4.0 * ((double)S + 2.0) * Cout * Kpad, (hipStream_t)stream);
    return wgrad_reduce_launch(partials, dw, S, Cout, Kpad, K, (hipStream_t)stream);
}

struct WgradPlan {
    int mt, gx, gy, gz, tpb, pad_rows;
    bool special;
    size_t lds;
    long ws_elems;          // partial-slab workspace worth using (0: direct atomics)
};

static int wgrad_plan(const WgradArgs& a, int use_tr, WgradPlan& p);

extern "C" long ieagan_conv_wgrad_workspace(const ieagan_wgrad_desc* d, int use_tr_read) {
    WgradPlan p;
    if (d == nullptr || wgrad_plan(*d, use_tr_read, p) != 0) return 0;
    return p.ws_elems;
}

// Launch geometry of one weight-gradient call (shared by the launcher and the workspace query).
static int wgrad_plan(const WgradArgs& a, int use_tr, WgradPlan& p) {
    CHECK_ARG(a.taps == 1 || a.taps == 9, "wgrad: taps must be 1 or 9");
    CHECK_ARG(a.Cin % 16 == 0 && a.Cout % 8 == 0, "wgrad: Cin %% 16 and Cout %% 8 required (%d,%d)", a.Cin, a.Cout);
    CHECK_ARG(a.Kpad >= a.taps * a.Cin, "wgrad: bad Kpad");
    CHECK_ARG(a.Cg >= a.Cout && a.Cg % 8 == 0, "wgrad: bad g channel stride %d", a.Cg);
    if (a.src.rs == 1) CHECK_ARG(a.H == 2 * a.src.Hs && a.W == 2 * a.src.Ws, "wgrad: upsample geometry mismatch");
    if (a.src.rs == 2) CHECK_ARG(2 * a.H == a.src.Hs && 2 * a.W == a.src.Ws, "wgrad: pool geometry mismatch");
    if (a.src.rs == 0) CHECK_ARG(a.H == a.src.Hs && a.W == a.src.Ws, "wgrad: geometry mismatch");
    // cout chunking: MT m-tiles per block (<= 8)
    int mt = 8;
    if (a.Cout % 128 != 0) mt = (a.Cout % 64 == 0) ? 4 : (a.Cout % 32 == 0) ? 2 : 1;
    p.mt = mt;
    p.gz = (a.Cout + mt * 16 - 1) / (mt * 16);
    const int nt_total = a.taps * (a.Cin / 16);
    // specialised 3x3 variants (conv_wgrad_kernel: NJ, CINV): the whole weight in ONE block column, next tile prefetched
    const bool relu_pro = a.src.relu != 0;
    p.special = use_tr && a.taps == 9 && a.src.rs != 2 && relu_pro && a.Cout == a.Cin &&
                ((a.Cin == 16 && mt == 1) || (a.Cin == 32 && mt == 2) || (a.Cin == 64 && mt == 4) || (a.Cin == 128 && mt == 8)) &&
                (a.src.scale == nullptr || a.Cin <= AFF_MAXC);
    const int slots = p.special ? 4 * (a.Cin == 16 ? 3 : a.Cin == 32 ? 5 : a.Cin == 64 ? 9 : 4) : 16;
    p.gy = (nt_total + slots - 1) / slots;
    const long tiles = (long)a.N * ((a.H + WG_TH - 1) / WG_TH) * ((a.W + WG_TW - 1) / WG_TW);
    // Pixel splits.  With a partial-slab workspace (two-stage accumulation) the blocks end with plain stores: split for parallelism
    // alone (~2 blocks per CU and dW column group).  Without it every block ends with one float atomicAdd per owned dW element,
    // which the chip retires at well under 1 TB/s: the larger dW is, the fewer splits pay off.
    const long dw_elems = (long)a.Cout * a.taps * a.Cin;
    const bool two_stage = a.partials != nullptr;
    long blocks_goal = dw_elems <= 4096 ? 2048 : dw_elems <= 12288 ? 1024 : (dw_elems <= 65536 || tiles > 128) ? 512 : 256;
    if (p.special && a.Cin <= 64) blocks_goal = a.Cin == 16 ? 1024 : (a.Cin == 32 ? 512 : 256);
    long target = blocks_goal / (p.gy * p.gz);
    if (target < 64) target = 64;
    {
        // Traffic rule: every pixel split costs one slab of dW written and read back.  The step as a whole is HBM-bound and these
        // launches run beside the main stream, so what they cost the step is the bytes they move, not their stand-alone latency
        // (measured: skipping them all saves 3.3 ms = their share of the step's traffic): no more splits than make the slabs as heavy
        // as the operands, but no block longer than `maxt` tiles.
        constexpr double tf = 2.0;          // slabs (written + read back) up to twice the operand bytes: in-step scan {0.5, 1, 2, 3, 4, 8} x
        constexpr long maxt = 16;           // {8, 16, 32, 64} serial tiles, 40-step A/B on one box: 33.64-33.78 ms against 33.96-33.98 without the rule
        const double in_bytes = 2.0 * a.N * ((double)a.src.Hs * a.src.Ws * a.Cin + (double)a.H * a.W * a.Cout);
        const double dw_bytes = 4.0 * a.Cout * a.Kpad;
        long cap = (long)(tf * in_bytes / (2.0 * dw_bytes));
        const long pmin = (tiles + maxt - 1) / maxt;
        if (cap < pmin) cap = pmin;
        if (cap < 2) cap = 2;
        if (target > cap) target = cap;
    }
    int tpb = (int)((tiles + target - 1) / target);
    if (tpb < 1) tpb = 1;
    p.tpb = tpb;
    p.gx = (int)((tiles + tpb - 1) / tpb);
    const int halo = (a.taps == 9) ? 1 : 0;
    size_t lds = (size_t)WG_TH * WG_TW * wg_stride(mt * 16) * 2 + (size_t)(WG_TH + 2 * halo) * (WG_TW + 2 * halo) * wg_stride(a.Cin) * 2;
    p.pad_rows = lds <= 160 * 1024;              // conflict-free padded LDS rows unless the tiles then exceed the 160 KB of a CU
    if (!p.pad_rows) lds = (size_t)WG_TH * WG_TW * mt * 16 * 2 + (size_t)(WG_TH + 2 * halo) * (WG_TW + 2 * halo) * a.Cin * 2;
    CHECK_ARG(lds <= 160 * 1024, "wgrad: LDS request %zu too large", lds);
    p.lds = lds;
    // workspace worth using: the slabs of the split count a two-stage launch would take, when the atomic volume of the direct
    // form is large (>= 2 MB of adds) -- small dW x few splits stays with the direct atomics (one launch less)
    {
        const long direct_bytes = (long)p.gx * dw_elems * 4;
        p.ws_elems = (two_stage || (direct_bytes >= WG_TWO_STAGE_MIN_BYTES && p.gx > 1)) ? (long)p.gx * a.Cout * a.Kpad : 0;
    }
    return 0;
}

int conv_wgrad_launch(const WgradArgs& a0, hipStream_t st, int use_tr) {
    WgradArgs a = a0;
    WgradPlan p;
    const int prc = wgrad_plan(a, use_tr, p);
    if (prc != 0) return prc;
    const int mt = p.mt;
    const bool special = p.special;
    a.tiles_per_block = p.tpb;
    a.pad_rows = p.pad_rows;
    const size_t lds = p.lds;
    const double flops = 2.0 * a.N * a.H * a.W * (double)a.Cout * a.taps * a.Cin;
    const double bytes = 2.0 * a.N * ((double)a.src.Hs * a.src.Ws * a.Cin + (double)a.H * a.W * a.Cout);
    char tag[64] = "";
    if (prof_tags_on()) snprintf(tag, sizeof(tag), "ci%d co%d %dx%d rs%d grid%dx%dx%d%s", a.Cin, a.Cout, a.H, a.W, a.src.rs, p.gx, p.gy, p.gz, a.partials ? " 2st" : "");
    ProfScope prof(a.taps == 9 ? "conv3x3_wgrad" : "conv1x1_wgrad", flops, bytes, st, tag);
    dim3 grid(p.gx, p.gy, p.gz);
#define WG_DISPATCH(TR)                                                                  \
    if (a.taps == 9) {                                                                   \
        if (a.src.rs == 0) launch_wgrad_pro<9, 0, TR>(a, st, mt, grid, lds, special);             \
        else if (a.src.rs == 1) launch_wgrad_pro<9, 1, TR>(a, st, mt, grid, lds, special);        \
        else { ieagan_set_error("wgrad: 3x3 with pooled source not instantiated"); return IEAGAN_EINVAL; } \
    } else {                                                                             \
        if (a.src.rs == 0) launch_wgrad_pro<1, 0, TR>(a, st, mt, grid, lds, false);             \
        else if (a.src.rs == 2) launch_wgrad_pro<1, 2, TR>(a, st, mt, grid, lds, false);        \
        else { ieagan_set_error("wgrad: 1x1 with upsampled source not instantiated"); return IEAGAN_EINVAL; } \
    }
    if (use_tr) { WG_DISPATCH(true) } else { WG_DISPATCH(false) }
#undef WG_DISPATCH
    CHECK_LAUNCH("conv_wgrad");
    if (a.partials != nullptr) {          // second stage: fold the gx slabs into dw
        const int rc = wgrad_reduce_launch(a.partials, a.dw, p.gx, a.Cout, a.Kpad, a.taps * a.Cin, st);
        if (rc != 0) return rc;
    }
    return 0;
}
