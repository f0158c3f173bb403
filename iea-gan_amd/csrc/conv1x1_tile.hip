// conv1x1_tile: the 1x1 convolutions that conv1x1_stream does not take -- Cin >= 256, the 2x2-pooled source of a DBlock's conv4
// (model.py:541-557), mid-size feature maps -- as an LDS-tiled implicit GEMM.
//
// conv_gather fetches every operand fragment straight from global memory: per k-step and wave two A fragments (16 pixels x 64 bytes
// each, the pixels Cin*2 bytes apart) and NT weight fragments (16 rows x 64 bytes, Kpad*2 bytes apart) -- the weight fragments again
// in all four waves and in every block.  For 256 -> 128 at 32x96 that is 192 KB of fragment-shaped L2 -> CU traffic per 128-pixel
// block against 96 KB of operands, and the address path (one 128-byte line per 64 useful bytes), not HBM or the MFMA, bounds the
// launch (the non-local attention kernels had the same disease: attention.hip).  Here a block moves its 128 pixels x KC channels and
// its NT*16 x KC weight slice into LDS ONCE per K chunk, in full rows with the prologue applied (per-image BatchNorm scale / shift,
// ReLU, 2x2 average pool of the activated source), and all fragments are ds_read_b128 from XOR-swizzled tiles.
// Same operator contract and epilogue (conv_common.h: bias, residuals, ReLU mask, BatchNorm-backward, statistics) as conv_gather for
// taps == 1; the accumulation order over k is the same, so the two kernels agree bit for bit.
#include "common.h"
#include "conv_args.h"
#include "conv_common.h"

namespace {

// 16-byte chunk c of row r of a [rows][KC] bf16 tile lives at chunk position c ^ key(r): 16 consecutive rows read by the 16 lanes of
// an A / B fragment (one chunk each, rows KC*2 bytes apart) then cover all banks
template <int KC>
__device__ __forceinline__ int tkey(int r) { return KC == 128 ? (r & 15) : KC == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
template <int KC>
__device__ __forceinline__ int toff(int r, int c) { return r * KC + ((c ^ tkey<KC>(r)) << 3); }      // in bf16 elements

template <bool AFF, bool RELU>
__device__ __forceinline__ void pro8(float (&v)[8], const bf16x8& raw, const SrcDesc& s, int n, int c, const float* aff) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(raw[i]);
    xform8<AFF, RELU>(v, s, n, c, aff);
}

}  // namespace

template <bool AFF, bool RELU, int RS, int NT, int KC, bool BNB>
__global__ __launch_bounds__(256, 3) void conv1x1_tile_kernel(ConvArgs a) {
    constexpr int CPR = KC / 8;                               // 16-byte chunks per tile row
    constexpr int EROWS = 16;
    constexpr int ABYTES = 128 * KC * 2, BBYTES = NT * 16 * KC * 2;
    constexpr int EPIB = 4 * EROWS * EpiLds<NT>::LDW * 4;
    constexpr int ASZ = ABYTES > EPIB ? ABYTES : EPIB;        // the epilogue's transpose buffer reuses the A tile
    static_assert(EPIB >= 4 * STATS_SX_FLOATS * 4, "epilogue buffer doubles as the statistics scratch");
    __shared__ __attribute__((aligned(16))) char smem[ASZ + BBYTES];
    __shared__ float red[4 * NT * 16 * 2];
    __shared__ __attribute__((aligned(32))) float aff_s[AFF ? 2 * AFF_MAXC : 8];
    bf16* atile = (bf16*)smem;
    bf16* btile = (bf16*)(smem + ASZ);
    float* epi = (float*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W;
    const long M = (long)a.N * HW;
    // XCD-aware order with the n-tile column as the fast index (see conv1x1_stream.hip): the gy blocks that read the same 128 pixels
    // run side by side on one XCD and share them through its L2
    const int gy = a.Cout / (NT * 16);
    int vid = blockIdx.x;
    {
        const int tot = gridDim.x, q = tot / 8, r = tot % 8, xcd = vid % 8;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + vid / 8;
    }
    const int bx = vid / gy;
    const long m_blk = (long)bx * 128;
    const long m_base = m_blk + wave * 32;                    // this wave's 32 output pixels
    const int n_base = (vid % gy) * NT * 16;
    const int n_img = (int)(m_blk / HW);                      // AFF: the launcher guarantees HW % 128 == 0 (one image per block)
    if (AFF) stage_aff(aff_s, a.src, n_img, a.Cin);           // (the first barrier of the K loop publishes it)
    const float* affp = AFF ? aff_s : nullptr;
    const bf16* xsrc = (const bf16*)a.src.x;

    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int IT = (128 * CPR) / 256;                     // A staging items (16-byte chunks) per thread and K chunk
    constexpr int ITB = (NT * 16 * CPR + 255) / 256;          // weight staging items per thread and K chunk
    for (int kc0 = 0; kc0 < a.Cin; kc0 += KC) {
        __syncthreads();                                      // the previous chunk's fragment reads are done (first trip: aff_s is complete)
        // ---- weights: rows [n_base, n_base + NT*16) x k in [kc0, kc0 + KC)
#pragma unroll
        for (int j = 0; j < ITB; ++j) {
            const int idx = threadIdx.x + j * 256;
            if (idx < NT * 16 * CPR) {
                const int row = idx / CPR, c = idx - row * CPR;
                bf16x8 v = zero8();
                if (n_base + row < a.Cout && kc0 + c * 8 < a.Kpad) v = *(const bf16x8*)((const bf16*)a.w + (long)(n_base + row) * a.Kpad + kc0 + c * 8);
                *(bf16x8*)(btile + toff<KC>(row, c)) = v;
            }
        }
        // ---- activations: 128 pixels x KC channels with the prologue applied; consecutive threads take consecutive 16-byte chunks
        // of consecutive pixels (full 128-byte lines), every load of a batch is requested before the first one is used
        if (RS == 0) {
            bf16x8 raw[IT];
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int idx = threadIdx.x + j * 256;
                const int p = idx / CPR, c = idx - p * CPR;
                const long m = min(m_blk + p, M - 1);
                raw[j] = zero8();
                if (kc0 + c * 8 < a.Cin) raw[j] = *(const bf16x8*)(xsrc + m * a.src.Cx + kc0 + c * 8);      // (Cin = 16: half of a 32-deep k-step is padding)
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int idx = threadIdx.x + j * 256;
                const int p = idx / CPR, c = idx - p * CPR;
                const int ch = kc0 + c * 8;
                bf16x8 o = raw[j];
                if (AFF || RELU) {
                    if (!AFF) {
                        o = relu8(o);
                    } else {
                        float v[8];
                        pro8<AFF, RELU>(v, raw[j], a.src, n_img, min(ch, a.Cin - 8), affp);
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
                    }
                }
                if (m_blk + p >= M || ch >= a.Cin) o = zero8();
                *(bf16x8*)(atile + toff<KC>(p, c)) = o;
            }
        } else {            // RS == 2: conv pixel = mean of the 2x2 block of the ACTIVATED source (AvgPool2d(2), model.py:547-550)
            constexpr int BT = 2;                             // items per batch: 8 loads in flight per thread
#pragma unroll
            for (int j0 = 0; j0 < IT; j0 += BT) {
                bf16x8 raw[BT][4];
#pragma unroll
                for (int jj = 0; jj < BT; ++jj) {
                    const int idx = threadIdx.x + (j0 + jj) * 256;
                    const int p = idx / CPR, c = idx - p * CPR;
                    const long m = min(m_blk + p, M - 1);
                    const int n = (int)(m / HW);
                    const int rem = (int)(m - (long)n * HW);
                    const int h = rem / W, w = rem - h * W;
                    const bf16* b = xsrc + (((long)n * a.src.Hs + 2 * h) * a.src.Ws + 2 * w) * a.src.Cx + min(kc0 + c * 8, a.Cin - 8);
                    if (kc0 + c * 8 < a.Cin) {          // (Cin = 16: half of a 32-deep k-step is padding)
                        raw[jj][0] = *(const bf16x8*)b;
                        raw[jj][1] = *(const bf16x8*)(b + a.src.Cx);
                        raw[jj][2] = *(const bf16x8*)(b + (long)a.src.Ws * a.src.Cx);
                        raw[jj][3] = *(const bf16x8*)(b + (long)a.src.Ws * a.src.Cx + a.src.Cx);
                    } else {
                        raw[jj][0] = raw[jj][1] = raw[jj][2] = raw[jj][3] = zero8();
                    }
                }
#pragma unroll
                for (int jj = 0; jj < BT; ++jj) {
                    const int idx = threadIdx.x + (j0 + jj) * 256;
                    const int p = idx / CPR, c = idx - p * CPR;
                    const int ch = kc0 + c * 8;
                    const int n = AFF ? n_img : 0;
                    float s8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) s8[i] = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[8];
                        pro8<AFF, RELU>(v, raw[jj][q], a.src, n, min(ch, a.Cin - 8), affp);
#pragma unroll
                        for (int i = 0; i < 8; ++i) s8[i] += v[i];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = f2bf(0.25f * s8[i]);
                    if (m_blk + p >= M || ch >= a.Cin) o = zero8();
                    *(bf16x8*)(atile + toff<KC>(p, c)) = o;
                }
            }
        }
        __syncthreads();
        // ---- MFMA over the chunk, every fragment from LDS
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            bf16x8 af[2], bf[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = *(const bf16x8*)(atile + toff<KC>(wave * 32 + mt * 16 + lr, ks * 4 + lg));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const bf16x8*)(btile + toff<KC>(nt * 16 + lr, ks * 4 + lg));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();                                          // the A tile is dead: its memory becomes the epilogue's transpose buffer

    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    const bool need_hw = (a.ra != nullptr && a.ra_rs != 0) || (BNB && a.bnb_scale != nullptr);
    auto pix = [&](int row, long& m, int& n, int& h, int& w) -> bool {
        m = m_base + row;
        n = h = w = 0;
        if (m >= M) return false;
        if (need_hw) {
            n = (int)(m / HW);
            const int rem = (int)(m - (long)n * HW);
            h = rem / W;
            w = rem - h * W;
        }
        return true;
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        auto pixh = [&](int row, long& m, int& n, int& h, int& w) -> bool { return pix(row + 16 * half, m, n, h, w); };
        const f32x4(&sub)[1][NT] = *reinterpret_cast<const f32x4(*)[1][NT]>(&acc[half]);
        conv_epilogue<BNB, NT, 1, (RS != 2)>(a, sub, epi + wave * EROWS * EpiLds<NT>::LDW, n_base, pixh, s1, s2);
    }
    if (a.stats != nullptr) {
        __syncthreads();          // every wave is done with its part of the epilogue buffer, which now serves as fold scratch
        const int event = (a.n_per_event > 0) ? (int)(m_blk / ((long)a.n_per_event * HW)) : 0;
        stats_flush<NT>(a, s1, s2, n_base, red, epi, bx, event);
    }
}

template <bool AFF, bool RELU, int RS, bool BNB>
static int tile_dispatch(const ConvArgs& a, hipStream_t st) {
    const long M = (long)a.N * a.H * a.W;
    const unsigned gx = (unsigned)((M + 127) / 128);
    const int n_events_t = (a.stats != nullptr && a.n_per_event > 0) ? a.N / a.n_per_event : 1;
#define TL(NTV, KCV)                                                                                                                    \
    {                                                                                                                                   \
        CONV_PLAN_POINT((int)(gx / n_events_t), 1)                                                                                      \
        hipLaunchKernelGGL((conv1x1_tile_kernel<AFF, RELU, RS, NTV, KCV, BNB>), dim3(gx * (a.Cout / (NTV * 16))), dim3(256), 0, st, a);     \
        return 1;                                                                                                                       \
    }
#define BY_KC(NTV)                      \
    if (a.Cin <= 32) TL(NTV, 32)        \
    else if (a.Cin == 64) TL(NTV, 64)   \
    else TL(NTV, 128)
    if (a.Cout % 64 == 0) { BY_KC(4) }
    else { BY_KC(2) }
#undef BY_KC
#undef TL
    return 0;
}

// 1 = launched, 0 = not applicable (the caller falls back to conv_gather)
int conv1x1_tile_launch(const ConvArgs& a, hipStream_t st) {
    if (a.taps != 1 || (a.src.rs != 0 && a.src.rs != 2) || a.Cout % 32 != 0 || a.Cin % 8 != 0) return 0;
    if (!(a.Cin <= 32 || a.Cin == 64 || a.Cin % 128 == 0) || a.Kpad != ((a.Cin + 31) / 32) * 32) return 0;
    const long HW = (long)a.H * a.W, M = (long)a.N * HW;
    // tiny maps stay with conv_gather's split-K form (4x12: 512 -> 128 7.3 us there, 14.0 here); from 8x24 up this kernel wins even with 60
    // pixel blocks (8x24: 256 -> 256 19.7 -> 8.7 us, pooled source 40.3 -> 13.3; 128 -> 512 13.2 -> 9.4), pooled sources already at 4x12 (13.0 -> 8.5)
    if (M < (a.src.rs == 2 ? 1024 : 4096)) return 0;
    if (M < 16384 && a.Cin > 256) return 0;                    // four serial K chunks on 120 blocks: in-step 17.7 -> 26.5 us for 512 -> 128 at 8x24
    const bool aff = a.src.scale != nullptr, relu = a.src.relu != 0;
    if (aff && (HW % 128 != 0 || a.Cin > AFF_MAXC)) return 0;  // one image (one BatchNorm table) per 128-pixel block
    if (a.bnb_scale != nullptr) {
        if (aff || relu || a.src.rs != 0) return 0;
        return tile_dispatch<false, false, 0, true>(a, st);
    }
    if (a.src.rs == 0) {
        if (aff && relu) return tile_dispatch<true, true, 0, false>(a, st);
        if (aff) return tile_dispatch<true, false, 0, false>(a, st);
        if (relu) return tile_dispatch<false, true, 0, false>(a, st);
        return tile_dispatch<false, false, 0, false>(a, st);
    }
    if (aff && relu) return tile_dispatch<true, true, 2, false>(a, st);
    if (aff) return tile_dispatch<true, false, 2, false>(a, st);
    if (relu) return tile_dispatch<false, true, 2, false>(a, st);
    return tile_dispatch<false, false, 2, false>(a, st);
}
