// Batched modified-orthogonal regularisation for every weight matrix of a network (reference utils/__init__.py:843-859):
//     grad_W += 2*strength * ((W W^T) (.) (1 - I)) W
// One call serves all layers of the flat fp32 arena through a layer table; two launches (Gram tiles, apply tiles).
//
//   R <= K ("row" form) : P = W W^T  [R,R] with the diagonal dropped,   out = P W
//   R  > K ("col" form) : Q = W^T W  [K,K],                              out = W Q - diag(|w_i|^2) W
// (the second form keeps G.linear's Gram at 256x256 instead of 24576x24576).  All arithmetic fp32.
//
// Layer table (int64 x 4 per layer): { weight offset in the arena (floats), R, K, Gram offset in the scratch (floats) }.
// Tile lists (int32 x 4 per tile):   { layer, tile_i, tile_j, k-split index }; 64x64 output tiles, reduce chunks of 16.
#include "common.h"

namespace {

constexpr int TM = 64;        // tile edge
constexpr int TK = 32;        // reduce chunk (two 16-deep MFMA groups)
constexpr int LDK = TK + 4;   // padded LDS row: 16 consecutive rows of a fragment read land on distinct banks
constexpr int KSPLIT = 512;   // reduce length per Gram work item
constexpr int NI = (TM * TK) / 256;      // staged elements per thread, operand and chunk

typedef float f4 __attribute__((ext_vector_type(4)));

struct Operand {              // element (m, k) of a [M x Kd] operand lives at p[m * rs + k * cs]
    const float* p;
    long rs, cs;
    int M, Kd;
};

// One [TM x TK] chunk of an operand (rows m0.., reduce k0..): `fetch` requests this thread's NI elements into registers (out-of-range
// -> 0), `put` writes them to LDS as s[m][k].  Split so that the NEXT chunk is in flight while the MFMAs of the current one run: as
// one load -> LDS -> barrier -> MFMA sequence per 16-deep chunk every chunk paid a full memory round trip (128 of them per work item:
// the two launches took 0.41 ms at the end of the generator phase).
__device__ __forceinline__ void elem(const Operand& o, int e, int& m, int& k) {
    if (o.cs == 1) { m = e / TK; k = e % TK; } else { k = e / TM; m = e % TM; }      // consecutive threads along the contiguous axis
}
__device__ __forceinline__ void fetch(const Operand& o, int m0, int k0, int kend, float (&r)[NI], int tid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int m, k;
        elem(o, tid + 256 * i, m, k);
        const int gm = m0 + m, gk = k0 + k;
        const bool ok = gm < o.M && gk < kend;
        const float v = o.p[ok ? (long)gm * o.rs + (long)gk * o.cs : 0];          // (address always valid: no branch around the load)
        r[i] = ok ? v : 0.f;
    }
}
__device__ __forceinline__ void put(const Operand& o, const float (&r)[NI], float (*s)[LDK], int tid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int m, k;
        elem(o, tid + 256 * i, m, k);
        s[m][k] = r[i];
    }
}

// The 64 x 64 tile on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).  Wave w owns rows
// [16w, 16w+16) x all 64 columns: acc[nt][r] = C[16w + 4*kq + r][16*nt + i] for lane (i = lane & 15, kq = lane >> 4).  The four MFMAs
// of a 16-deep group take the k sets {4*kq + j : kq} (j = 0..3) -- any assignment of k to the slots is valid as long as both operands
// use it -- so a lane's operands are ONE 16-byte LDS read per fragment (A[16w + i][4kq .. 4kq+3], B[16nt + i][4kq .. 4kq+3]).
// The VALU form of this kernel (4 x 4 register tile per thread, 8 floats of LDS traffic per 16 FMAs) ran at 13 TFLOP/s.
// rowsq (optional): |A row|^2 of the accumulator rows of this lane, i.e. rows 16w + 4*kq + r.
template <bool ROWSQ>
__device__ __forceinline__ void tile_gemm(const Operand& A, const Operand& B, int m0, int n0, int kbeg, int kend, float (*As)[LDK],
                                          float (*Bs)[LDK], f4 (&acc)[4], float (&rowsq)[4], int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    float rs_part = 0.f;                          // this lane's share of |A[16w + i]|^2
    float ra[NI], rb[NI];
    fetch(A, m0, kbeg, kend, ra, tid);
    fetch(B, n0, kbeg, kend, rb, tid);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        put(A, ra, As, tid);
        put(B, rb, Bs, tid);
        __syncthreads();
        if (k0 + TK < kend) {                     // in flight during the MFMAs below
            fetch(A, m0, k0 + TK, kend, ra, tid);
            fetch(B, n0, k0 + TK, kend, rb, tid);
        }
#pragma unroll
        for (int kk = 0; kk < TK; kk += 16) {
            const f4 a = *(const f4*)&As[16 * wave + i][kk + 4 * kq];
            if (ROWSQ) rs_part += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f4 b = *(const f4*)&Bs[16 * nt + i][kk + 4 * kq];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (ROWSQ) {
        rs_part += __shfl_xor(rs_part, 16, 64);
        rs_part += __shfl_xor(rs_part, 32, 64);   // every lane with this i now holds |A[16w + i]|^2
#pragma unroll
        for (int r = 0; r < 4; ++r) rowsq[r] = __shfl(rs_part, 4 * kq + r, 64);
    }
}

__global__ __launch_bounds__(256) void ortho_gram_kernel(const float* __restrict__ flat, const long* __restrict__ table,
                                                         const int* __restrict__ tiles, float* __restrict__ gram) {
    __shared__ __attribute__((aligned(16))) float As[TM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[TM][LDK];
    const int* t = tiles + 4 * blockIdx.x;
    const long* L = table + 4 * t[0];
    const float* W = flat + L[0];
    const int R = (int)L[1], K = (int)L[2];
    float* G = gram + L[3];
    const bool rowform = R <= K;
    const int M = rowform ? R : K, red = rowform ? K : R;
    Operand X = rowform ? Operand{W, (long)K, 1, M, red} : Operand{W, 1, (long)K, M, red};
    const int m0 = t[1] * TM, n0 = t[2] * TM, kbeg = t[3] * KSPLIT, kend = min(red, kbeg + KSPLIT);
    f4 acc[4] = {};
    float dummy[4];
    tile_gemm<false>(X, X, m0, n0, kbeg, kend, As, Bs, acc, dummy, threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = m0 + 16 * wave + 4 * kq + r, gj = n0 + 16 * nt + i;
            if (gi < M && gj < M && !(rowform && gi == gj)) atomicAdd(&G[(long)gi * M + gj], acc[nt][r]);
        }
}

__global__ __launch_bounds__(256) void ortho_apply_kernel(const float* __restrict__ flat, float* __restrict__ grad,
                                                          const long* __restrict__ table, const int* __restrict__ tiles,
                                                          const float* __restrict__ gram, float coef) {
    __shared__ __attribute__((aligned(16))) float As[TM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[TM][LDK];
    const int* t = tiles + 4 * blockIdx.x;
    const long* L = table + 4 * t[0];
    const float* W = flat + L[0];
    float* dW = grad + L[0];
    const int R = (int)L[1], K = (int)L[2];
    const float* G = gram + L[3];
    const bool rowform = R <= K;
    const int m0 = t[1] * TM, n0 = t[2] * TM;
    f4 acc[4] = {};
    float rowsq[4] = {0.f, 0.f, 0.f, 0.f};
    if (rowform) {      // out[i,c] = sum_j P[i,j] W[j,c]:  A = P [R x R], B(c, j) = W[j, c]
        Operand A{G, (long)R, 1, R, R}, B{W, 1, (long)K, K, R};
        tile_gemm<false>(A, B, m0, n0, 0, R, As, Bs, acc, rowsq, threadIdx.x);
    } else {            // out[i,c] = sum_a W[i,a] Q[a,c] - |w_i|^2 W[i,c]:  A = W [R x K], B(c, a) = Q[a, c] (symmetric)
        Operand A{W, (long)K, 1, R, K}, B{G, (long)K, 1, K, K};
        tile_gemm<true>(A, B, m0, n0, 0, K, As, Bs, acc, rowsq, threadIdx.x);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = m0 + 16 * wave + 4 * kq + r, gc = n0 + 16 * nt + i;
            if (gi < R && gc < K) {
                const long idx = (long)gi * K + gc;
                float v = acc[nt][r];
                if (!rowform) v -= rowsq[r] * W[idx];
                dW[idx] += coef * v;
            }
        }
}

}  // namespace

extern "C" int ieagan_ortho_ksplit(void) { return KSPLIT; }

extern "C" int ieagan_ortho_grad(const float* flat, float* grad, const long* table, const int* gram_tiles, int n_gram_tiles,
                                 const int* apply_tiles, int n_apply_tiles, float* gram, long gram_floats, float strength,
                                 void* stream) {
    hipStream_t st = (hipStream_t)stream;
    CHECK_ARG(flat && grad && table && gram_tiles && apply_tiles && gram, "ortho_grad: null pointer");
    CHECK_ARG(n_gram_tiles > 0 && n_apply_tiles > 0 && gram_floats > 0, "ortho_grad: empty work list");
    ProfScope prof("ortho_grad", 0.0, 0.0, st);
    CHECK_ARG(hipMemsetAsync(gram, 0, sizeof(float) * gram_floats, st) == hipSuccess, "ortho_grad: memset of the Gram scratch failed");
    hipLaunchKernelGGL(ortho_gram_kernel, dim3(n_gram_tiles), dim3(256), 0, st, flat, table, gram_tiles, gram);
    hipLaunchKernelGGL(ortho_apply_kernel, dim3(n_apply_tiles), dim3(256), 0, st, flat, grad, table, apply_tiles,
                       (const float*)gram, 2.f * strength);
    CHECK_LAUNCH("ortho_grad");
    return 0;
}
