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
constexpr int TK = 16;        // reduce chunk
constexpr int LDT = TM + 4;   // padded LDS row
constexpr int KSPLIT = 2048;  // reduce length per Gram work item

struct Operand {              // element (m, k) of a [M x Kd] operand lives at p[m * rs + k * cs]
    const float* p;
    long rs, cs;
    int M, Kd;
};

// Stage a [TM x TK] chunk (rows m0.., reduce k0..) into LDS as s[k][m]; out-of-range -> 0.
__device__ __forceinline__ void stage(const Operand& o, int m0, int k0, int kend, float (*s)[LDT], int tid) {
#pragma unroll
    for (int i = 0; i < (TM * TK) / 256; ++i) {
        int e = tid + 256 * i;
        int m, k;
        if (o.cs == 1) { m = e / TK; k = e % TK; } else { k = e / TM; m = e % TM; }
        int gm = m0 + m, gk = k0 + k;
        float v = 0.f;
        if (gm < o.M && gk < kend) v = o.p[(long)gm * o.rs + (long)gk * o.cs];
        s[k][m] = v;
    }
}

// acc[4][4] (+)= A[rows ty*4.., :] . B[cols tx*4.., :]^T over the reduce range [kbeg, kend); optional row square sums of A.
template <bool ROWSQ>
__device__ __forceinline__ void tile_gemm(const Operand& A, const Operand& B, int m0, int n0, int kbeg, int kend, float (*As)[LDT],
                                          float (*Bs)[LDT], float acc[4][4], float rowsq[4], int tid) {
    const int ty = tid / 16, tx = tid % 16;
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        stage(A, m0, k0, kend, As, tid);
        stage(B, n0, k0, kend, Bs, tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
                if (ROWSQ) rowsq[i] = fmaf(av[i], av[i], rowsq[i]);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void ortho_gram_kernel(const float* __restrict__ flat, const long* __restrict__ table,
                                                         const int* __restrict__ tiles, float* __restrict__ gram) {
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
    const int* t = tiles + 4 * blockIdx.x;
    const long* L = table + 4 * t[0];
    const float* W = flat + L[0];
    const int R = (int)L[1], K = (int)L[2];
    float* G = gram + L[3];
    const bool rowform = R <= K;
    const int M = rowform ? R : K, red = rowform ? K : R;
    Operand X = rowform ? Operand{W, (long)K, 1, M, red} : Operand{W, 1, (long)K, M, red};
    const int m0 = t[1] * TM, n0 = t[2] * TM, kbeg = t[3] * KSPLIT, kend = min(red, kbeg + KSPLIT);
    float acc[4][4] = {}, dummy[4];
    tile_gemm<false>(X, X, m0, n0, kbeg, kend, As, Bs, acc, dummy, threadIdx.x);
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gi = m0 + ty * 4 + i, gj = n0 + tx * 4 + j;
            if (gi < M && gj < M && !(rowform && gi == gj)) atomicAdd(&G[(long)gi * M + gj], acc[i][j]);
        }
}

__global__ __launch_bounds__(256) void ortho_apply_kernel(const float* __restrict__ flat, float* __restrict__ grad,
                                                          const long* __restrict__ table, const int* __restrict__ tiles,
                                                          const float* __restrict__ gram, float coef) {
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
    const int* t = tiles + 4 * blockIdx.x;
    const long* L = table + 4 * t[0];
    const float* W = flat + L[0];
    float* dW = grad + L[0];
    const int R = (int)L[1], K = (int)L[2];
    const float* G = gram + L[3];
    const bool rowform = R <= K;
    const int m0 = t[1] * TM, n0 = t[2] * TM;
    float acc[4][4] = {}, rowsq[4] = {0.f, 0.f, 0.f, 0.f};
    if (rowform) {      // out[i,c] = sum_j P[i,j] W[j,c]:  A = P [R x R], B(c, j) = W[j, c]
        Operand A{G, (long)R, 1, R, R}, B{W, 1, (long)K, K, R};
        tile_gemm<false>(A, B, m0, n0, 0, R, As, Bs, acc, rowsq, threadIdx.x);
    } else {            // out[i,c] = sum_a W[i,a] Q[a,c] - |w_i|^2 W[i,c]:  A = W [R x K], B(c, a) = Q[a, c] (symmetric)
        Operand A{W, (long)K, 1, R, K}, B{G, (long)K, 1, K, K};
        tile_gemm<true>(A, B, m0, n0, 0, K, As, Bs, acc, rowsq, threadIdx.x);
    }
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gi = m0 + ty * 4 + i, gc = n0 + tx * 4 + j;
            if (gi < R && gc < K) {
                long idx = (long)gi * K + gc;
                float v = acc[i][j];
                if (!rowform) v -= rowsq[i] * W[idx];
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
