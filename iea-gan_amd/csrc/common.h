// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the IEA-GAN train step.
// Activations: bf16, NHWC.  Accumulation / statistics / parameters: fp32.  Wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define IEAGAN_OK 0
#define IEAGAN_EINVAL (-1)
#define IEAGAN_ELAUNCH (-2)

// thread-local last-error text, set by the launchers in api.hip
void ieagan_set_error(const char* fmt, ...);
// optional per-kernel event profiling (api.hip); name must be a string literal
// (the scope OWNS its two events and hands the finished record to the collector in its destructor: a scope that is open on the autograd or
//  side-stream thread while another thread collects / resets never touches the collector's storage)
struct ProfScope {
    ProfScope(const char* name, double flops, double bytes, hipStream_t s, const char* tag = nullptr, double bytes_min = -1.0);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
    bool active;
    hipStream_t stream;
    hipEvent_t ev_a, ev_b;
    double flops, bytes, bytes_min;
    char name[96];
};

bool prof_tags_on();

#define CHECK_ARG(cond, ...)                                   \
    do {                                                       \
        if (!(cond)) {                                         \
            ieagan_set_error(__VA_ARGS__);                     \
            return IEAGAN_EINVAL;                              \
        }                                                      \
    } while (0)

#define CHECK_LAUNCH(name)                                                          \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ieagan_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return IEAGAN_ELAUNCH;                                                  \
        }                                                                           \
    } while (0)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// ReLU of 8 packed bf16 values on their bit patterns: as signed 16-bit integers every negative float (sign bit set, incl. -0
// and negative NaNs) is a negative integer and every non-negative float keeps its order, so max(bits, 0) is exact --
// four v_pk_max_i16 instead of unpack / v_max_f32 / repack per element.
typedef short i16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 relu8(bf16x8 v) {
    i16x8 s = __builtin_bit_cast(i16x8, v);
    const i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    s = __builtin_elementwise_max(s, z);
    return __builtin_bit_cast(bf16x8, s);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

// sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Replicated statistics buffers: float atomics on a single address serialise (MI355X: ~14x slower
// than spread adds), so per-channel sums are spread over STAT_REPL replicas keyed by the block
// index and folded by the consumer (bn_finalize / the python side).
#define STAT_REPL 32
#define BNB_REPL 8          // replicas of the per-image (d shift, d scale) accumulators of a BatchNorm-backward dgrad launch

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
