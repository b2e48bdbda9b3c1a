// Probe: sustained bf16 MFMA rate of the whole chip under its power limit, by instruction shape
// and operand data (gfx950).  Every wave spins on independent MFMAs whose operands stay in
// registers (no memory traffic), two waves per SIMD, for about a second per variant, so the
// figure is what the matrix pipes hold under the package power limit, not a burst.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_power.hip -o tools/probe_mfma_power && tools/probe_mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// SHAPE 0: 16x16x32 (16 independent accumulators of 4 registers); 1: 32x32x16 (4 accumulators of 16)
template <int SHAPE>
__global__ void __launch_bounds__(512) spin(const uint4 *src, float *sink, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    uint4 a4[4], b4[4];
    for (int i = 0; i < 4; ++i) { a4[i] = src[(t * 8 + i) & 0xffff]; b4[i] = src[(t * 8 + 4 + i) & 0xffff]; }
    float total = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4[i & 3]), __builtin_bit_cast(bf16x8, b4[i >> 2]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) total += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4[(i & 1) + 2 * r]), __builtin_bit_cast(bf16x8, b4[(i >> 1) + 2 * r]), acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
    }
    if (total == 12345.678f) sink[t] = total;
}

int main() {
    const int N = 1 << 16;
    std::vector<unsigned short> h(N * 8);
    uint4 *src; float *sink;
    CK(hipMalloc(&src, N * 16)); CK(hipMalloc(&sink, 512 * 2048 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *dnames[3] = {"zeros", "small integers (few mantissa bits)", "random bf16 in [-2, 2)"};
    for (int data = 0; data < 3; ++data) {
        unsigned s = 12345u;
        for (size_t i = 0; i < h.size(); ++i) {
            s = s * 1664525u + 1013904223u;
            if (data == 0) h[i] = 0;
            else if (data == 1) h[i] = (unsigned short)(0x3f80 + ((s >> 20) & 0x3) * 0x80);          // 1, 2, 4, 8
            else h[i] = (unsigned short)(((s >> 16) & 0x807f) | ((0x7c + ((s >> 8) & 3)) << 7));   // exponent 124..127
        }
        CK(hipMemcpy(src, h.data(), N * 16, hipMemcpyHostToDevice));
        for (int shape = 0; shape < 2; ++shape) {
            // per wave and iteration: 16 x 16x16x32 or 8 x 32x32x16 = 262 144 flops
            const int iters = 20000, grid = 512;                 // two workgroups of 8 waves per CU
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                for (int l = 0; l < (rep ? 100 : 2); ++l) {
                    if (shape == 0) hipLaunchKernelGGL(spin<0>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
                    else hipLaunchKernelGGL(spin<1>, dim3(grid), dim3(512), 0, 0, src, sink, iters);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) {
                    const double flops = 100.0 * grid * 8 * (double)iters * 262144.0;
                    printf("%-38s %s: %8.1f ms  %7.1f TFLOP/s\n", dnames[data], shape ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
