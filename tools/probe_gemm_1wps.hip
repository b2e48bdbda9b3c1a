// Probe: bf16 GEMM C[M][N] = A[M][K] * B[N][K]^T with ONE wave per SIMD (4 waves per
// workgroup, 128 x 128 outputs and 256 accumulator registers per wave), operands staged
// global -> registers -> LDS (no LDS-DMA), one barrier per 64-deep K tile.
// Question it answers: does this structure, written in HIP source, beat the 8-wave
// phase-staggered LDS-DMA structure the convolution kernels use (about 1.3 PFLOP/s on a
// plain GEMM)?  Not part of the product.
// Answer (MI355X, random data): no.  256 accumulators + 128 fragment registers (two K
// halves in flight) + 64 staging registers (a tile of global loads in flight, which
// LDS-DMA keeps out of the register file) + addresses fill all 512 registers:
//   unpipelined (reads, MFMAs, writes, barrier)            655 / 698 TFLOP/s at 4096^3 / 8192^3
//   + LDS reads of K half 1 under the MFMAs of half 0      723 / 770
//   + one barrier per tile, writes and next reads under MFMAs, compiler order   723 / 830
//     (4 spills); the same with sched_group_barrier interleave   578 / 603 (60 spills)
// LDS-DMA instead of staging registers would cost one wave 16 x 60-180 cycles of blocked
// issue per tile with no second wave on the SIMD to fill the matrix pipe.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o probe_gemm_1wps probe_gemm_1wps.hip
// Run:   ./probe_gemm_1wps [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#ifndef SCHED
#define SCHED 0
#endif

__global__ void __launch_bounds__(256, 1)
gemm_1wps(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B, float *__restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A 32 KB + B 32 KB)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / 256;
    const int m0 = (blockIdx.x / ntn) * 256, n0 = (blockIdx.x % ntn) * 256;
    const int nk = K / 64;

    // staging: chunk id = t + 256 i -> row (t >> 3) + 32 i, slot t & 7: one per-thread offset,
    // the step from i to i+1 is a scalar (32 rows), so no per-chunk address registers
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void *)B, 0, 0x7fffffff, 0x00020000);
    const int srow = t >> 3, sslot = t & 7;
    const unsigned a_off = (unsigned)(((m0 + srow) * K + sslot * 8) * 2);
    const unsigned b_off = (unsigned)(((n0 + srow) * K + sslot * 8) * 2);
    const int rstep = 32 * K * 2;                               // bytes between chunks i and i+1
    const int sdst = srow * 128 + ((sslot ^ (srow & 7)) << 4);  // + 4096 i (32 rows; the swizzle repeats every 8)
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 ga[8], gb[8];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ga[i] = __builtin_amdgcn_raw_buffer_load_b128(ar, a_off + (unsigned)(kt * 128), i * rstep, 0);
            gb[i] = __builtin_amdgcn_raw_buffer_load_b128(br, b_off + (unsigned)(kt * 128), i * rstep, 0);
        }
    };
    auto lstore = [&](int buf) {
        char *base = smem + buf * 65536 + sdst;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *(u32x4 *)(base + i * 4096) = ga[i];
            *(u32x4 *)(base + 32768 + i * 4096) = gb[i];
        }
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: row = block*16 + (lane & 15), chunk = kk*4 + (lane >> 4)
    const int frow = lane & 15, q4 = lane >> 4;
    const int a_base = (wm * 128 + frow) * 128, b_base = 32768 + (wn * 128 + frow) * 128;
    const int sw = frow & 7;     // (block*16 + frow) & 7 == frow & 7
    const int c0 = ((q4 ^ sw) << 4), c1 = (((4 + q4) ^ sw) << 4);

    // Software pipeline, one barrier per K tile:
    //   phase A (K half 0 of tile kt): 64 MFMAs; under them the LDS reads of K half 1, then the
    //            LDS writes of tile kt+1 (its global loads were issued a whole tile earlier);
    //   barrier;
    //   phase B (K half 1): 64 MFMAs; under them the LDS reads of tile kt+1's K half 0 and the
    //            global loads of tile kt+2.
    // Buffer (kt+1)&1 was last read in phase A of iteration kt-1, a barrier ago.
    uint4 af[2][8], bf[2][8];
    auto lread = [&](const char *buf, int c, int set) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[set][i] = *(const uint4 *)(buf + a_base + i * 2048 + c);
            bf[set][i] = *(const uint4 *)(buf + b_base + i * 2048 + c);
        }
    };
    auto mfma64 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[set][i]),
                                                                   __builtin_bit_cast(bf16x8, bf[set][j]), acc[i][j], 0, 0, 0);
    };
    gload(0);
    lstore(0);
    gload(nk > 1 ? 1 : 0);
    __syncthreads();
    lread(smem, c0, 0);

    for (int kt = 0; kt < nk; ++kt) {
        const char *cur = smem + (kt & 1) * 65536;
        const char *nxt = smem + ((kt + 1) & 1) * 65536;
        // ---- phase A
        lread(cur, c1, 1);
        mfma64(0);
        lstore((kt + 1) & 1);
#if SCHED
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS read
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   // 2 DS write
        }
#endif
        __syncthreads();
        // ---- phase B
        lread(nxt, c0, 0);
        gload(kt + 2 < nk ? kt + 2 : nk - 1);
        mfma64(1);
#if SCHED
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        }
#endif
    }

    // C layout: lane holds C[row = 4*(lane>>4) + r][col = lane & 15] of each 16x16 block
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = n0 + wn * 128 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 128 + i * 16 + 4 * q4 + r;
                C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
}

static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char **argv) {
    int M = 4096, N = 4096, K = 4096;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    if (M % 256 || N % 256 || K % 64) { printf("M, N multiples of 256, K of 64\n"); return 1; }
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = f2bf(rnd());
    for (auto &v : hB) v = f2bf(rnd());
    uint16_t *dA, *dB;
    float *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)gemm_1wps, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int grid = (M / 256) * (N / 256);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_1wps, dim3(grid), dim3(256), 131072, 0, dA, dB, dC, M, N, K);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_1wps, dim3(grid), dim3(256), 131072, 0, dA, dB, dC, M, N, K);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int c = 0; c < 64; ++c) {
        const int r = (c * 977 + 13) % M, q = (c * 613 + 7) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)q * K + k]);
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)r * N + q]));
    }
    printf("gemm_1wps %dx%dx%d: %.3f ms  %.0f TFLOP/s  max |err| on 64 samples %.3g  (sched hints %d)\n", M, N, K, ms,
           2.0 * M * N * K / ms / 1e9, maxerr, SCHED);
    return 0;
}
