// Second probe of the packed-fp32 hazard found by probe_pkfma.hip (gfx950): which operand
// combinations of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 return wrong halves while another
// kernel's MFMAs run on the same SIMD.  One instruction per variant, inline asm, results compared
// with scalar arithmetic; counts per 16-lane group for the low (x) and high (y) result.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe_pkfma2.hip -o tools/probe_pkfma2 && tools/probe_pkfma2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NV = 14;
static const char *NAMES[NV] = {
    "0  fma  s[lo,lo] * q + p(swapped)                      [the failing form]",
    "1  fma  s[lo,lo] * q + p(hi,hi)    low result reads the high register",
    "2  fma  s[lo,lo] * q + p(lo,lo)    high result reads the low register",
    "3  fma  q * s[lo,lo] + p(swapped)  scalar pair as second source",
    "4  fma  s[lo,hi] * q + p(swapped)  scalar pair read as a real pair",
    "5  fma  2.0 * q + p(swapped)       inline constant instead of the scalar pair",
    "6  fma  v[lo,lo] * q + p(swapped)  vector pair read lo,lo",
    "7  fma  s[lo,lo] * q(swapped) + p",
    "8  mul  s[lo,lo] * p(swapped)",
    "9  add  s[lo,lo] + p(swapped)",
    "10 fma  q * p(swapped) + s[lo,lo]  scalar pair as addend",
    "11 fma  s[lo,lo] * q + p           no swap anywhere",
    "12 fma  s[lo,lo] * q + p(swapped), s_nop 7 in front",
    "13 fma  s[lo,lo] * q + p(swapped), two in a row to different destinations (second checked)",
};

template <int V>
__device__ __forceinline__ void one(f32x2 q, f32x2 p, float s, unsigned long long sp_lo, unsigned long long sp_pair, f32x2 &r, f32x2 &e) {
    const f32x2 sv = {s, -7.0f};
    if constexpr (V == 0) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.x)};
    } else if constexpr (V == 1) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.y)};
    } else if constexpr (V == 2) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.x), __builtin_fmaf(s, q.y, p.x)};
    } else if constexpr (V == 3) {
        asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.x)};
    } else if constexpr (V == 4) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(r) : "s"(sp_pair), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.x)};
    } else if constexpr (V == 5) {
        asm volatile("v_pk_fma_f32 %0, 2.0, %1, %2 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=&v"(r) : "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(2.0f, q.x, p.y), __builtin_fmaf(2.0f, q.y, p.x)};
    } else if constexpr (V == 6) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=&v"(r) : "v"(sv), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.x)};
    } else if constexpr (V == 7) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.y, p.x), __builtin_fmaf(s, q.x, p.y)};
    } else if constexpr (V == 8) {
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=&v"(r) : "s"(sp_lo), "v"(p));
        e = f32x2{__fmul_rn(s, p.y), __fmul_rn(s, p.x)};
    } else if constexpr (V == 9) {
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=&v"(r) : "s"(sp_lo), "v"(p));
        e = f32x2{__fadd_rn(s, p.y), __fadd_rn(s, p.x)};
    } else if constexpr (V == 10) {
        asm volatile("v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,0,0]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(q.x, p.y, s), __builtin_fmaf(q.y, p.x, s)};
    } else if constexpr (V == 11) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.x), __builtin_fmaf(s, q.y, p.y)};
    } else if constexpr (V == 12) {
        asm volatile("s_nop 7\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=&v"(r) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.x)};
    } else {
        f32x2 r0;
        asm volatile("v_pk_fma_f32 %1, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[0,1,0]\n\t"
                     "v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[0,1,0]" : "=&v"(r), "=&v"(r0) : "s"(sp_lo), "v"(q), "v"(p));
        e = f32x2{__builtin_fmaf(s, q.x, p.y), __builtin_fmaf(s, q.y, p.x)};
    }
}

template <int V>
__global__ void __launch_bounds__(256) probe_kernel(unsigned *bad, float s, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned sbits = __builtin_amdgcn_readfirstlane(__float_as_uint(s));
    const unsigned long long sp_lo = (unsigned long long)sbits | (0xc0e00000ull << 32);      // high word: -7.0 (must never be read)
    const unsigned long long sp_pair = (unsigned long long)sbits | ((unsigned long long)sbits << 32);
    unsigned nx = 0, ny = 0;
    f32x2 q = {0.3f + 0.01f * lane, 0.7f + 0.02f * lane};
    f32x2 p = {3.0f + 0.5f * lane, 100.0f + 7.0f * lane};
    for (int it = 0; it < iters; ++it) {
        q.x += 0.001f; q.y += 0.002f; p.x += 0.25f; p.y += 0.125f;
        asm volatile("" : "+v"(q), "+v"(p));
        f32x2 r, e;
        one<V>(q, p, s, sp_lo, sp_pair, r, e);
        if (r.x != e.x) ++nx;
        if (r.y != e.y) ++ny;
    }
    const int g = lane >> 4;
    if (nx) atomicAdd(&bad[(V * 4 + g) * 2 + 0], nx);
    if (ny) atomicAdd(&bad[(V * 4 + g) * 2 + 1], ny);
}

__global__ void __launch_bounds__(256) spin_mfma(float *out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.02f * (threadIdx.x - i)); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

template <int V>
static void launch_all(unsigned *bad, int iters, hipStream_t s) {
    hipLaunchKernelGGL(probe_kernel<V>, dim3(1024), dim3(256), 0, s, bad, 8.0f, iters);
    if constexpr (V + 1 < NV) launch_all<V + 1>(bad, iters, s);
}

int main() {
    unsigned *bad; float *sink;
    CK(hipMalloc(&bad, NV * 4 * 2 * sizeof(unsigned)));
    CK(hipMalloc(&sink, 1024 * 256 * sizeof(float)));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *co[2] = {"alone", "beside an MFMA spin kernel"};
    for (int c = 0; c < 2; ++c) {
        CK(hipMemset(bad, 0, NV * 4 * 2 * sizeof(unsigned)));
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 4; ++rep) {
            if (c == 1) hipLaunchKernelGGL(spin_mfma, dim3(1024), dim3(256), 0, sb, sink, 1000000);
            launch_all<0>(bad, 20000, sa);
            CK(hipDeviceSynchronize());
        }
        std::vector<unsigned> h(NV * 4 * 2);
        CK(hipMemcpy(h.data(), bad, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        printf("== probe kernels %s: wrong low / high results per 16-lane group ==\n", co[c]);
        for (int v = 0; v < NV; ++v) {
            printf("%-92s", NAMES[v]);
            for (int g = 0; g < 4; ++g) printf(" | %2d-%2d: %6u / %6u", g * 16, g * 16 + 15, h[(v * 4 + g) * 2], h[(v * 4 + g) * 2 + 1]);
            printf("\n");
        }
    }
    printf("done\n");
    return 0;
}
