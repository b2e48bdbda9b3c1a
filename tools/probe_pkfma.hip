// Probe: does v_pk_fma_f32 with a half-swapped (op_sel) addend return wrong lanes while another
// kernel's MFMAs run on the same SIMD?  (gfx950)
//
// The decode kernel's box corners were once formed by hipcc as
//     v_pk_fma_f32 v[14:15], s[12:13], v[16:17], v[12:13] op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]
//     v_pk_fma_f32 v[12:13], s[12:13], v[16:17], v[12:13] op_sel:[0,0,1] op_sel_hi:[0,1,0]
// and under a co-running matrix kernel 16-lane groups of lanes 32-63 came out with the addend of
// the LOW result (the one fetched from the HIGH register of the pair) missing.  This program runs
// that instruction and variations of it in a loop beside (a) nothing, (b) a vector-only spin
// kernel, (c) an MFMA spin kernel, and counts wrong results per variant and 16-lane group.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe_pkfma.hip -o tools/probe_pkfma && tools/probe_pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NV = 8;

// bad[v][group][which]: which 0 = low result (addend from the high register), 1 = high result
template <int V>
__device__ __forceinline__ void one(f32x2 q, f32x2 p, unsigned long long spair, f32x2 &lo_form, f32x2 &hi_form) {
    // lo_form = {s*q.x - p.y, s*q.y - p.x}, hi_form = {s*q.x + p.y, s*q.y + p.x}
    if constexpr (V == 0) {          // as compiled: scalar pair, addend swapped by op_sel, the second in place
        f32x2 h = p;
        asm volatile("v_pk_mul_f32 %1, %1, 0.5 op_sel_hi:[1,0]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,0,1] op_sel_hi:[0,1,0]"
                     : "=&v"(lo_form), "+v"(h) : "s"(spair), "v"(q));
        hi_form = h;
    } else if constexpr (V == 1) {   // the same two FMAs, addend long since written, separate destinations
        asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_pk_fma_f32 %1, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[0,1,0]"
                     : "=&v"(lo_form), "=&v"(hi_form) : "s"(spair), "v"(q), "v"(p));
    } else if constexpr (V == 2) {   // no swap in the instruction: the addend pair is swapped beforehand
        const f32x2 ps = {p.y, p.x};
        asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[0,1,1]"
                     : "=&v"(lo_form), "=&v"(hi_form) : "s"(spair), "v"(q), "v"(ps));
    } else if constexpr (V == 3) {   // multiplier in a vector pair instead of a scalar pair, addend swapped by op_sel
        const float s = __uint_as_float((unsigned)spair);
        const f32x2 sv = {s, s};
        asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_pk_fma_f32 %1, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[1,1,0]"
                     : "=&v"(lo_form), "=&v"(hi_form) : "v"(sv), "v"(q), "v"(p));
    } else if constexpr (V == 4) {   // the swap on the multiplicand instead: {s*q.y - p.x, s*q.x - p.y}
        f32x2 a, b;
        asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_pk_fma_f32 %1, %2, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,0,1]"
                     : "=&v"(a), "=&v"(b) : "s"(spair), "v"(q), "v"(p));
        lo_form = f32x2{a.y, a.x};
        hi_form = f32x2{b.y, b.x};
    } else if constexpr (V == 5) {   // packed add with a swapped operand: {q.x + p.y, q.y + p.x}; scaled outside
        const float s = __uint_as_float((unsigned)spair);
        const f32x2 sq = {__fmul_rn(s, q.x), __fmul_rn(s, q.y)};
        const f32x2 np = {-p.x, -p.y};
        asm volatile("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f32 %1, %2, %4 op_sel:[0,1] op_sel_hi:[1,0]"
                     : "=&v"(lo_form), "=&v"(hi_form) : "v"(sq), "v"(np), "v"(p));
    } else if constexpr (V == 6) {   // scalar (unpacked) FMAs
        const float s = __uint_as_float((unsigned)spair);
        float a, b, c, d;
        asm volatile("v_fma_f32 %0, %4, %5, -%8\n\tv_fma_f32 %1, %4, %6, -%7\n\t"
                     "v_fma_f32 %2, %4, %5, %8\n\tv_fma_f32 %3, %4, %6, %7"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "s"(s), "v"(q.x), "v"(q.y), "v"(p.x), "v"(p.y));
        lo_form = f32x2{a, b};
        hi_form = f32x2{c, d};
    } else {                         // V == 7: as V1 but behind a transcendental instruction (as v_rcp_f32 sits before the second FMA)
        float dummy = q.x;
        asm volatile("v_pk_fma_f32 %0, %3, %4, %5 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                     "v_rcp_f32 %2, %2\n\t"
                     "v_pk_fma_f32 %1, %3, %4, %5 op_sel:[0,0,1] op_sel_hi:[0,1,0]"
                     : "=&v"(lo_form), "=&v"(hi_form), "+v"(dummy) : "s"(spair), "v"(q), "v"(p));
    }
}

template <int V>
__global__ void __launch_bounds__(256) probe_kernel(unsigned *bad, float *first_bad, float s, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned sbits = __builtin_amdgcn_readfirstlane(__float_as_uint(s));
    const unsigned long long spair = (unsigned long long)sbits;            // high word: 0, as in the compiled kernel
    unsigned nlo = 0, nhi = 0, nx = 0;
    float seen = 0.f;
    f32x2 q = {0.3f + 0.01f * lane, 0.7f + 0.02f * lane};
    f32x2 p = {3.0f + 0.5f * lane, 100.0f + 7.0f * lane};
    for (int it = 0; it < iters; ++it) {
        q.x += 0.001f; q.y += 0.002f; p.x += 0.25f; p.y += 0.125f;
        asm volatile("" : "+v"(q), "+v"(p));
        f32x2 lo, hi;
        const f32x2 pin = V == 0 ? f32x2{2.0f * p.x, 2.0f * p.y} : p;      // V0 halves its addend itself
        one<V>(q, pin, spair, lo, hi);
        float e_lo = __builtin_fmaf(s, q.x, -p.y), e_hi = __builtin_fmaf(s, q.x, p.y);
        float e_lo2 = __builtin_fmaf(s, q.y, -p.x), e_hi2 = __builtin_fmaf(s, q.y, p.x);
        if (V == 5) {     // product rounded before the add
            e_lo = __fadd_rn(__fmul_rn(s, q.x), -p.y); e_hi = __fadd_rn(__fmul_rn(s, q.x), p.y);
            e_lo2 = __fadd_rn(__fmul_rn(s, q.y), -p.x); e_hi2 = __fadd_rn(__fmul_rn(s, q.y), p.x);
        }
        if (lo.x != e_lo) { ++nlo; seen = lo.x - s * q.x; }
        if (hi.x != e_hi) { ++nhi; seen = hi.x - s * q.x; }
        if (lo.y != e_lo2 || hi.y != e_hi2) ++nx;
    }
    const int g = lane >> 4;
    if (nlo) atomicAdd(&bad[(V * 4 + g) * 3 + 0], nlo);
    if (nhi) atomicAdd(&bad[(V * 4 + g) * 3 + 1], nhi);
    if (nx) atomicAdd(&bad[(V * 4 + g) * 3 + 2], nx);
    if (nlo | nhi) first_bad[V] = seen;      // any one: the addend the wrong result carries (0 = none)
}

// co-runners: one wave per SIMD on every CU, few registers, no LDS, so that probe waves share the SIMDs
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
__global__ void __launch_bounds__(256) spin_valu(float *out, int iters) {
    float x = threadIdx.x * 0.001f, y = 1.0f;
    for (int it = 0; it < iters; ++it) {
        x = __builtin_fmaf(x, 0.999f, 0.001f); y = __builtin_fmaf(y, 1.0001f, -0.0001f);
        x = __builtin_fmaf(x, 0.999f, 0.001f); y = __builtin_fmaf(y, 1.0001f, -0.0001f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x + y;
}

template <int V>
static void launch_probe(unsigned *bad, float *fb, int iters, hipStream_t s) {
    hipLaunchKernelGGL(probe_kernel<V>, dim3(1024), dim3(256), 0, s, bad, fb, 8.0f, iters);
}

int main() {
    unsigned *bad; float *fb, *sink;
    CK(hipMalloc(&bad, NV * 4 * 3 * sizeof(unsigned)));
    CK(hipMalloc(&fb, NV * sizeof(float)));
    CK(hipMalloc(&sink, 1024 * 256 * sizeof(float)));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *names[NV] = {"0 compiled form (pk_mul; pk_fma op_sel addend; in place)", "1 pk_fma, addend swapped by op_sel, scalar-pair multiplier",
                             "2 pk_fma, addend pre-swapped (no op_sel swap)", "3 pk_fma, addend swapped by op_sel, vector multiplier",
                             "4 pk_fma, multiplicand swapped by op_sel", "5 pk_add with a swapped operand", "6 four scalar v_fma_f32",
                             "7 form 1 with v_rcp_f32 between the two"};
    const char *co[3] = {"alone", "beside a vector-only spin kernel", "beside an MFMA spin kernel"};
    const int iters = 20000;
    for (int c = 0; c < 3; ++c) {
        CK(hipMemset(bad, 0, NV * 4 * 3 * sizeof(unsigned)));
        CK(hipMemset(fb, 0, NV * sizeof(float)));
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 3; ++rep) {
            if (c == 1) hipLaunchKernelGGL(spin_valu, dim3(1024), dim3(256), 0, sb, sink, 4000000);
            if (c == 2) hipLaunchKernelGGL(spin_mfma, dim3(1024), dim3(256), 0, sb, sink, 600000);
            launch_probe<0>(bad, fb, iters, sa); launch_probe<1>(bad, fb, iters, sa); launch_probe<2>(bad, fb, iters, sa);
            launch_probe<3>(bad, fb, iters, sa); launch_probe<4>(bad, fb, iters, sa); launch_probe<5>(bad, fb, iters, sa);
            launch_probe<6>(bad, fb, iters, sa); launch_probe<7>(bad, fb, iters, sa);
            CK(hipDeviceSynchronize());
        }
        std::vector<unsigned> h(NV * 4 * 3);
        std::vector<float> hf(NV);
        CK(hipMemcpy(h.data(), bad, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hf.data(), fb, hf.size() * sizeof(float), hipMemcpyDeviceToHost));
        printf("== probe kernels %s ==\n", co[c]);
        for (int v = 0; v < NV; ++v) {
            printf("%-62s", names[v]);
            for (int g = 0; g < 4; ++g)
                printf(" | lanes %2d-%2d: low %7u high %7u x %5u", g * 16, g * 16 + 15, h[(v * 4 + g) * 3], h[(v * 4 + g) * 3 + 1], h[(v * 4 + g) * 3 + 2]);
            printf(" | addend seen in a wrong result: %g\n", hf[v]);
        }
    }
    printf("done\n");
    return 0;
}
