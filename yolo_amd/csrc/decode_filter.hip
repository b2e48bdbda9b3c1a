// Head decode, score threshold and order-preserving compaction for gfx950.
//
// One launch handles every scale and every image.  A workgroup of 256 threads owns a
// *segment* of 256 consecutive candidates of one scale of one image and walks it in
// four sub-steps of 64 candidates.  Per sub-step the 64 rows of (5+C) logits, which
// are contiguous in memory, are copied to LDS with coalesced 16-byte loads; then
// wave q (one lane per candidate) evaluates class slice q, so LDS is read at a row
// stride of 5+C words (odd for C = 80: conflict-free) and the class-major score
// matrix is written with one 4-byte element per lane to consecutive addresses.
//
// Compaction: each candidate's max score is reduced over the four class slices
// through LDS, a wavefront ballot turns the 64 keep flags into a bit mask, and a
// survivor's slot is  segment base + survivors in earlier sub-steps + popcount of
// the mask below its lane.  Segment bases come from a decoupled look-back over
// 8-byte {status, count} words (one relaxed agent-scope store / load each, so a
// word is never torn and needs no fence).  Segments are handed out by a per-image
// ticket, so every predecessor of a waiting segment is already running; spins are
// bounded and raise an error word instead of hanging.
//
// Algorithmic bytes: N*(5+C)*e read once, M*(4+C+1)*4 written once.
#include "common.h"

namespace ymx {

typedef unsigned long long u64;

#define YMX_ST_AGG (1ull << 62)
#define YMX_ST_PREFIX (2ull << 62)

template <typename HT>
__device__ __forceinline__ void stage_rows(float *lds, const HT *src, int count, int t) {
    // src is element-aligned only; peel to a 16-byte boundary, then 16-byte loads.
    constexpr int EPV = 16 / (int)sizeof(HT);
    const uintptr_t addr = (uintptr_t)src;
    int head = (int)(((16 - (addr & 15)) & 15) / sizeof(HT));
    if (head > count) head = count;
    const int nvec = (count - head) / EPV;
    const int tail0 = head + nvec * EPV;
    if (t < head) lds[t] = elem<HT>::load(src + t);
    const uint4 *vsrc = (const uint4 *)(src + head);
    for (int v = t; v < nvec; v += 256) {
        const uint4 u = vsrc[v];
        float *d = lds + head + v * EPV;
        if constexpr (sizeof(HT) == 4) {
            d[0] = __uint_as_float(u.x); d[1] = __uint_as_float(u.y);
            d[2] = __uint_as_float(u.z); d[3] = __uint_as_float(u.w);
        } else {
            d[0] = __uint_as_float(u.x << 16); d[1] = __uint_as_float(u.x & 0xffff0000u);
            d[2] = __uint_as_float(u.y << 16); d[3] = __uint_as_float(u.y & 0xffff0000u);
            d[4] = __uint_as_float(u.z << 16); d[5] = __uint_as_float(u.z & 0xffff0000u);
            d[6] = __uint_as_float(u.w << 16); d[7] = __uint_as_float(u.w & 0xffff0000u);
        }
    }
    if (t < count - tail0) lds[tail0 + t] = elem<HT>::load(src + tail0 + t);
}

// MODE 0: decode only (every candidate written at its own index, boxes = xywh pixels)
// MODE 1: decode + filter + compaction (boxes = yxyx / image_size)
// MODE 2: filter + compaction of already decoded input (HT unused)
template <typename HT, int CPT, int MODE>
__global__ void __launch_bounds__(256)
decode_kernel(const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int D = 5 + a.C;
    float *stage = (float *)smem_raw;                       // [64*D] (MODE 0/1)
    const int stage_words = (MODE == 2) ? 0 : ((64 * D + 22 * a.max_pad + 3) & ~3);
    float *smax = stage + stage_words;                      // [2][4][64], by sub-step parity
    int *sint = (int *)(smax + 512);                        // [0] segment, [1] base

    const int t = threadIdx.x, lane = t & 63;
    const int q = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.y;

    int seg;
    if (MODE == 0) {
        seg = blockIdx.x;
    } else {
        if (t == 0) sint[0] = atomicAdd(&a.ticket[b], 1);
        __syncthreads();
        seg = sint[0];
    }
    int sc = 0;
    if (a.n_scales > 1 && seg >= a.seg_begin[1]) sc = 1;
    if (a.n_scales > 2 && seg >= a.seg_begin[2]) sc = 2;
    const int S = a.S[sc];
    const int Ns = (MODE == 2) ? a.N : 3 * S * S;
    const int c_local0 = (seg - a.seg_begin[sc]) * 256;
    const int c_global0 = a.cand_begin[sc] + c_local0;
    const float xs = a.xyscale[sc], strd = a.stride[sc];
    const float half = 0.5f * (xs - 1.0f);
    const float aw0 = a.anchor[sc][0][0], ah0 = a.anchor[sc][0][1];
    const float aw1 = a.anchor[sc][1][0], ah1 = a.anchor[sc][1][1];
    const float aw2 = a.anchor[sc][2][0], ah2 = a.anchor[sc][2][1];

    float score[4][CPT];
    float4 box[4];
    u64 mask[4];

#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
        const int c0 = c_local0 + ss * 64;
        int nvalid = Ns - c0;
        nvalid = nvalid < 0 ? 0 : (nvalid > 64 ? 64 : nvalid);
        const bool live = lane < nvalid;
        float m = -INFINITY;
        box[ss] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != 2) {
            // The 64 candidates' rows: contiguous when the cells are dense (cell_ld = 3*D);
            // with padded cells the span from the first to the last candidate is staged as
            // it lies, padding included, and each lane finds its row inside it.
            const int ld = a.cell_ld[sc] ? a.cell_ld[sc] : 3 * D;
            const int cell0 = c0 / 3, a0 = c0 - cell0 * 3;
            const HT *src = (const HT *)a.head[sc] + ((size_t)b * S * S + cell0) * ld + a0 * D;
            int span = 0;
            if (nvalid > 0) {
                const int cl = c0 + nvalid - 1, cellL = cl / 3;
                span = (cellL - cell0) * ld + (cl - cellL * 3 + 1) * D - a0 * D;
            }
            stage_rows<HT>(stage, src, span, t);
            __syncthreads();
            if (live) {
                const int cme = c0 + lane, cellm = cme / 3;
                const float *row = stage + (cellm - cell0) * ld + (cme - cellm * 3) * D - a0 * D;
                const float conf = sigmoid_f(row[4]);
                const float cmul = (MODE == 0 && a.conf) ? 1.0f : conf;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = q * CPT + k;
                    float s = -INFINITY;
                    if (c < a.C) s = cmul * sigmoid_f(row[5 + c]);
                    score[ss][k] = s;
                    m = fmaxf(m, s);
                }
                if (q == 0) {
                    const int n = c0 + lane;
                    const int an = n % 3, cell = n / 3;
                    const int col = cell % S, rw = cell / S;
                    const float cx = ((sigmoid_f(row[0]) * xs) - half + (float)col) * strd;
                    const float cy = ((sigmoid_f(row[1]) * xs) - half + (float)rw) * strd;
                    const float bw = __expf(row[2]) * (an == 0 ? aw0 : (an == 1 ? aw1 : aw2));
                    const float bh = __expf(row[3]) * (an == 0 ? ah0 : (an == 1 ? ah1 : ah2));
                    if (MODE == 0) {
                        box[ss] = make_float4(cx, cy, bw, bh);
                        if (a.conf) a.conf[(size_t)b * a.cap + c_global0 + ss * 64 + lane] = conf;
                    } else
                        box[ss] = make_float4((cy - bh / 2.0f) / a.image_size, (cx - bw / 2.0f) / a.image_size,
                                              (cy + bh / 2.0f) / a.image_size, (cx + bw / 2.0f) / a.image_size);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; ++k) score[ss][k] = -INFINITY;
            }
        } else {
            if (live) {
                const size_t n = (size_t)(c0 + lane);
                const float *sp = a.in_scores + (size_t)b * a.sb + n * a.sn;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = q * CPT + k;
                    float s = -INFINITY;
                    if (c < a.C) s = sp[(size_t)c * a.sc];
                    score[ss][k] = s;
                    m = fmaxf(m, s);
                }
                if (q == 0) {
                    const float4 v = *(const float4 *)(a.in_xywh + ((size_t)b * a.N + n) * 4);
                    box[ss] = make_float4((v.y - v.w / 2.0f) / a.image_size, (v.x - v.z / 2.0f) / a.image_size,
                                          (v.y + v.w / 2.0f) / a.image_size, (v.x + v.z / 2.0f) / a.image_size);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPT; ++k) score[ss][k] = -INFINITY;
            }
        }
        // double-buffered by parity: the next sub-step's writes cannot overtake these reads
        float *sm = smax + (ss & 1) * 256;
        sm[q * 64 + lane] = m;
        __syncthreads();
        const float mm = fmaxf(fmaxf(sm[lane], sm[64 + lane]), fmaxf(sm[128 + lane], sm[192 + lane]));
        const bool keep = live && (MODE == 0 || mm >= a.thr);
        mask[ss] = __ballot(keep);
    }

    const int n0 = __popcll(mask[0]), n1 = __popcll(mask[1]), n2 = __popcll(mask[2]), n3 = __popcll(mask[3]);
    const int cnt = n0 + n1 + n2 + n3;
    int base;
    if (MODE == 0) {
        base = c_global0;
        if (seg == 0 && t == 0) a.counts[b] = a.N;
    } else {
        if (t == 0) {
            u64 *st = a.state + (size_t)b * a.nseg;
            u64 run = 0;
            if (seg > 0) {
                __hip_atomic_store(&st[seg], YMX_ST_AGG | (u64)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int p = seg - 1; p >= 0; --p) {
                    u64 v = 0;
                    for (int spin = 0; spin < (1 << 20); ++spin) {
                        v = __hip_atomic_load(&st[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v >> 62) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (!(v >> 62)) { atomicExch(a.err, 1); break; }
                    run += (v & 0xffffffffull);
                    if ((v >> 62) == 2) break;
                }
            }
            __hip_atomic_store(&st[seg], YMX_ST_PREFIX | (run + (u64)cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sint[1] = (int)run;
            if (seg == a.nseg - 1) a.counts[b] = (int)run + cnt;
        }
        __syncthreads();
        base = sint[1];
    }

    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int pre = 0;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
        const u64 mk = mask[ss];
        if ((mk >> lane) & 1ull) {
            const int pos = base + pre + __popcll(mk & below);
            if (pos < a.cap) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = q * CPT + k;
                    if (c < a.C) a.scores[((size_t)b * a.C + c) * a.cap + pos] = score[ss][k];
                }
                if (q == 0) {
                    *(float4 *)(a.boxes + ((size_t)b * a.cap + pos) * 4) = box[ss];
                    if (a.src_idx) a.src_idx[(size_t)b * a.cap + pos] = c_global0 + ss * 64 + lane;
                }
            }
        }
        pre += __popcll(mk);
    }
}

template <typename HT, int CPT>
static int launch_mode(const DecodeArgs &a, hipStream_t s) {
    const int D = 5 + a.C;
    const size_t stage_words = (a.filter == 2) ? 0 : ((64 * (size_t)D + 22 * (size_t)a.max_pad + 3) & ~(size_t)3);
    const size_t lds = (stage_words + 512 + 4) * 4;
    if (lds > 64 * 1024) return fail(YMX_EUNSUPPORTED, "decode: %d classes need %zu bytes of LDS", a.C, lds);
    dim3 grid(a.nseg, a.B), block(256);
    if (a.filter == 0)
        hipLaunchKernelGGL((decode_kernel<HT, CPT, 0>), grid, block, lds, s, a);
    else if (a.filter == 1)
        hipLaunchKernelGGL((decode_kernel<HT, CPT, 1>), grid, block, lds, s, a);
    else
        hipLaunchKernelGGL((decode_kernel<HT, CPT, 2>), grid, block, lds, s, a);
    return check_launch("decode");
}

template <typename HT>
static int launch_cpt(const DecodeArgs &a, hipStream_t s) {
    if (a.C <= 4) return launch_mode<HT, 1>(a, s);
    if (a.C <= 20) return launch_mode<HT, 5>(a, s);
    if (a.C <= 80) return launch_mode<HT, 20>(a, s);
    if (a.C <= 128) return launch_mode<HT, 32>(a, s);
    return fail(YMX_EUNSUPPORTED, "decode: at most 128 classes, got %d", a.C);
}

int launch_decode(const DecodeArgs &a, hipStream_t s) {
    if (a.filter == 2 || a.head_dtype == YMX_F32) return launch_cpt<float>(a, s);
    if (a.head_dtype == YMX_BF16) return launch_cpt<bf16_t>(a, s);
    return fail(YMX_EUNSUPPORTED, "decode: head dtype %d", a.head_dtype);
}

}  // namespace ymx
