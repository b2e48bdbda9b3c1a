// Combined (per-class, then per-image top-k) non-maximum suppression for gfx950.
//
// Kernel 1, one workgroup per (class, image): streams the class's score row (the
// score matrix is class-major, so the row is contiguous: coalesced 16-byte loads) and
// selects candidates in exact (score desc, index asc) order without sorting the row:
//   - every candidate is a 64-bit key  ord(score) << 32 | ~index,  so one unsigned
//     compare gives the visiting order;
//   - a first pass over a SAMPLE of the row (every k-th 1024-candidate chunk, k up to 8)
//     builds a 1024-bin histogram of the candidates' score bits (LDS atomics); a suffix sum
//     gives the highest bin edge above which the sample holds its share of 4 * per_class
//     candidates plus three standard deviations, and that edge is the starting floor;
//   - a second pass appends the keys above the score threshold and the floor to an LDS
//     buffer (wavefront ballot + one LDS atomic per wave, no barrier inside the loop).  Should
//     the buffer overflow (many equal scores in one bin) the pass is repeated in a careful
//     form that sorts and cuts the buffer whenever it could fill up;
//   - the buffer's few hundred keys are not sorted as a whole: a second histogram over
//     exactly the scores present picks the best 3 * per_class or so, those are sorted by
//     rank and visited; if they do not yield per_class boxes the next group follows;
//   - wave 0 runs the greedy suppression in batches of 64: a lane tests its candidate
//     against every box kept so far (the kept boxes live one per lane in registers and are
//     broadcast with readlane), a ballot gives the bit mask of still-alive lanes, and the
//     mask is resolved leader by leader;
//   - if the buffer runs out before per_class boxes are kept and keys were left out, the
//     scan repeats below the last visited key with twice the target (a further tier).  The
//     result is the exact greedy NMS for any input; floor, sample and tiers only matter for
//     speed.
// Kernel 2, one workgroup per image: cuts the kept entries of all classes with a histogram
// spanning their scores, sorts the survivors of the cut by (score desc, class asc, index asc)
// in LDS and writes the first `total`.
//
// IoU arithmetic is fp32, in a fixed order, compiled without contraction, with a
// correctly rounded division, so keep decisions are reproducible bit for bit.
#include <atomic>
#include <cstring>

#include "lds_dma.h"

namespace ymx {

// Diagnostic build only (make stamps): the 100 MHz wall clock at points of a workgroup's life.
#ifdef YMX_STAMPS
#define NSTAMP(k)                                                                               \
    do {                                                                                        \
        const unsigned lin_ = blockIdx.x + gridDim.x * blockIdx.y;                              \
        if (threadIdx.x == 0 && lin_ < YMX_STAMP_BLOCKS) ymx_stamp_buf[lin_ * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define NSTAMP(k) do { } while (0)
#endif

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int NMS_KEEP = 512;     // keys that survive a cut
constexpr int NMS_BUF = 2048;     // LDS key buffer
constexpr int NMS_CHUNK = 1024;   // candidates scanned between capacity checks
constexpr int NMS_BINS = 1024;    // histogram bins of the floor search

__device__ __forceinline__ u32 ord_bits(float s) {
    const u32 u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_bits(u32 u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Descending bitonic sort of n (a power of two) keys in LDS by 256 threads.
__device__ __forceinline__ void bitonic_desc(u64 *buf, int n, int t) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < (n >> 1); i += 256) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const bool desc = (lo & k) == 0;
                const u64 x = buf[lo], y = buf[hi];
                if (desc ? (x < y) : (x > y)) { buf[lo] = y; buf[hi] = x; }
            }
            __syncthreads();
        }
    }
}

// true iff IoU(a, b) > thr, where IoU is the correctly rounded quotient inter / union (the
// arithmetic of the oracle).  Boxes are normalised (min corner first); areas given.  The
// division is only carried out when the products cannot decide: inter and thr * union further
// apart than a few ulps give the same answer as the rounded quotient (a correctly rounded
// quotient is within half an ulp of the true one, the product within half an ulp of its own).
__device__ __forceinline__ bool iou_gt(float ay1, float ax1, float ay2, float ax2, float aar,
                                       float by1, float bx1, float by2, float bx2, float bar, float thr) {
    if (!(aar > 0.0f) || !(bar > 0.0f)) return false;
    const float iy1 = fmaxf(ay1, by1), ix1 = fmaxf(ax1, bx1);
    const float iy2 = fminf(ay2, by2), ix2 = fminf(ax2, bx2);
    const float inter = fmaxf(iy2 - iy1, 0.0f) * fmaxf(ix2 - ix1, 0.0f);
    const float uni = (aar + bar) - inter;
    const float prod = thr * uni;
    // |prod - thr*uni| <= 2^-24 |prod| and |fl(inter/uni) - inter/uni| <= 2^-24 |inter/uni|: outside a
    // relative band of 2^-20 around equality both tests agree; inside it (and for non-finite or
    // non-positive operands, where the band test is false) the quotient decides
    const float band = 9.5367431640625e-07f * fabsf(prod);
    if (prod > 1e-30f) {                       // normal range only (uni > 0 and thr > 0 follow)
        if (inter > prod + band) return true;
        if (inter < prod - band) return false;
    }
    return (inter / uni) > thr;
}

// Descending sort of n keys (all different, zero = padding allowed only as ties at the end)
// by rank: every thread counts, for each key it owns, the keys greater than it -- n
// broadcast reads of LDS, no barrier inside -- and writes the key to its rank in `out`.
// For a few hundred keys this is several times quicker than a sorting network, whose every
// stage ends in a barrier.
__device__ __forceinline__ void rank_sort_desc(const u64 *in, u64 *out, int n, int t) {
    // in[n .. n8) must be zero (n8 = n rounded up to 8): the scan reads eight keys per step
    const int n8 = (n + 7) & ~7;
    const ulonglong2 *in2 = (const ulonglong2 *)in;
    for (int i0 = 0; i0 < n; i0 += 512) {
        const int ia = i0 + t, ib = i0 + 256 + t;
        const u64 ka = ia < n ? in[ia] : ~0ull, kb = ib < n ? in[ib] : ~0ull;
        int ra = 0, rb = 0;
        for (int j = 0; j < n8; j += 8) {
            const ulonglong2 p0 = in2[(j >> 1)], p1 = in2[(j >> 1) + 1], p2 = in2[(j >> 1) + 2], p3 = in2[(j >> 1) + 3];
            ra += (p0.x > ka) + (p0.y > ka) + (p1.x > ka) + (p1.y > ka) + (p2.x > ka) + (p2.y > ka) + (p3.x > ka) + (p3.y > ka);
            rb += (p0.x > kb) + (p0.y > kb) + (p1.x > kb) + (p1.y > kb) + (p2.x > kb) + (p2.y > kb) + (p3.x > kb) + (p3.y > kb);
        }
        if (ia < n) out[ra] = ka;
        if (ib < n) out[rb] = kb;
    }
}

// Suffix scan of a 1024-bin histogram by 256 threads: returns (through *s_cutbin, which
// must be 0 on entry, and after a barrier) the highest bin b such that bins b.. hold at
// least `target` entries, or leaves 0 if all bins together hold fewer.
__device__ __forceinline__ void hist_cut(const int *hist, int target, int t, int *s_wsum, int *s_cutbin) {
    const int lane = t & 63;
    int h[4], local = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = hist[NMS_BINS - 1 - (4 * t + e)]; local += h[e]; }
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wsum[t >> 6] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < (t >> 6); ++w) run += s_wsum[w];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool was_below = run < target;
        run += h[e];
        if (was_below && run >= target) *s_cutbin = NMS_BINS - 1 - (4 * t + e);
    }
    __syncthreads();
}

// One 1024-candidate chunk of a score row as this thread sees it: four consecutive scores.
__device__ __forceinline__ float4 load_chunk(const float *__restrict__ col, int i0, int M) {
    // rows are padded to a multiple of 4 (cap % 4 == 0), so a vector that starts below M is in bounds
    return i0 < M ? *(const float4 *)(col + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256, 4)   // four workgroups per CU (128 registers); a fifth does not pay and three lose 8 %
nms_class_kernel(const NmsArgs a) {
    __shared__ __attribute__((aligned(16))) u64 buf[NMS_BUF + 8];
    __shared__ int hist[NMS_BINS];
    __shared__ int s_wsum[4];
    __shared__ u64 s_last, s_last2, s_kmin[4];
    __shared__ u32 s_red[8];
    __shared__ int s_cnt, s_cnt2, s_nkept, s_cutbin;
    // greedy pass: the 64 candidates of a batch and the boxes kept so far (normalised corners,
    // area), the candidates' suppression rows and who is dead on arrival
    __shared__ float4 s_cbox[64], s_kbox[64];
    __shared__ float s_car[64], s_kar[64];
    __shared__ u64 s_ckey[64], s_kkey[64], s_row[64], s_dead[4];

    const int c = blockIdx.x, b = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63;
    int M = a.counts ? a.counts[b] : a.cap;
    M = M > a.cap ? a.cap : (M < 0 ? 0 : M);
    const float *__restrict__ col = a.scores + ((size_t)b * a.C + c) * a.cap;
    const float4 *__restrict__ boxes4 = (const float4 *)a.boxes + (size_t)b * a.cap;
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int PC = a.per_class;

    // Histogram geometry: bin = (ord(score) - ord(thr)) >> hshift, clamped to the top bin;
    // scores normally lie in (thr, 1], anything above lands in the top bin.
    const u32 ob_thr = ord_bits(a.score_thr);
    int hshift = 0;
    {
        const u32 range = (a.score_thr < 1.0f) ? ord_bits(1.0f) - ob_thr : (1u << 23);
        while ((range >> hshift) >= (u32)NMS_BINS) ++hshift;
    }
    // candidates wanted above the floor: four times what may be kept (the greedy pass seldom
    // visits more before per_class boxes stand); every further tier doubles it
    int target = 4 * PC;
    target = target < 64 ? 64 : (target > NMS_KEEP ? NMS_KEEP : target);

    // The floor comes from a SAMPLE of the row: every k-th chunk (k grows with the row, up to
    // 8), and the bin edge above which the sample holds its share of `target` candidates plus
    // three standard deviations.  A sampled floor can only be too low (more keys than needed:
    // the buffer is cut) or too high (fewer than per_class boxes survive: a further tier
    // rescans below it); the result never depends on it, and a long row is read 1 + 1/k times
    // instead of twice.
    const int nchunk = (M + NMS_CHUNK - 1) / NMS_CHUNK;
    int kstep = nchunk / 4;
    kstep = kstep < 1 ? 1 : (kstep > 8 ? 8 : kstep);

    if (t == 0) s_nkept = 0;
    u64 ceil_key = ~0ull;           // keys must be < ceil_key

    NSTAMP(0);
    for (;;) {   // one pass per tier
        // ---- starting floor: the highest bin edge with at least `starget` sampled candidates above it
        int starget = target;
        if (kstep > 1) {
            const float share = (float)target / (float)kstep;
            starget = (int)(share + 3.0f * sqrtf(share)) + 4;
        }
        for (int i = t; i < NMS_BINS; i += 256) hist[i] = 0;
        if (t == 0) { s_cnt = 0; s_cutbin = 0; }
        __syncthreads();
        for (int j0 = 0; j0 < nchunk; j0 += 4 * kstep) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load_chunk(col, (j0 + u * kstep) * NMS_CHUNK + t * 4, (j0 + u * kstep) < nchunk ? M : 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i0 = (j0 + u * kstep) * NMS_CHUNK + t * 4;
                const float sv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = i0 + e;
                    const float s = sv[e];
                    const u32 ob = ord_bits(s);
                    const u64 key = ((u64)ob << 32) | (u64)(0xffffffffu - (u32)idx);
                    if ((j0 + u * kstep) < nchunk && idx < M && s > a.score_thr && key < ceil_key) {
                        const u32 bin = (ob - ob_thr) >> hshift;
                        atomicAdd(&hist[bin < (u32)NMS_BINS ? bin : NMS_BINS - 1], 1);
                    }
                }
            }
        }
        __syncthreads();
        NSTAMP(1);
        hist_cut(hist, starget, t, s_wsum, &s_cutbin);
        const int cutbin = s_cutbin;   // 0: fewer than `starget` sampled candidates in all, take everything
        u64 floor_key = cutbin > 0 ? (((u64)(ob_thr + ((u32)cutbin << hshift))) << 32) - 1ull : 0ull;   // keys must be > floor_key
        bool cut = cutbin > 0;

        // ---- key pass, fast form: no barrier inside the loop.  Keys above the floor are
        // appended (ballot + one LDS atomic per wave); positions beyond the buffer are counted
        // but not stored, and if that happened the pass is repeated in the careful form below.
        // The next four chunks are requested before the current four are looked at, so a
        // workgroup always has loads in flight; a 16-byte group of scores is examined
        // element by element only if some lane's group holds a score at or above the floor
        // (one float compare per element; a superset of the exact key test that follows).
        {
            const float floor_f = cut ? unord_bits((u32)((floor_key + 1ull) >> 32)) : -INFINITY;
            const float qthr = fmaxf(floor_f, a.score_thr);
            float4 v[4], nx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load_chunk(col, u * NMS_CHUNK + t * 4, M);
            for (int j0 = 0; j0 < nchunk; j0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) nx[u] = load_chunk(col, (j0 + 4 + u) * NMS_CHUNK + t * 4, M);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float mx = fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
                    if (__ballot(mx >= qthr) == 0ull) continue;
                    const int i0 = (j0 + u) * NMS_CHUNK + t * 4;
                    const float sv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = i0 + e;
                        const float s = sv[e];
                        const u64 key = ((u64)ord_bits(s) << 32) | (u64)(0xffffffffu - (u32)idx);
                        const bool pred = idx < M && s > a.score_thr && key > floor_key && key < ceil_key;
                        const u64 bal = __ballot(pred);
                        if (bal) {
                            const int leader = __ffsll((long long)bal) - 1;
                            int pos = 0;
                            if (lane == leader) pos = atomicAdd(&s_cnt, __popcll(bal));
                            pos = __shfl(pos, leader);
                            pos += __popcll(bal & below);
                            if (pred && pos < NMS_BUF) buf[pos] = key;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = nx[u];
            }
        }
        __syncthreads();
        NSTAMP(2);
        if (s_cnt > NMS_BUF) {
            // ---- careful form: capacity checked after every chunk; when the buffer could
            // overflow it is sorted, cut to its best NMS_KEEP keys and the floor is raised
            __syncthreads();
            if (t == 0) s_cnt = 0;
            __syncthreads();
            for (int base = 0; base < M; base += NMS_CHUNK) {
                const int i0 = base + t * 4;
                const float4 v = load_chunk(col, i0, M);
                const float sv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = i0 + e;
                    const float s = sv[e];
                    const u64 key = ((u64)ord_bits(s) << 32) | (u64)(0xffffffffu - (u32)idx);
                    const bool pred = idx < M && s > a.score_thr && key > floor_key && key < ceil_key;
                    const u64 bal = __ballot(pred);
                    if (bal) {
                        const int leader = __ffsll((long long)bal) - 1;
                        int pos = 0;
                        if (lane == leader) pos = atomicAdd(&s_cnt, __popcll(bal));
                        pos = __shfl(pos, leader);
                        if (pred) buf[pos + __popcll(bal & below)] = key;
                    }
                }
                __syncthreads();
                const int cnt = s_cnt;
                // every wave must have read the count before any wave appends the next chunk:
                // the branch below contains barriers, so the decision has to be uniform
                __syncthreads();
                if (cnt > NMS_BUF - NMS_CHUNK) {
                    for (int i = cnt + t; i < NMS_BUF; i += 256) buf[i] = 0;
                    __syncthreads();
                    bitonic_desc(buf, NMS_BUF, t);
                    floor_key = buf[NMS_KEEP - 1];
                    cut = true;
                    __syncthreads();
                    if (t == 0) s_cnt = NMS_KEEP;
                    __syncthreads();
                }
            }
        }
        const int n = s_cnt;
        // Greedy suppression over `cnt` keys in visiting order, by the whole workgroup, 64
        // candidates at a time: every IoU test of a batch runs in parallel -- candidate x box
        // already kept (each wave takes a quarter of the kept list) and candidate x earlier
        // candidate (each wave takes 16 rows of the 64 x 64 matrix, one ballot per row) -- and
        // what is left for one wave to do in order is bit arithmetic: the first live candidate
        // is kept and its row is cleared from the live mask.  Same decisions as testing one by
        // one: a candidate is kept iff no box kept before it overlaps it beyond the threshold.
        auto greedy = [&](const u64 *sorted, int cnt) {
            for (int base = 0; base < cnt; base += 64) {
                const int nk0 = s_nkept;
                if (nk0 >= PC) break;
                if (t < 64) {
                    const int e = base + t;
                    const bool valid = e < cnt;
                    const u64 key = valid ? sorted[e] : 0ull;
                    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid) bx = boxes4[0xffffffffu - (u32)key];
                    const float y1 = fminf(bx.x, bx.z), x1 = fminf(bx.y, bx.w);
                    const float y2 = fmaxf(bx.x, bx.z), x2 = fmaxf(bx.y, bx.w);
                    s_cbox[t] = make_float4(y1, x1, y2, x2);
                    s_car[t] = (y2 - y1) * (x2 - x1);
                    s_ckey[t] = key;
                }
                __syncthreads();
                {
                    const float4 cb4 = s_cbox[lane];
                    const float car = s_car[lane];
                    const int w = t >> 6;
                    // against the boxes kept in earlier batches: this wave's quarter of them
                    bool dead = false;
                    const int per = (nk0 + 3) >> 2;
                    for (int j = w * per; j < (w + 1) * per && j < nk0; ++j) {
                        const float4 kb = s_kbox[j];
                        if (iou_gt(cb4.x, cb4.y, cb4.z, cb4.w, car, kb.x, kb.y, kb.z, kb.w, s_kar[j], a.iou_thr)) dead = true;
                    }
                    const u64 dm = __ballot(dead);
                    if (lane == 0) s_dead[w] = dm;
                    // within the batch: row i = the later candidates that candidate i suppresses
                    for (int r = 0; r < 16; ++r) {
                        const int i = w * 16 + r;
                        const float4 lb = s_cbox[i];
                        const bool hit = lane > i && iou_gt(cb4.x, cb4.y, cb4.z, cb4.w, car, lb.x, lb.y, lb.z, lb.w, s_car[i], a.iou_thr);
                        const u64 rm = __ballot(hit);
                        if (lane == 0) s_row[i] = rm;
                    }
                }
                __syncthreads();
                if (t < 64) {
                    const int nv = cnt - base < 64 ? cnt - base : 64;
                    u64 alive = (nv == 64 ? ~0ull : ((1ull << nv) - 1ull)) & ~(s_dead[0] | s_dead[1] | s_dead[2] | s_dead[3]);
                    const u64 myrow = s_row[lane];
                    u64 keptmask = 0;
                    int nk = nk0;
                    while (alive != 0 && nk < PC) {
                        const int i = __ffsll((long long)alive) - 1;
                        keptmask |= 1ull << i;
                        ++nk;
                        const u32 rlo = __shfl((u32)myrow, i), rhi = __shfl((u32)(myrow >> 32), i);
                        alive &= ~((((u64)rhi) << 32) | rlo);
                        alive &= ~(1ull << i);
                    }
                    if ((keptmask >> lane) & 1ull) {
                        const int slot = nk0 + __popcll(keptmask & below);
                        s_kbox[slot] = s_cbox[lane];
                        s_kar[slot] = s_car[lane];
                        s_kkey[slot] = s_ckey[lane];
                    }
                    if (t == 0) s_nkept = nk;
                }
                __syncthreads();
            }
        };

        if (n > 0 && n <= NMS_BUF / 2 - 8) {
            // ---- the usual case.  The buffer holds a few hundred unsorted keys (a histogram bin is
            // coarse next to the few dozen keys the greedy pass will visit), so they are not sorted
            // as a whole: a second histogram, spanning exactly the scores in the buffer, picks the
            // best `t2` or so, only those are sorted (by rank) and visited; if that does not yield
            // per_class boxes the next best group below them follows, all inside LDS.
            u64 *sel = buf + NMS_BUF / 2;                  // [<= 1024] selection, its sorted copy behind it
            const int t2 = PC < 22 ? 64 : 3 * PC;
            u64 lceil = ~0ull;                             // this group's keys are < lceil
            for (;;) {
                u32 lmin = 0xffffffffu, lmax = 0u;
                u64 kmin = ~0ull;
                for (int i = t; i < n; i += 256) {
                    const u64 k = buf[i];
                    kmin = k < kmin ? k : kmin;
                    if (k < lceil) {
                        const u32 o = (u32)(k >> 32);
                        lmin = o < lmin ? o : lmin;
                        lmax = o > lmax ? o : lmax;
                    }
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const u32 omn = __shfl_xor(lmin, d), omx = __shfl_xor(lmax, d);
                    const u32 klo = __shfl_xor((u32)kmin, d), khi = __shfl_xor((u32)(kmin >> 32), d);
                    const u64 ok = ((u64)khi << 32) | klo;
                    lmin = omn < lmin ? omn : lmin;
                    lmax = omx > lmax ? omx : lmax;
                    kmin = ok < kmin ? ok : kmin;
                }
                for (int i = t; i < NMS_BINS; i += 256) hist[i] = 0;
                if (lane == 0) { s_red[t >> 6] = lmin; s_red[4 + (t >> 6)] = lmax; s_kmin[t >> 6] = kmin; }
                if (t == 0) { s_cnt2 = 0; s_cutbin = 0; }
                __syncthreads();
                u32 omin = s_red[0], omax = s_red[4];
                kmin = s_kmin[0];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    omin = s_red[w] < omin ? s_red[w] : omin;
                    omax = s_red[4 + w] > omax ? s_red[4 + w] : omax;
                    kmin = s_kmin[w] < kmin ? s_kmin[w] : kmin;
                }
                if (omax < omin) {                         // nothing left below lceil: the buffer is used up
                    if (t == 0) s_last = kmin;
                    break;
                }
                int sh2 = 0;
                while (((omax - omin) >> sh2) >= (u32)NMS_BINS) ++sh2;
                for (int i = t; i < n; i += 256) {
                    const u64 k = buf[i];
                    if (k < lceil) atomicAdd(&hist[((u32)(k >> 32) - omin) >> sh2], 1);
                }
                __syncthreads();
                hist_cut(hist, t2, t, s_wsum, &s_cutbin);
                const int cb2 = s_cutbin;                  // 0: fewer than t2 keys left, take them all
                for (int i0 = 0; i0 < n; i0 += 256) {
                    const int i = i0 + t;
                    const u64 k = i < n ? buf[i] : 0ull;
                    const bool pred = i < n && k < lceil && (int)(((u32)(k >> 32) - omin) >> sh2) >= cb2;
                    const u64 bal = __ballot(pred);
                    if (bal) {
                        const int leader = __ffsll((long long)bal) - 1;
                        int pos = 0;
                        if (lane == leader) pos = atomicAdd(&s_cnt2, __popcll(bal));
                        pos = __shfl(pos, leader);
                        if (pred) sel[pos + __popcll(bal & below)] = k;
                    }
                }
                __syncthreads();
                NSTAMP(5);
                const int nsel = s_cnt2;
                const u64 *sorted;
                if (nsel <= 500) {
                    if (t < 8) sel[nsel + t] = 0;          // the rank scan's padding
                    __syncthreads();
                    rank_sort_desc(sel, sel + 512, nsel, t);
                    sorted = sel + 512;
                } else {
                    int nsort = 512;
                    while (nsort < nsel) nsort <<= 1;
                    for (int i = nsel + t; i < nsort; i += 256) sel[i] = 0;
                    __syncthreads();
                    bitonic_desc(sel, nsort, t);
                    sorted = sel;
                }
                __syncthreads();
                NSTAMP(6);
                greedy(sorted, nsel);
                NSTAMP(7);
                if (t == 0) { s_last = kmin; s_last2 = sorted[nsel - 1]; }
                __syncthreads();
                if (s_nkept >= PC || cb2 == 0) break;      // done, or every key of the buffer visited
                lceil = s_last2;
                __syncthreads();
            }
        } else {
            // ---- a buffer more than half full (many equal scores): sort all of it
            int nsort = 64;
            while (nsort < n) nsort <<= 1;
            for (int i = n + t; i < nsort; i += 256) buf[i] = 0;
            __syncthreads();
            bitonic_desc(buf, nsort, t);
            greedy(buf, n);
            if (t == 0) s_last = n > 0 ? buf[n - 1] : 0ull;
        }
        NSTAMP(3);
        __syncthreads();
        NSTAMP(4);
        const int nk = s_nkept;
        if (nk >= PC || !cut || n == 0) break;
        ceil_key = s_last;
        target = target * 2 > NMS_KEEP ? NMS_KEEP : target * 2;
        __syncthreads();
    }
    if (t < 64) {
        // every slot of the class is written (zero = empty), so the merge reads them unconditionally
        const int nkept = s_nkept;
        if (lane < PC) a.kept_keys[((size_t)b * a.C + c) * PC + lane] = lane < nkept ? s_kkey[lane] : 0ull;
        if (lane == 0) a.kept_cnt[b * a.C + c] = nkept;
    }
}

// One workgroup per image: the per-class survivors (already in visiting order per class) are
// merged by (score desc, class asc, index asc).  Only the first `total` are wanted, so the
// keys are first cut at the highest score-histogram bin edge that leaves at least `total` of
// them (same histogram as kernel 1), and only those -- a few dozen instead of C * per_class --
// are sorted.
__global__ void __launch_bounds__(256)
nms_merge_kernel(const NmsArgs a, const int npow2) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64 *keys = (u64 *)smem_raw;                 // [npow2] all keys, then the sorted selection
    u64 *sel = keys + npow2;                     // [npow2] selection
    int *hist = (int *)(sel + npow2);            // [NMS_BINS]
    int *s_misc = hist + NMS_BINS;               // [0..3] wave sums, [4] cut bin, [5] selected, [8..11] / [12..15] wave minima / maxima
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int PC = a.per_class, nslots = a.C * PC;
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int i = t; i < NMS_BINS; i += 256) hist[i] = 0;
    if (t < 8) s_misc[t] = 0;                    // [8..15] are written by every wave before they are read
    // all slots of the image are contiguous and empty ones are zero: plain coalesced loads,
    // independent of each other (several in flight per thread)
    const u64 *__restrict__ kk = a.kept_keys + (size_t)b * nslots;
    u32 lmin = 0xffffffffu, lmax = 0u;
#pragma unroll 4
    for (int i = t; i < npow2; i += 256) {
        u64 key = 0;
        const u64 k1 = i < nslots ? kk[i] : 0ull;
        if (k1 != 0) {
            const int c = i / PC;
            const u32 idx = 0xffffffffu - (u32)k1;
            key = (k1 & 0xffffffff00000000ull) | ((u64)(0x7ffu - (u32)c) << 21) | (u64)(0x1fffffu - idx);
            const u32 o = (u32)(k1 >> 32);
            lmin = o < lmin ? o : lmin;
            lmax = o > lmax ? o : lmax;
        }
        keys[i] = key;
    }
    // The histogram spans exactly the scores present (the survivors of a long candidate list
    // all sit within a hair of each other, far above the threshold): bins of the smallest
    // power-of-two width that covers [min, max] in NMS_BINS bins.
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const u32 omn = __shfl_xor(lmin, d), omx = __shfl_xor(lmax, d);
        lmin = omn < lmin ? omn : lmin;
        lmax = omx > lmax ? omx : lmax;
    }
    if (lane == 0) { s_misc[8 + (t >> 6)] = (int)lmin; s_misc[12 + (t >> 6)] = (int)lmax; }
    __syncthreads();
    u32 omin = (u32)s_misc[8], omax = (u32)s_misc[12];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const u32 mn = (u32)s_misc[8 + w], mx = (u32)s_misc[12 + w];
        omin = mn < omin ? mn : omin;
        omax = mx > omax ? mx : omax;
    }
    int hshift = 0;
    if (omax >= omin)
        while (((omax - omin) >> hshift) >= (u32)NMS_BINS) ++hshift;
    auto bin_of = [&](u64 key) {
        const u32 bin = ((u32)(key >> 32) - omin) >> hshift;
        return bin < (u32)NMS_BINS ? (int)bin : NMS_BINS - 1;
    };
    for (int i = t; i < npow2; i += 256) {
        const u64 key = keys[i];
        if (key != 0) atomicAdd(&hist[bin_of(key)], 1);
    }
    __syncthreads();
    hist_cut(hist, a.total, t, s_misc, s_misc + 4);
    const int cutbin = s_misc[4];
    // order-free compaction of the keys at or above the cut
    for (int i0 = 0; i0 < npow2; i0 += 256) {
        const u64 key = keys[i0 + t];
        const bool pred = key != 0 && bin_of(key) >= cutbin;
        const u64 bal = __ballot(pred);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            int pos = 0;
            if (lane == leader) pos = atomicAdd(&s_misc[5], __popcll(bal));
            pos = __shfl(pos, leader);
            if (pred) sel[pos + __popcll(bal & below)] = key;
        }
    }
    __syncthreads();
    const int nsel = s_misc[5];
    int nsort = 64;
    while (nsort < nsel) nsort <<= 1;
    for (int i = nsel + t; i < nsort; i += 256) sel[i] = 0;
    __syncthreads();
    bitonic_desc(sel, nsort, t);
    const int nvalid = nsel < a.total ? nsel : a.total;
    if (t == 0) a.out_valid[b] = nvalid;
    for (int j = t; j < a.total; j += 256) {
        float4 ob = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f, cl = 0.f;
        int kidx = -1;
        if (j < nvalid) {
            const u64 key = sel[j];
            const u32 c = 0x7ffu - (u32)((key >> 21) & 0x7ffu);
            const u32 idx = 0x1fffffu - (u32)(key & 0x1fffffu);
            const float4 bx = ((const float4 *)a.boxes)[(size_t)b * a.cap + idx];
            ob = make_float4(fminf(fmaxf(bx.x, 0.f), 1.f), fminf(fmaxf(bx.y, 0.f), 1.f),
                             fminf(fmaxf(bx.z, 0.f), 1.f), fminf(fmaxf(bx.w, 0.f), 1.f));
            sc = unord_bits((u32)(key >> 32));
            cl = (float)c;
            kidx = a.src_idx ? a.src_idx[(size_t)b * a.cap + idx] : (int)idx;
        }
        ((float4 *)a.out_boxes)[(size_t)b * a.total + j] = ob;
        a.out_scores[(size_t)b * a.total + j] = sc;
        a.out_classes[(size_t)b * a.total + j] = cl;
        a.out_keep[(size_t)b * a.total + j] = kidx;
    }
}

int launch_nms(const NmsArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(nms_class_kernel, dim3(a.C, a.B), dim3(256), 0, s, a);
    int rc = check_launch("nms_class");
    if (rc) return rc;
    int npow2 = 256;
    while (npow2 < a.C * a.per_class) npow2 <<= 1;
    const size_t merge_lds = (size_t)npow2 * 16 + NMS_BINS * 4 + 64;
    if (merge_lds > 64 * 1024) {
        if (merge_lds > 160 * 1024) return fail(YMX_EUNSUPPORTED, "nms: C * per_class = %d too large", a.C * a.per_class);
        static LdsReserve reserved;
        if (const int rc = reserve_lds(reserved, (const void *)nms_merge_kernel, 160 * 1024, "nms_merge")) return rc;
    }
    hipLaunchKernelGGL(nms_merge_kernel, dim3(a.B), dim3(256), merge_lds, s, a, npow2);
    return check_launch("nms_merge");
}

size_t nms_workspace_bytes(int B, int C, int per_class) {
    if (B <= 0 || C <= 0 || per_class <= 0) return 0;
    return (size_t)B * C * per_class * 8 + (((size_t)B * C * 4 + 15) & ~(size_t)15);
}

int nms_entry(const float *boxes, const float *scores, const int *counts, const int *src_idx, int B, int cap, int C,
              int per_class, int total, float iou_thr, float score_thr, float *out_boxes, float *out_scores,
              float *out_classes, int *out_valid, int *out_keep_idx, void *workspace, size_t ws_bytes, hipStream_t s) {
    if (!boxes || !scores || !out_boxes || !out_scores || !out_classes || !out_valid || !out_keep_idx)
        return fail(YMX_EARG, "nms_combined: null pointer");
    if (B <= 0 || cap <= 0 || C <= 0 || per_class <= 0 || total <= 0) return fail(YMX_EARG, "nms_combined: non-positive size");
    if (cap & 3) return fail(YMX_EARG, "nms_combined: cap %d must be a multiple of 4", cap);
    if (per_class > 64 || C > 2048 || (long long)C * per_class > 8192 || cap > (1 << 21))
        return fail(YMX_EUNSUPPORTED, "nms_combined: limits per_class<=64, C<=2048, C*per_class<=8192, cap<=2097152");
    if (!aligned16(boxes) || !aligned16(scores) || !aligned16(out_boxes)) return fail(YMX_EARG, "nms_combined: 16-byte alignment");
    const size_t need = nms_workspace_bytes(B, C, per_class);
    if (!workspace || ws_bytes < need) return fail(YMX_EWORKSPACE, "nms_combined: workspace %zu < %zu bytes", ws_bytes, need);
    if (!aligned16(workspace)) return fail(YMX_EARG, "nms_combined: workspace must be 16-byte aligned");
    NmsArgs a;
    memset(&a, 0, sizeof(a));
    a.boxes = boxes; a.scores = scores; a.counts = counts; a.src_idx = src_idx;
    a.B = B; a.cap = cap; a.C = C; a.per_class = per_class; a.total = total;
    a.iou_thr = iou_thr; a.score_thr = score_thr;
    a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_classes = out_classes;
    a.out_valid = out_valid; a.out_keep = out_keep_idx;
    a.kept_keys = (unsigned long long *)workspace;
    a.kept_cnt = (int *)((char *)workspace + (size_t)B * C * per_class * 8);
    return launch_nms(a, s);
}

// ---- the entry point that allocates nothing ---------------------------------------------
// Per-class survivors (keys and counts, the same layout ymx_nms_combined carves out of the
// caller's workspace) live in a device array that is part of the code object: every GPU the
// library runs on holds its own copy, at its own address, until the process ends.
static __device__ __align__(256) unsigned char scratch[YMX_NMS_STATIC_SCRATCH_BYTES];

// `scratch` names a host-side handle, not device memory: the device address is looked up once
// per device (with that device current) and only that address is ever handed to a kernel.
static std::atomic<void *> g_scratch_ptr[64];

static int scratch_address(void **out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return fail(YMX_ELAUNCH, "nms_combined_static: no current device");
    }
    if (dev < 0 || dev >= 64) return fail(YMX_EUNSUPPORTED, "nms_combined_static: device index %d (at most 64 devices)", dev);
    void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            (void)hipGetLastError();
            return fail(YMX_ELAUNCH, "nms_combined_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        }
        g_scratch_ptr[dev].store(p, std::memory_order_release);      // every thread finds the same address
    }
    *out = p;
    return YMX_OK;
}

}  // namespace ymx

extern "C" int ymx_nms_combined_static(const float *boxes, const float *scores, const int *counts, const int *src_idx,
                                       int B, int cap, int C, int per_class, int total, float iou_thr,
                                       float score_thr, float *out_boxes, float *out_scores, float *out_classes,
                                       int *out_valid, int *out_keep_idx, void *stream) {
    using namespace ymx;
    if (B <= 0 || C <= 0 || per_class <= 0) return fail(YMX_EARG, "nms_combined_static: non-positive size");
    const size_t need = nms_workspace_bytes(B, C, per_class);
    if (need > (size_t)YMX_NMS_STATIC_SCRATCH_BYTES)
        return fail(YMX_EUNSUPPORTED, "nms_combined_static: B*C*per_class = %lld needs %zu bytes of scratch, the built-in "
                    "buffer holds %d (use ymx_nms_combined with a workspace)", (long long)B * C * per_class, need,
                    YMX_NMS_STATIC_SCRATCH_BYTES);
    void *ws = nullptr;
    if (const int rc = scratch_address(&ws)) return rc;
    // Nothing in the buffer has to start at zero: kernel 1 writes the count of every (image,
    // class) and kernel 2 reads only the keys below that count.
    return nms_entry(boxes, scores, counts, src_idx, B, cap, C, per_class, total, iou_thr, score_thr, out_boxes,
                     out_scores, out_classes, out_valid, out_keep_idx, ws, need, (hipStream_t)stream);
}

YMX_STAMP_EXPORT(ymx_debug_stamps_nms)
