// Head decode, score threshold and order-preserving compaction for gfx950.
//
// One launch handles every scale and every image.  A *segment* is 64 consecutive candidates
// of one scale of one image.  Workgroups of 256 threads are persistent: blockIdx.x is the
// image, and the workgroups of an image take its segments one after another by ticket.  The
// 64 rows of (5+C) logits of a segment, contiguous in memory, go to LDS (fp32 maps: by
// LDS-DMA, no registers involved); wave q (one lane per candidate) evaluates class slice q,
// so LDS is read at a row stride of 5+C words (odd for C = 80: conflict-free) and the
// class-major score matrix is written with one 4-byte element per lane to consecutive
// addresses.
//
// Compaction: each candidate's max score is reduced over the four class slices through LDS,
// a wavefront ballot turns the 64 keep flags into a bit mask, and a survivor's slot is
// segment base + popcount of the mask below its lane.  Segment bases come from a decoupled
// look-back over 8-byte {status, count} words (one relaxed agent-scope store / load each, so
// a word is never torn and needs no fence), read a window of 64 predecessors at a time by
// wave 0.  Tickets are drawn per image in increasing order, so every predecessor of a waiting
// segment is already being worked on; spins are bounded and raise an error word instead of
// hanging.
//
// The loop is software-pipelined around the two round trips that would otherwise idle the
// CU's LDS: a segment's count is published as soon as it is known; then wave 1 draws the next
// ticket and starts that segment's rows towards LDS; and the look-back and the stores are
// those of the PREVIOUS segment, whose predecessors published an iteration ago (its scores
// wait in registers).  Measured on 32 x 100 800 candidates, 80 classes: 0.61 ms for the
// one-segment-per-workgroup form with a serial look-back, 0.51 ms for this one; without any
// look-back the same traffic takes 0.42 ms, the reads alone 0.21 ms.
//
// Algorithmic bytes: N*(5+C)*e read once, M*(4+C+1)*4 written once.
#include "lds_dma.h"

namespace ymx {

typedef unsigned long long u64;

// Diagnostic build only (make stamps): the 100 MHz wall clock at six points of a workgroup's
// life, read back by tools/stamp_decode.py.
#ifdef YMX_STAMPS
#define DSTAMP(k)                                                                               \
    do {                                                                                        \
        const unsigned lin_ = blockIdx.x + gridDim.x * blockIdx.y;                              \
        if (threadIdx.x == 0 && lin_ < YMX_STAMP_BLOCKS && dstamp_it == 3) ymx_stamp_buf[lin_ * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif

#define YMX_ST_AGG (1ull << 62)
#define YMX_ST_PREFIX (2ull << 62)

// Words of LDS for the rows of 64 candidates: the run itself (padded cells included), up to
// three words of alignment shift, rounded up to whole kilobytes (an LDS-DMA instruction
// writes 64 lanes x 16 bytes).
__host__ __device__ inline int decode_stage_words(int D, int max_pad) { return ((64 * D + 22 * max_pad + 4) * 4 + 1023) / 1024 * 256; }

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
    auto put = [&](int v, const uint4 u) {
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
    };
    // every load of a batch is issued before the first LDS write waits for one
    constexpr int NB = 6;
    for (int v0 = 0; v0 < nvec; v0 += NB * 256) {
        uint4 u[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = v0 + i * 256 + t;
            u[i] = v < nvec ? vsrc[v] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = v0 + i * 256 + t;
            if (v < nvec) put(v, u[i]);
        }
    }
    if (t < count - tail0) lds[tail0 + t] = elem<HT>::load(src + tail0 + t);
}

// Where a segment of 64 candidates lies: its scale, its candidates, and (MODE 0/1) the run of
// memory that holds their rows.
struct SegGeom {
    int sc, S, Ns;          // scale, grid side, candidates of the scale (MODE 2: of the image)
    int c0, c_global0;      // first candidate: within the scale / in the concatenated list
    int nvalid;             // candidates of the segment (<= 64)
    int cell0, a0, ld;      // first grid cell, anchor of the first candidate, elements per cell
    int span;               // elements from the first row's start to the last row's end
    int shift;              // words between the start of the LDS image and the first row (LDS-DMA path)
};

// MODE 0: decode only (every candidate written at its own index, boxes = xywh pixels)
// MODE 1: decode + filter + compaction (boxes = yxyx / image_size)
// MODE 2: filter + compaction of already decoded input (HT unused)
//
// Workgroups are persistent: blockIdx.x is the image, and the gridDim.y workgroups of an
// image take its segments one after another (MODE 0: statically interleaved; otherwise by
// ticket).  The loop is software-pipelined around the look-back: a segment's survivor count
// is published as soon as it is known, then the rows of the workgroup's NEXT segment are
// started on their way from memory to LDS, and only then does wave 0 wait for the counts of
// the segments before it -- so the wait overlaps a transfer instead of idling the CU's LDS.
template <typename HT, int CPT, int MODE, bool DMA>
// six workgroups per CU is what the LDS allows at 80 classes: the LDS-DMA form keeps its
// registers within that; the register-staged form needs more and runs four per CU
__global__ void __launch_bounds__(256, DMA ? 6 : 4)
decode_kernel(const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int D = 5 + a.C;
    float *stage = (float *)smem_raw;                       // rows of 64 candidates (MODE 0/1)
    const int stage_words = (MODE == 2) ? 0 : decode_stage_words(D, a.max_pad);
    float *smax = stage + stage_words;                      // [4][64]
    int *sint = (int *)(smax + 256);                        // [0] next segment, [1] base
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem_raw;

    const int t = threadIdx.x, lane = t & 63;
    const int q = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.x;
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    auto geometry = [&](int seg) {
        SegGeom g;
        g.sc = 0;
        if (a.n_scales > 1 && seg >= a.seg_begin[1]) g.sc = 1;
        if (a.n_scales > 2 && seg >= a.seg_begin[2]) g.sc = 2;
        g.S = a.S[g.sc];
        g.Ns = (MODE == 2) ? a.N : 3 * g.S * g.S;
        g.c0 = (seg - a.seg_begin[g.sc]) * DEC_SEG;
        g.c_global0 = a.cand_begin[g.sc] + g.c0;
        int nv = g.Ns - g.c0;
        g.nvalid = nv < 0 ? 0 : (nv > 64 ? 64 : nv);
        g.ld = a.cell_ld[g.sc] ? a.cell_ld[g.sc] : 3 * D;
        g.cell0 = g.c0 / 3;
        g.a0 = g.c0 - g.cell0 * 3;
        g.span = 0;
        if (g.nvalid > 0) {
            // contiguous when the cells are dense (cell_ld = 3*D); with padded cells the span from
            // the first to the last candidate is staged as it lies, padding included
            const int cl = g.c0 + g.nvalid - 1, cellL = cl / 3;
            g.span = (cellL - g.cell0) * g.ld + (cl - cellL * 3 + 1) * D - g.a0 * D;
        }
        g.shift = 0;
        if (DMA) {
            const unsigned pre = (unsigned)((unsigned long long)a.head[g.sc] & 15ull);
            const unsigned o0 = pre + (unsigned)((((size_t)b * g.S * g.S + g.cell0) * g.ld + g.a0 * D) * 4);
            g.shift = (int)((o0 & 15u) >> 2);
        }
        return g;
    };

    // fp32 maps (DMA) go from memory to LDS without passing through registers: the run is
    // copied from the 16-byte boundary below its first element (g.shift = the words in
    // between), 1 KB per wave-instruction, out-of-range bytes arrive as zeros.  The calling
    // wave issues every instruction of the transfer and does not wait.
    auto dma_start = [&](const SegGeom &g) {
        const unsigned long long haddr = (unsigned long long)a.head[g.sc];
        const unsigned pre = (unsigned)(haddr & 15ull);
        const i32x4_t rs = make_rsrc((const void *)(haddr - pre), a.head_bytes[g.sc] + pre);
        const unsigned o0 = pre + (unsigned)((((size_t)b * g.S * g.S + g.cell0) * g.ld + g.a0 * D) * 4);
        const unsigned o0a = o0 & ~15u;
        const int ninstr = ((g.shift + g.span) * 4 + 1023) >> 10;
        for (int i = 0; i < ninstr; ++i)
            lds_dma16(rs, lds0 + (unsigned)(i * 1024), o0a + (unsigned)(i * 1024 + lane * 16));
    };
    // the register-staged copy (bf16 maps, maps beyond 4 GiB): every thread takes part, synchronous
    auto copy_rows = [&](const SegGeom &g) {
        const HT *src = (const HT *)a.head[g.sc] + ((size_t)b * g.S * g.S + g.cell0) * g.ld + g.a0 * D;
        stage_rows<HT>(stage, src, g.span, t);
    };

    // ---- first segment --------------------------------------------------------------------
    int seg;
    if (MODE == 0) {
        seg = blockIdx.y;
    } else {
        if (t == 0) sint[0] = atomicAdd(&a.ticket[b], 1);
        __syncthreads();
        seg = sint[0];
        __syncthreads();
    }
    if (seg >= a.nseg) return;
    SegGeom g = geometry(seg);
    if (MODE != 2) {
        if (DMA) {
            if (q == 1) dma_start(g);
        } else {
            copy_rows(g);
        }
    }

    // Loop state of the segment computed in the previous iteration and not yet written out.
    bool p_valid = false;
    int p_seg = 0, p_cnt = 0, p_cglobal0 = 0;
    u64 p_mask = 0;
    float p_score[CPT];
    float4 p_box = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < CPT; ++k) p_score[k] = 0.f;
    u64 *st = a.state + (size_t)b * a.nseg;
    const int l2_polls = (gridDim.x % 8 == 0) ? 16 : 0;     // see the look-back below

    int dstamp_it = 0;       // iteration counter (only the diagnostic build reads it)
    (void)dstamp_it;
    for (;;) {
        const bool have_cur = seg < a.nseg;                // uniform
        float score[CPT];
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        u64 mask = 0;
        int cnt = 0;
        DSTAMP(0);
        if (have_cur) {
            // only wave 1 issued the transfer
            if (DMA && q == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                               // rows of `seg` are in LDS
            DSTAMP(1);
            const float xs = a.xyscale[g.sc], strd = a.stride[g.sc];
            const float half = 0.5f * (xs - 1.0f);
            const bool live = lane < g.nvalid;
            float m = -INFINITY;
            if (MODE != 2) {
                if (live) {
                    const int cme = g.c0 + lane, cellm = cme / 3;
                    const float *row = stage + g.shift + (cellm - g.cell0) * g.ld + (cme - cellm * 3) * D - g.a0 * D;
                    const float conf = sigmoid_f(row[4]);
                    const float cmul = (MODE == 0 && a.conf) ? 1.0f : conf;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int c = q * CPT + k;
                        float s = -INFINITY;
                        if (c < a.C) s = cmul * sigmoid_f(row[5 + c]);
                        score[k] = s;
                        m = fmaxf(m, s);
                    }
                    if (q == 0) {
                        const int an = cme - cellm * 3;
                        const int col = cellm % g.S, rw = cellm / g.S;
                        const float aw = a.anchor[g.sc][an][0], ah = a.anchor[g.sc][an][1];
                        const float cx = ((sigmoid_f(row[0]) * xs) - half + (float)col) * strd;
                        const float cy = ((sigmoid_f(row[1]) * xs) - half + (float)rw) * strd;
                        const float bw = __expf(row[2]) * aw;
                        const float bh = __expf(row[3]) * ah;
                        if (MODE == 0) {
                            box = make_float4(cx, cy, bw, bh);
                            if (a.conf) a.conf[(size_t)b * a.cap + g.c_global0 + lane] = conf;
                        } else {
                            box = make_float4((cy - bh / 2.0f) / a.image_size, (cx - bw / 2.0f) / a.image_size,
                                              (cy + bh / 2.0f) / a.image_size, (cx + bw / 2.0f) / a.image_size);
                        }
                        // (This file is compiled with SLP vectorisation off, see the Makefile: left on, hipcc
                        // packs the two corner pairs into v_pk_fma_f32 with the stride in a scalar pair read
                        // (low, low) and the half sizes swapped by op_sel, a form that gfx950 gets wrong on
                        // lanes 32-63 whenever another wave's MFMAs occupy the SIMD: tools/isa_lint.py.)
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < CPT; ++k) score[k] = -INFINITY;
                }
            } else {
                if (live) {
                    const size_t n = (size_t)(g.c0 + lane);
                    const float *sp = a.in_scores + (size_t)b * a.sb + n * a.sn;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int c = q * CPT + k;
                        float s = -INFINITY;
                        if (c < a.C) s = sp[(size_t)c * a.sc];
                        score[k] = s;
                        m = fmaxf(m, s);
                    }
                    if (q == 0) {
                        const float4 v = *(const float4 *)(a.in_xywh + ((size_t)b * a.N + n) * 4);
                        box = make_float4((v.y - v.w / 2.0f) / a.image_size, (v.x - v.z / 2.0f) / a.image_size,
                                          (v.y + v.w / 2.0f) / a.image_size, (v.x + v.z / 2.0f) / a.image_size);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < CPT; ++k) score[k] = -INFINITY;
                }
            }
            smax[q * 64 + lane] = m;
            __syncthreads();                               // maxima visible; the rows in LDS are free
            const float mm = fmaxf(fmaxf(smax[lane], smax[64 + lane]), fmaxf(smax[128 + lane], smax[192 + lane]));
            const bool keep = live && (MODE == 0 || mm >= a.thr);
            mask = __ballot(keep);
            cnt = __popcll(mask);
            DSTAMP(2);
            // publish this segment's count at once: every later segment of the image needs it
            if (MODE != 0 && t == 0)
                __hip_atomic_store(&st[seg], (seg > 0 ? YMX_ST_AGG : YMX_ST_PREFIX) | (u64)cnt, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);

            // The workgroup's next segment: wave 1 draws the ticket and (DMA) starts its rows
            // towards LDS, so the transfer runs under the look-back and stores below.
            if (DMA || MODE == 2) {
                if (q == 1) {
                    int tk = (MODE == 0) ? seg + (int)gridDim.y : 0;
                    if (MODE != 0) {
                        if (lane == 0) tk = atomicAdd(&a.ticket[b], 1);
                        tk = __builtin_amdgcn_readfirstlane(tk);
                    }
                    if (lane == 0) sint[0] = tk;
                    if (MODE != 2 && tk < a.nseg) dma_start(geometry(tk));
                }
            } else {
                if (t == 0) sint[0] = (MODE == 0) ? seg + (int)gridDim.y : atomicAdd(&a.ticket[b], 1);
                __syncthreads();
                const int tk = sint[0];
                if (tk < a.nseg) copy_rows(geometry(tk));
            }
        }

        // ---- retire the PREVIOUS segment: its predecessors published their counts an iteration
        // ago, so the look-back reads them without waiting, and its stores leave while the next
        // rows arrive ----
        if (MODE != 0 && p_valid && q == 0) {
            const int lseg = p_seg, lcnt = p_cnt;
                // Wave-wide decoupled look-back: lane l reads the word of segment seg-1-l (then
                // seg-65-l, ...).  A window is usable once its nearest inclusive prefix and every
                // aggregate in front of it are published; the lanes behind that prefix are ignored.
                unsigned run = 0;
                bool failed = false;
                for (int hi = lseg - 1; hi >= 0 && !failed; hi -= 64) {
                    const int p = hi - lane;
                    u64 v = YMX_ST_PREFIX;                      // before the first segment: prefix 0
                    int spin = 0;
                    for (;;) {
                        // The first polls read through the L2 of this XCD only (workgroup scope skips just the
                        // per-CU cache): the workgroups of an image are dispatched to one XCD when the batch
                        // is a multiple of 8 (workgroup id mod 8 = image mod 8; other batches poll
                        // device-coherently from the start), where the publishing stores
                        // have updated that L2 on their way to memory -- a fraction of the round trip of a
                        // device-coherent load.  A word can only look older than it is (not yet published, or
                        // an aggregate where a prefix exists), both of which are safe to act on; whatever is
                        // still missing after sixteen short polls is read device-coherently as before.
                        if (p >= 0)
                            v = spin < l2_polls ? __hip_atomic_load(&st[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                         : __hip_atomic_load(&st[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const u64 ready = __ballot((v >> 62) != 0);
                        const u64 prefix = __ballot((v >> 62) == 2);
                        const u64 pfirst = prefix & (~prefix + 1);            // lowest set bit, 0 if none
                        const u64 need = pfirst ? (pfirst - 1) : ~0ull;
                        if ((ready & need) == need) {
                            const bool use = pfirst ? (((1ull << lane) & (pfirst | (pfirst - 1))) != 0) : true;
                            unsigned part = use ? (unsigned)(v & 0xffffffffull) : 0u;
#pragma unroll
                            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
                            run += part;
                            if (pfirst) hi = -1;                               // reached a prefix: done
                            break;
                        }
                        if (++spin > (1 << 20)) { failed = true; break; }
                        if (spin <= l2_polls) __builtin_amdgcn_s_sleep(4);
                        else __builtin_amdgcn_s_sleep(16);
                    }
                }
                if (lane == 0) {
                    if (failed) atomicExch(a.err, 1);
                    __hip_atomic_store(&st[lseg], YMX_ST_PREFIX | (u64)(run + (unsigned)lcnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sint[1] = (int)run;
                    if (lseg == a.nseg - 1) a.counts[b] = (int)run + lcnt;
                }
        }
        __syncthreads();                                   // base of the previous segment and the next ticket visible
        DSTAMP(3);
        if (p_valid) {
            int base = p_cglobal0;
            if (MODE != 0) base = sint[1];
            if ((p_mask >> lane) & 1ull) {
                const int pos = base + __popcll(p_mask & below);
                if (pos < a.cap) {
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int c = q * CPT + k;
                        if (c < a.C) a.scores[((size_t)b * a.C + c) * a.cap + pos] = p_score[k];
                    }
                    if (q == 0) {
                        *(float4 *)(a.boxes + ((size_t)b * a.cap + pos) * 4) = p_box;
                        if (a.src_idx) a.src_idx[(size_t)b * a.cap + pos] = p_cglobal0 + lane;
                    }
                }
            }
        }
        DSTAMP(4);
        ++dstamp_it;
        if (!have_cur) break;
        p_valid = true; p_seg = seg; p_cnt = cnt; p_mask = mask; p_cglobal0 = g.c_global0; p_box = box;
#pragma unroll
        for (int k = 0; k < CPT; ++k) p_score[k] = score[k];
        seg = sint[0];
        if (seg < a.nseg) g = geometry(seg);
        __syncthreads();                                   // everyone has read the ticket before the next one is written
    }
    if (MODE == 0 && blockIdx.y == 0 && t == 0) a.counts[b] = a.N;
}

template <typename HT, int CPT>
static int launch_mode(const DecodeArgs &a, hipStream_t s) {
    const int D = 5 + a.C;
    const size_t stage_words = (a.filter == 2) ? 0 : (size_t)decode_stage_words(D, a.max_pad);
    const size_t lds = (stage_words + 256 + 4) * 4;
    if (lds > 64 * 1024) return fail(YMX_EUNSUPPORTED, "decode: %d classes need %zu bytes of LDS", a.C, lds);
    // persistent workgroups: as many per image as keep every CU's LDS full (256 CUs, 160 KB each, at most 8 of
    // these workgroups per CU), never more than the image has segments
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    int gy = (256 * per_cu + a.B - 1) / a.B;
    gy = gy > a.nseg ? a.nseg : (gy < 1 ? 1 : gy);
    if (gy > 65535) gy = 65535;
    dim3 grid(a.B, gy), block(256);
    bool dma = sizeof(HT) == 4 && a.filter != 2;
    for (int i = 0; i < a.n_scales; ++i) dma = dma && a.head_bytes[i] != 0;
    if (a.filter == 0) {
        if (dma) hipLaunchKernelGGL((decode_kernel<HT, CPT, 0, sizeof(HT) == 4>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((decode_kernel<HT, CPT, 0, false>), grid, block, lds, s, a);
    } else if (a.filter == 1) {
        if (dma) hipLaunchKernelGGL((decode_kernel<HT, CPT, 1, sizeof(HT) == 4>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((decode_kernel<HT, CPT, 1, false>), grid, block, lds, s, a);
    } else {
        hipLaunchKernelGGL((decode_kernel<HT, CPT, 2, false>), grid, block, lds, s, a);
    }
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

YMX_STAMP_EXPORT(ymx_debug_stamps_decode)
