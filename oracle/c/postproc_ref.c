/* Plain-C restatement of the reference's head post-processing.
 *
 * TEST INFRASTRUCTURE ONLY: linked or loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, never by the product path.  Parity unpinned:
 * the reference (TensorFlow) cannot run here and ships no golden vectors; this
 * file is a second restatement, written independently of oracle/postproc_np.py
 * from the text of /root/reference/utils.py, and the two are checked against
 * each other.
 *
 *   yo_decode_scale   utils.py:107-126 (decode) + 133-140 (decode_predict)
 *   yo_filter         utils.py:142-167 (filter_bboxes), per image
 *   yo_combined_nms   utils.py:184-192 (call parameters); inner semantics are
 *                     fixed by this build: strict '>' score test, descending
 *                     score with ties by ascending index, suppress on IoU >
 *                     threshold, merge by (score desc, class asc, index asc)
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no fast-math): every float
 * operation rounds to binary32 exactly once.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* conv [B,S,S,3,5+C] -> xywh [B,3S^2,4] (pixels), score [B,3S^2,C] = conf*prob.
 * Candidate order is (row, col, anchor); grid value is (x=col, y=row). */
void yo_decode_scale(const float *conv, int B, int S, int C, float stride, float xyscale,
                     const float *anchors3x2, float *xywh, float *score)
{
    const int D = 5 + C;
    const float half = 0.5f * (xyscale - 1.0f);
    for (int b = 0; b < B; ++b)
        for (int row = 0; row < S; ++row)
            for (int col = 0; col < S; ++col)
                for (int a = 0; a < 3; ++a) {
                    size_t n = (((size_t)b * S + row) * S + col) * 3 + a;
                    const float *t = conv + n * D;
                    float *o = xywh + n * 4;
                    o[0] = ((sigmoidf_(t[0]) * xyscale) - half + (float)col) * stride;
                    o[1] = ((sigmoidf_(t[1]) * xyscale) - half + (float)row) * stride;
                    o[2] = expf(t[2]) * anchors3x2[a * 2 + 0];
                    o[3] = expf(t[3]) * anchors3x2[a * 2 + 1];
                    float conf = sigmoidf_(t[4]);
                    float *s = score + n * C;
                    for (int c = 0; c < C; ++c) s[c] = conf * sigmoidf_(t[5 + c]);
                }
}

/* xywh [B,N,4], score [B,N,C] -> boxes [B,cap,4] yxyx/size, scores_out [B,cap,C],
 * counts [B], src_idx [B,cap] (-1 padding).  Returns the largest count, or -1 if
 * some image has more than cap survivors. */
int yo_filter(const float *xywh, const float *score, int B, int N, int C, float thr,
              float image_size, float *boxes, float *scores_out, int *counts, int *src_idx,
              int cap)
{
    int mmax = 0;
    for (int b = 0; b < B; ++b) {
        int m = 0;
        for (int i = 0; i < cap; ++i) src_idx[(size_t)b * cap + i] = -1;
        memset(boxes + (size_t)b * cap * 4, 0, sizeof(float) * cap * 4);
        memset(scores_out + (size_t)b * cap * C, 0, sizeof(float) * (size_t)cap * C);
        for (int n = 0; n < N; ++n) {
            const float *s = score + ((size_t)b * N + n) * C;
            if (C <= 0) break;
            float mx = s[0];
            for (int c = 1; c < C; ++c) mx = s[c] > mx ? s[c] : mx;
            if (!(mx >= thr)) continue;
            if (m >= cap) return -1;
            const float *x = xywh + ((size_t)b * N + n) * 4;
            float *o = boxes + ((size_t)b * cap + m) * 4;
            o[0] = (x[1] - (x[3] / 2.0f)) / image_size;
            o[1] = (x[0] - (x[2] / 2.0f)) / image_size;
            o[2] = (x[1] + (x[3] / 2.0f)) / image_size;
            o[3] = (x[0] + (x[2] / 2.0f)) / image_size;
            memcpy(scores_out + ((size_t)b * cap + m) * C, s, sizeof(float) * C);
            src_idx[(size_t)b * cap + m] = n;
            ++m;
        }
        counts[b] = m;
        if (m > mmax) mmax = m;
    }
    return mmax;
}

static float iou_(const float *a, const float *b)
{
    float y1a = fminf(a[0], a[2]), x1a = fminf(a[1], a[3]);
    float y2a = fmaxf(a[0], a[2]), x2a = fmaxf(a[1], a[3]);
    float y1b = fminf(b[0], b[2]), x1b = fminf(b[1], b[3]);
    float y2b = fmaxf(b[0], b[2]), x2b = fmaxf(b[1], b[3]);
    float area_a = (y2a - y1a) * (x2a - x1a);
    float area_b = (y2b - y1b) * (x2b - x1b);
    if (!(area_a > 0.0f) || !(area_b > 0.0f)) return 0.0f;
    float iy1 = fmaxf(y1a, y1b), ix1 = fmaxf(x1a, x1b);
    float iy2 = fminf(y2a, y2b), ix2 = fminf(x2a, x2b);
    float inter = fmaxf(iy2 - iy1, 0.0f) * fmaxf(ix2 - ix1, 0.0f);
    float uni = (area_a + area_b) - inter;
    return inter / uni;
}

typedef struct { float s; int i; } cand_t;

/* heap order: the root is the candidate to visit next
 * (higher score first, then lower index) */
static int before_(cand_t a, cand_t b) { return a.s > b.s || (a.s == b.s && a.i < b.i); }

static void sift_(cand_t *h, int n, int k)
{
    for (;;) {
        int l = 2 * k + 1, r = l + 1, m = k;
        if (l < n && before_(h[l], h[m])) m = l;
        if (r < n && before_(h[r], h[m])) m = r;
        if (m == k) return;
        cand_t t = h[k]; h[k] = h[m]; h[m] = t;
        k = m;
    }
}

typedef struct { float s; int c; int i; } kept_t;

static int kept_cmp_(const void *pa, const void *pb)
{
    const kept_t *a = (const kept_t *)pa, *b = (const kept_t *)pb;
    if (a->s != b->s) return a->s > b->s ? -1 : 1;
    if (a->c != b->c) return a->c < b->c ? -1 : 1;
    return a->i < b->i ? -1 : (a->i > b->i ? 1 : 0);
}

/* boxes [B,M,4] yxyx, scores [B,M,C], counts [B] or NULL, src_idx [B,M] or NULL.
 * Outputs: out_boxes [B,T,4], out_scores [B,T], out_classes [B,T], out_valid [B],
 * out_keep [B,T] (-1 padding). */
void yo_combined_nms(const float *boxes, const float *scores, const int *counts, int B, int M,
                     int C, int per_class, int total, float iou_thr, float score_thr,
                     const int *src_idx, float *out_boxes, float *out_scores,
                     float *out_classes, int *out_valid, int *out_keep)
{
    cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)(M > 0 ? M : 1));
    int *kept = (int *)malloc(sizeof(int) * (size_t)(per_class > 0 ? per_class : 1));
    kept_t *merged = (kept_t *)malloc(sizeof(kept_t) * (size_t)(C > 0 ? C : 1) *
                                      (size_t)(per_class > 0 ? per_class : 1));
    for (int b = 0; b < B; ++b) {
        const int m = counts ? counts[b] : M;
        const float *bx = boxes + (size_t)b * M * 4;
        int nm = 0;
        for (int c = 0; c < C; ++c) {
            int n = 0;
            for (int i = 0; i < m; ++i) {
                float s = scores[((size_t)b * M + i) * C + c];
                if (s > score_thr) { heap[n].s = s; heap[n].i = i; ++n; }
            }
            for (int k = n / 2 - 1; k >= 0; --k) sift_(heap, n, k);
            int nk = 0;
            while (n > 0 && nk < per_class) {
                cand_t top = heap[0];
                heap[0] = heap[n - 1]; --n; sift_(heap, n, 0);
                int sup = 0;
                for (int k = 0; k < nk && !sup; ++k)
                    if (iou_(bx + (size_t)top.i * 4, bx + (size_t)kept[k] * 4) > iou_thr) sup = 1;
                if (sup) continue;
                kept[nk++] = top.i;
                merged[nm].s = top.s; merged[nm].c = c; merged[nm].i = top.i; ++nm;
            }
        }
        qsort(merged, (size_t)nm, sizeof(kept_t), kept_cmp_);
        int v = nm < total ? nm : total;
        out_valid[b] = v;
        for (int j = 0; j < total; ++j) {
            float *ob = out_boxes + ((size_t)b * total + j) * 4;
            if (j < v) {
                const float *src = bx + (size_t)merged[j].i * 4;
                for (int k = 0; k < 4; ++k) ob[k] = fminf(fmaxf(src[k], 0.0f), 1.0f);
                out_scores[(size_t)b * total + j] = merged[j].s;
                out_classes[(size_t)b * total + j] = (float)merged[j].c;
                out_keep[(size_t)b * total + j] =
                    src_idx ? src_idx[(size_t)b * M + merged[j].i] : merged[j].i;
            } else {
                ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
                out_scores[(size_t)b * total + j] = 0.0f;
                out_classes[(size_t)b * total + j] = 0.0f;
                out_keep[(size_t)b * total + j] = -1;
            }
        }
    }
    free(heap); free(kept); free(merged);
}
