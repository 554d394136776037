/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of class-aware greedy NMS.
 *
 * Contract (call site yolort/models/box_head.py:422 ->
 * torchvision.ops.batched_nms, third-party, not vendored => PARITY UNPINNED):
 *   - candidates arrive in (anchor asc, class asc) order (box_head.py:418 torch.where row-major);
 *   - order = stable sort by score descending (ties keep candidate order);
 *   - box i suppresses a later box j of the SAME label iff
 *       inter / (area_i + area_j - inter) > thr   (strict, fp32 arithmetic),
 *       inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1)), area = (x2-x1)*(y2-y1);
 *   - output = kept candidate indices in sorted order.
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -shared -fPIC nms_ref.c -o libnms_ref.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int64_t i; } key_t_;

static int cmp_desc(const void* a, const void* b) {
    const key_t_* x = (const key_t_*)a; const key_t_* y = (const key_t_*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);   /* stable: original order on ties */
}

int ymi_ref_batched_nms(const float* boxes, const float* scores, const int64_t* labels, int n,
                        float thr, int64_t* keep_out) {
    if (n <= 0) return 0;
    key_t_* keys = (key_t_*)malloc(sizeof(key_t_) * (size_t)n);
    unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        keys[i].s = scores[i]; keys[i].i = i;
        const float* b = boxes + 4 * (size_t)i;
        area[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    qsort(keys, (size_t)n, sizeof(key_t_), cmp_desc);
    int nk = 0;
    for (int a = 0; a < n; ++a) {
        int64_t i = keys[a].i;
        if (sup[i]) continue;
        keep_out[nk++] = i;
        const float* bi = boxes + 4 * (size_t)i;
        for (int c = a + 1; c < n; ++c) {
            int64_t j = keys[c].i;
            if (sup[j] || labels[j] != labels[i]) continue;
            const float* bj = boxes + 4 * (size_t)j;
            float xx1 = bi[0] > bj[0] ? bi[0] : bj[0];
            float yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
            float xx2 = bi[2] < bj[2] ? bi[2] : bj[2];
            float yy2 = bi[3] < bj[3] ? bi[3] : bj[3];
            float w = xx2 - xx1; if (w < 0.f) w = 0.f;
            float h = yy2 - yy1; if (h < 0.f) h = 0.f;
            float inter = w * h;
            float iou = inter / (area[i] + area[j] - inter);
            if (iou > thr) sup[j] = 1;
        }
    }
    free(keys); free(sup); free(area);
    return nk;
}
