/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (practical-collab-perception_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-C restatement of the reference's rotated BEV IoU and greedy bit-mask NMS:
 *   - box_overlap / iou_bev        : /root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:128-230
 *                                    (same arithmetic as iou3d_nms_kernel.cu:104-234)
 *   - greedy suppression           : /root/reference/pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:103-135
 *                                    (row i suppresses every later j with iou(i,j) > thresh)
 *   - score ordering / post-max    : /root/reference/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:84-99 and
 *                                    /root/reference/pcdet/models/model_utils/model_nms_utils.py:6-25
 * Pinned against oracle/_ref (the reference's own iou3d_cpu.cpp compiled where it lies) by
 * tests/test_oracle_pins.py and against tests/golden/g3_nms.npz.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC nms_oracle.c -o _build/liboracle_nms.so -lm
 * All arithmetic is float32 with one rounding per operation (no FMA contraction), like the x86 reference build.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS 1e-8f
#define ORC_MARGIN 1e-2f

typedef struct { float x, y; } vec2;

static float cross3(vec2 a, vec2 b, vec2 o) {          /* (a-o) x (b-o) */
    return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

static float fmin2(float a, float b) { return a > b ? b : a; }
static float fmax2(float a, float b) { return a > b ? a : b; }

static int bounding_rects_touch(vec2 p1, vec2 p2, vec2 q1, vec2 q2) {
    return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
           fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

/* segment (p0,p1) against segment (q0,q1); strict straddle test on both */
static int seg_intersection(vec2 p1, vec2 p0, vec2 q1, vec2 q0, vec2 *out) {
    if (!bounding_rects_touch(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > ORC_EPS) {
        out->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        out->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        out->x = (b0 * c1 - b1 * c0) / D;
        out->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static int corner_inside(const float *box, vec2 p) {
    float cx = box[0], cy = box[1];
    float c = cosf(-box[6]), s = sinf(-box[6]);
    float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    float ry = (p.x - cx) * s + (p.y - cy) * c;
    return fabsf(rx) < box[3] / 2 + ORC_MARGIN && fabsf(ry) < box[4] / 2 + ORC_MARGIN;
}

static void box_corners(const float *box, vec2 *c /* [5] */) {
    float hx = box[3] / 2, hy = box[4] / 2;
    float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    float ca = cosf(box[6]), sa = sinf(box[6]);
    vec2 raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; k++) {
        float dx = raw[k].x - box[0], dy = raw[k].y - box[1];
        c[k].x = dx * ca + dy * (-sa) + box[0];
        c[k].y = dx * sa + dy * ca + box[1];
    }
    c[4] = c[0];
}

float orc_box_overlap(const float *a, const float *b) {
    vec2 ca[5], cb[5], poly[16], ctr = {0.f, 0.f};
    int cnt = 0;
    box_corners(a, ca);
    box_corners(b, cb);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &poly[cnt])) {
                ctr.x = ctr.x + poly[cnt].x;
                ctr.y = ctr.y + poly[cnt].y;
                cnt++;
            }
    for (int k = 0; k < 4; k++) {
        if (corner_inside(a, cb[k])) { ctr.x = ctr.x + cb[k].x; ctr.y = ctr.y + cb[k].y; poly[cnt++] = cb[k]; }
        if (corner_inside(b, ca[k])) { ctr.x = ctr.x + ca[k].x; ctr.y = ctr.y + ca[k].y; poly[cnt++] = ca[k]; }
    }
    ctr.x /= cnt;
    ctr.y /= cnt;
    /* bubble sort by polar angle about the mean vertex (ascending) */
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++) {
            float ai = atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x);
            float an = atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x);
            if (ai > an) { vec2 t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t; }
        }
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        float ux = poly[k].x - poly[0].x, uy = poly[k].y - poly[0].y;
        float vx = poly[k + 1].x - poly[0].x, vy = poly[k + 1].y - poly[0].y;
        area += ux * vy - uy * vx;
    }
    return (float)(fabsf(area) / 2.0);
}

float orc_iou_bev(const float *a, const float *b) {
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float ov = orc_box_overlap(a, b);
    return ov / fmaxf(sa + sb - ov, ORC_EPS);
}

void orc_iou_matrix(const float *a, int na, const float *b, int nb, float *out) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) out[(size_t)i * nb + j] = orc_iou_bev(a + 7 * i, b + 7 * j);
}

void orc_overlap_matrix(const float *a, int na, const float *b, int nb, float *out) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) out[(size_t)i * nb + j] = orc_box_overlap(a + 7 * i, b + 7 * j);
}

/* boxes already in descending-score order. keep[] receives indices into that order; returns the count. */
int orc_nms_sorted(const float *boxes, int n, float thresh, int64_t *keep) {
    uint8_t *dead = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int i = 0; i < n; i++) {
        if (dead[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; j++)
            if (!dead[j] && orc_iou_bev(boxes + 7 * i, boxes + 7 * j) > thresh) dead[j] = 1;
    }
    free(dead);
    return nk;
}

/* same, but from a precomputed (n x n) IoU matrix (lets tests exclude near-threshold pairs) */
int orc_nms_from_iou(const float *iou, int n, float thresh, int64_t *keep) {
    uint8_t *dead = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int i = 0; i < n; i++) {
        if (dead[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; j++)
            if (iou[(size_t)i * n + j] > thresh) dead[j] = 1;
    }
    free(dead);
    return nk;
}
