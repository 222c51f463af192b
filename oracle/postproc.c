/*
 * oracle/postproc.c — mask clean-up, output composition, IoU.
 * TEST INFRASTRUCTURE ONLY (see ggc_oracle.h).
 *
 * Follows reference src/gcn_grabcut/pipeline.py clean_mask :189-227
 * (cv2.connectedComponentsWithStats, connectivity 8 — absent here, PARITY
 * UNPINNED; components are numbered by the raster position of their first
 * pixel, which only matters for area ties), grabcut.py overlay_mask :180-188 and
 * crop_foreground :190-195, metrics.py IoU :79-84.
 */
#include "ggc_oracle.h"
#include <stdlib.h>
#include <string.h>

void ggo_clean_mask(int H, int W, const uint8_t* mask, float min_area_ratio, int keep_largest, uint8_t* out) {
    const size_t P = (size_t)H * W;
    size_t total = 0;
    for (size_t p = 0; p < P; ++p) total += mask[p] != 0;
    if (total == 0 || (min_area_ratio <= 0.0f && !keep_largest)) { memmove(out, mask, P); return; }
    int32_t* lab = (int32_t*)malloc(P * sizeof(int32_t));
    int32_t* stack = (int32_t*)malloc(P * sizeof(int32_t));
    for (size_t p = 0; p < P; ++p) lab[p] = 0;
    int n = 0;
    int64_t* area = (int64_t*)malloc((P / 1 + 2) * sizeof(int64_t));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p0 = (size_t)y * W + x;
            if (!mask[p0] || lab[p0]) continue;
            ++n;
            int sp = 0;
            int64_t a = 0;
            stack[sp++] = (int32_t)p0; lab[p0] = n;
            while (sp) {
                const int p = stack[--sp];
                ++a;
                const int py = p / W, px = p % W;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = py + dy, xx = px + dx;
                        if ((dy == 0 && dx == 0) || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                        const size_t q = (size_t)yy * W + xx;
                        if (mask[q] && !lab[q]) { lab[q] = n; stack[sp++] = (int32_t)q; }
                    }
            }
            area[n] = a;
        }
    int best = 1;
    for (int i = 2; i <= n; ++i) if (area[i] > area[best]) best = i;   /* argmax: first maximum */
    const double min_area = (double)min_area_ratio * (double)P;
    int any = 0;
    if (!keep_largest) for (int i = 1; i <= n; ++i) any |= (double)area[i] >= min_area;
    for (size_t p = 0; p < P; ++p) {
        const int l = lab[p];
        int keep = 0;
        if (l) keep = (keep_largest || !any) ? (l == best) : ((double)area[l] >= min_area);
        out[p] = (uint8_t)keep;
    }
    free(lab); free(stack); free(area);
}

void ggo_compose(int H, int W, const uint8_t* bgr, const uint8_t* binary, float alpha,
                 int tb, int tg, int tr, uint8_t* overlay, uint8_t* rgba) {
    const size_t P = (size_t)H * W;
    const float tint[3] = {(float)tb, (float)tg, (float)tr};
    for (size_t p = 0; p < P; ++p) {
        const float m = binary[p] ? 1.0f : 0.0f;
        for (int c = 0; c < 3; ++c) {
            if (overlay) {
                float v = (float)bgr[3 * p + c] * (1.0f - alpha * m) + (tint[c] * alpha) * m;
                v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
                overlay[3 * p + c] = (uint8_t)v;      /* astype(uint8): truncation */
            }
            if (rgba) rgba[4 * p + c] = bgr[3 * p + c];
        }
        if (rgba) rgba[4 * p + 3] = binary[p] ? 255 : 0;
    }
}

double ggo_iou(int n, const uint8_t* pred, const uint8_t* gt) {
    int64_t tp = 0, fp = 0, fn = 0;
    for (int i = 0; i < n; ++i) {
        const int p = pred[i] != 0, g = gt[i] != 0;
        tp += p & g; fp += p & !g; fn += !p & g;
    }
    return (double)tp / ((double)(tp + fp + fn) + 1e-8);
}

/* Evaluation counts (reference metrics.py:58-129, 152-201): integer tallies behind evaluate / boundary_f1 /
 * evaluate_trimap for one image.  out[14]:
 *   0 tp 1 fp 2 fn                      binary confusion, pred/gt != 0 (:72-75)
 *   3 |pred boundary| 4 |gt boundary| 5 |both|   boundary = m - erode(m, ones(2w+1)^2), cv2.erode's default border
 *                                        (pixels outside the image never erode) (:117-125); zeros when width <= 0
 *   6 fg_tp 7 fg_fp 8 fg_fn 9 bg_tp 10 bg_fp 11 bg_fn 12 probable pixels 13 pixels where (FG|PR_FG) == gt value
 *                                        (:171-194); zeros when trimap == NULL.  Trimap values: 0 BG, 1 FG, 2 PR_BG, 3 PR_FG. */
static int ggo_is_boundary(int h, int w, const uint8_t* m, int y, int x, int width) {
    if (!m[(size_t)y * w + x]) return 0;
    for (int dy = -width; dy <= width; ++dy)
        for (int dx = -width; dx <= width; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            if (!m[(size_t)yy * w + xx]) return 1;
        }
    return 0;
}

void ggo_eval_counts(int h, int w, const uint8_t* pred, const uint8_t* gt, const uint8_t* trimap, int width, int64_t* out) {
    for (int i = 0; i < 14; ++i) out[i] = 0;
    uint8_t* pb = (uint8_t*)malloc((size_t)h * w);
    uint8_t* gb = (uint8_t*)malloc((size_t)h * w);
    for (size_t i = 0; i < (size_t)h * w; ++i) { pb[i] = pred[i] != 0; gb[i] = gt[i] != 0; }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const int p = pb[i], g = gb[i];
            out[0] += p & g; out[1] += p & (!g); out[2] += (!p) & g;
            if (width > 0) {
                const int bp = ggo_is_boundary(h, w, pb, y, x, width), bg = ggo_is_boundary(h, w, gb, y, x, width);
                out[3] += bp; out[4] += bg; out[5] += bp & bg;
            }
            if (trimap) {
                const int t = trimap[i];
                const int pf = t == 1, pbg = t == 0, prob = t == 2 || t == 3;
                out[6] += pf & g; out[7] += pf & (!g); out[8] += (!pf) & g;
                out[9] += pbg & (!g); out[10] += pbg & g; out[11] += (!pbg) & (!g);
                out[12] += prob;
                out[13] += ((t == 1 || t == 3) ? 1 : 0) == gt[i];
            }
        }
    free(pb); free(gb);
}
