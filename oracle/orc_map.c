/*
 * orc_map.c — surfel-map operations (oracle; test infrastructure only).
 * Follows GlobalModel::{initialise,fuse,clean} (Core/src/GlobalModel.cpp:214-288,355-688),
 * IndexMap::predictIndices (Core/src/IndexMap.cpp:193-267) and their shaders.
 *
 * GL rasterisation semantics fixed by this restatement:
 *  - a 1-px GL point lands on the pixel that contains its window coordinate after the viewport transform and the snap to
 *    the 1/256-pixel grid (hd_gl_point_window_coord, include/hrbf_detmath.h — the rule llvmpipe executes and NVIDIA's
 *    GL_SUBPIXEL_BITS = 8 implies); it is clipped when its centre leaves the view volume;
 *  - GL_LESS z-test: smallest camera-space z wins, ties go to the lower surfel index (draw order);
 *    the reference's 24-bit depth quantisation is not modelled;
 *  - the 4596^2 scatter target of fuse stage 1 is "first primitive in draw order wins"
 *    (all fragments z = 0, GL_LESS; GUI/src/Tools/GUI.h:69-71), draw order is column-major over
 *    pixels (GlobalModel.cpp:89-96);
 *  - transform feedback is an order-preserving compaction.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_vec.h"

float orc_get_radius(float depth, float norm_z, float camz, float camw);
float orc_radial_confidence(float x, float y, float cx, float cy, float max_dist, float weighting);
f4 orc_record_normal(const orc_ctx *c, int px, int py, f4 image);

#define SURF(buf, i, k) ((buf)[(size_t)(i) * 5 + (k)])

/* ---- F4: init_unstableTex.vert:51-98 + .geom ---------------------------------------------- */
void orc_initialise(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float cx = c->prm.cx, cy = c->prm.cy, thr = c->prm.curv_valid_threshold;
    const float max_dist = sqrtf(((float)H * 0.5f) * ((float)H * 0.5f) + ((float)W * 0.5f) * ((float)W * 0.5f));
    f4 *out = c->map[c->target];
    uint32_t n = 0;
    for (int px = 0; px < W; ++px)          /* column-major draw order */
        for (int py = 0; py < H; ++py) {
            int i = py * W + px;
            f4 vl = c->vertex_raw[i];
            f3 pg = xform(c->pose, xyz(vl));
            float conf = orc_radial_confidence(hd_px_attribute(px, W), hd_px_attribute(py, H), cx, cy, max_dist, 1.0f);   /* init_unstableTex.vert:53-54: x = texcoord.x * cols of the uv attribute */
            if (c->prm.use_conf_eval > 0) conf = conf * hd_expf(-c->prm.conf_eval_epsilon / sqrtf(c->gradmag[i]));
            f4 nl = c->normal[i];
            f3 ng = rot_mul(c->pose, xyz(nl));
            f4 k1 = c->curv1[i], k2 = c->curv2[i];
            if (len3(ng) > 0.5f && k1.w > -thr && k1.w < thr && k2.w > -thr && k2.w < thr) {
                if (n >= c->cap) continue;
                const uint8_t *rgb = &c->rgb[i * 3];
                SURF(out, n, 0) = v4(pg.x, pg.y, pg.z, conf);
                SURF(out, n, 1) = v4(encode_color_bytes(rgb[0], rgb[1], rgb[2]), 0.0f, 1.0f, 1.0f);
                SURF(out, n, 2) = v4(ng.x, ng.y, ng.z, nl.w);
                SURF(out, n, 3) = k1;
                SURF(out, n, 4) = k2;
                n++;
            }
        }
    c->count = n;
}

/* ---- M1: index_map.vert:34-66, index_map.frag:35-43 --------------------------------------- */
void orc_predict_indices(orc_ctx *c)
{
    const int W = c->W, H = c->H, P = c->P;
    const float fx = c->prm.fx, fy = c->prm.fy, cx = c->prm.cx, cy = c->prm.cy;
    const float maxDepth = c->prm.max_depth_processed;
    float tinv[16];
    rigid_inverse(c->pose, tinv);
    const f4 *m = c->map[c->target];
    float *zbuf = (float *)malloc(sizeof(float) * P);
    int64_t *win = (int64_t *)malloc(sizeof(int64_t) * P);
    for (int i = 0; i < P; ++i) { zbuf[i] = 0.0f; win[i] = -1; }
    for (uint32_t s = 0; s < c->count; ++s) {
        f4 p = SURF(m, s, 0);
        f3 h = xform(tinv, xyz(p));
        /* active-submap mask (index_map.vert:41-45, IndexMap.cpp:222-237): KeyFrameIDMap holds 1 for the ids in
           lActiveKFID and 0 elsewhere.  No mask installed = every submap active (the scoped configs: one submap) */
        if (c->submap_active) {
            const uint32_t sm = hd_cvt_u32(SURF(m, s, 1).y);
            if (sm >= (uint32_t)c->n_submap_active || c->submap_active[sm] == 0) continue;
        }
        if (h.z > maxDepth || h.z < 0.0f) continue;
        float u = ((fx * h.x) / h.z) + cx;
        float v = ((fy * h.y) / h.z) + cy;
        int clip_u, clip_v;
        u = hd_gl_point_window_coord(u, (float)W, &clip_u);
        v = hd_gl_point_window_coord(v, (float)H, &clip_v);
        if (clip_u || clip_v || !(u >= 0.0f && u < (float)W && v >= 0.0f && v < (float)H)) continue;
        int ix = (int)floorf(u), iy = (int)floorf(v);
        int pi = iy * W + ix;
        if (win[pi] < 0 || h.z < zbuf[pi]) { zbuf[pi] = h.z; win[pi] = s; }
    }
    for (int i = 0; i < P; ++i) {
        if (win[i] < 0) {
            c->idx[i] = 0;
            c->im_vertconf[i] = c->im_colortime[i] = c->im_normrad[i] = c->im_curvmax[i] = c->im_curvmin[i] =
                v4(0, 0, 0, 0);
            continue;
        }
        uint32_t s = (uint32_t)win[i];
        f4 p = SURF(m, s, 0), nr = SURF(m, s, 2);
        f3 h = xform(tinv, xyz(p));
        f3 n = normalize3(rot_mul(tinv, xyz(nr)));
        c->idx[i] = s;
        c->im_vertconf[i] = v4(h.x, h.y, h.z, p.w);
        c->im_colortime[i] = SURF(m, s, 1);
        c->im_normrad[i] = v4(n.x, n.y, n.z, nr.w);
        c->im_curvmax[i] = SURF(m, s, 3);
        c->im_curvmin[i] = SURF(m, s, 4);
    }
    free(zbuf); free(win);
}

/* ---- F1: data.vert:63-198 (association) + F2: update.vert:51-115 (merge) ------------------- */
void orc_fuse(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float cx = c->prm.cx, cy = c->prm.cy;
    const float camz = (float)(1.0 / (double)c->prm.fx), camw = (float)(1.0 / (double)c->prm.fy);
    const float maxDepth = c->prm.max_depth_processed;
    const float timef = (float)c->tick;
    const int tpar = c->tick % 2;
    const int QH = H / 2;
    f4 *m = c->map[c->target];
    memset(c->rec_flag, 0, sizeof(int32_t) * c->Q);
    uint32_t merged = 0;
    /* first-primitive-wins table */
    uint32_t *slot = (uint32_t *)malloc(sizeof(uint32_t) * (c->count ? c->count : 1));
    for (uint32_t i = 0; i < c->count; ++i) slot[i] = 0xFFFFFFFFu;

    for (int px = 0; px < W; ++px)
        for (int py = 0; py < H; ++py) {
            if (!(px % 2 == tpar && py % 2 == tpar)) continue;
            int i = py * W + px;
            int q = (px / 2) * QH + (py / 2);
            float x = hd_px_attribute(px, W), y = hd_px_attribute(py, H);   /* data.vert:66-67: texcoord (the uv attribute) * cols, rows */
            float zr = c->depth_metric[i];
            f3 vl = v3((x - cx) * zr * camz, (y - cy) * zr * camw, zr);
            f4 npca = orc_record_normal(c, px, py, c->normal_pca[i]);   /* data.vert:83-96 recomputes it */
            f3 nl = xyz(npca);
            f4 k1 = c->curv1[i], k2 = c->curv2[i];
            if (!(len3(nl) > 0.8f && vl.z > 0.3f && vl.z <= maxDepth && k1.w > -300.0f && k1.w < 300.0f &&
                  k2.w > -300.0f && k2.w < 300.0f))
                continue;
            float bestDist = 1000.0f;
            uint32_t best = 0;
            int counter = 0;
            float xl = (x - cx) * camz, yl = (y - cy) * camw;
            float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);
            f3 ray = v3(xl, yl, 1.0f);
            float lray = len3(ray);
            /* data.vert:108-138, literally: indexXStep = (1 / (cols * scale)) * 0.5, scale = 1, windowMultiplier = 2;
                   for (i = texcoord.x - scale * indexXStep * windowMultiplier; i < texcoord.x + ...; i += indexXStep), same in j —
               half-pixel steps accumulated in fp32 from the uv attribute; every other sample sits ON a texel edge and NEAREST
               reads texel floor(fl(i * cols)) (hd_window_texel).  In exact arithmetic that is {p-1, p, p, p+1} per axis (what
               rounds 1-3 walked); in fp32 the sample at p + 1/2 lands in texel p for about a fifth of the columns of a 640-wide
               image, and unless the accumulated i ends an ulp below the bound (a fifth sample, at p + 1) texel p+1 is never
               visited.  Both rasterisers of the image execute it this way. */
            const float stepx = (1.0f / ((float)W * 1.0f)) * 0.5f, stepy = (1.0f / ((float)H * 1.0f)) * 0.5f;
            const float tcx = hd_uv_attribute(px, W), tcy = hd_uv_attribute(py, H);
            const float ihi = tcx + (1.0f * stepx * 2.0f), jhi = tcy + (1.0f * stepy * 2.0f);
            for (float wi = tcx - (1.0f * stepx * 2.0f); wi < ihi; wi += stepx)
                for (float wj = tcy - (1.0f * stepy * 2.0f); wj < jhi; wj += stepy) {
                    int sx = hd_window_texel(wi, W), sy = hd_window_texel(wj, H);
                    int si = sy * W + sx;
                    uint32_t current = c->idx[si];
                    if (current > 0u) {
                        f4 vcf = c->im_vertconf[si];
                        if (fabsf((vcf.z * lambda) - (vl.z * lambda)) < 0.05f) {
                            float dist = len3(cross3(ray, xyz(vcf))) / lray;
                            f4 nr = c->im_normrad[si];
                            int ok = fabsf(nr.z) < 0.75f;
                            if (!ok) {
                                float ang = hd_acosf(dot3(xyz(nr), nl) / (len3(xyz(nr)) * len3(nl)));
                                ok = fabsf(ang) < 0.5f;
                            }
                            if (dist < bestDist && ok) { counter++; bestDist = dist; best = current; }
                        }
                    }
                }
            f3 pg = xform(c->pose, vl);
            f3 ng = rot_mul(c->pose, nl);
            const uint8_t *rgb = &c->rgb[i * 3];
            f4 *r = &c->rec[(size_t)q * 5];
            r[0] = v4(pg.x, pg.y, pg.z, c->confidence[i]);
            r[1] = v4(encode_color_bytes(rgb[0], rgb[1], rgb[2]), (float)c->index_submap, timef,
                      counter > 0 ? -1.0f : -2.0f);
            r[2] = v4(ng.x, ng.y, ng.z, npca.w);
            r[3] = k1; r[4] = k2;
            c->rec_flag[q] = counter > 0 ? 1 : 2;
            c->rec_best[q] = best;
            if (counter > 0 && slot[best] == 0xFFFFFFFFu) slot[best] = (uint32_t)q;
        }

    /* F2: update.vert — only surfels holding a winning record change */
    for (uint32_t s = 0; s < c->count; ++s) {
        if (slot[s] == 0xFFFFFFFFu) continue;
        const f4 *r = &c->rec[(size_t)slot[s] * 5];
        f4 vp = SURF(m, s, 0), vc = SURF(m, s, 1), vn = SURF(m, s, 2), c1 = SURF(m, s, 3), c2 = SURF(m, s, 4);
        float c_k = vp.w, a = r[0].w, sum = c_k + a;
        if (r[2].w < (1.0f + 0.5f) * vn.w) {
            SURF(m, s, 0) = v4(((c_k * vp.x) + (a * r[0].x)) / sum, ((c_k * vp.y) + (a * r[0].y)) / sum,
                               ((c_k * vp.z) + (a * r[0].z)) / sum, sum);
            f3 oc = decode_color(vc.x), nc = decode_color(r[1].x);
            f3 avg = v3(((c_k * oc.x) + (a * nc.x)) / sum, ((c_k * oc.y) + (a * nc.y)) / sum,
                        ((c_k * oc.z) + (a * nc.z)) / sum);
            SURF(m, s, 1) = v4(encode_color(avg), vc.y, vc.z, (float)c->tick);
            f3 nn = normalize3(v3(((c_k * vn.x) + (a * r[2].x)) / sum, ((c_k * vn.y) + (a * r[2].y)) / sum,
                                  ((c_k * vn.z) + (a * r[2].z)) / sum));
            SURF(m, s, 2) = v4(nn.x, nn.y, nn.z, ((c_k * vn.w) + (a * r[2].w)) / sum);
            SURF(m, s, 3) = v4(((c_k * c1.x) + (a * r[3].x)) / sum, ((c_k * c1.y) + (a * r[3].y)) / sum,
                               ((c_k * c1.z) + (a * r[3].z)) / sum, ((c_k * c1.w) + (a * r[3].w)) / sum);
            SURF(m, s, 4) = v4(((c_k * c2.x) + (a * r[4].x)) / sum, ((c_k * c2.y) + (a * r[4].y)) / sum,
                               ((c_k * c2.z) + (a * r[4].z)) / sum, ((c_k * c2.w) + (a * r[4].w)) / sum);
        } else {
            SURF(m, s, 0) = v4(vp.x, vp.y, vp.z, sum);
            SURF(m, s, 1) = v4(vc.x, vc.y, vc.z, (float)c->tick);
        }
        merged++;
    }
    free(slot);
    c->fuse_stats[1] = merged;
}

/* ---- F3: copy_unstable.vert:62-166 + .geom ------------------------------------------------ */
static int clean_test(const orc_ctx *c, const float *tinv, f4 vp, f4 *vcol, f4 vn, f4 k1, f4 k2)
{
    const int W = c->W, H = c->H;
    const float fx = c->prm.fx, fy = c->prm.fy, cx = c->prm.cx, cy = c->prm.cy;
    const float maxDepth = c->prm.max_depth_processed, confThr = c->prm.confidence_threshold;
    const float thr = c->prm.curv_valid_threshold;
    const int time = c->tick;
    int test = 1;
    f3 lp = xform(tinv, xyz(vp));
    float x = ((fx * lp.x) / lp.z) + cx;
    float y = ((fy * lp.y) / lp.z) + cy;
    f3 ln = normalize3(rot_mul(tinv, xyz(vn)));
    int count = 0, zCount = 0;
    float active = 1.0f;   /* KeyFrameIDMap lookup of the surfel's own submap (copy_unstable.vert:98-101); no mask = all active */
    if (c->submap_active) {
        const uint32_t sm = hd_cvt_u32(vcol->y);
        active = (sm < (uint32_t)c->n_submap_active && c->submap_active[sm]) ? 1.0f : 0.0f;
    }
    if (lp.z < maxDepth && lp.z > 0.0f && x > 0.0f && y > 0.0f && x < (float)W && y < (float)H) {
        /* copy_unstable.vert:85-141: the half-pixel walk, literally (hd_halfpixel_walk: an fp32-accumulated loop that takes
           2 wm samples per axis, or one more where the accumulated coordinate ends an ulp below the bound) */
        const hd_walk wx = hd_halfpixel_walk(x, W, c->prm.clean_window_multiplier), wy = hd_halfpixel_walk(y, H, c->prm.clean_window_multiplier);
        for (float fi = wx.lo; fi < wx.hi; fi += wx.step) {
            int sx = hd_window_texel(fi, W);
            for (float fj = wy.lo; fj < wy.hi; fj += wy.step) {
                int sy = hd_window_texel(fj, H);
                int si = sy * W + sx;
                uint32_t current = c->idx[si];
                if (current > 0u) {
                    f4 vcf = c->im_vertconf[si], ct = c->im_colortime[si];
                    float dx = vcf.x - lp.x, dy = vcf.y - lp.y;
                    if (ct.z < vcol->z && vcf.w > confThr && vcf.z > lp.z && vcf.z - lp.z < 0.01f &&
                        sqrtf(dx * dx + dy * dy) < vn.w * 1.4f)
                        count++;
                    if (ct.w == (float)time && vcf.w > confThr && vcf.z > lp.z && vcf.z - lp.z > 0.01f &&
                        fabsf(ln.z) > 0.85f && active > 0.0f)
                        zCount++;
                }
            }
        }
    }
    if (k1.w < -thr || k1.w > thr || k2.w < -thr || k2.w > thr) test = 0;
    if (count > 8 || zCount > 4) test = 0;
    if (vcol->w == -2.0f) vcol->w = (float)time;
    if (vcol->w == -1.0f || (((float)time - vcol->w) > 200.0f && vp.w < confThr)) test = 0;
    return test;
}

void orc_clean(orc_ctx *c)
{
    float tinv[16];
    rigid_inverse(c->pose, tinv);
    const f4 *m = c->map[c->target];
    f4 *out = c->map[1 - c->target];
    uint32_t n = 0, appended = 0;
    c->fuse_stats[0] = c->count;
    for (uint32_t s = 0; s < c->count; ++s) {
        f4 vp = SURF(m, s, 0), vc = SURF(m, s, 1), vn = SURF(m, s, 2), k1 = SURF(m, s, 3), k2 = SURF(m, s, 4);
        if (clean_test(c, tinv, vp, &vc, vn, k1, k2)) {
            SURF(out, n, 0) = vp; SURF(out, n, 1) = vc; SURF(out, n, 2) = vn; SURF(out, n, 3) = k1; SURF(out, n, 4) = k2;
            n++;
        }
    }
    for (int q = 0; q < c->Q; ++q) {
        if (c->rec_flag[q] == 0) continue;
        const f4 *r = &c->rec[(size_t)q * 5];
        f4 vc = r[1];
        if (clean_test(c, tinv, r[0], &vc, r[2], r[3], r[4])) {
            if (n >= c->cap) continue;
            SURF(out, n, 0) = r[0]; SURF(out, n, 1) = vc; SURF(out, n, 2) = r[2]; SURF(out, n, 3) = r[3]; SURF(out, n, 4) = r[4];
            n++; appended++;
        }
    }
    c->count = n;
    c->target = 1 - c->target;
    c->fuse_stats[2] = appended;
    c->fuse_stats[3] = n;
    /* records are consumed: a second clean without a fuse must not re-append them */
    memset(c->rec_flag, 0, sizeof(int32_t) * c->Q);
}

/* ---- GlobalModel::updateModel (GlobalModel.cpp:690-767) -> update_delta_trans.vert:41-104 ------------------
 * Every surfel is moved by the rigid correction of ITS submap (colour_time.y): position <- T p (w kept),
 * normal <- R n (radius kept); colour/time and both curvature vectors are copied unchanged (the shader does not
 * rotate the principal directions - kept).  delta: n column-major 4x4 matrices, the layout updateModel() uploads
 * (m(k,j), j outer).  A submap id >= n reads texels the reference never uploaded (undefined there): unchanged here. */
void orc_update_model(orc_ctx *c, const float *delta, int n)
{
    f4 *m = c->map[c->target];
    for (uint32_t s = 0; s < c->count; ++s) {
        const uint32_t sm = hd_cvt_u32(SURF(m, s, 1).y);
        if (sm >= (uint32_t)n) continue;
        const float *T = delta + (size_t)sm * 16;   /* T[col*4 + row] */
        const f4 p = SURF(m, s, 0), nr = SURF(m, s, 2);
        f4 po, no;
        po.x = ((T[0] * p.x + T[4] * p.y) + T[8] * p.z) + T[12] * 1.0f;
        po.y = ((T[1] * p.x + T[5] * p.y) + T[9] * p.z) + T[13] * 1.0f;
        po.z = ((T[2] * p.x + T[6] * p.y) + T[10] * p.z) + T[14] * 1.0f;
        po.w = p.w;
        no.x = (T[0] * nr.x + T[4] * nr.y) + T[8] * nr.z;
        no.y = (T[1] * nr.x + T[5] * nr.y) + T[9] * nr.z;
        no.z = (T[2] * nr.x + T[6] * nr.y) + T[10] * nr.z;
        no.w = nr.w;
        SURF(m, s, 0) = po; SURF(m, s, 2) = no;
    }
}
