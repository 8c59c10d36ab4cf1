/*
 * orc_predict.c — HRBF ray-cast prediction and fill-in (oracle; test infrastructure only).
 * Follows IndexMap::predictHRBF (Core/src/IndexMap.cpp:413-518) -> predict_hrbf.frag:40-311 and
 * FillIn::{vertex,normal,curvature,image} (Core/src/Shaders/FillIn.cpp:93-297) -> fill_*.frag.
 */
#include <string.h>
#include "oracle.h"
#include "orc_vec.h"

/* predict_hrbf.frag:75-113 — neighbour gathering, ring by ring, with the reference's
   "break only the innermost loop" quirk.  Returns the count; slot order = visiting order. */
static int gather(const orc_ctx *c, int px, int py, f4 *vc, f4 *nr, f4 *ct, f4 *cmax, f4 *cmin)
{
    const int W = c->W, H = c->H;
    const int win = (int)c->prm.predict_window_multiplier;
    const int maxn = c->prm.predict_max_neighbors;
    const float cthr = c->prm.predict_conf_threshold;
    int n = 0;
    for (int i = 0; i <= win; ++i)
        for (int dj = -i; dj <= i; ++dj)
            for (int dk = -i; dk <= i; ++dk) {
                if (!(dj == -i || dk == -i || dj == i || dk == i)) continue;
                int sx = px + dj, sy = py + dk;
                if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                int si = sy * W + sx;
                f4 p = c->im_vertconf[si], nn = c->im_normrad[si];
                if (p.z < 0.1f || len3(xyz(nn)) < 0.1f || p.w < cthr || nn.z < 0.0f) continue;
                vc[n] = p; nr[n] = nn; ct[n] = c->im_colortime[si]; cmax[n] = c->im_curvmax[si];
                cmin[n] = c->im_curvmin[si];
                n++;
                if (n > maxn) break;
            }
    return n;
}

void orc_predict_hrbf(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float cx = c->prm.cx, cy = c->prm.cy;
    const float camz = (float)(1.0 / (double)c->prm.fx), camw = (float)(1.0 / (double)c->prm.fy);
    const int minn = c->prm.predict_min_neighbors;
    const float lambda = c->prm.icp_curv_weight_lambda;
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int pi = py * W + px;
            float x = hd_px_fragment(px, W), y = hd_px_fragment(py, H);   /* predict_hrbf.frag:42-43: texcoord * cols, rows */
            float xl = (x - cx) * camz, yl = (y - cy) * camw;
            f3 ray = normalize3(v3(xl, yl, 1.0f));
            f4 vc[100], nr[100], ct[100], cmax[100], cmin[100];
            int n = gather(c, px, py, vc, nr, ct, cmax, cmin);

            uint8_t img[4] = {0, 0, 0, 0};
            f3 p_surface = v3(0, 0, 0), p_normal = v3(0, 0, 0);
            f4 cmx = v4(0, 0, 0, 1000.0f), cmn = v4(0, 0, 0, 1000.0f);
            float icpw = 0.0f, confidence = 0.0f, radius = 0.0f;
            uint32_t tm = 0;   /* `time` is not written when no surface is found (predict_hrbf.frag:296): cleared value */

            f3 closest = v3(0, 0, 0);
            float pmin = 1000000.0f;
            for (int i = 0; i < n; ++i) {
                float pj = fabsf(dot3(xyz(vc[i]), ray));
                if (pj < pmin) { closest = scale3(ray, pj); pmin = pj; }
            }
            int find_interval = 0, found = 0;
            f3 sp = v3(0, 0, 0), ep = v3(0, 0, 0), p_temp = v3(0, 0, 0), ntemp = v3(0, 0, 0);
            int nsup = 0;
            if (n > minn) {
                float pc[3] = {closest.x, closest.y, closest.z};
                float v0 = orc_hrbf_value(pc, vc, nr, n, &nsup);
                if (nsup > minn) {
                    if (v0 > 0.0f) {
                        ep = closest;
                        int sfound = 0;
                        for (int i = 0; i < 25; ++i) {
                            f3 p1 = sub3(ep, scale3(ray, 0.004f * (float)i));
                            float q[3] = {p1.x, p1.y, p1.z};
                            float v1 = orc_hrbf_value(q, vc, nr, n, &nsup);
                            if (v1 < 0.0f) { sp = p1; sfound = 1; break; }
                        }
                        if (sfound)
                            for (int i = 1; i < 11; ++i) {
                                f3 p2 = add3(sp, scale3(ray, 0.0004f * (float)i));
                                float q[3] = {p2.x, p2.y, p2.z};
                                float v2 = orc_hrbf_value(q, vc, nr, n, &nsup);
                                if (v2 > 0.0f) { ep = p2; find_interval = 1; break; }
                            }
                    } else {
                        sp = closest;
                        int efound = 0;
                        for (int i = 0; i < 25; ++i) {
                            f3 p1 = add3(sp, scale3(ray, 0.004f * (float)i));
                            float q[3] = {p1.x, p1.y, p1.z};
                            float v1 = orc_hrbf_value(q, vc, nr, n, &nsup);
                            if (v1 > 0.0f) { ep = p1; efound = 1; break; }
                        }
                        if (efound)
                            for (int i = 1; i < 11; ++i) {
                                f3 p2 = sub3(ep, scale3(ray, 0.0004f * (float)i));
                                float q[3] = {p2.x, p2.y, p2.z};
                                float v2 = orc_hrbf_value(q, vc, nr, n, &nsup);
                                if (v2 < 0.0f) { sp = p2; find_interval = 1; break; }
                            }
                    }
                }
            }
            if (find_interval) {
                for (int j = 0; j < 10; ++j) {
                    f3 step = sub3(ep, sp);
                    if (len3(step) < 0.00001f) {
                        float q[3] = {p_temp.x, p_temp.y, p_temp.z}, g[3];
                        orc_hrbf_gradient(q, vc, nr, n, g);
                        ntemp = v3(g[0], g[1], g[2]);
                        found = 1;
                        break;
                    }
                    p_temp = add3(sp, scale3(step, 0.5f));
                    float q[3] = {p_temp.x, p_temp.y, p_temp.z};
                    float f_temp = orc_hrbf_value(q, vc, nr, n, &nsup);
                    if (fabsf(f_temp) < 0.00001f) {
                        float g[3];
                        orc_hrbf_gradient(q, vc, nr, n, g);
                        ntemp = v3(g[0], g[1], g[2]);
                        found = 1;
                        break;
                    }
                    if (f_temp < 0.0f) sp = p_temp; else ep = p_temp;
                }
            }
            if (found) {
                p_surface = p_temp;
                p_normal = normalize3(ntemp);
                float dsm = 1000000.0f;
                for (int it = 0; it < n; ++it) {
                    float dx = p_surface.x - vc[it].x, dy = p_surface.y - vc[it].y, dz = p_surface.z - vc[it].z;
                    float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
                    if (dist < dsm) {
                        cmx = cmax[it]; cmn = cmin[it];
                        confidence = vc[it].w; radius = nr[it].w;
                        int ci = hd_cvt_i32(ct[it].x);
                        img[0] = (uint8_t)((ci >> 16) & 0xFF); img[1] = (uint8_t)((ci >> 8) & 0xFF);
                        img[2] = (uint8_t)(ci & 0xFF); img[3] = 255;
                        tm = hd_cvt_u32(ct[it].z);
                        dsm = dist;
                    }
                }
                float a1 = fabsf(cmx.w), a2 = fabsf(cmn.w);
                float cm = a1 > a2 ? a1 : a2;
                icpw = (1.0f / (p_surface.z * p_surface.z)) *
                       (confidence / 256.0f + hd_expf(-0.5f * (lambda * lambda) / (cm * cm)));
            }
            memcpy(&c->pr_image[pi * 4], img, 4);
            c->pr_vertex[pi] = v4(p_surface.x, p_surface.y, p_surface.z, confidence);
            c->pr_normal[pi] = v4(p_normal.x, p_normal.y, p_normal.z, radius);
            c->pr_curv1[pi] = cmx; c->pr_curv2[pi] = cmn;
            c->pr_time[pi] = tm;
            c->pr_icpw[pi] = icpw;
        }
}

/* fill_vertex.frag:43-72, fill_normal.frag:36-49, fill_curvature.frag:35-51, fill_rgb.frag:29-37;
   passthrough = lost = false always (HRBFFusion.cpp:35). */
void orc_fillin(orc_ctx *c)
{
    const float thr = c->prm.curv_valid_threshold, lambda = c->prm.icp_curv_weight_lambda;
    for (int i = 0; i < c->P; ++i) {
        f4 s = c->pr_vertex[i];
        if (s.z == 0.0f) {
            f4 fv = c->vertex_filtered[i], r1 = c->curv1[i], r2 = c->curv2[i];
            f4 outv = v4(0, 0, 0, 0);
            float outw = 0.0f;
            if (r1.w > -thr && r1.w < thr && r2.w > -thr && r2.w < thr) {
                float vConf = c->confidence[i];
                float a1 = fabsf(r1.w), a2 = fabsf(r2.w);
                float cm = a1 > a2 ? a1 : a2;
                outw = (1.0f / (fv.z * fv.z)) * (vConf / 256.0f + hd_expf(-0.5f * (lambda * lambda) / (cm * cm)));
                outv = v4(fv.x, fv.y, fv.z, vConf);
            }
            c->fi_vertex[i] = outv; c->fi_icpw[i] = outw;
        } else { c->fi_vertex[i] = s; c->fi_icpw[i] = c->pr_icpw[i]; }

        f4 n = c->pr_normal[i];
        c->fi_normal[i] = (len3(xyz(n)) < 0.8f) ? c->normal[i] : n;

        f4 k1 = c->pr_curv1[i], k2 = c->pr_curv2[i];
        if (k1.w > 300.0f || k2.w > 300.0f) { c->fi_curv1[i] = c->curv1[i]; c->fi_curv2[i] = c->curv2[i]; }
        else { c->fi_curv1[i] = k1; c->fi_curv2[i] = k2; }

        const uint8_t *e = &c->pr_image[i * 4];
        if ((int)e[0] + (int)e[1] + (int)e[2] == 0 || c->prm.frame_to_frame_rgb) {
            c->fi_image[i * 4 + 0] = c->rgb[i * 3 + 0]; c->fi_image[i * 4 + 1] = c->rgb[i * 3 + 1];
            c->fi_image[i * 4 + 2] = c->rgb[i * 3 + 2]; c->fi_image[i * 4 + 3] = 255;
        } else memcpy(&c->fi_image[i * 4], e, 4);
    }
}

/* Resize::vertex (Core/src/Shaders/Resize.cpp:106, resize.frag) + HRBFFusion::denseEnough
   (HRBFFusion.cpp:974-988): nearest sample at the centre of each 20x20 cell. */
int orc_dense_enough(const orc_ctx *c)
{
    const int cs = 20;
    int w = c->W / cs, h = c->H / cs, sum = 0;
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            int sx = (int)floorf(((float)i + 0.5f) * (float)c->W / (float)w);
            int sy = (int)floorf(((float)j + 0.5f) * (float)c->H / (float)h);
            sum += c->pr_vertex[sy * c->W + sx].z > 0.0f;
        }
    float per = (float)sum / (float)(w * h);
    return per > c->prm.dense_enough_thresh;
}
