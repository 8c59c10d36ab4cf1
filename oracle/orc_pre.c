/*
 * orc_pre.c — per-frame pre-processing passes (oracle; test infrastructure only).
 * Follows HRBFFusion::filterDepth/metriciseDepth/computeVertexNormalRadius/
 * computeCurvatureGradient/updateNormalRad/VertexConfidence (Core/src/HRBFFusion.cpp:1263-1345)
 * and the fragment shaders they run.
 *
 * Canonical conventions fixed by this restatement (the reference leaves them to the GL driver):
 *  - a fragment at pixel (px,py) sees x = px + 0.5, y = py + 0.5 exactly; int(x) = px;
 *  - texture fetches are nearest-texel, texel = floor(coord * size) under exact arithmetic,
 *    clamp-to-edge outside [0,1].
 */
#include <stdlib.h>
#include "oracle.h"
#include "orc_vec.h"

/* ---- P1: depth_bilateral.frag:16-67 / depth_guass.frag:21-74 ---------------------------- */
void orc_filter_depth(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float depthFactor = c->prm.depth_scale;
    const float maxD = c->prm.depth_cutoff;
    const float adj = 1.0f / (depthFactor * 1000.0f);
    const int bilateral = c->prm.use_bilateral;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float value = (float)c->depth_raw[y * W + x] / adj;
            float out;
            if (value > maxD * 1000.0f || value < 300.0f) out = 0.0f;
            else if (bilateral) {
                const float ss = 0.024691358f, sc = 0.000555556f;
                const int D = 13;
                int tx = x - D / 2 + D; if (tx > W) tx = W;
                int ty = y - D / 2 + D; if (ty > H) ty = H;
                float sum1 = 0.0f, sum2 = 0.0f;
                for (int cy = (y - D / 2 > 0 ? y - D / 2 : 0); cy < ty; ++cy)
                    for (int cx = (x - D / 2 > 0 ? x - D / 2 : 0); cx < tx; ++cx) {
                        /* depth_bilateral.frag:51-54: texture(gSampler, vec2(float(cx) / cols, float(cy) / rows)), NEAREST — a
                           coordinate exactly ON the texel's edge; the texel is floor(fl(fl(c / n) * n)) (hd_tap_texel) */
                        float tmp = (float)c->depth_raw[hd_tap_texel(cy, H) * W + hd_tap_texel(cx, W)] / adj;
                        float dx = (float)x - (float)cx, dy = (float)y - (float)cy;
                        float space2 = dx * dx + dy * dy;
                        float dv = value - tmp;
                        float color2 = dv * dv;
                        float weight = hd_expf(-(space2 * ss + color2 * sc));
                        sum1 += tmp * weight;
                        sum2 += weight;
                    }
                out = (sum1 / sum2) * adj;
            } else {
                const int D = 9;
                int tx = x - D / 2 + D; if (tx > W) tx = W;
                int ty = y - D / 2 + D; if (ty > H) ty = H;
                float sum1 = 0.0f, sum2 = 0.0f;
                for (int cy = (y - D / 2 > 0 ? y - D / 2 : 0); cy < ty; ++cy)
                    for (int cx = (x - D / 2 > 0 ? x - D / 2 : 0); cx < tx; ++cx) {
                        float tmp = (float)c->depth_raw[hd_tap_texel(cy, H) * W + hd_tap_texel(cx, W)] / adj;   /* depth_guass.frag:54-58 */
                        if (tmp > 300.0f && fabsf(tmp - value) < 100.0f) {
                            float dx = (float)x - (float)cx, dy = (float)y - (float)cy;
                            float weight = hd_expf(-((dx * dx + dy * dy) / (2.0f * 3.0f * 3.0f)));
                            sum1 += tmp * weight;
                            sum2 += weight;
                        }
                    }
                out = (sum1 / sum2) * adj;
            }
            c->depth_filtered[y * W + x] = out;
        }
    }
}

/* ---- P2: depth_metric_raw.frag:29-42, depth_metric_filtered.frag:29-41 -------------------- */
void orc_metricise(orc_ctx *c)
{
    const int P = c->P;
    const float depthFactor = c->prm.depth_scale, maxD = c->prm.depth_cutoff;
    const uint32_t hi = (uint32_t)(maxD / depthFactor), lo = (uint32_t)(0.3f / depthFactor);
    const float fhi = maxD / depthFactor, flo = 0.3f / depthFactor;
    for (int i = 0; i < P; ++i) {
        uint32_t v = c->depth_raw[i];
        c->depth_metric[i] = (v > hi || v < lo) ? 0.0f : (float)v * depthFactor;
        float f = c->depth_filtered[i];
        c->depth_metric_filtered[i] = (f > fhi || f < flo) ? 0.0f : f * depthFactor;
    }
}

/* ---- helpers: surfels.glsl:19-46, geometry.glsl ------------------------------------------ */
float orc_get_radius(float depth, float norm_z, float camz, float camw)
{
    float meanFocal = ((1.0f / fabsf(camz)) + (1.0f / fabsf(camw))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius / fabsf(norm_z);
    float two_r = 2.0f * radius;
    return two_r < radius_n ? two_r : radius_n;   /* GLSL min(): y < x ? y : x with x = 2r */
}

float orc_radial_confidence(float x, float y, float cx, float cy, float max_dist, float weighting)
{
    float px = x - cx, py = y - cy;
    float radialDist = sqrtf(px * px + py * py) / max_dist;
    return hd_expf(-(radialDist * radialDist) / 0.72f) * weighting;
}

/* geometry.glsl:53-61 computeRoots2 */
static f3 roots2(float b, float cc)
{
    float d = b * b - 4.0f * cc;
    if (d < 0.0f) d = 0.0f;
    float sd = sqrtf(d);
    return v3(0.0f, 0.5f * (b + sd), 0.5f * (b - sd));
}

/* geometry.glsl:63-146 computeRoots; m[c][r] symmetric: m00 m10 m20 m11 m21 m22 */
static f3 compute_roots(float m00, float m10, float m20, float m11, float m21, float m22)
{
    float c0 = (((m00 * m11 * m22 + 2.0f * m10 * m20 * m21) - m00 * m21 * m21) - m11 * m20 * m20) - m22 * m10 * m10;
    float c1 = ((((m00 * m11 - m10 * m10) + m00 * m22) - m20 * m20) + m11 * m22) - m21 * m21;
    float c2 = (m00 + m11) + m22;
    if (fabsf(c0) < 0.000001f) return roots2(c2, c1);
    const float s_inv3 = 1.0f / 3.0f;
    const float s_sqrt3 = 1.7320508075688772f;
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    float theta = hd_atan2f(sqrtf(-q), half_b) * s_inv3;
    float st, ct;
    hd_sincosf(theta, &st, &ct);
    f3 r;
    r.x = c2_over_3 + 2.0f * rho * ct;
    r.y = c2_over_3 - rho * (ct + s_sqrt3 * st);
    r.z = c2_over_3 - rho * (ct - s_sqrt3 * st);
    if (r.x >= r.y) { float t = r.x; r.x = r.y; r.y = t; }
    if (r.y >= r.z) {
        float t = r.y; r.y = r.z; r.z = t;
        if (r.x >= r.y) { float t1 = r.x; r.x = r.y; r.y = t1; }
    }
    if (r.x <= 0.0f) return roots2(c2, c1);
    return r;
}

/* geometry.glsl:190-244 getNormalPCA, window = 3 */
static f3 normal_pca(const float *depth, int W, int H, float tx, float ty, float zc,
                     float cx, float cy, float camz, float camw)
{
    /* sample set: interior offsets -3..3; left/top clamp starts on texel 0 with integer
       coordinates (tx_min = max(0, ..) lands on texel boundaries, geometry.glsl:196-200) */
    /* geometry.glsl:198-213: the float-stepped 7 x 7 walk, literally (hd_window_axis); a sample's vertex is formed at the
       float position i * cols, j * rows — the float overload of getVertex (:21-25), unlike the pixel's own vertex */
    const hd_window wx = hd_window_axis_t(tx, W, 3.0f), wy = hd_window_axis_t(ty, H, 3.0f);   /* tx, ty: the shader's texcoord */
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
    int n = 0;
    for (float i = wx.lo; i <= wx.hi; i += wx.step) {
        const int ix = hd_window_texel(i, W);
        const float xf = i * (float)W;
        for (float j = wy.lo; j <= wy.hi; j += wy.step) {
            const int iy = hd_window_texel(j, H);
            const float yf = j * (float)H;
            float z = depth[iy * W + ix];
            if (z > 0.3f && fabsf(z - zc) < 0.05f) {
                float X = (xf - cx) * z * camz;
                float Y = (yf - cy) * z * camw;
                a0 += X * X; a1 += X * Y; a2 += X * z; a3 += Y * Y; a4 += Y * z; a5 += z * z;
                a6 += X; a7 += Y; a8 += z;
                n++;
            }
        }
    }
    if (n < 8) return v3(0.0f, 0.0f, 0.0f);
    float fn = (float)n;
    a0 /= fn; a1 /= fn; a2 /= fn; a3 /= fn; a4 /= fn; a5 /= fn; a6 /= fn; a7 /= fn; a8 /= fn;
    float m00 = a0 - a6 * a6, m10 = a1 - a6 * a7, m20 = a2 - a6 * a8;
    float m11 = a3 - a7 * a7, m21 = a4 - a7 * a8, m22 = a5 - a8 * a8;
    float s01 = m00 > m10 ? m00 : m10, s23 = m20 > m11 ? m20 : m11;
    float s0123 = s01 > s23 ? s01 : s23;
    float s45 = m21 > m22 ? m21 : m22;
    float scale = s0123 > s45 ? s0123 : s45;
    float n00 = m00 / scale, n10 = m10 / scale, n20 = m20 / scale, n11 = m11 / scale, n21 = m21 / scale,
          n22 = m22 / scale;
    f3 ev = compute_roots(m00, m10, m20, m11, m21, m22);   /* on the UNscaled matrix (geometry.glsl:217) */
    float eigenvalue = ev.x * scale;
    n00 -= eigenvalue; n11 -= eigenvalue; n22 -= eigenvalue;
    f3 row0 = v3(n00, n10, n20), row1 = v3(n10, n11, n21), row2 = v3(n20, n21, n22);
    f3 v1 = cross3(row0, row1), v2 = cross3(row0, row2), v3_ = cross3(row1, row2);
    float l1 = len3(v1), l2 = len3(v2), l3 = len3(v3_);
    f3 nrm;
    if (l1 >= l2 && l1 >= l3) nrm = v1;
    else if (l2 >= l1 && l2 >= l3) nrm = v2;
    else nrm = v3_;
    if (nrm.z < 0.0f) nrm = v3(-nrm.x, -nrm.y, -nrm.z);
    return normalize3(nrm);
}

/* geometry.glsl:36-51 getNormal (central difference), used when PCA is off; utils.glsl:23-41 */
static f3 normal_cd(const float *depth, int W, int H, int px, int py, f3 vpos, float fx_, float fy_,
                    float cx, float cy, float camz, float camw)
{
    int xf = clampi(px + 1, 0, W - 1), xb = clampi(px - 1, 0, W - 1);
    int yf = clampi(py + 1, 0, H - 1), yb = clampi(py - 1, 0, H - 1);
    float z;
    z = depth[py * W + xf]; f3 vxf = v3(((float)(px + 1) - cx) * z * camz, ((float)py - cy) * z * camw, z);
    z = depth[py * W + xb]; f3 vxb = v3(((float)(px - 1) - cx) * z * camz, ((float)py - cy) * z * camw, z);
    z = depth[yf * W + px]; f3 vyf = v3(((float)px - cx) * z * camz, ((float)(py + 1) - cy) * z * camw, z);
    z = depth[yb * W + px]; f3 vyb = v3(((float)px - cx) * z * camz, ((float)(py - 1) - cy) * z * camw, z);
    (void)fx_; (void)fy_;
    f3 del_x = sub3(scale3(add3(vxb, vpos), 0.5f), scale3(add3(vxf, vpos), 0.5f));
    f3 del_y = sub3(scale3(add3(vyb, vpos), 0.5f), scale3(add3(vyf, vpos), 0.5f));
    return normalize3(cross3(del_x, del_y));
}

/* the float overload (geometry.glsl:36-51 with float x, y): data.vert:91-94 passes x = texcoord.x * cols, i.e. half-pixel
   coordinates, where depth_vertex_normal_radius.frag passes int(x) */
static f3 normal_cd_float(const float *depth, int W, int H, int px, int py, float x, float y, f3 vpos,
                          float cx, float cy, float camz, float camw)
{
    int xf = clampi(px + 1, 0, W - 1), xb = clampi(px - 1, 0, W - 1);
    int yf = clampi(py + 1, 0, H - 1), yb = clampi(py - 1, 0, H - 1);
    float z;
    z = depth[py * W + xf]; f3 vxf = v3(((x + 1.0f) - cx) * z * camz, (y - cy) * z * camw, z);
    z = depth[py * W + xb]; f3 vxb = v3(((x - 1.0f) - cx) * z * camz, (y - cy) * z * camw, z);
    z = depth[yf * W + px]; f3 vyf = v3((x - cx) * z * camz, ((y + 1.0f) - cy) * z * camw, z);
    z = depth[yb * W + px]; f3 vyb = v3((x - cx) * z * camz, ((y - 1.0f) - cy) * z * camw, z);
    f3 del_x = sub3(scale3(add3(vxb, vpos), 0.5f), scale3(add3(vxf, vpos), 0.5f));
    f3 del_y = sub3(scale3(add3(vyb, vpos), 0.5f), scale3(add3(vyf, vpos), 0.5f));
    return normalize3(cross3(del_x, del_y));
}

static int check_neighbours(const float *depth, int W, int H, int px, int py)
{
    if (depth[py * W + clampi(px - 1, 0, W - 1)] == 0.0f) return 0;
    if (depth[clampi(py - 1, 0, H - 1) * W + px] == 0.0f) return 0;
    if (depth[py * W + clampi(px + 1, 0, W - 1)] == 0.0f) return 0;
    if (depth[clampi(py + 1, 0, H - 1) * W + px] == 0.0f) return 0;
    return 1;
}

/* data.vert:83-96: the normal and radius of a new point, recomputed by the vertex shader from the filtered depth with ITS
   texcoord — the host-computed uv attribute (hd_uv_attribute) — and ITS x, y = texcoord * cols, rows as floats.  `image` is what
   the fragment shader left for this pixel (NORMAL_PCA); it is the answer where the inputs coincide: PCA at every pixel of a
   power-of-two image and wherever the attribute equals the fragment's coordinate.  Central differences always differ (they
   run on half-pixel coordinates here, on integer ones in the fragment shader). */
f4 orc_record_normal(const orc_ctx *c, int px, int py, f4 image)
{
    const int W = c->W, H = c->H;
    const float cx = c->prm.cx, cy = c->prm.cy;
    const float camz = (float)(1.0 / (double)c->prm.fx), camw = (float)(1.0 / (double)c->prm.fy);
    const float rm = c->prm.init_radius_multiplier;
    const float tax = hd_uv_attribute(px, W), tay = hd_uv_attribute(py, H);
    const float zf = c->depth_metric_filtered[py * W + px];
    f3 nr;
    if (c->prm.normal_estimation_pca > 0.0f) {
        if (tax == hd_uv_fragment(px, W) && tay == hd_uv_fragment(py, H)) nr = v3(image.x, image.y, image.z);   /* the radius is formed below either way */
        else nr = normal_pca(c->depth_metric_filtered, W, H, tax, tay, zf, cx, cy, camz, camw);
    } else {
        const float xa = tax * (float)W, ya = tay * (float)H;
        nr = v3(0.0f, 0.0f, 0.0f);
        if (check_neighbours(c->depth_metric, W, H, px, py))
            nr = normal_cd_float(c->depth_metric_filtered, W, H, px, py, xa, ya,
                                 v3((xa - cx) * zf * camz, (ya - cy) * zf * camw, zf), cx, cy, camz, camw);
    }
    return v4(nr.x, nr.y, nr.z, rm * orc_get_radius(zf, nr.z, camz, camw));
}

/* ---- P3: depth_vertex_normal_radius.frag:23-68 -------------------------------------------- */
void orc_vertex_normal_radius(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float cx = c->prm.cx, cy = c->prm.cy;
    const float camz = (float)(1.0 / (double)c->prm.fx), camw = (float)(1.0 / (double)c->prm.fy);
    const float rm = c->prm.init_radius_multiplier;
    const float max_dist = sqrtf(((float)H * 0.5f) * ((float)H * 0.5f) + ((float)W * 0.5f) * ((float)W * 0.5f));
#pragma omp parallel for schedule(static)
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int i = py * W + px;
            const float tfx = c->frag_tc ? c->frag_tc[2 * i] : hd_uv_fragment(px, W), tfy = c->frag_tc ? c->frag_tc[2 * i + 1] : hd_uv_fragment(py, H);
            float x = tfx * (float)W, y = tfy * (float)H;     /* texcoord * cols: p + 0.5 up to an ulp where the size is no power of two */
            float zr = c->depth_metric[i], zf = c->depth_metric_filtered[i];
            f3 vr = v3(((float)px - cx) * zr * camz, ((float)py - cy) * zr * camw, zr);
            f3 vf = v3(((float)px - cx) * zf * camz, ((float)py - cy) * zf * camw, zf);
            f3 n = v3(0.0f, 0.0f, 0.0f);
            if (c->prm.normal_estimation_pca > 0.0f)
                n = normal_pca(c->depth_metric_filtered, W, H, tfx, tfy, vf.z, cx, cy, camz, camw);
            else if (check_neighbours(c->depth_metric, W, H, px, py))
                n = normal_cd(c->depth_metric_filtered, W, H, px, py, vf, 0, 0, cx, cy, camz, camw);
            float radius_init = rm * orc_get_radius(vf.z, n.z, camz, camw);
            /* build-specific side output: the un-invalidated normal + radius.  data.vert RECOMPUTES both for a new point
               (data.vert:83-96); where its inputs are the fragment shader's the result is this one, elsewhere the association
               recomputes (orc_record_normal) */
            c->normal_pca[i] = v4(n.x, n.y, n.z, radius_init);
            if (len3(n) < 0.3f || vr.z < 0.3f || vf.z < 0.3f) {
                vr = v3(0, 0, 0); vf = v3(0, 0, 0); n = v3(0, 0, 0); radius_init = 0.0f;
            }
            c->vertex_raw[i] = v4(vr.x, vr.y, vr.z, orc_radial_confidence(x, y, cx, cy, max_dist, 1.0f));
            c->vertex_filtered[i] = v4(vf.x, vf.y, vf.z, 1.0f);
            c->normal[i] = v4(n.x, n.y, n.z, radius_init);
            c->radius[i] = radius_init;
        }
}

/* ---- P4 + P5: depth_curvature_gradient.frag:28-142, depth_update_normalrad.frag ----------- */
void orc_curvature(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float camz = (float)(1.0 / (double)c->prm.fx), camw = (float)(1.0 / (double)c->prm.fy);
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int i = py * W + px;
            f4 vfil = c->vertex_filtered[i], vn = c->normal[i];
            f4 pcmax = v4(0, 0, 0, 1000.0f), pcmin = v4(0, 0, 0, 1000.0f), nopt = v4(0, 0, 0, 0);
            float gmag = 0.0f;
            if (vfil.z > 0.3f && len3(xyz(vn)) > 0.5f) {
                float k1 = 1000.0f, k2 = 1000.0f;
                f3 pmax = v3(0, 0, 0), pmin = v3(0, 0, 0);
                f4 vc[100], nr[100];
                int n = 0;
                /* depth_curvature_gradient.frag:48-73: the float-stepped window, literally (hd_window_axis) */
                const hd_window wx = hd_window_axis_t(c->frag_tc ? c->frag_tc[2 * i] : hd_uv_fragment(px, W), W, c->prm.curv_estimation_window);
                const hd_window wy = hd_window_axis_t(c->frag_tc ? c->frag_tc[2 * i + 1] : hd_uv_fragment(py, H), H, c->prm.curv_estimation_window);
                for (float fi = wx.lo; fi <= wx.hi; fi += wx.step) {
                    const int ix = hd_window_texel(fi, W);
                    for (float fj = wy.lo; fj <= wy.hi; fj += wy.step) {
                        const int iy = hd_window_texel(fj, H);
                        f4 v = c->vertex_filtered[iy * W + ix], nn = c->normal[iy * W + ix];
                        if (fabsf(v.z - vfil.z) < 0.10f && v.z > 0.3f && len3(xyz(nn)) > 0.8f) {
                            vc[n] = v4(v.x, v.y, v.z, 1.0f);
                            nr[n] = nn;
                            n++;
                        }
                    }
                }
                if (n > 15) {
                    float p[3] = {vfil.x, vfil.y, vfil.z};
                    float gr[3], g[9];
                    orc_hrbf_gradient(p, vc, nr, n, gr);
                    gmag = fabsf((gr[0] * vn.x + gr[1] * vn.y) + gr[2] * vn.z);
                    f3 g1 = normalize3(v3(gr[0], gr[1], gr[2]));
                    nopt = v4(g1.x, g1.y, g1.z, vn.w);
                    (void)camz; (void)camw;
                    orc_hrbf_hessian(p, vc, nr, n, g);
                    float g0 = gr[0], g1_ = gr[1], g2 = gr[2];
                    float g2c = g2 * g2 * g2;
                    float h_x = -g0 / g2, h_y = -g1_ / g2;
                    float h_xx = (((2.0f * g0 * g2 * g[2] - g0 * g0 * g[8]) - g2 * g2 * g[0])) / g2c;
                    float h_xy = (((g0 * g2 * g[5] + g1_ * g2 * g[2]) - g0 * g1_ * g[8]) - g2 * g2 * g[1]) / g2c;
                    float h_yy = (((2.0f * g1_ * g2 * g[5] - g1_ * g1_ * g[8]) - g2 * g2 * g[4])) / g2c;
                    f3 r_u = v3(1.0f, 0.0f, h_x), r_v = v3(0.0f, 1.0f, h_y);
                    float E = 1.0f + h_x * h_x, F = h_x * h_y, G = 1.0f + h_y * h_y;
                    float length = sqrtf((h_x * h_x + h_y * h_y) + 1.0f);
                    float L = h_xx / length, M = h_xy / length, N = h_yy / length;
                    float den = E * G - F * F;
                    float curvature_g = (L * N - M * M) / den;
                    float curvature_m = ((E * N + G * L) - 2.0f * F * M) / (2.0f * den);
                    if (!hd_isnanf(curvature_g) && !hd_isnanf(curvature_m)) {
                        float delta = curvature_m * curvature_m - curvature_g;
                        if (delta < 0.0f) delta = 0.0f;
                        float sd = sqrtf(delta);
                        k1 = curvature_m + sd;
                        k2 = curvature_m - sd;
                        float lmax = -(M - k1 * F) / (N - k1 * G);
                        float lmin = -(M - k2 * F) / (N - k2 * G);
                        pmax = normalize3(add3(r_u, scale3(r_v, lmax)));
                        pmin = normalize3(add3(r_u, scale3(r_v, lmin)));
                    }
                }
                pcmax = v4(pmax.x, pmax.y, pmax.z, k1);
                pcmin = v4(pmin.x, pmin.y, pmin.z, k2);
            }
            c->curv1[i] = pcmax; c->curv2[i] = pcmin; c->gradmag[i] = gmag; c->normal_opt[i] = nopt;
        }
    /* updateNormalRad: NORMAL <- NORMAL_OPT (HRBFFusion.cpp:1301-1310) */
    for (int i = 0; i < c->P; ++i) c->normal[i] = c->normal_opt[i];
}

/* ---- VertexConfidence: depth_confidence_evaluation.frag:37-50 ----------------------------- */
void orc_confidence(orc_ctx *c)
{
    const int W = c->W, H = c->H;
    const float cx = c->prm.cx, cy = c->prm.cy;
    const float max_dist = sqrtf(((float)H * 0.5f) * ((float)H * 0.5f) + ((float)W * 0.5f) * ((float)W * 0.5f));
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            int i = py * W + px;
            float conf = orc_radial_confidence(hd_px_fragment(px, W), hd_px_fragment(py, H), cx, cy, max_dist, c->weighting);
            if (c->prm.use_conf_eval > 0) conf = conf * hd_expf(-c->prm.conf_eval_epsilon / sqrtf(c->gradmag[i]));
            c->confidence[i] = conf;
        }
}
