// tri_precompute.cpp — see tri_precompute.h. Formulas: src/shapes/triangle_mesh.inl:93-131,:168,
// src/intersection.cpp:41, src/frame.h:11-22.
#include "tri_precompute.h"
#include <cmath>

namespace gdpt {

void precompute_tri_constants(const double pos[3][3], const float e1[3], const float e2[3], DevTriShade *ts) {
    // geometric normal: fp32 cross product (what the traversal reports as Ng), normalised in fp64
    float ngf[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    double ng[3] = {(double)ngf[0], (double)ngf[1], (double)ngf[2]};
    double l = std::sqrt(ng[0] * ng[0] + ng[1] * ng[1] + ng[2] * ng[2]);
    if (l <= 0) { ts->gn[0] = ts->gn[1] = ts->gn[2] = 0; }
    else { double inv = 1.0 / l; for (int k = 0; k < 3; k++) ts->gn[k] = ng[k] * inv; }
    double dsx = ts->uv[2][0] - ts->uv[0][0], dsy = ts->uv[2][1] - ts->uv[0][1];   // duvds
    double dtx = ts->uv[2][0] - ts->uv[1][0], dty = ts->uv[2][1] - ts->uv[1][1];   // duvdt
    double det = dsx * dty - dtx * dsy;
    if (std::fabs(det) > (double)1e-8f) {
        double dsdu = dty / det, dtdu = -dsy / det, dsdv = dtx / det, dtdv = -dsx / det;
        for (int k = 0; k < 3; k++) {
            double dpds = pos[2][k] - pos[0][k], dpdt = pos[2][k] - pos[1][k];
            ts->dpdu[k] = dpds * dsdu + dpdt * dtdu;
            ts->dpdv[k] = dpds * dsdv + dpdt * dtdv;
        }
    } else {
        const double *n = ts->gn;   // coordinate_system(vertex.geometric_normal): before the flip to the shading side
        if (n[2] < (-1 + 1e-6)) {
            ts->dpdu[0] = 0; ts->dpdu[1] = -1; ts->dpdu[2] = 0;
            ts->dpdv[0] = -1; ts->dpdv[1] = 0; ts->dpdv[2] = 0;
        } else {
            double a = 1 / (1 + n[2]);
            double b = -n[0] * n[1] * a;
            ts->dpdu[0] = 1 - n[0] * n[0] * a; ts->dpdu[1] = b; ts->dpdu[2] = -n[0];
            ts->dpdv[0] = b; ts->dpdv[1] = 1 - n[1] * n[1] * a; ts->dpdv[2] = -n[1];
        }
    }
    // flat triangles: compute_shading_info's frame (:133-166) once, at st = (1/3, 1/3)
    bool flat = true;
    if (ts->has_normals)
        for (int k = 0; k < 3; k++) if (ts->n[0][k] != ts->n[1][k] || ts->n[0][k] != ts->n[2][k]) flat = false;
    ts->flat_frame = flat ? 1 : 0; ts->pad = 0;
    if (flat) {
        double sn[3];
        if (ts->has_normals) {
            const double s = 1.0 / 3.0, t = 1.0 / 3.0, b0 = 1 - s - t;
            double v[3];
            for (int k = 0; k < 3; k++) v[k] = b0 * ts->n[0][k] + s * ts->n[1][k] + t * ts->n[2][k];
            double ln = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            double inv = ln > 0 ? 1.0 / ln : 0.0;
            for (int k = 0; k < 3; k++) sn[k] = v[k] * inv;
        } else {
            for (int k = 0; k < 3; k++) sn[k] = ts->gn[k];
        }
        double d = sn[0] * ts->dpdu[0] + sn[1] * ts->dpdu[1] + sn[2] * ts->dpdu[2];
        double tg[3] = {ts->dpdu[0] - sn[0] * d, ts->dpdu[1] - sn[1] * d, ts->dpdu[2] - sn[2] * d};
        double lt = std::sqrt(tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2]);
        double it = lt > 0 ? 1.0 / lt : 0.0;
        for (int k = 0; k < 3; k++) tg[k] *= it;
        double bt[3] = {sn[1] * tg[2] - sn[2] * tg[1], sn[2] * tg[0] - sn[0] * tg[2], sn[0] * tg[1] - sn[1] * tg[0]};
        double lb = std::sqrt(bt[0] * bt[0] + bt[1] * bt[1] + bt[2] * bt[2]);
        double ib = lb > 0 ? 1.0 / lb : 0.0;
        for (int k = 0; k < 3; k++) { ts->n[0][k] = tg[k]; ts->n[1][k] = bt[k] * ib; ts->n[2][k] = sn[k]; }
    }
    double lu = std::sqrt(ts->dpdu[0] * ts->dpdu[0] + ts->dpdu[1] * ts->dpdu[1] + ts->dpdu[2] * ts->dpdu[2]);
    double lv = std::sqrt(ts->dpdv[0] * ts->dpdv[0] + ts->dpdv[1] * ts->dpdv[1] + ts->dpdv[2] * ts->dpdv[2]);
    ts->inv_uv_size = lu > lv ? lu : lv;
}

} // namespace gdpt
