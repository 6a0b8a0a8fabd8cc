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
    double lu = std::sqrt(ts->dpdu[0] * ts->dpdu[0] + ts->dpdu[1] * ts->dpdu[1] + ts->dpdu[2] * ts->dpdu[2]);
    double lv = std::sqrt(ts->dpdv[0] * ts->dpdv[0] + ts->dpdv[1] * ts->dpdv[1] + ts->dpdv[2] * ts->dpdv[2]);
    ts->inv_uv_size = lu > lv ? lu : lv;
}

} // namespace gdpt
