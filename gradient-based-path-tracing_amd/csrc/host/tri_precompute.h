// tri_precompute.h — per-triangle constants of compute_shading_info evaluated once on the host.
#pragma once
#include "../device_scene.h"
namespace gdpt {
// pos: fp64 vertex positions; e1/e2: the fp32 traversal edges (v1-v0, v2-v0). Fills dpdu, dpdv, gn, inv_uv_size
// of `ts` (its uv[][] must already be set). Compiled with -ffp-contract=off: one rounding per operation, like the
// reference built for x86-64.
void precompute_tri_constants(const double pos[3][3], const float e1[3], const float e2[3], DevTriShade *ts);
}
