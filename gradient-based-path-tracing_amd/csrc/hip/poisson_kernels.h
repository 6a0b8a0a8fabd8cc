// poisson_kernels.h — launch interface of the screened-Poisson solver (host side of poisson_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gdpt {

struct PoissonResult { int iterations; int solver; double rel_residual; double solve_ms; };

// c=img; cx=cx0(x,y)+cx1(x-1,y); cy=cy0(x,y)+cy1(x,y-1)   (src/render.cpp:340-350)
void launch_assemble(int w, int h, const double *img, const double *cx0, const double *cy0, const double *cx1, const double *cy1,
                     double *c, double *cx, double *cy, hipStream_t stream);

// Screened Poisson solve on device buffers (W*H*3 doubles, interleaved RGB), reproducing fourierSolve
// (src/render.cpp:172-254): same operator, mirror boundaries and DC override. Synchronises `stream`
// (convergence is checked on the host between chunks of iterations).
PoissonResult poisson_solve_device(int w, int h, const double *d_c, const double *d_gx, const double *d_gy, double alpha,
                                   double *d_out, int solver, double tol, int max_iters, hipStream_t stream);

void poisson_release_workspace();

} // namespace gdpt
