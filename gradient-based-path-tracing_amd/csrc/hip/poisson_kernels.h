// poisson_kernels.h — launch interface of the screened-Poisson solver (host side of poisson_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gdpt {

struct PoissonResult { int iterations; int solver; double rel_residual; double solve_ms; };

// c=img; cx=cx0(x,y)+cx1(x-1,y); cy=cy0(x,y)+cy1(x,y-1)   (src/render.cpp:340-350), rows [row_begin,row_end) (0,0 = all)
void launch_assemble(int w, int h, int row_begin, int row_end, const double *img, const double *cx0, const double *cy0, const double *cx1, const double *cy1,
                     double *c, double *cx, double *cy, hipStream_t stream);

// Screened Poisson solve on device buffers (W*H*3 doubles, interleaved RGB), reproducing fourierSolve
// (src/render.cpp:172-254): same operator, mirror boundaries and DC override.
// GDPT_SOLVER_DCT only enqueues work on `stream` (no event, no host wait) unless `timed`, which brackets the solve with
// HIP events and waits for it (PoissonResult::solve_ms; 0 otherwise). GDPT_SOLVER_CG synchronises `stream` in any
// case: convergence is checked on the host between chunks of iterations. Scratch state is kept per (device, stream).
PoissonResult poisson_solve_device(int w, int h, const double *d_c, const double *d_gx, const double *d_gy, double alpha,
                                   double *d_out, int solver, double tol, int max_iters, hipStream_t stream, bool timed);

// launch_assemble over the whole film followed by poisson_solve_device on its outputs, with the first two passes over the film fused
// for the DCT solvers (c / cx / cy are still written; same bits as the two calls).
PoissonResult assemble_solve_device(int w, int h, const double *img, const double *cx0, const double *cy0, const double *cx1, const double *cy1,
                                    double *d_c, double *d_cx, double *d_cy, double alpha, double *d_out, int solver, double tol, int max_iters,
                                    hipStream_t stream, bool timed);

// Drops the (device, stream) pair's scratch state; call before destroying a stream the solver has run on (nothing in flight).
void poisson_forget_stream(int dev, hipStream_t stream);
void poisson_release_workspace();

} // namespace gdpt
