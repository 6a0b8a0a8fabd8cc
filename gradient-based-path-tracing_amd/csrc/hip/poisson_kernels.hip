// poisson_kernels.hip — gradient assembly and the screened-Poisson reconstruction for gfx950.
//
// Reference: fourierSolve (src/render.cpp:172-254) solves, per channel,
//     argmin_f  alpha*|f-u|^2 + |grad f - g|^2
// with a DCT-I (FFTW REDFT00), whose basis implies whole-sample-symmetric (mirror, edge not
// repeated) boundaries, and then overrides the DC coefficient with sum(w*u) (:205-211,:239).
// The same f is the solution of the linear system
//     W (alpha I - L) f = W h,     h = alpha*u - D(g)                                 (:213-224)
// where L is the 5-point Laplacian with mirror boundaries (x-part 2(f[1]-f[0]) at x=0), W = diag(w_x w_y)
// with w = 1 on borders and 2 inside (it makes the operator symmetric), followed by the constant
//     f += (sum(w u) - sum(w h)/alpha) / (4 (W-1)(H-1))                                per channel
// which reproduces the DC override (derivation: SURVEY.md §8(a) P1, Appendix A.2).
// Here: conjugate gradients on that system, all three channels as one block-diagonal system
// (N = W*H*3 unknowns, interleaved RGB exactly as Image3::data), fp64, x0 = u.
//
// Per iteration two kernels and no host round trip:
//   cg_step_a: p' = r + beta p (recomputed on the 5 taps, p double-buffered), q = A p', partial <p',q>
//   cg_step_b: x += a p', r -= a q, partial <r,r>
// Scalars are reduced from per-block partials by every block in the same fixed order (deterministic).
#include "poisson_kernels.h"
#include "../capi_common.h"
#include "../../../include/gdpt.h"

#include <rocblas/rocblas.h>

#include <chrono>
#include <cmath>
#include <vector>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

namespace gp {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;

struct CgState {
    double rr[2];         // <r,r> ping-pong by iteration parity
    double bb;            // <b,b>
    double wu[3], wh[3];  // sum(w*u), sum(w*h) per channel
    double tol2;          // tol^2
    int iters;
    int converged;
};

__device__ __forceinline__ double block_sum(double v, double *red) {
    // fixed-order: xor tree inside each wave, then the 4 wave totals in index order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int k = 1; k < kBlock / 64; k++) s += red[k];
    return s;
}
__device__ __forceinline__ double reduce_partials(const double *part, int n, double *red) {
    double v = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) v += part[i];
    return block_sum(v, red);
}

struct Geo { int w, h, n3, row; };   // row = 3*w doubles per image row

__device__ __forceinline__ double weight(const Geo &g, int x, int y) {
    double wx = (x > 0 && x < g.w - 1) ? 2.0 : 1.0, wy = (y > 0 && y < g.h - 1) ? 2.0 : 1.0;
    return wx * wy;
}
// (L v)[i] with mirror boundaries; `at(j)` returns v[j]
template <class F>
__device__ __forceinline__ double laplace(const Geo &g, int i, int x, int y, double vi, F at) {
    double lx, ly;
    if (x > 0 && x < g.w - 1) lx = (at(i - 3) + at(i + 3)) - 2.0 * vi;
    else if (x == 0) lx = 2.0 * (at(i + 3) - vi);
    else lx = 2.0 * (at(i - 3) - vi);
    if (y > 0 && y < g.h - 1) ly = (at(i - g.row) + at(i + g.row)) - 2.0 * vi;
    else if (y == 0) ly = 2.0 * (at(i + g.row) - vi);
    else ly = 2.0 * (at(i - g.row) - vi);
    return lx + ly;
}

// rows [row_begin, row_end) only: a rank of the sharded tile loop assembles its own band (cy of the band's first row
// reads the halo row row_begin-1 of cy1, which the rank above has sent)
__global__ __launch_bounds__(kBlock) void assemble_kernel(Geo g, int row_begin, int row_end, const double *img, const double *cx0, const double *cy0,
                                                          const double *cx1, const double *cy1, double *c, double *cx, double *cy) {
    const int lo = row_begin * g.row, hi = row_end * g.row;
    for (int i = lo + blockIdx.x * kBlock + threadIdx.x; i < hi; i += gridDim.x * kBlock) {
        int y = i / g.row, x = (i - y * g.row) / 3;
        c[i] = img[i];
        cx[i] = (x == 0) ? cx0[i] : cx0[i] + cx1[i - 3];
        cy[i] = (y == 0) ? cy0[i] : cy0[i] + cy1[i - g.row];
    }
}

// h = alpha*u - D(g) (src/render.cpp:213-224); b = w*h; x0 = u; r0 = b - W(alpha I - L)x0; p = 0.
// partials layout: [0] rr, [1] bb, [2..4] wu, [5..7] wh, each gridDim.x long.
__global__ __launch_bounds__(kBlock) void cg_init_kernel(Geo g, double alpha, const double *u, const double *gx, const double *gy,
                                                         double *x, double *r, double *p0, double *p1, double *partials) {
    __shared__ double red[kBlock / 64];
    double s_rr = 0, s_bb = 0, s_wu[3] = {0, 0, 0}, s_wh[3] = {0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        int y = i / g.row, col = i - y * g.row, xx = col / 3, ch = col - xx * 3;
        double ui = u[i];
        double hv = alpha * ui;
        if (xx > 0 && xx < g.w - 1) hv -= (gx[i + 3] - gx[i]); else hv -= (-2.0 * gx[i]);
        if (y > 0 && y < g.h - 1) hv -= (gy[i + g.row] - gy[i]); else hv -= (-2.0 * gy[i]);
        double wgt = weight(g, xx, y);
        double b = wgt * hv;
        double Au = wgt * (alpha * ui - laplace(g, i, xx, y, ui, [&](int j) { return u[j]; }));
        double ri = b - Au;
        x[i] = ui; r[i] = ri; p0[i] = 0.0; p1[i] = 0.0;
        s_rr += ri * ri; s_bb += b * b;
#pragma unroll
        for (int k = 0; k < 3; k++) if (ch == k) { s_wu[k] += wgt * ui; s_wh[k] += wgt * hv; }
    }
    const int nb = gridDim.x;
    double v;
    v = block_sum(s_rr, red); if (threadIdx.x == 0) partials[0 * nb + blockIdx.x] = v;
    v = block_sum(s_bb, red); if (threadIdx.x == 0) partials[1 * nb + blockIdx.x] = v;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        v = block_sum(s_wu[k], red); if (threadIdx.x == 0) partials[(2 + k) * nb + blockIdx.x] = v;
        v = block_sum(s_wh[k], red); if (threadIdx.x == 0) partials[(5 + k) * nb + blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(kBlock) void cg_init_reduce_kernel(int nb, const double *partials, CgState *st, double tol) {
    __shared__ double red[kBlock / 64];
    double bb = reduce_partials(partials + 1 * nb, nb, red);
    double wu[3], wh[3];
    for (int k = 0; k < 3; k++) { wu[k] = reduce_partials(partials + (2 + k) * nb, nb, red); wh[k] = reduce_partials(partials + (5 + k) * nb, nb, red); }
    if (threadIdx.x == 0) {
        st->bb = bb; st->rr[0] = 1.0; st->rr[1] = 1.0;
        for (int k = 0; k < 3; k++) { st->wu[k] = wu[k]; st->wh[k] = wh[k]; }
        st->tol2 = tol * tol; st->iters = 0; st->converged = 0;
    }
}

// iteration `it` (parity selects the p buffers): p_out = r + beta p_in; q = A p_out; partial <p_out,q>
__global__ __launch_bounds__(kBlock) void cg_step_a(Geo g, double alpha, int it, const double *r, const double *p_in, double *p_out,
                                                    double *q, const double *part_rr, double *part_pq, CgState *st) {
    __shared__ double red[kBlock / 64];
    const int nb = gridDim.x;
    double rr_new = reduce_partials(part_rr, nb, red);
    const double rr_old = st->rr[(it + 1) & 1];
    const double bb = st->bb, tol2 = st->tol2;
    if (rr_new <= tol2 * bb || st->converged) {        // uniform over the whole grid: same inputs in every block
        if (blockIdx.x == 0 && threadIdx.x == 0) st->converged = 1;
        return;
    }
    const double beta = rr_new / rr_old;               // first iteration: p_in = 0, beta arbitrary
    if (blockIdx.x == 0 && threadIdx.x == 0) st->rr[it & 1] = rr_new;
    double s = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        int y = i / g.row, xx = (i - y * g.row) / 3;
        double pi = r[i] + beta * p_in[i];
        double lap = laplace(g, i, xx, y, pi, [&](int j) { return r[j] + beta * p_in[j]; });
        double qi = weight(g, xx, y) * (alpha * pi - lap);
        p_out[i] = pi; q[i] = qi;
        s += pi * qi;
    }
    double v = block_sum(s, red);
    if (threadIdx.x == 0) part_pq[blockIdx.x] = v;
}

__global__ __launch_bounds__(kBlock) void cg_step_b(Geo g, int it, const double *p, const double *q, double *x, double *r,
                                                    const double *part_pq, double *part_rr, CgState *st) {
    __shared__ double red[kBlock / 64];
    if (st->converged) return;
    const int nb = gridDim.x;
    double pq = reduce_partials(part_pq, nb, red);
    const double a = st->rr[it & 1] / pq;
    double s = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        x[i] += a * p[i];
        double ri = r[i] - a * q[i];
        r[i] = ri;
        s += ri * ri;
    }
    double v = block_sum(s, red);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = v;
        if (blockIdx.x == 0) st->iters = it + 1;
    }
}

// out = x + (sum(w u) - sum(w h)/alpha) / (4 (W-1)(H-1)) per channel  (the DC override, src/render.cpp:239)
__global__ __launch_bounds__(kBlock) void cg_finalize_kernel(Geo g, double alpha, const double *x, double *out,
                                                             const double *part_rr, int nb, CgState *st, double *rel_res) {
    __shared__ double red[kBlock / 64];
    double shift[3];
    double denom = 4.0 * (double)(g.w - 1) * (double)(g.h - 1);
    for (int k = 0; k < 3; k++) shift[k] = (st->wu[k] - st->wh[k] / alpha) / denom;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        int ch = i % 3;
        out[i] = x[i] + (ch == 0 ? shift[0] : (ch == 1 ? shift[1] : shift[2]));
    }
    if (blockIdx.x == 0) {
        double rr = reduce_partials(part_rr, nb, red);
        if (threadIdx.x == 0) *rel_res = (st->bb > 0) ? sqrt(rr / st->bb) : 0.0;
    }
}


// ------------------------------------------------------------------------------------------------
// Direct solver: the reference's own algorithm (src/render.cpp:172-254) with FFTW's REDFT00 replaced by its
// definition as a dense product, Y[k] = sum_j w_j X[j] cos(pi j k/(n-1)) (w = 1 at the ends, 2 inside), i.e.
//     H^ = Ch^T * H * Cw,   C[j][k] = w_j cos(pi j k/(n-1))
// evaluated with fp64 GEMMs (MFMA via rocBLAS) on channel-planar buffers. Includes the fp32-rounded Laplacian
// eigenvalue (:233) and the DC override (:239), so it matches the reference operator exactly.
// ------------------------------------------------------------------------------------------------
// planar h[c][y][x] = alpha*u - D(g) (:213-224); partials[c*nb + b] = block partial of sum(w*u) per channel
__global__ __launch_bounds__(kBlock) void dct_rhs_kernel(Geo g, double alpha, const double *u, const double *gx, const double *gy,
                                                         double *hplanar, double *partials) {
    __shared__ double red[kBlock / 64];
    const int plane = g.w * g.h;
    double s_wu[3] = {0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        int y = i / g.row, col = i - y * g.row, xx = col / 3, ch = col - xx * 3;
        double ui = u[i];
        double hv = alpha * ui;
        if (xx > 0 && xx < g.w - 1) hv -= (gx[i + 3] - gx[i]); else hv -= (-2.0 * gx[i]);
        if (y > 0 && y < g.h - 1) hv -= (gy[i + g.row] - gy[i]); else hv -= (-2.0 * gy[i]);
        hplanar[(size_t)ch * plane + (size_t)y * g.w + xx] = hv;
        double wgt = weight(g, xx, y);
#pragma unroll
        for (int k = 0; k < 3; k++) if (ch == k) s_wu[k] += wgt * ui;
    }
    const int nb = gridDim.x;
#pragma unroll
    for (int k = 0; k < 3; k++) { double v = block_sum(s_wu[k], red); if (threadIdx.x == 0) partials[k * nb + blockIdx.x] = v; }
}
// assemble_kernel + dct_rhs_kernel in one pass over the film (whole images on one device): the five rendered buffers are read
// once, c / cx / cy written once and not read back — every value by the expression the two kernels use, so the bits are theirs.
__global__ __launch_bounds__(kBlock) void assemble_rhs_kernel(Geo g, double alpha, const double *img, const double *cx0, const double *cy0, const double *cx1,
                                                              const double *cy1, double *c, double *cx, double *cy, double *hplanar, double *partials) {
    __shared__ double red[kBlock / 64];
    const int plane = g.w * g.h;
    double s_wu[3] = {0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        int y = i / g.row, col = i - y * g.row, xx = col / 3, ch = col - xx * 3;
        const double ui = img[i];
        const double gxi = (xx == 0) ? cx0[i] : cx0[i] + cx1[i - 3];
        const double gyi = (y == 0) ? cy0[i] : cy0[i] + cy1[i - g.row];
        c[i] = ui; cx[i] = gxi; cy[i] = gyi;
        double hv = alpha * ui;
        if (xx > 0 && xx < g.w - 1) hv -= ((cx0[i + 3] + cx1[i]) - gxi); else hv -= (-2.0 * gxi);
        if (y > 0 && y < g.h - 1) hv -= ((cy0[i + g.row] + cy1[i]) - gyi); else hv -= (-2.0 * gyi);
        hplanar[(size_t)ch * plane + (size_t)y * g.w + xx] = hv;
        double wgt = weight(g, xx, y);
#pragma unroll
        for (int k = 0; k < 3; k++) if (ch == k) s_wu[k] += wgt * ui;
    }
    const int nb = gridDim.x;
#pragma unroll
    for (int k = 0; k < 3; k++) { double v = block_sum(s_wu[k], red); if (threadIdx.x == 0) partials[k * nb + blockIdx.x] = v; }
}
// F^ = H^ / (alpha - (float)(lapY[y] + lapX[x]))  (:229-235)
__global__ __launch_bounds__(kBlock) void dct_scale_kernel(Geo g, double alpha, const double *lap_x, const double *lap_y, double *hhat) {
    const int plane = g.w * g.h;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < 3 * plane; i += gridDim.x * kBlock) {
        int r = i % plane, y = r / g.w, xx = r - y * g.w;
        float resp = (float)(lap_y[y] + lap_x[xx]);          // `float ftLapResponse` in the reference
        hhat[i] = hhat[i] / (alpha - resp);
    }
}
// F^[0,0] := sum(w*u) per channel (:239), after the scaling pass
__global__ void dct_dc_kernel(Geo g, double *hhat, const double *partials, int nb) {
    __shared__ double red[kBlock / 64];
    const int plane = g.w * g.h;
    for (int c = 0; c < 3; c++) {
        double dc = reduce_partials(partials + c * nb, nb, red);
        __syncthreads();
        if (threadIdx.x == 0) hhat[(size_t)c * plane] = dc;
    }
}
__global__ __launch_bounds__(kBlock) void dct_finalize_kernel(Geo g, const double *fplanar, double *out) {
    const int plane = g.w * g.h;
    const double denom = 4.0 * (double)(g.w - 1) * (double)(g.h - 1);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < g.n3; i += gridDim.x * kBlock) {
        int p = i / 3, ch = i - p * 3;
        out[i] = fplanar[(size_t)ch * plane + p] / denom;      // :245-247
    }
}


// ------------------------------------------------------------------------------------------------
// The 1-D transforms as hand-written fp64 MFMA GEMMs (v_mfma_f64_16x16x4_f64) on HALF the flops: the DCT-I matrix is
// even / odd symmetric about its middle column, cos(pi j (n-1-k)/(n-1)) = (-1)^j cos(pi j k/(n-1)) with w_{n-1-k} = w_k, so
//     Y[2q+p] = sum_{x < ceil(n/2)} E_p[x][q] * (X[x] + (-1)^p X[n-1-x]),   E_p[x][q] = w_x cos(pi x (2q+p)/(n-1))
// (the middle sample of an odd n pairs with itself: it counts once for p = 0 and not at all for p = 1). Each 1-D pass is
// two products of half the depth and half the outputs, one per parity p of the output index, and the fold rides on the
// operand staging: the thread that brings X[x] to LDS also fetches X[n-1-x] and stores the sum or the difference. Outputs go
// straight to rows / columns 2q+p of the natural layout, so the next pass sees an ordinary matrix.
//   FORM 0 (row pass,    T = X * Cw):    C[m][2q+p] = sum_x (X[m][x] +- X[m][K-1-x]) * E_p[x][q]      data operand = A, folded along its rows
//   FORM 1 (column pass, Y = Ch^T * T):  C[2q+p][n] = sum_y E_p^T[q][y] * (X[y][n] +- X[K-1-y][n])    data operand = B, folded along its columns
// Block = 8 waves (two per SIMD: one wave's LDS traffic and waits sit under the other's MFMAs), block tile 64 x 64 of one
// (channel, parity) pair, K step BK staged through double-buffered LDS. Two shapes of the same loop, picked by the host:
//   BK = 32, global loads TWO steps ahead in a ring of two register groups, 76 KB of LDS (2 blocks per CU): for grids of at
//           most two blocks per CU (512 x 512: 192 blocks), where nothing but the block itself can hide a load from L2 / the
//           Infinity Cache (a 16-deep step is 8 MFMAs = 512 cycles per wave; with loads one such step ahead the 512 x 512 solve
//           took 103 us, the same as the library, with this shape 93);
//   BK = 16, loads one step ahead, 39 KB of LDS and 61 VGPRs (4 blocks = 32 waves per CU): for larger grids, where the other
//           resident blocks cover the loads (1024 x 1024: 352 us against 373-383 with the deep shape).
// LDS images: A tile row-major [64][BK + 1] (the MFMA's A lanes read [row l&15][k l>>4]), B tile k-major [BK][80]
// ([k l>>4][col l&15]). The compiler fetches the operands with ds_read2_b64, which the LDS serves in groups of 16 lanes on 32
// banks (MI355X_MICROARCH.md, LDS table): 16 rows of the A tile must therefore start on 16 different even banks, which an ODD
// row stride gives (2 (BK+1) r mod 32 = 2 r) and the even stride of the first version did not (SQ_LDS_BANK_CONFLICT as many
// cycles as SQ_ACTIVE_INST_LDS: every A read a two-way conflict, profiles/r03_poisson_pmc.txt). A wave owns 16 x 32 = two MFMA tiles; result
// register r of lane l is C[(l>>4) + 4r][l&15] (the f64 map, not the f32 one).
// Epilogues (EPI, column pass only): 0 plain; 1 the spectral division  F^ = H^ / (alpha - (float)(lapY[y] + lapX[x]))  with
// the DC override F^[0,0] = sum(w u) (src/render.cpp:229-239: the block holding element (0,0) reduces the per-block partials
// of dct_rhs_kernel itself); 2 the final  out[(y*W+x)*3+ch] = f / (4 (W-1)(H-1))  (:245-247), planar -> interleaved.
// (History, whole 512x512x3 solve on MI355X: unfolded 64x64 tiles with 4 waves 132 us, rocBLAS dgemm_strided_batched 103 us.)
constexpr int kGemmBN = 64, kGemmLdB = 80, kGemmThreads = 512;
struct GemmEpi {
    double alpha; const double *lap_x, *lap_y;      // EPI 1
    const double *dc_partials; int dc_nb;            // EPI 1: [3][dc_nb] block partials of sum(w u)
    double denom; double *out; int out_w;            // EPI 2
};
// M x N = extent of the output; K = length of the folded dimension (FORM 0: K == N, FORM 1: K == M); X = data operand with
// row stride ldx and channel stride strideX; E0 / E1 = the parity tables (FORM 0: E_p, ceil(K/2) x ceil-or-floor(K/2);
// FORM 1: E_p^T), row stride ldE each.
// Block -> tile, XCD-aware. A 64 x 64 tile of depth K/2 reads 5 flops per operand byte; with every block fetching its operand
// panels from the Infinity Cache the four products of a 512 x 512 solve were bound by that traffic (75 MB per product, 21 us
// against 7 us of MFMA time). The blocks that share a DATA panel — FORM 0: the same 64 rows of X (all column tiles, both
// parities), FORM 1: the same 64 columns — are therefore given block ids that are equal modulo 8, which the hardware deals
// to one XCD (MI355X_MICROARCH.md: workgroup dispatch; for speed only — nothing depends on it): the panel is fetched into
// that XCD's L2 once and the other seven members hit there; the parity tables (1 MB at 512) end up in every L2.
// Grid = 8 x ceil(groups / 8) x members blocks, 1-D; ids whose group does not exist return at once.
template <int FORM, int EPI, int BK, int BM>
__global__ __launch_bounds__(kGemmThreads, BK == 32 ? 2 : 4) void dct_fold_gemm_f64(int M, int N, int K, int tiles_m, int tiles_n, const double *X, int ldx, long long strideX,
                                                                     const double *E0, const double *E1, int ldE,
                                                                     double *C, int ldc, long long strideC, GemmEpi e) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    static_assert((BM == 64 || BM == 32) && (BK == 16 || BK == 32), "tile shapes");
    // HB / HA staging passes per step and thread for the B / A tile; odd A row stride, see above; NT MFMA tiles per wave
    constexpr int kGemmBM = BM, kGemmBK = BK, kGemmLdA = BK + 1, HB = BK / 16, HA = BM * BK / 1024, NT = BM / 32;
    __shared__ __attribute__((aligned(16))) double sA[2][kGemmBM * kGemmLdA];
    __shared__ __attribute__((aligned(16))) double sB[2][kGemmBK * kGemmLdB];
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ch, p, tile_m, tile_n;
    {
        const int b = (int)blockIdx.x, xcd = b & 7, j = b >> 3;
        const int shared_tiles = FORM == 0 ? tiles_m : tiles_n, other_tiles = FORM == 0 ? tiles_n : tiles_m;
        const int members = other_tiles * 2, groups = shared_tiles * 3;
        const int t = j / members, mem = j - t * members;
        const int g = xcd + 8 * t;
        if (g >= groups) return;                             // (block-uniform)
        ch = g / shared_tiles;
        const int st = g - ch * shared_tiles;
        p = mem / other_tiles;
        const int ot = mem - p * other_tiles;
        tile_m = FORM == 0 ? st : ot; tile_n = FORM == 0 ? ot : st;
    }
    X += (long long)ch * strideX; C += (long long)ch * strideC;
    const double *E = p ? E1 : E0;
    const int Kf = (K + 1) >> 1;                         // folded depth
    const int mid = (K & 1) ? (K >> 1) : -1;             // the self-paired sample of an odd length
    const int Qn = (K + 1 - p) >> 1;                     // outputs of this parity along the folded dimension
    const int Mt = FORM == 0 ? M : Qn, Nt = FORM == 0 ? Qn : N;     // extent of this block's (parity-compact) output index space
    const int m0 = tile_m * kGemmBM, n0 = tile_n * kGemmBN;
    if (m0 >= Mt || n0 >= Nt) return;                    // (block-uniform: the odd parity has one output less)
    // a wave owns 16 rows x (16 NT) columns: 4 x 2 waves on a 64-row tile, 2 x 4 on a 32-row tile
    const int wm = (BM == 64 ? (wave >> 1) : (wave >> 2)) * 16, wn = (BM == 64 ? (wave & 1) * 32 : (wave & 3) * 16);
    // staging roles: A tile BM x BK -> thread (row tid / (BK/2) + (1024/BK) pass, 2 consecutive k at 2 (tid % (BK/2)));
    // B tile BK x 64 -> thread (k = (tid >> 5) + 16 pass, 2 consecutive n)
    const int am = tid / (BK / 2), ak = (tid % (BK / 2)) * 2;
    const int bk = tid >> 5, bn = (tid & 31) * 2;
    constexpr int kRowsPerPass = 1024 / BK;
    const double sgn = p ? -1.0 : 1.0;
    // vector loads when every 16-byte pair is aligned and the tile lies inside the matrices (block-uniform)
    const bool fast = ((ldx | ldE | K) & 1) == 0 && m0 + kGemmBM <= Mt && n0 + kGemmBN <= Nt &&
                      ((reinterpret_cast<unsigned long long>(X) | reinterpret_cast<unsigned long long>(E)) & 15ull) == 0;
    auto fold = [&](double a, double b, int x) { return (x == mid) ? (p ? 0.0 : a) : a + sgn * b; };
    struct Group { d2 a[HA], b[HB]; };                   // one K step of this thread
    auto fetch = [&](int k0, Group &g) {
        const bool vec = fast && k0 + kGemmBK <= Kf;      // (K even here: no middle sample)
#pragma unroll
        for (int h = 0; h < HA; h++) {                    // A tile
            const int gm = m0 + am + kRowsPerPass * h, ka0 = k0 + ak;
            if (vec) {
                if (FORM == 0) {
                    const double *row = X + (long long)gm * ldx;
                    const d2 u = *reinterpret_cast<const d2 *>(row + ka0), v = *reinterpret_cast<const d2 *>(row + K - 2 - ka0);   // v = (X[K-2-x], X[K-1-x])
                    g.a[h] = d2{u.x + sgn * v.y, u.y + sgn * v.x};
                } else g.a[h] = *reinterpret_cast<const d2 *>(E + (long long)gm * ldE + ka0);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int ka = ka0 + j;
                if (FORM == 0) g.a[h][j] = (gm < Mt && ka < Kf) ? fold(X[(long long)gm * ldx + ka], X[(long long)gm * ldx + (K - 1 - ka)], ka) : 0.0;
                else g.a[h][j] = (gm < Mt && ka < Kf) ? E[(long long)gm * ldE + ka] : 0.0;
            }
        }
#pragma unroll
        for (int h = 0; h < HB; h++) {                    // B tile
            const int kb = k0 + bk + 16 * h;
            if (vec) {
                if (FORM == 0) g.b[h] = *reinterpret_cast<const d2 *>(E + (long long)kb * ldE + n0 + bn);
                else {
                    const d2 u = *reinterpret_cast<const d2 *>(X + (long long)kb * ldx + n0 + bn), v = *reinterpret_cast<const d2 *>(X + (long long)(K - 1 - kb) * ldx + n0 + bn);
                    g.b[h] = d2{u.x + sgn * v.x, u.y + sgn * v.y};
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int gn = n0 + bn + j;
                if (FORM == 0) g.b[h][j] = (kb < Kf && gn < Nt) ? E[(long long)kb * ldE + gn] : 0.0;
                else g.b[h][j] = (kb < Kf && gn < Nt) ? fold(X[(long long)kb * ldx + gn], X[(long long)(K - 1 - kb) * ldx + gn], kb) : 0.0;
            }
        }
    };
    auto stage = [&](int buf, const Group &g) {
#pragma unroll
        for (int h = 0; h < HA; h++) {
            double *q = &sA[buf][(am + kRowsPerPass * h) * kGemmLdA + ak];      // (rows are only 8-byte aligned)
            q[0] = g.a[h].x; q[1] = g.a[h].y;
        }
#pragma unroll
        for (int h = 0; h < HB; h++) *reinterpret_cast<d2 *>(&sB[buf][(bk + 16 * h) * kGemmLdB + bn]) = g.b[h];
    };
    d4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
    const int steps = (Kf + kGemmBK - 1) / kGemmBK;
    auto compute = [&](int buf) {
        const double *pa = sA[buf] + (wm + (lane & 15)) * kGemmLdA + (lane >> 4);
        const double *pb = sB[buf] + (lane >> 4) * kGemmLdB + wn + (lane & 15);
        double a[kGemmBK / 4], bb[NT][kGemmBK / 4];
#pragma unroll
        for (int ks = 0; ks < kGemmBK / 4; ks++) {                 // all operand reads of the step first: one LDS latency, not eight
            a[ks] = pa[ks * 4];
#pragma unroll
            for (int j = 0; j < NT; j++) bb[j][ks] = pb[ks * 4 * kGemmLdB + 16 * j];
        }
#pragma unroll
        for (int ks = 0; ks < kGemmBK / 4; ks++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], bb[j][ks], acc[j], 0, 0, 0);
    };
    Group g0, g1;
    if (BK == 32) {
        // step s computes from LDS buffer s & 1, which was filled from register group s & 1; group s & 1 is refilled with step
        // s + 2 as soon as it has been staged, i.e. two whole steps before it is needed again
        fetch(0, g0);
        if (steps > 1) fetch(kGemmBK, g1);
        stage(0, g0);
        if (steps > 2) fetch(2 * kGemmBK, g0);
        __syncthreads();
        for (int s = 0; s < steps; s += 2) {
            compute(0);
            if (s + 1 < steps) { stage(1, g1); if (s + 3 < steps) fetch((s + 3) * kGemmBK, g1); }
            __syncthreads();
            if (s + 1 >= steps) break;
            compute(1);
            if (s + 2 < steps) { stage(0, g0); if (s + 4 < steps) fetch((s + 4) * kGemmBK, g0); }
            __syncthreads();
        }
    } else {
        fetch(0, g0);
        stage(0, g0);
        __syncthreads();
        for (int s = 0; s < steps; s++) {
            if (s + 1 < steps) fetch((s + 1) * kGemmBK, g0);          // in flight during the MFMAs below
            compute(s & 1);
            if (s + 1 < steps) stage((s & 1) ^ 1, g0);
            __syncthreads();
        }
    }
    double dc = 0.0;
    if (EPI == 1 && p == 0 && m0 == 0 && n0 == 0) {                // block-uniform branch: this block holds element (0,0)
        // (red[] has one slot per wave of a 256-thread block: the first four waves reduce, all eight pass the barriers)
        double v = 0;
        if (tid < kBlock) for (int i = tid; i < e.dc_nb; i += kBlock) v += e.dc_partials[(size_t)ch * e.dc_nb + i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && wave < kBlock / 64) red[wave] = v;
        __syncthreads();
        dc = red[0];
#pragma unroll
        for (int k = 1; k < kBlock / 64; k++) dc += red[k];
    }
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int tr = m0 + wm + (lane >> 4) + 4 * r, tc = n0 + wn + 16 * j + (lane & 15);       // parity-compact indices
            if (tr >= Mt || tc >= Nt) continue;
            const int row = FORM == 0 ? tr : 2 * tr + p, col = FORM == 0 ? 2 * tc + p : tc;
            double v = acc[j][r];
            if (EPI == 1) {
                const float resp = (float)(e.lap_y[row] + e.lap_x[col]);          // `float ftLapResponse` in the reference (:233)
                v = v / (e.alpha - resp);
                if (row == 0 && col == 0) v = dc;
            }
            if (EPI == 2) e.out[((size_t)row * e.out_w + col) * 3 + ch] = v / e.denom;
            else C[(long long)row * ldc + col] = v;
        }
}

} // namespace gp

namespace gdpt {

namespace {

void ck(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

struct Workspace {
    size_t n3 = 0;
    double *x = nullptr, *r = nullptr, *q = nullptr, *p0 = nullptr, *p1 = nullptr, *partials = nullptr, *rel = nullptr;
    gp::CgState *state = nullptr;
    gp::CgState *h_state = nullptr;   // pinned
    double *h_rel = nullptr;          // pinned
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // timing pair + chunk marker, created once
    void release() {
        for (double *p : {x, r, q, p0, p1, partials, rel}) if (p) hipFree(p);
        if (state) hipFree(state);
        if (h_state) hipHostFree(h_state);
        if (h_rel) hipHostFree(h_rel);
        for (auto &e : ev) if (e) hipEventDestroy(e);
        *this = Workspace();
    }
    void ensure(size_t n) {
        if (n <= n3) return;
        release();
        n3 = n;
        for (double **p : {&x, &r, &q, &p0, &p1}) ck(hipMalloc((void **)p, n * sizeof(double)), "hipMalloc(poisson workspace)");
        ck(hipMalloc((void **)&partials, 10 * gp::kMaxBlocks * sizeof(double)), "hipMalloc(partials)");
        ck(hipMalloc((void **)&rel, sizeof(double)), "hipMalloc(rel)");
        ck(hipMalloc((void **)&state, sizeof(gp::CgState)), "hipMalloc(state)");
        ck(hipHostMalloc((void **)&h_state, sizeof(gp::CgState)), "hipHostMalloc");
        ck(hipHostMalloc((void **)&h_rel, sizeof(double)), "hipHostMalloc");
        for (auto &e : ev) ck(hipEventCreate(&e), "hipEventCreate");
    }
};

} // namespace


// ---- direct DCT-I solver (GEMM) ---------------------------------------------------------------------
namespace {

struct DctPlan {
    int n = 0;
    double *d_mat = nullptr;   // C[j][k] = w_j cos(pi j k/(n-1)), row-major n x n
    double *d_lap = nullptr;   // 2 cos(pi i/(n-1)) (the caller adds -4 on the y axis)
    // parity tables of the folded products (dct_fold_gemm_f64): E_p[x][q] = w_x cos(pi x (2q+p)/(n-1)), x < ceil(n/2),
    // and their transposes; row strides padded to an even number of doubles
    double *d_e[2] = {nullptr, nullptr}, *d_et[2] = {nullptr, nullptr};
    int ld_e = 0, ld_et = 0;
};
// Transform matrices and eigenvalue tables depend on the extent only: one set per device, shared (read-only) by every
// stream. Scratch buffers, the rocBLAS handle and the timing events belong to one (device, stream) pair, so solves on
// different streams or devices (one host thread per GPU, gdpt_gradient_path_render_multi) never share mutable state.
struct DctTables {
    std::mutex mu;
    std::vector<std::unique_ptr<DctPlan>> plans;
    std::vector<std::pair<int, double *>> lap_y;
    void release() {
        for (auto &p : plans) {
            if (p->d_mat) hipFree(p->d_mat);
            if (p->d_lap) hipFree(p->d_lap);
            for (int k = 0; k < 2; k++) { if (p->d_e[k]) hipFree(p->d_e[k]); if (p->d_et[k]) hipFree(p->d_et[k]); }
        }
        for (auto &l : lap_y) if (l.second) hipFree(l.second);
        plans.clear(); lap_y.clear();
    }
};
struct DctWorkspace {
    rocblas_handle handle = nullptr;
    size_t elems = 0;
    double *buf[2] = {nullptr, nullptr};
    double *partials = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    void release() {
        for (auto &b : buf) { if (b) hipFree(b); b = nullptr; }
        if (partials) hipFree(partials);
        if (handle) rocblas_destroy_handle(handle);
        for (auto &e : ev) if (e) hipEventDestroy(e);
        *this = DctWorkspace();
    }
};
struct StreamState {                  // everything a solve on one (device, stream) mutates
    std::mutex mu;                    // held while a solve is being enqueued on this pair
    Workspace cg;
    DctWorkspace dct;
};
std::mutex g_registry_mu;
std::map<std::pair<int, hipStream_t>, std::unique_ptr<StreamState>> g_streams;
std::map<int, std::unique_ptr<DctTables>> g_tables;

StreamState &stream_state(int dev, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_registry_mu);
    auto &slot = g_streams[{dev, stream}];
    if (!slot) slot.reset(new StreamState());
    return *slot;
}
DctTables &device_tables(int dev) {
    std::lock_guard<std::mutex> lk(g_registry_mu);
    auto &slot = g_tables[dev];
    if (!slot) slot.reset(new DctTables());
    return *slot;
}

void rb(rocblas_status st, const char *what) {
    if (st != rocblas_status_success) throw std::runtime_error(std::string(what) + ": rocBLAS status " + std::to_string((int)st));
}

// host tables, with the argument reduced exactly: cos(pi * ((j*k) mod 2(n-1)) / (n-1))
const DctPlan &get_plan(DctTables &t, int n) {
    for (auto &p : t.plans) if (p->n == n) return *p;
    std::unique_ptr<DctPlan> p(new DctPlan());
    std::vector<double> m((size_t)n * n), lap(n), ctab(2 * (size_t)(n - 1));
    for (size_t i = 0; i < ctab.size(); i++) ctab[i] = std::cos(M_PI * (double)i / (double)(n - 1));
    for (int j = 0; j < n; j++) {
        double wj = (j > 0 && j < n - 1) ? 2.0 : 1.0;
        for (int k = 0; k < n; k++) m[(size_t)j * n + k] = wj * ctab[((size_t)j * k) % ctab.size()];
    }
    for (int i = 0; i < n; i++) lap[i] = 2.0 * std::cos(M_PI * i / (n - 1));     // ftLapX (:184-186)
    ck(hipMalloc((void **)&p->d_mat, m.size() * sizeof(double)), "hipMalloc(dct matrix)");
    ck(hipMalloc((void **)&p->d_lap, lap.size() * sizeof(double)), "hipMalloc(dct lap)");
    ck(hipMemcpy(p->d_mat, m.data(), m.size() * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(dct matrix)");
    ck(hipMemcpy(p->d_lap, lap.data(), lap.size() * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(dct lap)");
    {
        const int kf = (n + 1) / 2;
        p->ld_e = (kf + 1) & ~1; p->ld_et = (kf + 1) & ~1;       // (both parities share the wider stride)
        for (int par = 0; par < 2; par++) {
            const int qn = (n + 1 - par) / 2;
            std::vector<double> e((size_t)kf * p->ld_e, 0.0), et((size_t)std::max(qn, 1) * p->ld_et, 0.0);
            for (int x = 0; x < kf; x++) {
                const double wx = (x > 0 && x < n - 1) ? 2.0 : 1.0;
                for (int q = 0; q < qn; q++) {
                    const double v = wx * ctab[((size_t)x * (size_t)(2 * q + par)) % ctab.size()];
                    e[(size_t)x * p->ld_e + q] = v; et[(size_t)q * p->ld_et + x] = v;
                }
            }
            ck(hipMalloc((void **)&p->d_e[par], e.size() * sizeof(double)), "hipMalloc(dct parity table)");
            ck(hipMalloc((void **)&p->d_et[par], et.size() * sizeof(double)), "hipMalloc(dct parity table^T)");
            ck(hipMemcpy(p->d_e[par], e.data(), e.size() * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(dct parity table)");
            ck(hipMemcpy(p->d_et[par], et.data(), et.size() * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(dct parity table^T)");
        }
    }
    p->n = n;
    t.plans.push_back(std::move(p));
    return *t.plans.back();
}
const double *get_lap_y(DctTables &t, int h) {   // ftLapY = -4 + 2 cos(pi y/(h-1)) (:187-189)
    for (auto &l : t.lap_y) if (l.first == h) return l.second;
    std::vector<double> ly(h);
    for (int y = 0; y < h; y++) ly[y] = -4.0 + (2.0 * std::cos(M_PI * y / (h - 1)));
    double *d = nullptr;
    ck(hipMalloc((void **)&d, h * sizeof(double)), "hipMalloc(lap_y)");
    ck(hipMemcpy(d, ly.data(), h * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(lap_y)");
    t.lap_y.push_back({h, d});
    return d;
}

// Enqueue-only unless `timed`: no event is created, recorded or waited for on the product path (stats == NULL).
// `raw` (nullable): the five rendered buffers — then d_c / d_gx / d_gy are OUTPUTS of the fused first pass (assemble_rhs_kernel).
PoissonResult poisson_dct(int dev, DctWorkspace &ws, int w, int h, const double *d_c, const double *d_gx, const double *d_gy, double alpha,
                          double *d_out, hipStream_t stream, bool timed, bool library_gemm, const double *const *raw = nullptr) {
    if (library_gemm) {
        if (!ws.handle) rb(rocblas_create_handle(&ws.handle), "rocblas_create_handle");
        rb(rocblas_set_stream(ws.handle, stream), "rocblas_set_stream");
    }
    const double *Cw, *Ch, *lap_x, *lap_y, *Ew[2], *EhT[2];
    int ld_ew, ld_eht;
    {
        DctTables &t = device_tables(dev);
        std::lock_guard<std::mutex> lk(t.mu);
        const DctPlan &pw = get_plan(t, w);
        Cw = pw.d_mat; lap_x = pw.d_lap; Ew[0] = pw.d_e[0]; Ew[1] = pw.d_e[1]; ld_ew = pw.ld_e;
        const DctPlan &ph = get_plan(t, h);
        Ch = ph.d_mat; EhT[0] = ph.d_et[0]; EhT[1] = ph.d_et[1]; ld_eht = ph.ld_et;
        lap_y = get_lap_y(t, h);
    }
    gp::Geo g{w, h, w * h * 3, w * 3};
    const size_t plane = (size_t)w * h;
    if (ws.elems < 3 * plane) {
        if (ws.buf[0]) ck(hipStreamSynchronize(stream), "hipStreamSynchronize");   // earlier solves may still use the old buffers
        for (auto &b : ws.buf) { if (b) hipFree(b); b = nullptr; }
        for (auto &b : ws.buf) ck(hipMalloc((void **)&b, 3 * plane * sizeof(double)), "hipMalloc(dct buffers)");
        ws.elems = 3 * plane;
    }
    if (!ws.partials) ck(hipMalloc((void **)&ws.partials, 3 * gp::kMaxBlocks * sizeof(double)), "hipMalloc(dct partials)");
    const int nb = std::min(gp::kMaxBlocks, (g.n3 + gp::kBlock - 1) / gp::kBlock);
    if (timed) {
        for (auto &e : ws.ev) if (!e) ck(hipEventCreate(&e), "hipEventCreate");
        ck(hipEventRecord(ws.ev[0], stream), "hipEventRecord");
    }
    double *A = ws.buf[0], *B = ws.buf[1];
    if (raw) hipLaunchKernelGGL(gp::assemble_rhs_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, alpha, raw[0], raw[1], raw[2], raw[3], raw[4],
                                const_cast<double *>(d_c), const_cast<double *>(d_gx), const_cast<double *>(d_gy), A, ws.partials);
    else hipLaunchKernelGGL(gp::dct_rhs_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, alpha, d_c, d_gx, d_gy, A, ws.partials);
    const double one = 1.0, zero = 0.0;
    // Row-major X (h x w) is the column-major matrix X^T (w x h, ld = w). Row transform T = X * Cw  <=>  T^T = Cw^T * X^T:
    // the row-major buffer of Cw read column-major IS Cw^T, so (N, N). Column transform Y = Ch^T * T  <=>  Y^T = T^T * Ch:
    // the buffer of Ch read column-major is Ch^T, hence op = T on it.
    auto transform = [&](const double *src, double *tmp, double *dst) {
        rb(rocblas_dgemm_strided_batched(ws.handle, rocblas_operation_none, rocblas_operation_none, w, h, w, &one,
                                         Cw, w, 0, src, w, (rocblas_stride)plane, &zero, tmp, w, (rocblas_stride)plane, 3), "dgemm(rows)");
        rb(rocblas_dgemm_strided_batched(ws.handle, rocblas_operation_none, rocblas_operation_transpose, w, h, h, &one,
                                         tmp, w, (rocblas_stride)plane, Ch, h, 0, &zero, dst, w, (rocblas_stride)plane, 3), "dgemm(cols)");
    };
    if (library_gemm) {                                      // GDPT_SOLVER_DCT: the two 1-D passes as rocBLAS dgemm_strided_batched
        transform(A, B, A);                                  // A = DCT2D(h)
        hipLaunchKernelGGL(gp::dct_scale_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, alpha, lap_x, lap_y, A);
        hipLaunchKernelGGL(gp::dct_dc_kernel, dim3(1), dim3(gp::kBlock), 0, stream, g, A, ws.partials, nb);
        transform(A, B, A);                                  // A = DCT2D(F^)
        hipLaunchKernelGGL(gp::dct_finalize_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, A, d_out);
    } else {
        // GDPT_SOLVER_DCT_MFMA: own folded fp64 MFMA GEMMs, per (channel, output parity) plane:  T = X * Cw  (rows),
        // Y = Ch^T * T  (columns); spectral division + DC override ride on the second product, the final scaling +
        // interleaving on the fourth
        gp::GemmEpi e{};
        e.alpha = alpha; e.lap_x = lap_x; e.lap_y = lap_y; e.dc_partials = ws.partials; e.dc_nb = nb;
        e.denom = 4.0 * (double)(w - 1) * (double)(h - 1); e.out = d_out; e.out_w = w;
        const int qw = (w + 1) / 2, qh = (h + 1) / 2;           // outputs of the even parity along the folded dimension
        const dim3 block(gp::kGemmThreads);
        const long long pl = (long long)plane;
        int num_cus = 256;
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) num_cus = prop.multiProcessorCount; }
        // The shape follows the grid (dct_fold_gemm_f64): with 64-row tiles, at most one block per CU -> 32-row tiles (twice the
        // blocks, two of them resident per CU) with deep prefetch; at most two per CU -> 64 rows, deep prefetch; more -> 64 rows, the
        // small-footprint shape. Test knobs dct_bk / dct_bm force a shape.
        const long long blocks64 = (long long)((h + 63) / 64) * ((qw + gp::kGemmBN - 1) / gp::kGemmBN) * 6;
        int bm = blocks64 <= num_cus ? 32 : 64, bk = blocks64 <= 2LL * num_cus ? 32 : 16;
        { const int f = gdpt::debug_knob_int("dct_bk", 0); if (f == 16 || f == 32) bk = f; }
        { const int f = gdpt::debug_knob_int("dct_bm", 0); if (f == 32 || f == 64) bm = f; }
        if (bm == 32) bk = 32;                                   // (the 32-row tile exists in the deep-prefetch shape only)
        // tile counts of the parity-compact index spaces (even parity: the larger one), 1-D XCD-aware grids
        const int tm_rows = (h + bm - 1) / bm, tn_rows = (qw + gp::kGemmBN - 1) / gp::kGemmBN;     // row pass: M = h, N' = ceil(w/2)
        const int tm_cols = (qh + bm - 1) / bm, tn_cols = (w + gp::kGemmBN - 1) / gp::kGemmBN;     // column pass: M' = ceil(h/2), N = w
        const dim3 grid_rows((unsigned)(8 * ((tm_rows * 3 + 7) / 8) * (tn_rows * 2)));
        const dim3 grid_cols((unsigned)(8 * ((tn_cols * 3 + 7) / 8) * (tm_cols * 2)));
#define GDPT_DCT_CHAIN(BK_, BM_)                                                                                                                              \
        hipLaunchKernelGGL((gp::dct_fold_gemm_f64<0, 0, BK_, BM_>), grid_rows, block, 0, stream, h, w, w, tm_rows, tn_rows, (const double *)A, w, pl, Ew[0], Ew[1], ld_ew, B, w, pl, e);      /* B = A * Cw */ \
        hipLaunchKernelGGL((gp::dct_fold_gemm_f64<1, 1, BK_, BM_>), grid_cols, block, 0, stream, h, w, h, tm_cols, tn_cols, (const double *)B, w, pl, EhT[0], EhT[1], ld_eht, A, w, pl, e);   /* A = Ch^T * B, / (alpha - lambda), DC */ \
        hipLaunchKernelGGL((gp::dct_fold_gemm_f64<0, 0, BK_, BM_>), grid_rows, block, 0, stream, h, w, w, tm_rows, tn_rows, (const double *)A, w, pl, Ew[0], Ew[1], ld_ew, B, w, pl, e);      /* B = A * Cw */ \
        hipLaunchKernelGGL((gp::dct_fold_gemm_f64<1, 2, BK_, BM_>), grid_cols, block, 0, stream, h, w, h, tm_cols, tn_cols, (const double *)B, w, pl, EhT[0], EhT[1], ld_eht, A, w, pl, e);   /* out = Ch^T * B / denom */
        if (bm == 32) { GDPT_DCT_CHAIN(32, 32) }
        else if (bk == 32) { GDPT_DCT_CHAIN(32, 64) }
        else { GDPT_DCT_CHAIN(16, 64) }
#undef GDPT_DCT_CHAIN
    }
    ck(hipGetLastError(), "dct kernel launch");
    float ms = 0;
    if (timed) {
        ck(hipEventRecord(ws.ev[1], stream), "hipEventRecord");
        ck(hipEventSynchronize(ws.ev[1]), "hipEventSynchronize");
        ck(hipEventElapsedTime(&ms, ws.ev[0], ws.ev[1]), "hipEventElapsedTime");
    }
    PoissonResult res;
    res.iterations = 0; res.solver = GDPT_SOLVER_DCT; res.rel_residual = 0.0; res.solve_ms = ms;
    return res;
}

} // namespace

// Drops what the solver keeps for one (device, stream) pair — scratch buffers, rocBLAS handle, timing events. Owners of a
// stream call it before destroying the stream: the registry is keyed by the handle's value, and a later stream that reuses
// that value must not inherit another stream's state. The caller guarantees no solve is in flight on the pair.
void poisson_forget_stream(int dev, hipStream_t stream) {
    std::unique_ptr<StreamState> gone;
    {
        std::lock_guard<std::mutex> lk(g_registry_mu);
        auto it = g_streams.find({dev, stream});
        if (it == g_streams.end()) return;
        gone = std::move(it->second);
        g_streams.erase(it);
    }
    int cur = 0;
    hipGetDevice(&cur);
    hipSetDevice(dev);
    gone->cg.release(); gone->dct.release();
    hipSetDevice(cur);
}

void poisson_release_workspace() {
    std::lock_guard<std::mutex> lk(g_registry_mu);
    int cur = 0;
    hipGetDevice(&cur);
    for (auto &kv : g_streams) { hipSetDevice(kv.first.first); kv.second->cg.release(); kv.second->dct.release(); }
    for (auto &kv : g_tables) { hipSetDevice(kv.first); kv.second->release(); }
    g_streams.clear(); g_tables.clear();
    hipSetDevice(cur);
}

void launch_assemble(int w, int h, int row_begin, int row_end, const double *img, const double *cx0, const double *cy0, const double *cx1,
                     const double *cy1, double *c, double *cx, double *cy, hipStream_t stream) {
    if (w <= 0 || h <= 0) throw std::runtime_error("assemble: empty image");
    if (row_begin == 0 && row_end == 0) row_end = h;
    if (row_begin < 0 || row_end > h || row_begin >= row_end) throw std::runtime_error("assemble: bad row band");
    gp::Geo g{w, h, w * h * 3, w * 3};
    const int n = (row_end - row_begin) * g.row;
    int nb = std::min(gp::kMaxBlocks * 2, (n + gp::kBlock - 1) / gp::kBlock);
    hipLaunchKernelGGL(gp::assemble_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, row_begin, row_end, img, cx0, cy0, cx1, cy1, c, cx, cy);
    ck(hipGetLastError(), "assemble kernel launch");
}

PoissonResult assemble_solve_device(int w, int h, const double *img, const double *cx0, const double *cy0, const double *cx1, const double *cy1,
                                    double *d_c, double *d_cx, double *d_cy, double alpha, double *d_out, int solver, double tol, int max_iters,
                                    hipStream_t stream, bool timed) {
    if (solver == GDPT_SOLVER_CG) {      // (the CG reads c / cx / cy several times: nothing to fuse)
        launch_assemble(w, h, 0, 0, img, cx0, cy0, cx1, cy1, d_c, d_cx, d_cy, stream);
        return poisson_solve_device(w, h, d_c, d_cx, d_cy, alpha, d_out, solver, tol, max_iters, stream, timed);
    }
    if (w < 2 || h < 2) throw std::runtime_error("poisson: width and height must be >= 2 (the reference divides by (W-1)(H-1))");
    if (!(alpha > 0)) throw std::runtime_error("poisson: dataCost must be > 0");
    if (solver != GDPT_SOLVER_DCT && solver != GDPT_SOLVER_DCT_MFMA) throw std::runtime_error("poisson: unknown solver");
    int dev = 0;
    ck(hipGetDevice(&dev), "hipGetDevice");
    StreamState &ss = stream_state(dev, stream);
    std::lock_guard<std::mutex> lk(ss.mu);
    const double *raw[5] = {img, cx0, cy0, cx1, cy1};
    PoissonResult r = poisson_dct(dev, ss.dct, w, h, d_c, d_cx, d_cy, alpha, d_out, stream, timed, solver == GDPT_SOLVER_DCT, raw);
    r.solver = solver;
    return r;
}

PoissonResult poisson_solve_device(int w, int h, const double *d_c, const double *d_gx, const double *d_gy, double alpha,
                                   double *d_out, int solver, double tol, int max_iters, hipStream_t stream, bool timed) {
    if (w < 2 || h < 2) throw std::runtime_error("poisson: width and height must be >= 2 (the reference divides by (W-1)(H-1))");
    if (!(alpha > 0)) throw std::runtime_error("poisson: dataCost must be > 0");
    if (solver != GDPT_SOLVER_CG && solver != GDPT_SOLVER_DCT && solver != GDPT_SOLVER_DCT_MFMA) throw std::runtime_error("poisson: unknown solver");
    if (tol <= 0) tol = 1e-10;
    if (max_iters <= 0) max_iters = 2000;
    int dev = 0;
    ck(hipGetDevice(&dev), "hipGetDevice");
    StreamState &ss = stream_state(dev, stream);
    std::lock_guard<std::mutex> lk(ss.mu);
    if (solver != GDPT_SOLVER_CG) {
        PoissonResult r = poisson_dct(dev, ss.dct, w, h, d_c, d_gx, d_gy, alpha, d_out, stream, timed, solver == GDPT_SOLVER_DCT);
        r.solver = solver;
        return r;
    }
    gp::Geo g{w, h, w * h * 3, w * 3};
    if ((size_t)g.n3 > ss.cg.n3 && ss.cg.n3) ck(hipStreamSynchronize(stream), "hipStreamSynchronize");
    ss.cg.ensure((size_t)g.n3);
    Workspace &ws = ss.cg;
    const int nb = std::min(gp::kMaxBlocks, (g.n3 + gp::kBlock - 1) / gp::kBlock);
    // init layout: 8 slots of nb doubles (slot 0 = <r,r>, reused as part_rr); <p,q> partials live in slot 8
    double *part_rr = ws.partials, *part_pq = ws.partials + 8 * (size_t)nb;
    hipEvent_t e0 = ws.ev[0], e1 = ws.ev[1];
    ck(hipEventRecord(e0, stream), "hipEventRecord");
    hipLaunchKernelGGL(gp::cg_init_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, alpha, d_c, d_gx, d_gy, ws.x, ws.r, ws.p0, ws.p1, ws.partials);
    hipLaunchKernelGGL(gp::cg_init_reduce_kernel, dim3(1), dim3(gp::kBlock), 0, stream, nb, ws.partials, ws.state, tol);
    ck(hipGetLastError(), "cg init launch");
    const int chunk = 32;          // even, so buffer parity is chunk-invariant
    int launched = 0;
    bool done = false;
    // one chunk is always in flight ahead of the status check of the previous one (no pipeline bubble)
    auto enqueue_chunk = [&]() {
        for (int k = 0; k < chunk; k++) {
            int it = launched + k;
            const double *pin = (it & 1) ? ws.p1 : ws.p0;
            double *pout = (it & 1) ? ws.p0 : ws.p1;
            hipLaunchKernelGGL(gp::cg_step_a, dim3(nb), dim3(gp::kBlock), 0, stream, g, alpha, it, ws.r, pin, pout, ws.q, part_rr, part_pq, ws.state);
            hipLaunchKernelGGL(gp::cg_step_b, dim3(nb), dim3(gp::kBlock), 0, stream, g, it, pout, ws.q, ws.x, ws.r, part_pq, part_rr, ws.state);
        }
        launched += chunk;
        ck(hipGetLastError(), "cg chunk launch");
    };
    hipEvent_t chunk_ev = ws.ev[2];
    enqueue_chunk();
    while (!done) {
        ck(hipMemcpyAsync(ws.h_state, ws.state, sizeof(gp::CgState), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(state)");
        ck(hipEventRecord(chunk_ev, stream), "hipEventRecord");
        bool more = launched < max_iters;
        if (more) enqueue_chunk();               // speculative next chunk; its kernels exit at once if converged
        ck(hipEventSynchronize(chunk_ev), "hipEventSynchronize");
        if (ws.h_state->converged || !more) done = true;
    }
    hipLaunchKernelGGL(gp::cg_finalize_kernel, dim3(nb), dim3(gp::kBlock), 0, stream, g, alpha, ws.x, d_out, part_rr, nb, ws.state, ws.rel);
    ck(hipGetLastError(), "cg finalize launch");
    ck(hipMemcpyAsync(ws.h_state, ws.state, sizeof(gp::CgState), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(state)");
    ck(hipMemcpyAsync(ws.h_rel, ws.rel, sizeof(double), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(rel)");
    ck(hipEventRecord(e1, stream), "hipEventRecord");
    ck(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0;
    ck(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
    PoissonResult res;
    res.iterations = ws.h_state->iters; res.solver = GDPT_SOLVER_CG; res.rel_residual = *ws.h_rel; res.solve_ms = ms;
    return res;
}

} // namespace gdpt
