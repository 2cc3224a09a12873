/* ORACLE (test infrastructure, not product): plain-C direct convolution in float64.
 *
 * An arithmetic ground truth that does not depend on torch's convolution kernels: tests use it to
 * confirm that oracle/wav2lip_ref.py (torch fp32) and the HIP path agree with the mathematical
 * definition of the layers in wav2lip/models/conv.py:5-44:
 *     y = act( BN_eval( conv(x, w) + b ) [+ x] )
 * Conv2d:           y[n][co][oy][ox] = sum_{ci,ky,kx} x[n][ci][oy*sh+ky-ph][ox*sw+kx-pw] * w[co][ci][ky][kx]
 * ConvTranspose2d:  y[n][co][iy*s+ky-p][ix*s+kx-p] += x[n][ci][iy][ix] * w[ci][co][ky][kx]
 * Layouts are NCHW, row-major, like the torch tensors the reference holds.  Only tests/ and
 * bench.py's cpu_baseline leg may load the shared object built from this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void ref_conv2d(const float* x, const float* w, const float* b, double* y, int N, int Cin, int H, int W, int Cout,
                int KH, int KW, int sh, int sw, int ph, int pw) {
    const int OH = (H + 2 * ph - KH) / sh + 1, OW = (W + 2 * pw - KW) / sw + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < OH; ++oy)
                for (int ox = 0; ox < OW; ++ox) {
                    double acc = b ? (double)b[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < KH; ++ky) {
                            const int iy = oy * sh + ky - ph;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < KW; ++kx) {
                                const int ix = ox * sw + kx - pw;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)x[((size_t)(n * Cin + ci) * H + iy) * W + ix] *
                                       (double)w[((size_t)(co * Cin + ci) * KH + ky) * KW + kx];
                            }
                        }
                    y[((size_t)(n * Cout + co) * OH + oy) * OW + ox] = acc;
                }
}

void ref_conv_transpose2d(const float* x, const float* w, const float* b, double* y, int N, int Cin, int H, int W,
                          int Cout, int K, int s, int p, int op) {
    const int OH = (H - 1) * s - 2 * p + K + op, OW = (W - 1) * s - 2 * p + K + op;
    for (size_t i = 0; i < (size_t)N * Cout * OH * OW; ++i) y[i] = 0.0;
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Cin; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    const double xv = x[((size_t)(n * Cin + ci) * H + iy) * W + ix];
                    for (int co = 0; co < Cout; ++co)
                        for (int ky = 0; ky < K; ++ky) {
                            const int oy = iy * s + ky - p;
                            if (oy < 0 || oy >= OH) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                const int ox = ix * s + kx - p;
                                if (ox < 0 || ox >= OW) continue;
                                y[((size_t)(n * Cout + co) * OH + oy) * OW + ox] +=
                                    xv * (double)w[((size_t)(ci * Cout + co) * K + ky) * K + kx];
                            }
                        }
                }
    if (b)
        for (int n = 0; n < N; ++n)
            for (int co = 0; co < Cout; ++co)
                for (int i = 0; i < OH * OW; ++i) y[(size_t)(n * Cout + co) * OH * OW + i] += (double)b[co];
}

/* y (float64, in place): eval BatchNorm (eps), optional residual x_res (fp32, same shape), ReLU. */
void ref_bn_res_relu(double* y, const float* g, const float* beta, const float* mean, const float* var, double eps,
                     const float* x_res, int N, int C, int HW) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < HW; ++i) {
                const size_t o = (size_t)(n * C + c) * HW + i;
                double v = (y[o] - (double)mean[c]) / sqrt((double)var[c] + eps) * (double)g[c] + (double)beta[c];
                if (x_res) v += (double)x_res[o];
                y[o] = v > 0.0 ? v : 0.0;
            }
}
