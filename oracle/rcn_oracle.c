/*
 * rcn_oracle.c -- loop-faithful f64 restatement of rcn's CPU path (see rcn_oracle.h).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 * Parity status: pinned by the reference KATs kernel.rs:402-417 and :436-441 only;
 * the rest is "parity unpinned" (see header / DESIGN.md).
 *
 * Loop orders, index quirks and floating-point expression shapes follow the Rust source
 * statement by statement so that f64 results agree to the last bit wherever the
 * reference itself is deterministic (everything except the rayon sum order in
 * train_batch, rcn.rs:190-205).
 */
#include "rcn_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define AT(m, rows, r, c) ((m)[(size_t)(c) * (size_t)(rows) + (size_t)(r)])

/* ------------------------------------------------------------------ kernel.rs */

/* kernel.rs:38-53 -- (3x1 column kernel, 1x3 row kernel) */
void rcn_o_sobel_separated(int op, double col3[3], double row3[3]) {
    const double n0 = 0.0, n1 = 1.0, n2 = n1 + n1, n1n = n0 - n1;
    switch (op) {
    case RCN_O_OP_TOP:    col3[0] = n1;  col3[1] = n0; col3[2] = n1n; row3[0] = n1;  row3[1] = n2; row3[2] = n1;  break;
    case RCN_O_OP_BOTTOM: col3[0] = n1n; col3[1] = n0; col3[2] = n1;  row3[0] = n1;  row3[1] = n2; row3[2] = n1;  break;
    case RCN_O_OP_LEFT:   col3[0] = n1;  col3[1] = n2; col3[2] = n1;  row3[0] = n1;  row3[1] = n0; row3[2] = n1n; break;
    default:              col3[0] = n1;  col3[1] = n2; col3[2] = n1;  row3[0] = n1n; row3[1] = n0; row3[2] = n1;  break;
    }
}

/* kernel.rs:56-59 -- the full 3x3 constants, returned column-major */
void rcn_o_sobel_full(int op, double k[9]) {
    static const double top[9]    = { 1, 2, 1,   0, 0, 0,  -1, -2, -1 };  /* row-major as written in the source */
    static const double bottom[9] = { -1, -2, -1, 0, 0, 0,  1, 2, 1 };
    static const double left[9]   = { 1, 0, -1,  2, 0, -2,  1, 0, -1 };
    static const double right[9]  = { -1, 0, 1, -2, 0, 2,  -1, 0, 1 };
    const double* s = op == RCN_O_OP_TOP ? top : op == RCN_O_OP_BOTTOM ? bottom : op == RCN_O_OP_LEFT ? left : right;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) AT(k, 3, r, c) = s[r * 3 + c];
}

int rcn_o_conv_out_shape(int R, int C, int kr, int kc, int padding, int* oR, int* oC) {
    /* kernel.rs:123-128 */
    if (kr <= 0 || kc <= 0 || kr > R || kc > C) return RCN_O_ERR_SHAPE;
    if (padding == RCN_O_PAD_SAME) {
        /* kernel.rs:131-135 */
        if (kr % 2 == 0 || kc % 2 == 0) return RCN_O_ERR_SHAPE;
        /* kernel.rs:154-158: the copy loop reads self[(cy-1,cx-1)] for cy in 1..R+pr, cx in 1..C+pc;
         * with pr >= 2 (or pc >= 2) that indexes row R (column C) -> nalgebra bounds panic. */
        const int pr = kr / 2, pc = kc / 2;
        const int cy_hi = R + pr - 1, cx_hi = C + pc - 1; /* largest cy / cx visited */
        if (cy_hi >= 1 && cx_hi >= 1 && (cy_hi - 1 >= R || cx_hi - 1 >= C)) return RCN_O_ERR_SHAPE;
        *oR = R; *oC = C;
    } else {
        *oR = R - kr + 1; *oC = C - kc + 1;
    }
    return RCN_O_OK;
}

/* kernel.rs:110-194 */
int rcn_o_convolve_2d(const double* m, int R, int C, const double* k, int kr, int kc, int padding, double* out) {
    int oR, oC;
    const int st = rcn_o_conv_out_shape(R, C, kr, kc, padding, &oR, &oC);
    if (st != RCN_O_OK) return st;
    memset(out, 0, sizeof(double) * (size_t)oR * (size_t)oC);
    if (padding == RCN_O_PAD_SAME) {
        const int pr = kr / 2, pc = kc / 2;
        const int PR = R + pr * 2, PC = C + pc * 2;
        double* P = (double*)calloc((size_t)PR * (size_t)PC, sizeof(double));
        if (!P) return RCN_O_ERR_SHAPE;
        /* kernel.rs:154-158 -- offset is hard-coded to 1 on both axes */
        for (int cy = 1; cy < R + pr; ++cy)
            for (int cx = 1; cx < C + pc; ++cx) AT(P, PR, cy, cx) = AT(m, R, cy - 1, cx - 1);
        /* kernel.rs:160-168 */
        for (int cy = 0; cy < oR; ++cy)
            for (int cx = 0; cx < oC; ++cx)
                for (int ky = 0; ky < kr; ++ky)
                    for (int kx = 0; kx < kc; ++kx)
                        AT(out, oR, cy, cx) += AT(P, PR, cy + ky, cx + kx) * AT(k, kr, ky, kx);
        free(P);
    } else {
        /* kernel.rs:182-190 */
        for (int cy = 0; cy < oR; ++cy)
            for (int cx = 0; cx < oC; ++cx)
                for (int ky = 0; ky < kr; ++ky)
                    for (int kx = 0; kx < kc; ++kx)
                        AT(out, oR, cy, cx) += AT(m, R, cy + ky, cx + kx) * AT(k, kr, ky, kx);
    }
    return RCN_O_OK;
}

/* kernel.rs:209-216 */
void rcn_o_relu(const double* m, size_t n, double* out) {
    for (size_t i = 0; i < n; ++i) out[i] = (m[i] >= 0.0) ? m[i] : 0.0;
}

/* kernel.rs:196-207: column (3x1) pass, row (1x3) pass, ReLU */
int rcn_o_convolve_2d_separated(const double* m, int R, int C, int op, int padding, double* out) {
    if (R < 3 || C < 3) return RCN_O_ERR_SHAPE; /* kernel.rs:199-201 */
    double col3[3], row3[3];
    rcn_o_sobel_separated(op, col3, row3);
    int r1, c1, r2, c2;
    int st = rcn_o_conv_out_shape(R, C, 3, 1, padding, &r1, &c1);
    if (st != RCN_O_OK) return st;
    double* t = (double*)malloc(sizeof(double) * (size_t)r1 * (size_t)c1);
    st = rcn_o_convolve_2d(m, R, C, col3, 3, 1, padding, t);
    if (st == RCN_O_OK) st = rcn_o_conv_out_shape(r1, c1, 1, 3, padding, &r2, &c2);
    if (st == RCN_O_OK) {
        double* u = (double*)malloc(sizeof(double) * (size_t)r2 * (size_t)c2);
        st = rcn_o_convolve_2d(t, r1, c1, row3, 1, 3, padding, u);
        if (st == RCN_O_OK) rcn_o_relu(u, (size_t)r2 * (size_t)c2, out);
        free(u);
    }
    free(t);
    return st;
}

int rcn_o_pool_out_shape(int R, int C, int padding, int* oR, int* oC) {
    if (R < 2 || C < 2) return RCN_O_ERR_SHAPE; /* kernel.rs:246-251 */
    if (padding == RCN_O_PAD_SAME) { *oR = (R + R % 2) / 2; *oC = (C + C % 2) / 2; }
    else                           { *oR = R / 2;           *oC = C / 2; }
    return RCN_O_OK;
}

/* kernel.rs:245-291 and __pooling_padded :298-349 */
int rcn_o_pool_2d(const double* m, int R, int C, int padding, int pooling, double* out) {
    int oR, oC;
    const int st = rcn_o_pool_out_shape(R, C, padding, &oR, &oC);
    if (st != RCN_O_OK) return st;
    if (pooling != RCN_O_POOL_MAX) return RCN_O_ERR_UNSUPPORTED; /* kernel.rs:283-285, 341-343 */
    const int rp = (padding == RCN_O_PAD_SAME) ? R % 2 : 0, cp = (padding == RCN_O_PAD_SAME) ? C % 2 : 0;
    const int PR = R + rp, PC = C + cp;
    const double* src = m;
    double* P = NULL;
    if (rp || cp) { /* kernel.rs:310-319: zero-pad bottom/right */
        P = (double*)calloc((size_t)PR * (size_t)PC, sizeof(double));
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < C; ++x) AT(P, PR, y, x) = AT(m, R, y, x);
        src = P;
    }
    for (int ry = 0; ry < oR; ++ry)
        for (int rx = 0; rx < oC; ++rx) {
            double pooler[4];
            for (int px = 0; px < 2; ++px)        /* kernel.rs:273-277 (px walks rows, py columns) */
                for (int py = 0; py < 2; ++py) pooler[py + px * 2] = AT(src, PR, ry * 2 + px, rx * 2 + py);
            double best = pooler[0];              /* max_by(partial_cmp): last maximal element */
            for (int i = 1; i < 4; ++i)
                if (!(pooler[i] < best)) best = pooler[i];
            AT(out, oR, ry, rx) = best;
        }
    free(P);
    return RCN_O_OK;
}

/* ------------------------------------------------------------------ lib.rs */

/* lib.rs:27-41: DMatrix::from_row_iterator(height, width, pixels) -> m[(y,x)] = pixel(x,y) */
void rcn_o_get_pixel_matrix(const uint8_t* px, int H, int W, double* m) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) AT(m, H, y, x) = (double)px[(size_t)y * (size_t)W + (size_t)x];
}

/* ------------------------------------------------------------------ rcn.rs: features */

static const int SEP_OPS[4] = { RCN_O_OP_TOP, RCN_O_OP_LEFT, RCN_O_OP_RIGHT, RCN_O_OP_BOTTOM }; /* rcn.rs:41-46 */

long rcn_o_feature_len(int H, int W, const rcn_o_layer* layers, int n_layers) {
    long maps = 0;
    int R = H, C = W;
    for (int i = 0; i < n_layers; ++i) {
        if (layers[i].kind == RCN_O_LAYER_CONV) {
            if (R < 3 || C < 3) return RCN_O_ERR_SHAPE;
            if (layers[i].arg == RCN_O_PAD_NONE) { R -= 2; C -= 2; }
            else if (layers[i].arg != RCN_O_PAD_SAME) return RCN_O_ERR_SHAPE;
            maps = maps ? maps * 4 : 4;
            if (R < 1 || C < 1) return RCN_O_ERR_SHAPE;
        } else if (layers[i].kind == RCN_O_LAYER_POOL) {
            if (maps == 0) continue; /* rcn.rs:343: loop over an empty feature_set is a no-op */
            if (layers[i].arg != RCN_O_POOL_MAX) return RCN_O_ERR_UNSUPPORTED;
            int oR, oC;
            if (rcn_o_pool_out_shape(R, C, RCN_O_PAD_SAME, &oR, &oC) != RCN_O_OK) return RCN_O_ERR_SHAPE;
            R = oR; C = oC;
        } else return RCN_O_ERR_SHAPE;
    }
    return maps * (long)R * (long)C;
}

/* rcn.rs:317-356 */
int rcn_o_flatten_feature_set(const double* m, int H, int W, const rcn_o_layer* layers, int n_layers, double* out) {
    const long flen = rcn_o_feature_len(H, W, layers, n_layers);
    if (flen < 0) return (int)flen;
    size_t cap = 4;
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].kind == RCN_O_LAYER_CONV) cap *= 4;
    double** fs = (double**)calloc(cap, sizeof(double*));
    size_t len = 0;
    int R = H, C = W, st = RCN_O_OK;
    for (int li = 0; li < n_layers && st == RCN_O_OK; ++li) {
        if (layers[li].kind == RCN_O_LAYER_CONV) {
            const int pad = layers[li].arg;
            int oR, oC;
            if (pad == RCN_O_PAD_SAME) { oR = R; oC = C; } else { oR = R - 2; oC = C - 2; }
            const size_t osz = (size_t)oR * (size_t)oC;
            if (len != 0) {
                const size_t curr_len = len;                       /* rcn.rs:325 */
                for (size_t i = 0; i < curr_len && st == RCN_O_OK; ++i) {
                    double* src = fs[i];
                    for (int o = 0; o < 4 && st == RCN_O_OK; ++o) { /* rcn.rs:330-336 */
                        double* dst = (double*)malloc(sizeof(double) * osz);
                        st = rcn_o_convolve_2d_separated(src, R, C, SEP_OPS[o], pad, dst);
                        if (o == 3) fs[i] = dst;                   /* last op replaces slot i ...  */
                        else fs[len++] = dst;                      /* ... the others are pushed    */
                    }
                    free(src);
                }
            } else {
                for (int o = 0; o < 4 && st == RCN_O_OK; ++o) {    /* rcn.rs:339 */
                    fs[len] = (double*)malloc(sizeof(double) * osz);
                    st = rcn_o_convolve_2d_separated(m, R, C, SEP_OPS[o], pad, fs[len]);
                    ++len;
                }
            }
            R = oR; C = oC;
        } else {
            if (len == 0) continue;
            int oR, oC;
            st = rcn_o_pool_out_shape(R, C, RCN_O_PAD_SAME, &oR, &oC);
            for (size_t i = 0; i < len && st == RCN_O_OK; ++i) {   /* rcn.rs:343-345 */
                double* dst = (double*)malloc(sizeof(double) * (size_t)oR * (size_t)oC);
                st = rcn_o_pool_2d(fs[i], R, C, RCN_O_PAD_SAME, layers[li].arg, dst);
                free(fs[i]);
                fs[i] = dst;
            }
            R = oR; C = oC;
        }
    }
    if (st == RCN_O_OK) { /* rcn.rs:350-355: concat maps, each in column-major iteration order */
        const size_t sz = (size_t)R * (size_t)C;
        for (size_t i = 0; i < len; ++i) memcpy(out + i * sz, fs[i], sizeof(double) * sz);
    }
    for (size_t i = 0; i < cap; ++i) free(fs[i]);
    free(fs);
    return st;
}

/* rcn.rs:230-251 */
void rcn_o_gen_scales(const double* feats, size_t n, size_t F, double* mean_out, double* sd_out) {
    double mean = 0.0, sd = 0.0;
    const double cnt = (double)F * (double)n;
    for (size_t v = 0; v < n; ++v)
        for (size_t r = 0; r < F; ++r) mean += feats[v * F + r];
    mean /= cnt;
    for (size_t v = 0; v < n; ++v)
        for (size_t r = 0; r < F; ++r) { const double d = feats[v * F + r] - mean; sd += d * d; } /* powi(.,2) */
    sd = sqrt(sd / cnt);
    *mean_out = mean;
    *sd_out = sd;
}

/* rcn.rs:407-412 / 86-89 */
void rcn_o_standardize(double* feats, size_t count, double mean, double sd) {
    for (size_t i = 0; i < count; ++i) {
        const double d = (feats[i] - mean) / sd;
        feats[i] = (d >= 0.0) ? d : 0.0;
    }
}

/* rcn.rs:466-471 */
void rcn_o_get_expected_vec(int class_idx, int classes, double* out) {
    for (int i = 0; i < classes; ++i) out[i] = (i == class_idx) ? 1.0 : 0.0;
}

/* ------------------------------------------------------------------ rcn.rs: dense net */

/* rcn.rs:429-443: usize::pow(4,c) / usize::pow(2,p) * l with p += 2 per pool layer */
long rcn_o_first_layer_fan_in(const rcn_o_layer* layers, int n_layers, long l) {
    unsigned c = 0, p = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (layers[i].kind == RCN_O_LAYER_CONV) c += 1; else p += 2;
    }
    unsigned long long num = 1, den = 1;
    for (unsigned i = 0; i < c; ++i) num *= 4ULL;
    for (unsigned i = 0; i < p; ++i) den *= 2ULL;
    return (long)(num / den * (unsigned long long)l);
}

/* rcn.rs:478-483: 1/(1 + E^(-x)) through powf */
double rcn_o_sigmoid(double x) { return 1.0 / (1.0 + pow(M_E, -x)); }

/* rcn.rs:490-492: sigmoid(v) (*) (1 - sigmoid(v)) */
double rcn_o_sigmoid_prime(double z) { const double s = rcn_o_sigmoid(z); return s * (1.0 - s); }

/* nalgebra gemv (blas.rs): y = a[:,0]*x[0]; y += a[:,j]*x[j] for j=1.. -- sequential over columns */
static void gemv(const double* W, int rows, int cols, const double* x, double* y) {
    for (int i = 0; i < rows; ++i) y[i] = AT(W, rows, i, 0) * x[0];
    for (int j = 1; j < cols; ++j) {
        const double xj = x[j];
        for (int i = 0; i < rows; ++i) y[i] = AT(W, rows, i, j) * xj + y[i];
    }
}

/* (W^T) * d : the reference materialises W.transpose() then runs the same gemv (rcn.rs:308) */
static void gemv_t(const double* W, int rows, int cols, const double* d, double* y) {
    for (int j = 0; j < cols; ++j) y[j] = AT(W, rows, 0, j) * d[0];
    for (int i = 1; i < rows; ++i) {
        const double di = d[i];
        for (int j = 0; j < cols; ++j) y[j] = AT(W, rows, i, j) * di + y[j];
    }
}

/* rcn.rs:105-116 */
void rcn_o_classify_test(const rcn_o_net* net, const double* x, double* out) {
    int maxd = 0;
    for (int l = 0; l <= net->n_layers; ++l) if (net->dims[l] > maxd) maxd = net->dims[l];
    double* a = (double*)malloc(sizeof(double) * (size_t)maxd);
    double* z = (double*)malloc(sizeof(double) * (size_t)maxd);
    memcpy(a, x, sizeof(double) * (size_t)net->dims[0]);
    for (int l = 0; l < net->n_layers; ++l) {
        const int in = net->dims[l], on = net->dims[l + 1];
        gemv(net->W[l], on, in, a, z);
        for (int i = 0; i < on; ++i) a[i] = rcn_o_sigmoid(z[i] + net->b[l][i]);
    }
    memcpy(out, a, sizeof(double) * (size_t)net->dims[net->n_layers]);
    free(a); free(z);
}

/* rcn.rs:92-97: max_by(total_cmp) keeps the LAST maximal element */
int rcn_o_classify_argmax(const double* out, int classes) {
    int best = 0;
    for (int i = 1; i < classes; ++i)
        if (!(out[i] < out[best])) best = i;
    return best;
}

/* rcn.rs:154-156 */
int rcn_o_eval_accept(const double* out, const double* expect, int classes) {
    double mx = out[0];
    for (int i = 1; i < classes; ++i) if (out[i] > mx) mx = out[i];
    for (int i = 0; i < classes; ++i) {
        const double onehot = (out[i] == mx) ? 1.0 : 0.0;
        if (onehot != expect[i]) return 0;
    }
    return 1;
}

/* rcn.rs:260-314 */
void rcn_o_backprop(const rcn_o_net* net, const double* x, const double* y, double** dW, double** db) {
    const int L = net->n_layers;
    double** act = (double**)malloc(sizeof(double*) * (size_t)(L + 1));
    double** zs = (double**)malloc(sizeof(double*) * (size_t)L);
    act[0] = (double*)malloc(sizeof(double) * (size_t)net->dims[0]);
    memcpy(act[0], x, sizeof(double) * (size_t)net->dims[0]);
    for (int l = 0; l < L; ++l) {                                     /* rcn.rs:281-291 */
        const int in = net->dims[l], on = net->dims[l + 1];
        zs[l] = (double*)malloc(sizeof(double) * (size_t)on);
        act[l + 1] = (double*)malloc(sizeof(double) * (size_t)on);
        gemv(net->W[l], on, in, act[l], zs[l]);
        for (int i = 0; i < on; ++i) { zs[l][i] = zs[l][i] + net->b[l][i]; act[l + 1][i] = rcn_o_sigmoid(zs[l][i]); }
    }
    int maxd = 0;
    for (int l = 0; l <= L; ++l) if (net->dims[l] > maxd) maxd = net->dims[l];
    double* delta = (double*)malloc(sizeof(double) * (size_t)maxd);
    double* nd = (double*)malloc(sizeof(double) * (size_t)maxd);
    {                                                                 /* rcn.rs:299-303 */
        const int on = net->dims[L], in = net->dims[L - 1];
        for (int i = 0; i < on; ++i) delta[i] = (act[L][i] - y[i]) * rcn_o_sigmoid_prime(zs[L - 1][i]);
        memcpy(db[L - 1], delta, sizeof(double) * (size_t)on);
        for (int j = 0; j < in; ++j)
            for (int i = 0; i < on; ++i) AT(dW[L - 1], on, i, j) = delta[i] * act[L - 1][j];
    }
    for (int k = 1; k < L; ++k) {                                     /* rcn.rs:305-311, l = k */
        const int li = L - 1 - k;                  /* layer whose delta we compute */
        const int on = net->dims[li + 1], in = net->dims[li];
        const int up_rows = net->dims[li + 2];     /* W[li+1] is up_rows x on */
        gemv_t(net->W[li + 1], up_rows, on, delta, nd);
        for (int i = 0; i < on; ++i) delta[i] = nd[i] * rcn_o_sigmoid_prime(zs[li][i]);
        memcpy(db[li], delta, sizeof(double) * (size_t)on);
        for (int j = 0; j < in; ++j)
            for (int i = 0; i < on; ++i) AT(dW[li], on, i, j) = delta[i] * act[li][j];
    }
    for (int l = 0; l < L; ++l) { free(zs[l]); free(act[l + 1]); }
    free(act[0]); free(act); free(zs); free(delta); free(nd);
}

static double sample_cost(const rcn_o_net* net, const double* x, const double* y) {
    const int C = net->dims[net->n_layers];
    double* o = (double*)malloc(sizeof(double) * (size_t)C);
    rcn_o_classify_test(net, x, o);
    double c = 0.0;
    for (int i = 0; i < C; ++i) { const double d = o[i] - y[i]; c += d * d; }
    free(o);
    return c;
}

static void alloc_grads(const rcn_o_net* net, double*** gW, double*** gb) {
    const int L = net->n_layers;
    *gW = (double**)malloc(sizeof(double*) * (size_t)L);
    *gb = (double**)malloc(sizeof(double*) * (size_t)L);
    for (int l = 0; l < L; ++l) {
        (*gW)[l] = (double*)calloc((size_t)net->dims[l] * (size_t)net->dims[l + 1], sizeof(double));
        (*gb)[l] = (double*)calloc((size_t)net->dims[l + 1], sizeof(double));
    }
}

static void free_grads(const rcn_o_net* net, double** gW, double** gb) {
    for (int l = 0; l < net->n_layers; ++l) { free(gW[l]); free(gb[l]); }
    free(gW); free(gb);
}

void rcn_o_batch_gradient(const rcn_o_net* net, const double* X, const double* Y, size_t B,
                          double** gW, double** gb, double* cost) {
    const int L = net->n_layers;
    const size_t F = (size_t)net->dims[0], C = (size_t)net->dims[L];
    double **dW, **db;
    alloc_grads(net, &dW, &db);
    for (int l = 0; l < L; ++l) {                                     /* rcn.rs:177-188 */
        memset(gW[l], 0, sizeof(double) * (size_t)net->dims[l] * (size_t)net->dims[l + 1]);
        memset(gb[l], 0, sizeof(double) * (size_t)net->dims[l + 1]);
    }
    double c = 0.0;
    for (size_t s = 0; s < B; ++s) {                                  /* rcn.rs:190-205, fixed order */
        rcn_o_backprop(net, X + s * F, Y + s * C, dW, db);
        if (cost) c += sample_cost(net, X + s * F, Y + s * C);
        for (int l = 0; l < L; ++l) {
            const size_t nw = (size_t)net->dims[l] * (size_t)net->dims[l + 1], nb = (size_t)net->dims[l + 1];
            for (size_t i = 0; i < nb; ++i) gb[l][i] = db[l][i] + gb[l][i];
            for (size_t i = 0; i < nw; ++i) gW[l][i] = dW[l][i] + gW[l][i];
        }
    }
    if (cost) *cost = c / (2.0 * (double)B);
    free_grads(net, dW, db);
}

static void sgd_update(rcn_o_net* net, double** gW, double** gb, size_t B, double eta) {
    const double scale = eta / (double)B;                             /* rcn.rs:214,221 */
    for (int l = 0; l < net->n_layers; ++l) {
        const size_t nw = (size_t)net->dims[l] * (size_t)net->dims[l + 1], nb = (size_t)net->dims[l + 1];
        for (size_t i = 0; i < nw; ++i) net->W[l][i] = net->W[l][i] - scale * gW[l][i];
        for (size_t i = 0; i < nb; ++i) net->b[l][i] = net->b[l][i] - scale * gb[l][i];
    }
}

/* rcn.rs:176-223 */
double rcn_o_train_batch(rcn_o_net* net, const double* X, const double* Y, size_t B, double eta) {
    double **gW, **gb, cost = 0.0;
    alloc_grads(net, &gW, &gb);
    rcn_o_batch_gradient(net, X, Y, B, gW, gb, &cost);
    sgd_update(net, gW, gb, B, eta);
    free_grads(net, gW, gb);
    return cost;
}

/* ---- threaded variant mirroring the rayon structure (for the CPU baseline timing) ---- */
typedef struct {
    const rcn_o_net* net; const double* X; const double* Y; size_t B;
    size_t* next; pthread_mutex_t* mu; double*** gW; double*** gb;
} mt_job;

static void* mt_worker(void* arg) {
    mt_job* j = (mt_job*)arg;
    const rcn_o_net* net = j->net;
    const int L = net->n_layers;
    const size_t F = (size_t)net->dims[0], C = (size_t)net->dims[L];
    for (;;) {
        pthread_mutex_lock(j->mu);
        const size_t s = (*j->next)++;
        pthread_mutex_unlock(j->mu);
        if (s >= j->B) break;
        double **dW, **db;
        alloc_grads(net, &dW, &db);                                   /* rcn.rs:265-274: fresh zeroed grads per sample */
        rcn_o_backprop(net, j->X + s * F, j->Y + s * C, dW, db);
        pthread_mutex_lock(j->mu);                                    /* rcn.rs:192-193 */
        double **nW, **nb;                                            /* rcn.rs:195-204: sums rebuilt by map+collect */
        alloc_grads(net, &nW, &nb);
        for (int l = 0; l < L; ++l) {
            const size_t nw = (size_t)net->dims[l] * (size_t)net->dims[l + 1], nbn = (size_t)net->dims[l + 1];
            for (size_t i = 0; i < nbn; ++i) nb[l][i] = db[l][i] + (*j->gb)[l][i];
            for (size_t i = 0; i < nw; ++i) nW[l][i] = dW[l][i] + (*j->gW)[l][i];
        }
        free_grads(net, *j->gW, *j->gb);
        *j->gW = nW; *j->gb = nb;
        pthread_mutex_unlock(j->mu);
        free_grads(net, dW, db);
    }
    return NULL;
}

/* persistent worker pool (rayon keeps its global pool alive between par_iter calls; spawning threads per batch would
 * charge the baseline for something the reference does not do) */
static struct {
    pthread_mutex_t mu; pthread_cond_t start, done;
    pthread_t* th; int n; unsigned long gen; int remaining; mt_job* job; int init;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, NULL, 0 };

static void* pool_main(void* arg) {
    (void)arg;
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.start, &g_pool.mu);
        seen = g_pool.gen;
        mt_job* job = g_pool.job;
        pthread_mutex_unlock(&g_pool.mu);
        if (job) mt_worker(job);
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.remaining == 0) pthread_cond_signal(&g_pool.done);
        pthread_mutex_unlock(&g_pool.mu);
    }
    return NULL;
}

static void pool_ensure(int n) {
    if (g_pool.n >= n) return;
    g_pool.th = (pthread_t*)realloc(g_pool.th, sizeof(pthread_t) * (size_t)n);
    for (int t = g_pool.n; t < n; ++t) pthread_create(&g_pool.th[t], NULL, pool_main, NULL);
    g_pool.n = n;
}

double rcn_o_train_batch_mt(rcn_o_net* net, const double* X, const double* Y, size_t B, double eta, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > B) threads = (int)B;                    /* par_iter over B items: at most B tasks */
    double **gW, **gb;
    alloc_grads(net, &gW, &gb);
    size_t next = 0;
    pthread_mutex_t mu;
    pthread_mutex_init(&mu, NULL);
    mt_job job = { net, X, Y, B, &next, &mu, &gW, &gb };
    const int helpers = threads - 1;
    if (helpers > 0) {
        pool_ensure(helpers);
        pthread_mutex_lock(&g_pool.mu);
        /* wake exactly the pool; workers beyond `helpers` find the batch drained immediately */
        g_pool.job = &job; g_pool.remaining = g_pool.n; g_pool.gen++;
        pthread_cond_broadcast(&g_pool.start);
        pthread_mutex_unlock(&g_pool.mu);
    }
    mt_worker(&job);
    if (helpers > 0) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.remaining != 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
        g_pool.job = NULL;
        pthread_mutex_unlock(&g_pool.mu);
    }
    pthread_mutex_destroy(&mu);
    sgd_update(net, gW, gb, B, eta);
    free_grads(net, gW, gb);
    return 0.0;
}
