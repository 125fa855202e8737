/*
 * sift_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference SIFT hot path.
 *
 * This file is the parity oracle for the MI355X HIP implementation.  It is NOT part of
 * the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  It restates, stage by stage, what the reference's OpenCL-CPU kernels compute
 * (pierrepaleo/sift_pyocl, cited as file:line below), in strict IEEE binary32 with no FMA
 * contraction (build with -ffp-contract=off), with f64 promotion exactly where the
 * reference source uses unsuffixed literals.  Transcendentals come from oracle_math.h
 * ("siftmath v1", see there for why libm is not used).
 *
 * Parity status: PINNED against the reference's own kernels compiled natively
 * (oracle/_ref, built by oracle/Makefile from the reference openCL kernel files) --
 * tests/test_oracle_vs_ref.py; golden vectors generated from that build are committed
 * under tests/golden/ (tests/golden/make_golden.py).
 *
 * Layout conventions: image planes are row-major (H,W) float32, index y*W+x
 * (convolution.cl:51).  A keypoint in flight is 4 floats; before orientation
 * (peak,row,col,sigma) (image.cl:11-19), after orientation (x,y,sigma*oct,angle)
 * (orientation_cpu.cl:141-145).  Final record: 144 bytes {x,y,scale,angle,desc[128]}
 * (plan.py:110-115).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_math.h"

#define SO_PI_F 3.14159274101257f   /* OpenCL M_PI_F   */
#define SO_1_PI_F 0.31830987334251f /* OpenCL M_1_PI_F */

typedef struct { float s0, s1, s2, s3; } so_kp4;
typedef struct { float x, y, scale, angle; uint8_t desc[128]; } so_record;

typedef struct {
    double init_sigma;   /* par.InitSigma as the Python double (param.py:55, plan.py:131) */
    float peak_thresh;   /* par.PeakThresh  (param.py:58)  */
    float edge_thresh0;  /* par.EdgeThresh1 -> kernel slot EdgeThresh0 (plan.py:633) */
    float edge_thresh;   /* par.EdgeThresh  -> kernel slot EdgeThresh  (plan.py:634) */
    float ori_sigma;     /* par.OriSigma    (param.py:69)  */
    int border_dist;     /* par.BorderDist  (param.py:56)  */
    int octave_max;      /* 0 = all octaves (reference behaviour, plan.py:213-224) */
    int pix_per_kp;      /* SiftPlan.PIX_PER_KP (plan.py:109) */
    int double_im_size;  /* par.DoubleImSize (param.py:53): the input counts as blurred by 1.0 instead of 0.5 (plan.py:254, 297, 534) */
} so_params;

/* ------------------------------------------------------------------ math exports */
float so_expf(float x) { return om_expf(x); }
float so_exp2f(float x) { return om_exp2f(x); }
float so_atan2f(float y, float x) { return om_atan2f(y, x); }
void so_sincosf(float x, float *s, float *c) { om_sincosf(x, s, c); }
/* array forms (the device fast paths are checked on 10^7 arguments) */
void so_expf_array(const float *x, float *out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = om_expf(x[i]);
}
void so_atan2f_array(const float *y, const float *x, float *out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = om_atan2f(y[i], x[i]);
}

/* ------------------------------------------------------------------ A3: Gaussian taps
 * gaussian.cl:56-140 launched with one work-group of nextpower(size) items
 * (plan.py:321-330).  Tree sum in LDS order, not a serial sum. */
static int so_nextpower(int n) { int p = 1; while (p < n) p <<= 1; return p; }

int so_gaussian_taps(float sigma, int size, float *out) {
    int P = so_nextpower(size);
    if (P < 2) P = 2;
    float *g = (float *)calloc((size_t)P, sizeof(float));
    float *sum = (float *)calloc((size_t)P, sizeof(float));
    if (!g || !sum) { free(g); free(sum); return -1; }
    const float norm = sqrtf(2.0f * SO_PI_F);
    for (int i = 0; i < size; i++) {
        float x = ((float)i - ((float)size - 1.0f) / 2.0f) / sigma;
        float y = om_expf(-x * x / 2.0f);
        g[i] = y / sigma / norm;
        sum[i] = g[i];
    }
    /* strides 512..2, each gated on SIZE > stride (gaussian.cl:78-135) */
    for (int stride = 512; stride >= 2; stride >>= 1) {
        if (size > stride) {
            for (int i = 0; i < stride && i + stride < P; i++) sum[i] += sum[i + stride];
        }
    }
    sum[0] += sum[1];
    for (int i = 0; i < size; i++) out[i] = g[i] / sum[0];
    free(g); free(sum);
    return 0;
}

/* ------------------------------------------------------------------ A2: min/max + normalise
 * reductions.cl:217-241 (serial variant; min/max are order independent so the two-stage
 * variant gives the same bits), preprocess.cl:239-252. */
void so_minmax(const float *img, int64_t n, float *mn, float *mx) {
    float lo = img[0], hi = img[0];
#pragma omp parallel for reduction(min : lo) reduction(max : hi) schedule(static)
    for (int64_t i = 1; i < n; i++) {
        float v = img[i];
        if (v > hi) hi = v;
        if (v < lo) lo = v;
    }
    *mn = lo; *mx = hi;
}

void so_normalize(float *img, int64_t n, float mn, float mx) {
    const float top = 255.0f;
    const float range = mx - mn;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) img[i] = top * (img[i] - mn) / range;
}

/* ------------------------------------------------------------------ A4: separable blur
 * convolution.cl:16-101.  Symmetric boundary: index -k -> k-1, W-1+k -> W-k.
 * Accumulation in ascending tap order from 0.0f, tap j multiplies filter[n-1-j]. */
static inline int so_reflect(int i, int n) {
    if (i < 0) return -i - 1;
    if (i > n - 1) return 2 * n - 1 - i;
    return i;
}

void so_convolve_h(const float *in, float *out, const float *taps, int n, int W, int H) {
    int c = (n & 1) ? n / 2 : n / 2 - 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float *row = in + (size_t)y * W;
        float *orow = out + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float acc = 0.0f;
            for (int j = 0; j < n; j++) acc += row[so_reflect(x - c + j, W)] * taps[n - 1 - j];
            orow[x] = acc;
        }
    }
}

void so_convolve_v(const float *in, float *out, const float *taps, int n, int W, int H) {
    int c = (n & 1) ? n / 2 : n / 2 - 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        float *orow = out + (size_t)y * W;
        for (int x = 0; x < W; x++) orow[x] = 0.0f;
        for (int j = 0; j < n; j++) {
            const float *row = in + (size_t)so_reflect(y - c + j, H) * W;
            const float t = taps[n - 1 - j];
            for (int x = 0; x < W; x++) orow[x] += row[x] * t;
        }
    }
}

/* plan.py:571-594: horizontal into tmp, vertical into out (in may alias out) */
void so_blur(const float *in, float *out, float *tmp, const float *taps, int n, int W, int H) {
    so_convolve_h(in, tmp, taps, n, W, H);
    so_convolve_v(tmp, out, taps, n, W, H);
}

/* ------------------------------------------------------------------ A5: DoG  (algebra.cl:18-37)
 * called as combine(blur[s+1], -1, blur[s], +1) (plan.py:619-623) */
void so_combine(const float *u, float a, const float *v, float b, float *w, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) w[i] = a * u[i] + b * v[i];
}

/* ------------------------------------------------------------------ A6: extrema  (image.cl:119-213)
 * dogs = 5 contiguous planes.  Appends (val,row,col,scale) at *counter; entries beyond
 * capacity are counted but not stored, as the reference does. */
static inline int so_is_keypoint(const float *P, const float *D, const float *N, size_t pos, int W,
                                 double contrast, float edth) {
    const float v = D[pos];
    if (!((double)fabsf(v) > contrast)) return 0;
    int ismax = (v > 0.0f), ismin = !ismax;
    for (int dr = -1; dr <= 1; dr++)
        for (int dc = -1; dc <= 1; dc++) {
            const size_t q = pos + (ptrdiff_t)dr * W + dc;
            if (ismax && (P[q] > v || D[q] > v || N[q] > v)) ismax = 0;
            if (ismin && (P[q] < v || D[q] < v || N[q] < v)) ismin = 0;
        }
    if (!(ismax || ismin)) return 0;
    /* 2-D Hessian of D; literals 2.0 and 4.0 are double in the source (image.cl:180-184) */
    const float up = D[pos - W], dn = D[pos + W], lf = D[pos - 1], rt = D[pos + 1];
    const float H00 = (float)(((double)up - 2.0 * (double)v) + (double)dn);
    const float H11 = (float)(((double)lf - 2.0 * (double)v) + (double)rt);
    const float dd = (D[pos + W + 1] - D[pos + W - 1]) - (D[pos - W + 1] - D[pos - W - 1]);
    const float H01 = (float)((double)dd / 4.0);
    const float det = H00 * H11 - H01 * H01;
    const float tr = H00 + H11;
    if (det < edth * tr * tr) return 0;
    return v != 0.0f;   /* res != 0.0f, image.cl:201 */
}

void so_local_maxmin(const float *dogs, so_kp4 *out, int border, float peak_thresh, int octsize,
                     float edge_thresh0, float edge_thresh, int *counter, int capacity,
                     int scale, int W, int H) {
    const size_t plane = (size_t)W * H;
    const float *P = dogs + (size_t)(scale - 1) * plane;
    const float *D = dogs + (size_t)scale * plane;
    const float *N = dogs + (size_t)(scale + 1) * plane;
    const double contrast = 0.8 * (double)peak_thresh;   /* image.cl:152: double literal */
    const float edth = (octsize <= 1) ? edge_thresh0 : edge_thresh;
    if (H <= 2 * border || W <= 2 * border) return;
    int *rowcnt = (int *)calloc((size_t)H + 1, sizeof(int));
    if (!rowcnt) return;
#pragma omp parallel for schedule(dynamic, 8)
    for (int r = border; r < H - border; r++) {
        int n = 0;
        for (int c = border; c < W - border; c++)
            n += so_is_keypoint(P, D, N, (size_t)r * W + c, W, contrast, edth);
        rowcnt[r] = n;
    }
    int base = *counter;
    for (int r = 0; r < H; r++) { int n = rowcnt[r]; rowcnt[r] = base; base += n; }
    *counter = base;                      /* atomic_inc counts even past capacity */
#pragma omp parallel for schedule(dynamic, 8)
    for (int r = border; r < H - border; r++) {
        int at = rowcnt[r];
        for (int c = border; c < W - border; c++) {
            const size_t pos = (size_t)r * W + c;
            if (!so_is_keypoint(P, D, N, pos, W, contrast, edth)) continue;
            if (at < capacity) {
                out[at].s0 = D[pos]; out[at].s1 = (float)r; out[at].s2 = (float)c; out[at].s3 = (float)scale;
            }
            at++;
        }
    }
    free(rowcnt);
}

/* ------------------------------------------------------------------ A7: refinement  (image.cl:235-369) */
void so_interp_keypoint(const float *dogs, so_kp4 *kps, int start, int end, float peak_thresh,
                        float init_sigma, int W, int H) {
    const size_t plane = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = start; i < end; i++) {
        so_kp4 k = kps[i];
        int r = (int)k.s1, c = (int)k.s2, scale = (int)k.s3;
        if (r == -1) continue;
        const float *P = dogs + (size_t)(scale - 1) * plane;
        const float *D = dogs + (size_t)scale * plane;
        const float *N = dogs + (size_t)(scale + 1) * plane;
        int newr = r, newc = c, moves = 5, again = 1;
        float s0 = 0, s1 = 0, s2 = 0, peak = 0;
        while (again) {
            r = newr; c = newc;
            const size_t pos = (size_t)r * W + c;
            const float g0 = (N[pos] - P[pos]) / 2.0f;
            const float g1 = (D[pos + W] - D[pos - W]) / 2.0f;
            const float g2 = (D[pos + 1] - D[pos - 1]) / 2.0f;
            const float H00 = P[pos] - 2.0f * D[pos] + N[pos];
            const float H11 = D[pos - W] - 2.0f * D[pos] + D[pos + W];
            const float H22 = D[pos - 1] - 2.0f * D[pos] + D[pos + 1];
            const float H01 = ((N[pos + W] - N[pos - W]) - (P[pos + W] - P[pos - W])) / 4.0f;
            const float H02 = ((N[pos + 1] - N[pos - 1]) - (P[pos + 1] - P[pos - 1])) / 4.0f;
            const float H12 = ((D[pos + W + 1] - D[pos + W - 1]) - (D[pos - W + 1] - D[pos - W - 1])) / 4.0f;
            const float H10 = H01, H20 = H02, H21 = H12;
            /* image.cl:298: six-term expansion, left to right */
            const float det = -(H02 * H11 * H20) + H01 * H12 * H20 + H02 * H10 * H21
                              - H00 * H12 * H21 - H01 * H10 * H22 + H00 * H11 * H22;
            const float K00 = H11 * H22 - H12 * H21;
            const float K01 = H02 * H21 - H01 * H22;
            const float K02 = H01 * H12 - H02 * H11;
            const float K10 = H12 * H20 - H10 * H22;
            const float K11 = H00 * H22 - H02 * H20;
            const float K12 = H02 * H10 - H00 * H12;
            const float K20 = H10 * H21 - H11 * H20;
            const float K21 = H01 * H20 - H00 * H21;
            const float K22 = H00 * H11 - H01 * H10;
            s0 = -(g0 * K00 + g1 * K01 + g2 * K02) / det;
            s1 = -(g0 * K10 + g1 * K11 + g2 * K12) / det;
            s2 = -(g0 * K20 + g1 * K21 + g2 * K22) / det;
            peak = D[pos] + 0.5f * (s0 * g0 + s1 * g1 + s2 * g2);
            if (s1 > 0.6f && newr < H - 3) newr++;
            else if (s1 < -0.6f && newr > 3) newr--;
            if (s2 > 0.6f && newc < W - 3) newc++;
            else if (s2 < -0.6f && newc > 3) newc--;
            if (moves > 0 && (newr != r || newc != c)) moves--;
            else again = 0;
        }
        so_kp4 o;
        if (fabsf(s0) <= 1.5f && fabsf(s1) <= 1.5f && fabsf(s2) <= 1.5f && fabsf(peak) >= peak_thresh) {
            o.s0 = peak;
            o.s1 = (float)r + s1;
            o.s2 = (float)c + s2;
            o.s3 = init_sigma * om_exp2f(((float)scale + s0) / 3.0f);
        } else {
            o.s0 = o.s1 = o.s2 = o.s3 = -1.0f;
        }
        kps[i] = o;
    }
}

/* ------------------------------------------------------------------ A8: compaction  (algebra.cl:57-84)
 * serial: keeps order; returns new count. */
int so_compact(const so_kp4 *in, so_kp4 *out, int start, int end) {
    int n = start;
    for (int i = 0; i < start; i++) out[i] = in[i];
    for (int i = start; i < end; i++)
        if (in[i].s1 != -1.0f) { if (n < end) out[n] = in[i]; n++; }
    return n;
}

/* ------------------------------------------------------------------ A9: gradient maps  (image.cl:47-80) */
static inline void so_gradient_at(const float *I, int x, int y, int W, int H, float *mag, float *ori) {
    const size_t pos = (size_t)y * W + x;
    float gx, gy;
    if (x == 0) gx = 2.0f * (I[pos + 1] - I[pos]);
    else if (x == W - 1) gx = 2.0f * (I[pos] - I[pos - 1]);
    else gx = I[pos + 1] - I[pos - 1];
    if (y == 0) gy = 2.0f * (I[pos] - I[pos + W]);
    else if (y == H - 1) gy = 2.0f * (I[pos - W] - I[pos]);
    else gy = I[pos - W] - I[pos + W];
    *mag = sqrtf(gx * gx + gy * gy);
    *ori = om_atan2f(-gy, gx);
}

void so_gradient(const float *img, float *grad, float *ori, int W, int H) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            so_gradient_at(img, x, y, W, H, grad + (size_t)y * W + x, ori + (size_t)y * W + x);
}

/* ------------------------------------------------------------------ A10: orientation  (orientation_cpu.cl:41-174)
 * Processes kps[start..end); extra orientations are appended at *counter (which the
 * caller sets to `end`, plan.py:675-690). */
typedef struct { int valid; int nextra; float angle; float extra[18]; } so_ori_result;

static void so_orientation_one(const so_kp4 k, const float *grad, const float *ori, float ori_sigma,
                               int W, int H, so_ori_result *res) {
    float hist[36];
    for (int b = 0; b < 36; b++) hist[b] = 0.0f;
    const int row = (int)((double)k.s1 + 0.5), col = (int)((double)k.s2 + 0.5);
    const float sigma = ori_sigma * k.s3;
    const int radius = (int)((double)sigma * 3.0);
    const int rmin = row - radius > 0 ? row - radius : 0;
    const int cmin = col - radius > 0 ? col - radius : 0;
    const int rmax = row + radius < H - 2 ? row + radius : H - 2;
    const int cmax = col + radius < W - 2 ? col + radius : W - 2;
    const float lim = (float)(radius * radius) + 0.5f;
    const float two_s2 = 2.0f * sigma * sigma;
    for (int r = rmin; r <= rmax; r++)
        for (int c = cmin; c <= cmax; c++) {
            const float gval = grad[(size_t)r * W + c];
            float dif = (float)r - k.s1;
            float distsq = dif * dif;
            dif = (float)c - k.s2;
            distsq += dif * dif;
            if (gval > 0.0f && distsq < lim) {
                const float a = ori[(size_t)r * W + c];
                int bin = (int)(36.0f * (a + SO_PI_F + 0.001f) / (2.0f * SO_PI_F));
                if (bin >= 0 && bin <= 36) {
                    if (bin > 35) bin = 35;
                    hist[bin] += om_expf(-distsq / two_s2) * gval;
                }
            }
        }
    /* six passes of [1,1,1]/3 circular smoothing; the wrap at i=35 sees the updated
     * hist[0]; division by the double literal 3.0 (orientation_cpu.cl:101-109) */
    for (int pass = 0; pass < 6; pass++) {
        float prev = hist[35];
        for (int b = 0; b < 36; b++) {
            const float cur = hist[b];
            const float nxt = hist[(b + 1 == 36) ? 0 : b + 1];
            hist[b] = (float)((double)((prev + cur) + nxt) / 3.0);
            prev = cur;
        }
    }
    float maxval = 0.0f; int argmax = 0;
    for (int b = 0; b < 36; b++) if (maxval < hist[b]) { maxval = hist[b]; argmax = b; }
    const float hp = hist[argmax == 0 ? 35 : argmax - 1];
    const float hn = hist[argmax == 35 ? 0 : argmax + 1];
    const float interp = 0.5f * (hp - hn) / (hp - 2.0f * maxval + hn);
    res->angle = 2.0f * SO_PI_F * ((float)argmax + 0.5f + interp) / 36.0f - SO_PI_F;
    res->nextra = 0;
    for (int b = 0; b < 36; b++) {
        const float hc = hist[b];
        const float hpp = hist[b == 0 ? 35 : b - 1], hnn = hist[b == 35 ? 0 : b + 1];
        if (hc > hpp && hc > hnn && hc >= 0.8f * maxval && b != argmax) {
            const float it = 0.5f * (hpp - hnn) / (hpp - 2.0f * hc + hnn);
            /* orientation_cpu.cl:166: "/36.0" is a double division */
            const float a2 = (float)((double)(2.0f * SO_PI_F * ((float)b + 0.5f + it)) / 36.0 - (double)SO_PI_F);
            if (a2 >= -SO_PI_F && a2 <= SO_PI_F) res->extra[res->nextra++] = a2;
        }
    }
}

void so_orientation(so_kp4 *kps, const float *grad, const float *ori, int *counter, int octsize,
                    float ori_sigma, int capacity, int start, int end, int W, int H) {
    if (end <= start) return;
    so_ori_result *res = (so_ori_result *)malloc((size_t)(end - start) * sizeof(so_ori_result));
    if (!res) return;
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = start; i < end; i++) {
        res[i - start].valid = (kps[i].s1 >= 0.0f);
        if (res[i - start].valid) so_orientation_one(kps[i], grad, ori, ori_sigma, W, H, &res[i - start]);
    }
    for (int i = start; i < end; i++) {
        const so_ori_result *q = &res[i - start];
        if (!q->valid) continue;
        so_kp4 k = kps[i];
        k.s0 = k.s2 * (float)octsize;      /* x = col*oct  (orientation_cpu.cl:141-145) */
        k.s1 = k.s1 * (float)octsize;      /* y = row*oct */
        k.s2 = k.s3 * (float)octsize;      /* sigma*oct   */
        k.s3 = q->angle;
        kps[i] = k;
        for (int e = 0; e < q->nextra; e++) {
            k.s3 = q->extra[e];
            int old = (*counter)++;
            if (old < capacity) kps[old] = k;
        }
    }
    free(res);
}

/* ------------------------------------------------------------------ A11: descriptor  (keypoints_cpu.cl:36-161) */
void so_descriptor(const so_kp4 *kps, uint8_t *desc, const float *grad, const float *orim, int octsize,
                   int start, int end, int W, int H) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = start; i < end; i++) {
        const so_kp4 k = kps[i];
        if (!(k.s1 >= 0.0f)) continue;
        float d[128];
        for (int b = 0; b < 128; b++) d[b] = 0.0f;
        const float row = k.s1 / (float)octsize, col = k.s0 / (float)octsize, angle = k.s3;
        const int irow = (int)(row + 0.5f), icol = (int)(col + 0.5f);
        float sine, cosine;
        om_sincosf(angle, &sine, &cosine);
        const float spacing = k.s2 / (float)octsize * 3.0f;
        const int iradius = (int)((1.414f * spacing * 2.5f) + 0.5f);
        const float drow = row - (float)irow, dcol = col - (float)icol;
        for (int ii = -iradius; ii <= iradius; ii++)
            for (int jj = -iradius; jj <= iradius; jj++) {
                const float rx = ((cosine * (float)ii - sine * (float)jj) - drow) / spacing + 1.5f;
                const float cx = ((sine * (float)ii + cosine * (float)jj) - dcol) / spacing + 1.5f;
                const int yy = irow + ii, xx = icol + jj;
                if (!(rx > -1.0f && rx < 4.0f && cx > -1.0f && cx < 4.0f && yy >= 0 && yy < H && xx >= 0 && xx < W))
                    continue;
                const size_t pos = (size_t)yy * W + xx;
                const float er = rx - 1.5f, ec = cx - 1.5f;
                const float mag = grad[pos] * om_expf(-0.125f * (er * er + ec * ec));
                float o = orim[pos] - angle;
                while (o > 2.0f * SO_PI_F) o -= 2.0f * SO_PI_F;
                while (o < 0.0f) o += 2.0f * SO_PI_F;
                const float oval = 4.0f * o * SO_1_PI_F;
                const int ri = (int)((rx >= 0.0f) ? rx : rx - 1.0f);
                const int ci = (int)((cx >= 0.0f) ? cx : cx - 1.0f);
                const int oi = (int)((oval >= 0.0f) ? oval : oval - 1.0f);
                const float rf = rx - (float)ri, cf = cx - (float)ci, of = oval - (float)oi;
                if (!(ri >= -1 && ri < 4 && oi >= 0 && oi <= 8 && rf >= 0.0f && rf <= 1.0f)) continue;
                for (int a = 0; a < 2; a++) {
                    const int rb = ri + a;
                    if (rb < 0 || rb >= 4) continue;
                    const float rw = mag * (a == 0 ? 1.0f - rf : rf);
                    for (int b = 0; b < 2; b++) {
                        const int cb = ci + b;
                        if (cb < 0 || cb >= 4) continue;
                        const float cw = rw * (b == 0 ? 1.0f - cf : cf);
                        for (int e = 0; e < 2; e++) {
                            int ob = oi + e;
                            if (ob >= 8) ob = 0;
                            d[(rb * 4 + cb) * 8 + ob] += cw * (e == 0 ? 1.0f - of : of);
                        }
                    }
                }
            }
        float norm = 0.0f;
        for (int b = 0; b < 128; b++) norm += d[b] * d[b];
        norm = 1.0f / sqrtf(norm);                 /* rsqrt */
        for (int b = 0; b < 128; b++) d[b] *= norm;
        int changed = 0;
        norm = 0.0f;
        for (int b = 0; b < 128; b++) {
            if (d[b] > 0.2f) { d[b] = 0.2f; changed = 1; }
            norm += d[b] * d[b];
        }
        if (changed) {
            norm = 1.0f / sqrtf(norm);
            for (int b = 0; b < 128; b++) d[b] *= norm;
        }
        for (int b = 0; b < 128; b++) {
            /* (int)(512.0*v): NaN (all-zero histogram) is undefined in C; x86 cvttss2si
             * yields INT_MIN -> MIN(255,.) -> (uchar) 0.  Made explicit here. */
            const float v = d[b];
            int iv = (v == v) ? (int)(512.0 * (double)v) : 0;
            desc[(size_t)i * 128 + b] = (uint8_t)(iv < 255 ? iv : 255);
        }
    }
}

/* ------------------------------------------------------------------ A12: shrink  (preprocess.cl:267-285) */
void so_shrink(const float *in, float *out, int LW, int LH, int SW, int SH) {
    (void)LH;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < SH; y++)
        for (int x = 0; x < SW; x++) out[(size_t)y * SW + x] = in[(size_t)(2 * y) * LW + 2 * x];
}

/* ------------------------------------------------------------------ sigma schedule (plan.py:534-539, 602-618) */
int so_kernel_size(double sigma) {           /* utils.py:54-64, odd=True, cutoff=4 */
    int size = (int)ceil(2.0 * 4.0 * sigma + 1.0);
    if (size % 2 == 0) size += 1;
    return size;
}

int so_octave_count(int H, int W) {            /* plan.py:213-224 */
    int n = 1, h = H, w = W;
    while ((h < w ? h : w) > 12) { h /= 2; w /= 2; n++; }
    return n - 1;
}

/* ------------------------------------------------------------------ whole pipeline  (plan.py:432-567, 596-756)
 * image: (H,W) float32, not modified.  Returns 0, or -1 on allocation failure.
 * *n_out receives the number of records written (<= cap); *overflow is set when an
 * octave exceeded kpsize = H*W/pix_per_kp (the reference silently drops, plan.py:771). */
int so_keypoints(const float *image, int H, int W, const so_params *par, so_record *out,
                 int64_t cap, int64_t *n_out, int *overflow) {
    const size_t N = (size_t)W * H;
    int n_oct = so_octave_count(H, W);
    if (par->octave_max > 0 && par->octave_max < n_oct) n_oct = par->octave_max;
    const int kpsize = (int)(N / (size_t)par->pix_per_kp);
    float *blur[6];
    for (int i = 0; i < 6; i++) blur[i] = (float *)malloc(N * sizeof(float));
    float *tmp = (float *)malloc(N * sizeof(float));
    float *oribuf = (float *)malloc(N * sizeof(float));
    float *dogs = (float *)malloc(5 * N * sizeof(float));
    so_kp4 *kp1 = (so_kp4 *)malloc((size_t)kpsize * sizeof(so_kp4));
    so_kp4 *kp2 = (so_kp4 *)malloc((size_t)kpsize * sizeof(so_kp4));
    uint8_t *desc = (uint8_t *)malloc((size_t)kpsize * 128);
    *n_out = 0; *overflow = 0;
    if (!tmp || !oribuf || !dogs || !kp1 || !kp2 || !desc || !blur[5]) return -1;

    memcpy(blur[0], image, N * sizeof(float));
    float mn, mx;
    so_minmax(blur[0], (int64_t)N, &mn, &mx);
    so_normalize(blur[0], (int64_t)N, mn, mx);

    float taps[6][64]; int ntaps[6];
    const double init_sigma = par->init_sigma;
    int have_init = 0;
    const double cur_sigma = par->double_im_size ? 1.0 : 0.5;   /* plan.py:534 */
    if (init_sigma > cur_sigma) {
        double s = sqrt(init_sigma * init_sigma - cur_sigma * cur_sigma);
        ntaps[5] = so_kernel_size(s);
        so_gaussian_taps((float)s, ntaps[5], taps[5]);
        have_init = 1;
    }
    {
        const double ratio = pow(2.0, 1.0 / 3.0);
        double prev = init_sigma;
        for (int s = 0; s < 5; s++) {
            double inc = prev * sqrt(ratio * ratio - 1.0);
            ntaps[s] = so_kernel_size(inc);
            so_gaussian_taps((float)inc, ntaps[s], taps[s]);
            prev *= ratio;
        }
    }
    if (have_init) so_blur(blur[0], blur[0], tmp, taps[5], ntaps[5], W, H);

    int w = W, h = H, octsize = 1;
    int64_t total = 0;
    for (int oct = 0; oct < n_oct; oct++) {
        const size_t n = (size_t)w * h;
        /* memset_float(Kp_1,-1) (plan.py:797) is not restated: no stage reads beyond `counter` */
        int counter = 0, last_start = 0;
        for (int s = 0; s < 5; s++) {
            so_blur(blur[s], blur[s + 1], tmp, taps[s], ntaps[s], w, h);
            so_combine(blur[s + 1], -1.0f, blur[s], 1.0f, dogs + (size_t)s * n, (int64_t)n);
        }
        for (int s = 1; s <= 3; s++) {
            so_local_maxmin(dogs, kp1, par->border_dist, par->peak_thresh, octsize, par->edge_thresh0,
                            par->edge_thresh, &counter, kpsize, s, w, h);
            if (counter > kpsize) { *overflow = 1; counter = kpsize; }
            so_interp_keypoint(dogs, kp1, last_start, counter, par->peak_thresh, (float)par->init_sigma, w, h);
            int newcnt = so_compact(kp1, kp2, last_start, counter);
            { so_kp4 *t = kp1; kp1 = kp2; kp2 = t; }
            counter = newcnt;
            so_gradient(blur[s], tmp, oribuf, w, h);
            if (newcnt > last_start) {
                so_orientation(kp1, tmp, oribuf, &counter, octsize, par->ori_sigma, kpsize, last_start, newcnt, w, h);
                if (counter > kpsize) { *overflow = 1; counter = kpsize; }
                so_descriptor(kp1, desc, tmp, oribuf, octsize, last_start, counter, w, h);
            }
            last_start = counter;
        }
        if (oct < n_oct - 1) {
            so_shrink(blur[3], blur[0], w, h, w / 2, h / 2);
        }
        /* plan.py:545-565: drop rows whose 4-float sum is NaN, pack records */
        for (int i = 0; i < last_start; i++) {
            const so_kp4 k = kp1[i];
            const float sum = ((k.s0 + k.s1) + k.s2) + k.s3;
            if (sum != sum) continue;
            if (total < cap) {
                so_record *r = out + total;
                r->x = k.s0; r->y = k.s1; r->scale = k.s2; r->angle = k.s3;
                memcpy(r->desc, desc + (size_t)i * 128, 128);
            }
            total++;
        }
        w /= 2; h /= 2; octsize *= 2;
    }
    *n_out = total < cap ? total : cap;
    for (int i = 0; i < 6; i++) free(blur[i]);
    free(tmp); free(oribuf); free(dogs); free(kp1); free(kp2); free(desc);
    return 0;
}

/* ------------------------------------------------------------------ A14: matching  (matching_cpu.cl:57-109)
 * Returns the number of pairs that passed the ratio test; at most cap are stored. */
int64_t so_match(const so_record *k1, int64_t n1, const so_record *k2, int64_t n2, float ratio_th,
                 int32_t *pairs, int64_t cap) {
    int32_t *best = (int32_t *)malloc((size_t)(n1 > 0 ? n1 : 1) * sizeof(int32_t));
    if (!best) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n1; i++) {
        float d1 = 1000000000000.0f, d2 = 1000000000000.0f;
        int cur = 0;
        const uint8_t *a = k1[i].desc;
        for (int64_t j = 0; j < n2; j++) {
            const uint8_t *b = k2[j].desc;
            int dist = 0;
            for (int t = 0; t < 128; t++) dist += a[t] > b[t] ? a[t] - b[t] : b[t] - a[t];
            if ((float)dist < d1) { d2 = d1; d1 = (float)dist; cur = (int)j; }
            else if ((float)dist < d2) d2 = (float)dist;
        }
        best[i] = (d2 != 0.0f && d1 / d2 < ratio_th) ? cur : -1;
    }
    int64_t count = 0;
    for (int64_t i = 0; i < n1; i++) {
        if (best[i] < 0) continue;
        if (count < cap) { pairs[2 * count] = (int32_t)i; pairs[2 * count + 1] = best[i]; }
        count++;
    }
    free(best);
    return count;
}

/* ------------------------------------------------------------------ ROI-masked / mutual matching
 * roi_mode 1 restates `matching_valid` (matching_cpu.cl:136-199) literally, quirks included:
 *   - a query (list 1) keypoint is dropped only when it lies inside the mask array AND the mask is 0 there
 *     (keypoints beyond roi_width / roi_height are processed);
 *   - a list-2 keypoint that fails `inside && mask != 0` is NOT skipped: its distance is 0 (the accumulation
 *     is what the mask guards), so it competes as a perfect match;
 *   - (c, r) = (int)x, (int)y.
 * roi_mode 2 ("strict", an extension): keypoints of either list that are not inside the array on a non-zero
 * mask pixel do not take part at all.  roi_mode 0: no mask (== so_match).
 * mutual != 0 (extension): a pair (i, j) is kept only if i is also the nearest list-1 keypoint of j, over the
 * same masked distances, ties to the smallest index (the reference's strict '<' scan order). */
static inline int so_roi_inside(const so_record *k, int rw, int rh) {
    const int c = (int)k->x, r = (int)k->y;
    return r < rh && c < rw && r >= 0 && c >= 0;
}
static inline int so_l1(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int t = 0; t < 128; t++) dist += a[t] > b[t] ? a[t] - b[t] : b[t] - a[t];
    return dist;
}

int64_t so_match_ex(const so_record *k1, int64_t n1, const so_record *k2, int64_t n2, float ratio_th,
                    const int8_t *valid, int rw, int rh, int roi_mode, int mutual, int32_t *pairs, int64_t cap) {
    if (!valid) roi_mode = 0;
    int32_t *best = (int32_t *)malloc((size_t)(n1 > 0 ? n1 : 1) * sizeof(int32_t));
    uint8_t *f1 = (uint8_t *)calloc((size_t)(n1 > 0 ? n1 : 1), 1);   /* 1: query dropped */
    uint8_t *f2 = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1);   /* 1: distance forced to 0, 2: excluded */
    int32_t *rev = (int32_t *)malloc((size_t)(n2 > 0 ? n2 : 1) * sizeof(int32_t));
    if (!best || !f1 || !f2 || !rev) { free(best); free(f1); free(f2); free(rev); return -1; }
    for (int64_t i = 0; roi_mode && i < n1; i++) {
        const int in = so_roi_inside(&k1[i], rw, rh);
        const int on = in && valid[(size_t)((int)k1[i].y) * rw + (int)k1[i].x] != 0;
        f1[i] = roi_mode == 1 ? (in && !on) : !on;
    }
    for (int64_t j = 0; roi_mode && j < n2; j++) {
        const int in = so_roi_inside(&k2[j], rw, rh);
        const int on = in && valid[(size_t)((int)k2[j].y) * rw + (int)k2[j].x] != 0;
        f2[j] = on ? 0 : (roi_mode == 1 ? 1 : 2);
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n1; i++) {
        best[i] = -1;
        if (f1[i]) continue;
        float d1 = 1000000000000.0f, d2 = 1000000000000.0f;
        int cur = 0;
        for (int64_t j = 0; j < n2; j++) {
            if (f2[j] == 2) continue;
            const int dist = f2[j] == 1 ? 0 : so_l1(k1[i].desc, k2[j].desc);
            if ((float)dist < d1) { d2 = d1; d1 = (float)dist; cur = (int)j; }
            else if ((float)dist < d2) d2 = (float)dist;
        }
        best[i] = (d2 != 0.0f && d1 / d2 < ratio_th) ? cur : -1;
    }
    if (mutual) {
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t j = 0; j < n2; j++) {
            rev[j] = -1;
            if (f2[j] == 2) continue;
            int dmin = 0x7fffffff;
            for (int64_t i = 0; i < n1; i++) {
                if (f1[i]) continue;
                const int dist = f2[j] == 1 ? 0 : so_l1(k1[i].desc, k2[j].desc);
                if (dist < dmin) { dmin = dist; rev[j] = (int32_t)i; }
            }
        }
    }
    int64_t count = 0;
    for (int64_t i = 0; i < n1; i++) {
        if (best[i] < 0) continue;
        if (mutual && rev[best[i]] != (int32_t)i) continue;
        if (count < cap) { pairs[2 * count] = (int32_t)i; pairs[2 * count + 1] = best[i]; }
        count++;
    }
    free(best); free(f1); free(f2); free(rev);
    return count;
}

/* ------------------------------------------------------------------ affine warp  (transform.cl:22-110, 116-204)
 * out[y][x] = bilinear(image, M*(y,x) + off) with the reference's quirks: (y,x) order of the matrix rows
 * ("Fortran convention"), reads guarded by fill at the right/bottom edge, and the final half-pixel cut.
 * matrix = {m0,m1,m2,m3}: ty = m0*y + m1*x + off[0];  tx = m2*y + m3*x + off[1]. */
static inline float so_transform_sample(float tx, float ty, int W, int H, float fill, int mode,
                                        float p, float px, float py, float pn) {
    (void)H; (void)W; (void)mode; (void)fill;
    const int tx_prev = (int)tx, tx_next = tx_prev + 1, ty_prev = (int)ty, ty_next = ty_prev + 1;
    const float i1 = ((float)tx_next - tx) * p + (tx - (float)tx_prev) * px;
    const float i2 = ((float)tx_next - tx) * py + (tx - (float)tx_prev) * pn;
    return ((float)ty_next - ty) * i1 + (ty - (float)ty_prev) * i2;
}

void so_transform(const float *image, float *out, const float *matrix, const float *offset, int W, int H,
                  int OW, int OH, float fill, int mode) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < OH; y++)
        for (int x = 0; x < OW; x++) {
            float tx = matrix[2] * (float)y + matrix[3] * (float)x;
            float ty = matrix[0] * (float)y + matrix[1] * (float)x;
            tx += offset[1]; ty += offset[0];
            float interp = fill;
            if (0.0f <= tx && tx < (float)W && 0.0f <= ty && ty < (float)H) {
                const int tx_prev = (int)tx, tx_next = tx_prev + 1, ty_prev = (int)ty, ty_next = ty_prev + 1;
                if (mode == 1) {
                    const float p = image[(size_t)ty_prev * W + tx_prev];
                    const float px = tx_next >= W ? fill : image[(size_t)ty_prev * W + tx_next];
                    const float py = ty_next >= H ? fill : image[(size_t)ty_next * W + tx_prev];
                    const float pn = (tx_next >= W || ty_next >= H) ? fill : image[(size_t)ty_next * W + tx_next];
                    interp = so_transform_sample(tx, ty, W, H, fill, mode, p, px, py, pn);
                } else interp = image[(size_t)ty_prev * W + tx_prev];
            }
            if (tx >= (float)W + -0.5f) interp = fill;
            if (ty >= (float)H + -0.5f) interp = fill;
            out[(size_t)y * OW + x] = interp;
        }
}

void so_transform_rgb(const uint8_t *image, uint8_t *out, const float *matrix, const float *offset, int W, int H,
                      int OW, int OH, float fill, int mode) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < OH; y++)
        for (int x = 0; x < OW; x++) {
            float tx = matrix[2] * (float)y + matrix[3] * (float)x;
            float ty = matrix[0] * (float)y + matrix[1] * (float)x;
            tx += offset[1]; ty += offset[0];
            const int inside = (0.0f <= tx && tx < (float)W && 0.0f <= ty && ty < (float)H);
            const int tx_prev = (int)tx, tx_next = tx_prev + 1, ty_prev = (int)ty, ty_next = ty_prev + 1;
            for (int c = 0; c < 3; c++) {
                float interp = fill;
                if (inside) {
                    if (mode == 1) {
                        const float p = (float)image[3 * ((size_t)ty_prev * W + tx_prev) + c];
                        const float px = tx_next >= W ? fill : (float)image[3 * ((size_t)ty_prev * W + tx_next) + c];
                        const float py = ty_next >= H ? fill : (float)image[3 * ((size_t)ty_next * W + tx_prev) + c];
                        const float pn = (tx_next >= W || ty_next >= H) ? fill : (float)image[3 * ((size_t)ty_next * W + tx_next) + c];
                        interp = so_transform_sample(tx, ty, W, H, fill, mode, p, px, py, pn);
                    } else interp = (float)image[3 * ((size_t)ty_prev * W + tx_prev) + c];
                }
                if (tx >= (float)W + -0.5f) interp = fill;
                if (ty >= (float)H + -0.5f) interp = fill;
                out[3 * ((size_t)y * OW + x) + c] = (uint8_t)interp;
            }
        }
}
