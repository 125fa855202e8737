// ref_shim.cpp -- TEST INFRASTRUCTURE.
//
// Host-side stand-in for an OpenCL runtime, so that the reference's OWN kernels
// (compiled from /root/reference/openCL/*.cl to x86-64 objects by oracle/Makefile, never
// copied into this repository) can be executed natively and used to pin the oracle.
//
// It provides (i) the OpenCL work-item / math / atomic builtins the kernel objects import
// (Itanium-mangled names given through asm labels), backed by glibc libm, and (ii) one
// `ref_<kernel>` entry point per reference kernel that runs a serial NDRange over it with the
// launch geometry of sift-src/plan.py / match.py.  `gaussian` uses work-group barriers, so it
// runs one pthread per work-item with a real barrier.
//
// Conventions: kernels built with -ffp-contract=off, rsqrt(x) = 1/sqrtf(x).
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <pthread.h>
#include <vector>

namespace {
thread_local size_t t_gid[3] = {0, 0, 0};
thread_local size_t t_lid[3] = {0, 0, 0};
thread_local size_t t_grp[3] = {0, 0, 0};
thread_local size_t t_lsz[3] = {1, 1, 1};
thread_local size_t t_ngr[3] = {1, 1, 1};
pthread_barrier_t *g_barrier = nullptr;
}  // namespace

// ---- work-item builtins -------------------------------------------------------------------
size_t cl_get_global_id(unsigned d) __asm__("_Z13get_global_idj");
size_t cl_get_global_id(unsigned d) { return d < 3 ? t_gid[d] : 0; }
size_t cl_get_local_id(unsigned d) __asm__("_Z12get_local_idj");
size_t cl_get_local_id(unsigned d) { return d < 3 ? t_lid[d] : 0; }
size_t cl_get_group_id(unsigned d) __asm__("_Z12get_group_idj");
size_t cl_get_group_id(unsigned d) { return d < 3 ? t_grp[d] : 0; }
size_t cl_get_local_size(unsigned d) __asm__("_Z14get_local_sizej");
size_t cl_get_local_size(unsigned d) { return d < 3 ? t_lsz[d] : 1; }
size_t cl_get_num_groups(unsigned d) __asm__("_Z14get_num_groupsj");
size_t cl_get_num_groups(unsigned d) { return d < 3 ? t_ngr[d] : 1; }
void cl_barrier(unsigned) __asm__("_Z7barrierj");
void cl_barrier(unsigned) { if (g_barrier) pthread_barrier_wait(g_barrier); }
int cl_atomic_inc(volatile int *p) __asm__("_Z10atomic_incPU8CLglobalVi");
int cl_atomic_inc(volatile int *p) { return __atomic_fetch_add(p, 1, __ATOMIC_SEQ_CST); }

// ---- math builtins ---------------------------------------------------------------------------
// Default build (libsiftclref.so): glibc libm.  -DREF_SIFTMATH (libsiftclref_sm.so): the five transcendental
// builtins are bound to the oracle's siftmath functions instead.  OpenCL leaves their last bits to the
// implementation, so either binding is "the reference"; the second one isolates libm: with it the reference's own
// kernels must reproduce the oracle byte for byte, i.e. every residual of the default build is a libm last-bit choice.
#ifdef REF_SIFTMATH
extern "C" {
#include "oracle_math.h"
}
static inline float sm_sin(float x) { float s, c; om_sincosf(x, &s, &c); return s; }
static inline float sm_cos(float x) { float s, c; om_sincosf(x, &s, &c); return c; }
float cl_exp(float x) __asm__("_Z3expf");       float cl_exp(float x) { return om_expf(x); }
float cl_sin(float x) __asm__("_Z3sinf");       float cl_sin(float x) { return sm_sin(x); }
float cl_cos(float x) __asm__("_Z3cosf");       float cl_cos(float x) { return sm_cos(x); }
// the only pow() of the hot path is pow(2.0f, y) (image.cl:354)
float cl_pow(float x, float y) __asm__("_Z3powff"); float cl_pow(float x, float y) { return x == 2.0f ? om_exp2f(y) : powf(x, y); }
#else
float cl_exp(float x) __asm__("_Z3expf");       float cl_exp(float x) { return expf(x); }
float cl_sin(float x) __asm__("_Z3sinf");       float cl_sin(float x) { return sinf(x); }
float cl_cos(float x) __asm__("_Z3cosf");       float cl_cos(float x) { return cosf(x); }
float cl_pow(float x, float y) __asm__("_Z3powff"); float cl_pow(float x, float y) { return powf(x, y); }
#endif
float cl_fabs(float x) __asm__("_Z4fabsf");     float cl_fabs(float x) { return fabsf(x); }
float cl_sqrt(float x) __asm__("_Z4sqrtf");     float cl_sqrt(float x) { return sqrtf(x); }
float cl_rsqrt(float x) __asm__("_Z5rsqrtf");   float cl_rsqrt(float x) { return 1.0f / sqrtf(x); }
float cl_atan2(float y, float x) __asm__("_Z5atan2ff");
#ifdef REF_SIFTMATH
float cl_atan2(float y, float x) { return om_atan2f(y, x); }
#else
float cl_atan2(float y, float x) { return atan2f(y, x); }
#endif
float cl_fmax(float a, float b) __asm__("_Z4fmaxff"); float cl_fmax(float a, float b) { return fmaxf(a, b); }
float cl_fmin(float a, float b) __asm__("_Z4fminff"); float cl_fmin(float a, float b) { return fminf(a, b); }
typedef float cl_float2 __attribute__((vector_size(8)));
float cl_dot2(cl_float2 a, cl_float2 b) __asm__("_Z3dotDv2_fS_"); float cl_dot2(cl_float2 a, cl_float2 b) { return a[0] * b[0] + a[1] * b[1]; }
unsigned cl_minu(unsigned a, unsigned b) __asm__("_Z3minjj"); unsigned cl_minu(unsigned a, unsigned b) { return a < b ? a : b; }

// ---- the reference kernels (symbols defined by the .cl objects) ----------------------------
extern "C" {
void gaussian(float *data, float sigma, int SIZE);
void max_min_serial(const float *data, unsigned SIZE, float *maximum, float *minimum);
void normalizes(float *image, const float *min_in, const float *max_in, const float *max_out, int W, int H);
void shrink(const float *in, float *out, int sw, int sh, int LW, int LH, int SW, int SH);
void u8_to_float(const unsigned char *in, float *out, int W, int H);
void u16_to_float(const unsigned short *in, float *out, int W, int H);
void u32_to_float(const unsigned int *in, float *out, int W, int H);
void u64_to_float(const unsigned long *in, float *out, int W, int H);
void s32_to_float(const int *in, float *out, int W, int H);
void s64_to_float(const long *in, float *out, int W, int H);
void rgb_to_float(const unsigned char *in, float *out, int W, int H);
void horizontal_convolution(const float *in, float *out, float *filter, int hlen, int W, int H);
void vertical_convolution(const float *in, float *out, float *filter, int hlen, int W, int H);
void combine(float *u, float a, float *v, float b, float *w, int dog, int W, int H);
void compact(void *kps, void *out, int *counter, int start, int end);
void compute_gradient_orientation(float *igray, float *grad, float *ori, int W, int H);
void local_maxmin(float *DOGS, void *out, int border, float peak, int octsize, float ET0, float ET,
                  int *counter, int nb_kp, int scale, int W, int H);
void interp_keypoint(float *DOGS, void *kps, int start, int end, float peak, float InitSigma, int W, int H);
void orientation_assignment(void *kps, float *grad, float *ori, int *counter, int octsize, float OriSigma,
                            int nb_kp, int start, int end, int W, int H);
void descriptor(void *kps, unsigned char *desc, float *grad, float *ori, int octsize, int start,
                int *end, int W, int H);
void matching(void *k1, void *k2, void *matchings, int *counter, int max_nb, float ratio, int size1, int size2);
void matching_valid(void *k1, void *k2, char *valid, int roi_w, int roi_h, void *matchings, int *counter, int max_nb, float ratio, int size1, int size2);
void transform(float *image, float *output, void *matrix, void *offset, int W, int H, int OW, int OH, float fill, int mode);
void transform_RGB(unsigned char *image, unsigned char *output, void *matrix, void *offset, int W, int H, int OW, int OH, float fill, int mode);
}

namespace {
template <class F> void run1d(size_t n, F f) {
    for (size_t i = 0; i < n; i++) { t_gid[0] = i; t_gid[1] = 0; f(); }
}
template <class F> void run2d(size_t nx, size_t ny, F f) {
    for (size_t y = 0; y < ny; y++)
        for (size_t x = 0; x < nx; x++) { t_gid[0] = x; t_gid[1] = y; f(); }
}
struct GaussArgs { float *data; float sigma; int size; size_t lid, lsz; };
void *gauss_thread(void *p) {
    GaussArgs *a = static_cast<GaussArgs *>(p);
    t_lid[0] = a->lid; t_gid[0] = a->lid; t_lsz[0] = a->lsz; t_grp[0] = 0; t_ngr[0] = 1;
    gaussian(a->data, a->sigma, a->size);
    return nullptr;
}
}  // namespace

extern "C" {

// plan.py:321-330: one work-group of nextpower(size) work-items
int ref_gaussian(float *data, float sigma, int size, int wg) {
    pthread_barrier_t bar;
    if (pthread_barrier_init(&bar, nullptr, (unsigned)wg)) return -1;
    g_barrier = &bar;
    std::vector<pthread_t> th((size_t)wg);
    std::vector<GaussArgs> args((size_t)wg);
    for (int i = 0; i < wg; i++) {
        args[(size_t)i] = GaussArgs{data, sigma, size, (size_t)i, (size_t)wg};
        pthread_create(&th[(size_t)i], nullptr, gauss_thread, &args[(size_t)i]);
    }
    for (int i = 0; i < wg; i++) pthread_join(th[(size_t)i], nullptr);
    g_barrier = nullptr;
    pthread_barrier_destroy(&bar);
    return 0;
}

void ref_max_min_serial(const float *data, unsigned n, float *mx, float *mn) { max_min_serial(data, n, mx, mn); }

void ref_normalizes(float *img, const float *mn, const float *mx, const float *top, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { normalizes(img, mn, mx, top, W, H); });
}
void ref_shrink(const float *in, float *out, int LW, int LH, int SW, int SH) {
    run2d((size_t)SW, (size_t)SH, [&] { shrink(in, out, 2, 2, LW, LH, SW, SH); });
}
void ref_u8_to_float(const unsigned char *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { u8_to_float(in, out, W, H); });
}
void ref_u16_to_float(const unsigned short *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { u16_to_float(in, out, W, H); });
}
void ref_u32_to_float(const unsigned int *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { u32_to_float(in, out, W, H); });
}
void ref_u64_to_float(const unsigned long *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { u64_to_float(in, out, W, H); });
}
void ref_s32_to_float(const int *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { s32_to_float(in, out, W, H); });
}
void ref_s64_to_float(const long *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { s64_to_float(in, out, W, H); });
}
void ref_rgb_to_float(const unsigned char *in, float *out, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { rgb_to_float(in, out, W, H); });
}
void ref_horizontal_convolution(const float *in, float *out, float *filt, int n, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { horizontal_convolution(in, out, filt, n, W, H); });
}
void ref_vertical_convolution(const float *in, float *out, float *filt, int n, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { vertical_convolution(in, out, filt, n, W, H); });
}
void ref_combine(float *u, float a, float *v, float b, float *w, int dog, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { combine(u, a, v, b, w, dog, W, H); });
}
void ref_compute_gradient_orientation(float *img, float *grad, float *ori, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { compute_gradient_orientation(img, grad, ori, W, H); });
}
void ref_local_maxmin(float *dogs, void *out, int border, float peak, int octsize, float et0, float et,
                      int *counter, int nb_kp, int scale, int W, int H) {
    run2d((size_t)W, (size_t)H, [&] { local_maxmin(dogs, out, border, peak, octsize, et0, et, counter, nb_kp, scale, W, H); });
}
void ref_interp_keypoint(float *dogs, void *kps, int start, int end, float peak, float init_sigma, int W, int H, int nitems) {
    run1d((size_t)nitems, [&] { interp_keypoint(dogs, kps, start, end, peak, init_sigma, W, H); });
}
void ref_compact(void *kps, void *out, int *counter, int start, int end, int nitems) {
    run1d((size_t)nitems, [&] { compact(kps, out, counter, start, end); });
}
void ref_orientation_assignment(void *kps, float *grad, float *ori, int *counter, int octsize, float ori_sigma,
                                int nb_kp, int start, int end, int W, int H, int nitems) {
    run1d((size_t)nitems, [&] { orientation_assignment(kps, grad, ori, counter, octsize, ori_sigma, nb_kp, start, end, W, H); });
}
void ref_descriptor(void *kps, unsigned char *desc, float *grad, float *ori, int octsize, int start, int *end,
                    int W, int H, int nitems) {
    run1d((size_t)nitems, [&] { descriptor(kps, desc, grad, ori, octsize, start, end, W, H); });
}
void ref_matching(void *k1, void *k2, void *matchings, int *counter, int max_nb, float ratio, int n1, int n2, int nitems) {
    run1d((size_t)nitems, [&] { matching(k1, k2, matchings, counter, max_nb, ratio, n1, n2); });
}

void ref_matching_valid(void *k1, void *k2, char *valid, int rw, int rh, void *matchings, int *counter, int max_nb, float ratio, int n1, int n2) {
    run1d((size_t)n1, [&] { matching_valid(k1, k2, valid, rw, rh, matchings, counter, max_nb, ratio, n1, n2); });
}

void ref_transform(float *image, float *out, void *matrix, void *offset, int W, int H, int OW, int OH, float fill, int mode) {
    run2d((size_t)OW, (size_t)OH, [&] { transform(image, out, matrix, offset, W, H, OW, OH, fill, mode); });
}
void ref_transform_RGB(unsigned char *image, unsigned char *out, void *matrix, void *offset, int W, int H, int OW, int OH, float fill, int mode) {
    for (size_t y = 0; y < (size_t)OH; y++)
        for (size_t x = 0; x < (size_t)OW; x++)
            for (size_t c = 0; c < 4; c++) { t_gid[0] = c; t_gid[1] = x; t_gid[2] = y; transform_RGB(image, out, matrix, offset, W, H, OW, OH, fill, mode); }
}

}  // extern "C"
