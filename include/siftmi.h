/*
 * siftmi.h -- C ABI of libsiftmi.so, the MI355X (gfx950) SIFT hot path.
 *
 * This is the drop-in boundary for the reference's SiftPlan.keypoints() / MatchPlan.match()
 * path.  The reference (pierrepaleo/sift_pyocl) has no FFI of its own: its boundary is
 * PyOpenCL -- pyopencl.Program(...).build() (sift-src/plan.py:364), kernel launches
 * program.kernel(queue, global, local, *args) (e.g. plan.py:586-591, match.py:246-255),
 * pyopencl.array allocation (plan.py:276-293) and blocking enqueue_copy read-backs
 * (plan.py:452-468, 642, 751; match.py:258-261).  Each entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions: every function returns 0 on success, a negative SIFTMI_E* code otherwise, with a
 * message retrievable from siftmi_last_error() (thread-local).  Handles are opaque, bound to one
 * HIP device and one HIP stream, and must be used by one host thread at a time (the Python
 * wrapper holds the reference's per-plan semaphore, plan.py:156,439 / match.py:117,215).
 * "is_device" pointers are HIP device pointers on the plan's device; the call orders itself
 * after work already submitted to the NULL stream by synchronising the device once on entry
 * when a device pointer is passed.
 */
#ifndef SIFTMI_H
#define SIFTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFTMI_OK 0
#define SIFTMI_EINVAL (-1)   /* bad argument                       -> RuntimeError / AssertionError in Python */
#define SIFTMI_ENOMEM (-2)   /* device or host allocation failed   -> MemoryError  (plan.py:365-366)           */
#define SIFTMI_EDEVICE (-3)  /* HIP runtime error / no gfx950 GPU  -> RuntimeError                             */
#define SIFTMI_ECAPACITY (-4)/* output buffer too small (results truncated, *n_out = number written)           */

/* input pixel types accepted by SiftPlan (plan.py:99-106, 450-488) */
enum siftmi_dtype {
    SIFTMI_F32 = 0, SIFTMI_U8 = 1, SIFTMI_U16 = 2, SIFTMI_U32 = 3, SIFTMI_U64 = 4,
    SIFTMI_I32 = 5, SIFTMI_I64 = 6, SIFTMI_F64 = 7, SIFTMI_RGB8 = 8
};

/* 144-byte keypoint record == SiftPlan.dtype_kp / MatchPlan.dtype_kp (plan.py:110-115, match.py:70-75),
 * == t_keypoint of openCL/matching_cpu.cl:22-25 */
typedef struct siftmi_keypoint {
    float x, y, scale, angle;
    uint8_t desc[128];
} siftmi_keypoint;

/* SIFT parameters read from sift_pyocl.param.par at call time (param.py:52-79) */
typedef struct siftmi_params {
    double init_sigma;    /* SiftPlan(init_sigma=) or par.InitSigma; Python double (plan.py:129-131) */
    float peak_thresh;    /* par.PeakThresh  -> local_maxmin / interp_keypoint (plan.py:631, 648)    */
    float edge_thresh0;   /* par.EdgeThresh1 -> kernel slot EdgeThresh0, octsize<=1 (plan.py:633)    */
    float edge_thresh;    /* par.EdgeThresh  -> kernel slot EdgeThresh (plan.py:634)                 */
    float ori_sigma;      /* par.OriSigma    (plan.py:681)                                           */
    int32_t border_dist;  /* par.BorderDist  (plan.py:630)                                           */
    int32_t octave_max;   /* 0 = every octave (reference behaviour, plan.py:213-224); >0 = extension */
    int32_t pix_per_kp;   /* SiftPlan.PIX_PER_KP: kpsize = H*W / pix_per_kp (plan.py:109, 243)       */
    int32_t double_im_size; /* par.DoubleImSize != 0: the frame counts as blurred by sigma 1.0 instead of 0.5, i.e. the
                             * initial blur is sqrt(init_sigma^2 - 1) wide (plan.py:254, 297, 534) -- all the reference does
                             * with that parameter; the image is not resampled                                           */
} siftmi_params;

typedef struct siftmi_plan siftmi_plan;
typedef struct siftmi_matcher siftmi_matcher;

/* ---- runtime ------------------------------------------------------------------------------
 * replaces sift-src/clinit.py device enumeration (ocl.select_device, clinit.py:360-400) */
int siftmi_device_count(void);
int siftmi_device_name(int device_id, char *buf, int64_t buflen);
const char *siftmi_last_error(void);
const char *siftmi_version(void);

/* ---- SiftPlan ------------------------------------------------------------------------------
 * siftmi_plan_create    <- SiftPlan.__init__ (plan.py:117-201): sizes the pyramid (_calc_scales
 *                          :213), allocates every device buffer (_allocate_buffers :268) and the six
 *                          Gaussian tap vectors (_init_gaussian :308).
 * siftmi_plan_keypoints <- SiftPlan.keypoints (plan.py:432-567) including _one_octave (:596-756) and
 *                          _compact (:758-795); one host<->device synchronisation per image instead of
 *                          >= 18 per octave.  Output order within the result is unspecified (as in the
 *                          reference, whose kernels append through atomic_inc).
 *                          image_dtype must be the plan's dtype or SIFTMI_F32 (plan.py:444 accepts both).
 * siftmi_plan_get_minmax<- buffers["min"] / buffers["max"] (plan.py:286-287; read by alignment.py:345).
 * siftmi_plan_profile   <- SiftPlan.log_profile (plan.py:826-847): "label\tms\n" lines of the last call.
 */
int siftmi_plan_create(int32_t height, int32_t width, int32_t in_dtype, int32_t device_id,
                       const siftmi_params *params, int32_t profile, siftmi_plan **out);
int siftmi_plan_info(const siftmi_plan *plan, int32_t *n_octaves, int64_t *kpsize, int64_t *bytes_allocated);
/* Capacity as the reference (plan.py:243, 797-804): kpsize = H*W / PIX_PER_KP entries PER OCTAVE -- for the candidates of a
 * detection scale appended behind the octave's oriented keypoints so far, and for those oriented keypoints.  An image within
 * that rule returns every record (up to n_octaves * kpsize of them); one beyond it raises `overflow` and keeps at most kpsize
 * records of each octave (the reference silently drops whatever its atomic counter places beyond the buffer, image.cl:203-205).
 * The device lists start at kpsize entries and grow when an image needs more (it is then run again inside the same call):
 * siftmi_plan_capacity reports the record list's current size and how often a list has grown. */
int siftmi_plan_capacity(const siftmi_plan *plan, int64_t *records, int64_t *growths);
/* How often the one-launch form of the small octaves (a workgroup per octave, chained through a flag with a bounded wait) gave up
 * waiting and the image ran again octave by octave -- the result is the same either way, the plan keeps the per-octave launches
 * from then on (*tail_enabled 0).  No reference counterpart: the reference launches every octave's kernels one by one. */
int siftmi_plan_tail_timeouts(const siftmi_plan *plan, int64_t *timeouts, int32_t *tail_enabled);
int siftmi_plan_set_params(siftmi_plan *plan, const siftmi_params *params);
/* Tuning / diagnostic option of one plan by name (the reference's counterparts are constructor keywords such as
 * max_workgroup_size, plan.py:117-131).  Results never depend on an option.  Unknown name -> SIFTMI_EINVAL.  Names:
 *   launch shapes      "march", "march_wgs", "xcd_map" (marching blur, extrema: every XCD takes a contiguous range of tiles; default 1), "march_prio" (marching blur: wave priority falls with a workgroup's progress -- 0 never, 1 = default: launches of about three workgroups per CU and the later octaves' chains, 2 every launch), "mm_blocks", "mm_threads", "ext_rows", "ext_strips",
 *                      "ori_blocks", "ori_small_blocks", "ori_pad", "ori_team", "desc_blocks", "desc_small_blocks", "desc_early_blocks",
 *                      "desc_dense_blocks", "desc_pad", "desc_team", "desc_dynamic", "desc_stream", "maps_blocks"
 *   kernel forms       "fused_convert", "fused_shrink", "fused_refine", "tail", "tail_pixels",
 *                      "maps" (0 never / 1 always / 2 by the previous image's count), "maps_density"
 *   stream schedule    "overlap" (0: one stream), "fork" (the octaves below octave 0 as two chains and groups -- octave 1 | the rest: 0 never, 1 always, 2 from five octaves),
 *                      "early_chain" (that chain starts at plane 3 of octave 0: 0 never, 1 always, 2 unless the previous image was keypoint-rich),
 *                      "split" (frames whose later octaves form ONE chain: the octaves below octave 1 built and searched on a stream of their own, one
 *                      orientation / descriptor launch for the group; default 0), "spin"
 *   diagnostics        "host_timing", "tail_fault" (treat the next n tail launches as timed out: exercises the re-run path) */
int siftmi_plan_set_option(siftmi_plan *plan, const char *name, int64_t value);
/* out_is_device of siftmi_plan_keypoints: where the result array lives.  SIFTMI_OUT_PINNED = pinned host memory from
 * siftmi_host_alloc: the descriptor kernels write every record straight into it while they run (zero-copy over PCIe),
 * so no device-to-host copy follows the last kernel (the reference copies keypoints and descriptors of every octave
 * back with blocking reads, plan.py:541-567). */
#define SIFTMI_OUT_HOST 0
#define SIFTMI_OUT_DEVICE 1
#define SIFTMI_OUT_PINNED 2
/* Pinned, device-writable host blocks from a size-bucketed pool (a power of two from 64 KiB to 1 MiB, a multiple of 2 MiB
 * above).  The reference returns ordinary numpy arrays (plan.py:553-565); page-locked result arrays are this build's way to
 * have no copy after the last kernel, and they cannot be swapped out: the pool holds at most `limit` bytes (default 2 GiB,
 * live + spare) and siftmi_host_alloc returns SIFTMI_ENOMEM beyond it -- the Python layer then hands out an ordinary array.
 * siftmi_host_free never calls the driver (it runs from destructors; hipHostFree synchronises the device): surplus blocks
 * are released by the next siftmi_host_alloc or by siftmi_host_pool_trim(keep_bytes). */
int siftmi_host_alloc(int64_t bytes, void **out);
int siftmi_host_free(void *ptr);
int siftmi_host_pool_limit(int64_t limit_bytes /* < 0: query only */, int64_t *live_bytes, int64_t *spare_bytes);
int siftmi_host_pool_trim(int64_t keep_bytes);
int siftmi_plan_keypoints(siftmi_plan *plan, const void *image, int32_t image_dtype, int32_t image_is_device,
                          siftmi_keypoint *out, int32_t out_is_device, int64_t capacity, int64_t *n_out,
                          int32_t *overflow);
/* two-step variant: siftmi_plan_keypoints(..., out = NULL, capacity = 0, ...) only returns the count and
 * leaves the records on the device; siftmi_plan_fetch copies records [first, first+count) of the last call
 * (to a host buffer, or to a device buffer with out_is_device) -- saves one host-side copy of the result */
int siftmi_plan_fetch(siftmi_plan *plan, siftmi_keypoint *out, int32_t out_is_device, int64_t first, int64_t count);
/* device address and count of the records of the last call; valid until the next siftmi_plan_keypoints on this plan
 * (lets MatchPlan consume them in place, as the reference matches pyopencl arrays: alignment.py:155-157,250) */
int siftmi_plan_records_device(const siftmi_plan *plan, const siftmi_keypoint **records, int64_t *count);
/* Affine warp with bilinear interpolation of an image of the plan's shape -- the `transform` / `transform_RGB`
 * kernels (openCL/transform.cl:22, :116) as LinearAlign.align launches them (sift-src/alignment.py:325-348).
 *   out[y][x] = bilinear(image, (ty, tx)),  ty = matrix[0]*y + matrix[1]*x + offset[0],
 *                                           tx = matrix[2]*y + matrix[3]*x + offset[1]
 * with `fill` outside the image, for taps right of / below it, and where tx >= W-0.5 or ty >= H-0.5.
 *   image        H x W float32 (channels 1) or H x W x 3 uint8 (channels 3); NULL = the host image most recently
 *                handed to siftmi_plan_keypoints, still staged on the device (the reference's buffers["input"])
 *   out          OH x OW (x3) of the same element type; the whole output is written (the reference only
 *                launches W x H work-items and leaves the `extra` margin of its output buffer undefined)
 *   mode         1 = bilinear (the only value the reference passes), else nearest-lower tap
 *   kernel_ms    optional, hipEvent duration of the kernel */
int siftmi_plan_transform(siftmi_plan *plan, const void *image, int32_t image_is_device, int32_t channels, void *out,
                          int32_t out_is_device, int32_t out_width, int32_t out_height, const float *matrix /*[4]*/,
                          const float *offset /*[2]*/, float fill, int32_t mode, double *kernel_ms);
int siftmi_plan_get_minmax(const siftmi_plan *plan, float *min_out, float *max_out);
int siftmi_plan_profile(const siftmi_plan *plan, char *buf, int64_t buflen);
/* device time (ms, hipEvent on the plan's stream) of the kernels of the last keypoints() call,
 * excluding host<->device copies; requires a profile level at creation.  Under the LIGHT level (profile = 1: one event
 * pair around the octave-0 blur launches and nothing else -- every further event record is a bubble between kernels)
 * the first and last kernels are not bracketed and *total_ms is 0: only the blur figures are measured. */
int siftmi_plan_last_kernel_ms(const siftmi_plan *plan, float *total_ms, float *blur_ms, int32_t *blur_launches,
                               double *blur_pixels);
/* the same restricted to the blur launches of one octave (octave < 0: all).  Octave-0 launches never run
 * concurrently with another kernel of the plan, later octaves overlap the detection stream. */
int siftmi_plan_blur_ms(const siftmi_plan *plan, int32_t octave, float *blur_ms, int32_t *blur_launches,
                        double *blur_pixels);
/* running totals of the two figures above over every keypoints() call since the last reset (light profile): what a
 * benchmark loop reads ONCE after its timed region instead of querying events after every call.
 *   calls, total_ms (first -> last kernel of each call, summed; 0 under the light level, see above), blur0_ms /
 *   blur0_launches / blur0_pixels (octave 0) */
int siftmi_plan_profile_totals(siftmi_plan *plan, int32_t reset, int64_t *calls, double *total_ms, double *blur0_ms,
                               int64_t *blur0_launches, double *blur0_pixels);
int siftmi_plan_destroy(siftmi_plan *plan);

/* ---- batched, pipelined keypoints -----------------------------------------------------------
 * The reference processes one image per SiftPlan.keypoints call and blocks on >= 18 counter read-backs per octave
 * (plan.py:642,689,731,767,777); a stack of frames is a Python loop (scripts/sift_pyocl.py, LinearAlign).  The batch
 * handle is the throughput form of that loop: `lanes` independent plans take the images round-robin,
 * nothing waits until a lane is reused, the records of the whole batch are parked on the device and handed back by one copy.
 *   siftmi_batch_keypoints  images[n]: all host or all device pointers of the batch's shape / dtype (or float32);
 *                           counts[n], offsets[n] (in records, into the parked result) and *total are returned
 *   siftmi_batch_fetch      copies records [first, first+count) of the parked result (host or device destination) */
typedef struct siftmi_batch siftmi_batch;
int siftmi_batch_create(int32_t height, int32_t width, int32_t in_dtype, int32_t device_id, const siftmi_params *params,
                        int32_t lanes, siftmi_batch **out);
int siftmi_batch_destroy(siftmi_batch *batch);
int siftmi_batch_set_params(siftmi_batch *batch, const siftmi_params *params);
int siftmi_batch_info(const siftmi_batch *batch, int32_t *lanes, int64_t *bytes_allocated);
/* siftmi_plan_tail_timeouts summed over the lanes; *lanes_with_tail: lanes that still use the one-launch form */
int siftmi_batch_tail_timeouts(const siftmi_batch *batch, int64_t *timeouts, int32_t *lanes_with_tail);
/* light profiling of the lanes (level 1: one hipEvent pair around the full-resolution blur launches of every frame, as
 * siftmi_plan_create's profile = 1); siftmi_batch_blur_ms returns their sum over the frames of the last batch */
int siftmi_batch_set_profile(siftmi_batch *batch, int32_t level);
/* siftmi_plan_set_option on every lane of the batch */
int siftmi_batch_set_option(siftmi_batch *batch, const char *name, int64_t value);
int siftmi_batch_blur_ms(const siftmi_batch *batch, double *blur_ms, int64_t *blur_launches, double *blur_pixels);
int siftmi_batch_keypoints(siftmi_batch *batch, const void *const *images, int32_t n_images, int32_t image_dtype,
                           int32_t images_are_device, int64_t *counts, int64_t *offsets, int64_t *total, int32_t *overflow);
/* same, delivering records into caller-owned host arrays while the batch runs: frame i goes to host_outs[i] if its
 * count fits host_caps[i] (offsets[i] = -1), otherwise it stays parked on the device at offsets[i] for siftmi_batch_fetch;
 * *total_parked = records parked.  The blocking per-frame copy overlaps the other lanes' kernels. */
int siftmi_batch_keypoints_into(siftmi_batch *batch, const void *const *images, int32_t n_images, int32_t image_dtype,
                                int32_t images_are_device, siftmi_keypoint *const *host_outs, const int64_t *host_caps,
                                int64_t *counts, int64_t *offsets, int64_t *total_parked, int32_t *overflow);
int siftmi_batch_fetch(siftmi_batch *batch, siftmi_keypoint *out, int32_t out_is_device, int64_t first, int64_t count);

/* ---- MatchPlan -----------------------------------------------------------------------------
 * siftmi_match_create <- MatchPlan.__init__ (match.py:77-139)
 * siftmi_match        <- MatchPlan.match (match.py:200-271) with the `matching` kernel
 *                        (matching_cpu.cl:57-109): L1 distance, best/second-best, ratio test
 *                        dist1/dist2 < ratio_th.  *n_out = pairs written (<= capacity); *n_total = pairs
 *                        that passed (the reference silently drops the excess, match.py:252).
 */
int siftmi_match_create(int64_t size, int32_t device_id, int32_t profile, siftmi_matcher **out);
int siftmi_match(siftmi_matcher *plan, const siftmi_keypoint *kp1, int64_t n1, int32_t kp1_is_device,
                 const siftmi_keypoint *kp2, int64_t n2, int32_t kp2_is_device, float ratio_th,
                 int32_t *pairs, int64_t capacity, int64_t *n_out, int64_t *n_total);
/* ROI-masked and mutual-best variants.
 * siftmi_match_set_roi <- MatchPlan.set_roi / unset_roi (match.py:312-327): uploads the int8 mask (roi = NULL unsets).
 * siftmi_match_ex      <- the `matching_valid` kernel (matching_cpu.cl:136-199), which the reference compiles but its
 *                         host code never launches (match.py:246 always calls `matching`).
 *   roi_mode 0  no mask (== siftmi_match)
 *   roi_mode 1  `matching_valid` literally: a list-1 keypoint is dropped iff it lies inside the mask array on a zero
 *               pixel; a list-2 keypoint that is not inside the array on a non-zero pixel keeps competing with
 *               distance 0 (the kernel guards the accumulation, not the candidate); (c, r) = (int)x, (int)y
 *   roi_mode 2  strict (extension): keypoints of either list that are not on a non-zero mask pixel do not take part
 *   mutual      (extension) keep (i, j) only if i is also the nearest list-1 keypoint of j over the same masked
 *               distances, ties to the smallest index */
int siftmi_match_set_roi(siftmi_matcher *m, const int8_t *roi, int32_t roi_width, int32_t roi_height);
int siftmi_match_ex(siftmi_matcher *m, const siftmi_keypoint *kp1, int64_t n1, int32_t kp1_is_device,
                    const siftmi_keypoint *kp2, int64_t n2, int32_t kp2_is_device, float ratio_th, int32_t roi_mode,
                    int32_t mutual, int32_t *pairs, int64_t capacity, int64_t *n_out, int64_t *n_total);
int siftmi_match_last_kernel_ms(const siftmi_matcher *plan, float *ms);
/* profile != 0 at creation: device time in ms of the last call's stages, in the order of the events the reference
 * appends under profile=True (sift-src/match.py:226-263): ms4[0] "copy H->D KP_1", [1] "copy H->D KP_2", [2] "matching",
 * [3] "copy D->H match"; -1 where the stage did not run (device-resident list, no pair to copy). */
int siftmi_match_last_stage_ms(const siftmi_matcher *plan, float *ms4);
int siftmi_match_destroy(siftmi_matcher *plan);

/* ---- per-stage entry points (host pointers in/out; golden-vector replay, one reference kernel each)
 * gaussian.cl:56 | reductions.cl:62-241 + preprocess.cl:239 | convolution.cl:16,62 | algebra.cl:18 |
 * image.cl:119 | image.cl:235 + algebra.cl:57 | image.cl:47 | orientation_cpu.cl:41 |
 * keypoints_cpu.cl:36 | preprocess.cl:267 | preprocess.cl:53-223 */
int siftmi_stage_gaussian_taps(float sigma, int32_t size, float *out);
/* (host only, no reference counterpart) position of workgroup `id` of a grid of `n` in the XCD-contiguous order the marching
 * blur and the extrema pass use (csrc/k_xcd.hpp): the tile it works on.  A bijection of [0, n) for every n. */
int32_t siftmi_stage_xcd_order(int32_t id, int32_t n);
int siftmi_stage_minmax_normalize(int32_t device_id, const float *in, float *out, int32_t W, int32_t H,
                                  float *min_out, float *max_out);
int siftmi_stage_blur(int32_t device_id, const float *in, float *out, int32_t W, int32_t H,
                      const float *taps, int32_t ntaps);
/* the same stage with a plan's launch choices exposed (test hook for the large-plane kernels): `in` holds a frame of
 * `in_dtype` (SIFTMI_F32, or an integer / RGB8 code: those enter through the normalising 15-tap blur only, as in a plan);
 * norm != 0: min/max of the frame first (reductions.cl:62-241), then the blur with `normalizes` (preprocess.cl:239-252)
 * applied to its inputs; xcd_map: bit 0 = workgroup order of the marching kernel (option "xcd_map"), bit 1 set = its priority
 * feedback off (option "march_prio" 0); march_wgs: its workgroup count (0: default);
 * *kernel_used (may be null): 0 generic two-pass, 1 tiled kernel, 2 marching team kernel. */
int siftmi_stage_blur_ex(int32_t device_id, const void *in, int32_t in_dtype, float *out, int32_t W, int32_t H,
                         const float *taps, int32_t ntaps, int32_t norm, int32_t xcd_map, int32_t march_wgs,
                         int32_t *kernel_used);
int siftmi_stage_dog(int32_t device_id, const float *blur_a, const float *blur_b, float *out, int64_t n);
/* blurs: 6 planes (H,W); out: (capacity,4) floats (peak,row,col,scale) for scales 1..3 */
int siftmi_stage_local_maxmin(int32_t device_id, const float *blurs, int32_t W, int32_t H, int32_t octsize,
                              const siftmi_params *params, float *out, int64_t capacity, int64_t *n_out);
/* candidates (n,4) -> refined (peak,row,col,sigma) + detection scale, holes removed */
int siftmi_stage_interp(int32_t device_id, const float *blurs, int32_t W, int32_t H, const float *cand, int64_t n,
                        const siftmi_params *params, float *out, int32_t *out_scale, int64_t *n_out);
/* `compact` (openCL/algebra.cl:57-84, host side plan.py:758-795): rows [start, end) of kps (n,4) whose row field is not
 * -1 are moved up to follow the first `start` rows; out receives *n_out rows (start + survivors), survivors unordered */
int siftmi_stage_compact(int32_t device_id, const float *kps, int64_t n, int64_t start, int64_t end, float *out, int64_t *n_out);
int siftmi_stage_gradient(int32_t device_id, const float *img, float *grad, float *ori, int32_t W, int32_t H);
/* refined (n,4)+scale -> oriented (x,y,sigma*oct,angle)+scale, extras appended (capacity rows) */
int siftmi_stage_orientation(int32_t device_id, const float *blurs, int32_t W, int32_t H, int32_t octsize,
                             const float *kps, const int32_t *kp_scale, int64_t n, const siftmi_params *params,
                             float *out, int32_t *out_scale, int64_t capacity, int64_t *n_out);
int siftmi_stage_descriptor(int32_t device_id, const float *blurs, int32_t W, int32_t H, int32_t octsize,
                            const float *kps, const int32_t *kp_scale, int64_t n, uint8_t *desc);
int siftmi_stage_shrink(int32_t device_id, const float *in, float *out, int32_t W, int32_t H);
int siftmi_stage_convert(int32_t device_id, const void *in, int32_t in_dtype, float *out, int32_t W, int32_t H);
/* device versions of the "siftmath v1" functions, elementwise over n floats (test hook) */
int siftmi_stage_math(int32_t device_id, int32_t fn /*0 exp,1 exp2,2 sin,3 cos,4 atan2*/, const float *a,
                      const float *b, float *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
