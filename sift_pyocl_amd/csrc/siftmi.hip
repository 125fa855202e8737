// siftmi.hip -- host side of libsiftmi.so: the C ABI of include/siftmi.h over the gfx950 kernels.
//
// One plan = one HIP device + one stream + every buffer pre-allocated (as SiftPlan.__init__ does,
// sift-src/plan.py:117-201,268-306).  keypoints() enqueues the whole pyramid / detection /
// description chain without reading anything back and synchronises once at the end; the
// reference's host loop (plan.py:596-756) reads a 4-byte counter back >= 18 times per octave.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/siftmi.h"
#include "k_extrema.hpp"
#include "k_keypoint.hpp"
#include "k_descriptor.hpp"
#include "k_align.hpp"
#include "k_match.hpp"
#include "k_pyramid.hpp"
#include "k_tail.hpp"
#include "siftmath.hpp"

using namespace siftk;

static_assert(sizeof(siftmi_keypoint) == 144, "siftmi_keypoint must be 144 bytes");

namespace {

thread_local std::string g_err;

// Events that order one stream of a plan behind another ON THE SAME DEVICE, kernel to kernel: no timing, and no system-scope fence when they are
// recorded.  INVARIANT: producer and consumer of such an event are kernels on this device (ev_pyr, ev_p3, ev_maps0).  Anything whose
// consumer is a DMA copy, the host or another device -- the upload ring's events, the events bracketing a profile -- uses plain events.
// (The kernels' own agent-scope release / acquire at their boundaries is what the consumers need; nothing the host or another
// device reads is ordered by these events -- results are waited for through the streams themselves.)
#ifndef SIFT_SYNC_EVENT
#define SIFT_SYNC_EVENT (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
#define SIFTMI_ETAILRETRY (-100)   // internal: plan_wait -> siftmi_plan_keypoints, never returned through the C ABI
#define SIFTMI_EGROW (-101)        // internal, likewise: a list was grown, the image has to run again
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? SIFTMI_ENOMEM : SIFTMI_EDEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                \
    } while (0)

// ---- Gaussian taps on the host: gaussian.cl:56-140 run as one work-group of nextpower(size)
// items (plan.py:321-330); LDS tree sum restated serially in the same association order.
int nextpower(int n) { int p = 1; while (p < n) p <<= 1; return p; }

int gaussian_taps(float sigma, int size, float *out) {
    if (size < 1 || size > 1024) return fail(SIFTMI_EINVAL, "gaussian size %d out of range", size);
    int P = nextpower(size);
    if (P < 2) P = 2;
    std::vector<float> g((size_t)P, 0.0f), sum((size_t)P, 0.0f);
    const float norm = sqrtf(2.0f * SM_PI_F);
    for (int i = 0; i < size; i++) {
        const float x = ((float)i - ((float)size - 1.0f) / 2.0f) / sigma;
        const float y = siftmath::expf_(-x * x / 2.0f);
        g[(size_t)i] = y / sigma / norm;
        sum[(size_t)i] = g[(size_t)i];
    }
    for (int stride = 512; stride >= 2; stride >>= 1)
        if (size > stride)
            for (int i = 0; i < stride && i + stride < P; i++) sum[(size_t)i] += sum[(size_t)(i + stride)];
    sum[0] += sum[1];
    for (int i = 0; i < size; i++) out[i] = g[(size_t)i] / sum[0];
    return SIFTMI_OK;
}

int kernel_size(double sigma) {   // utils.py:54-64 with odd=True, cutoff=4
    int size = (int)std::ceil(2.0 * 4.0 * sigma + 1.0);
    if (size % 2 == 0) size += 1;
    return size;
}

struct Taps { int n = 0; float t[64] = {0}; float *dev = nullptr; };

// Per-plan tuning / diagnostic options (siftmi_plan_set_option).  Defaults are the measured best; nothing on the launch
// path reads the environment.
struct Options {
    int fused_convert = 1;   // typed frames converted at the point of use (0: separate convert pass)
    int overlap = 1;         // detection / description streams beside the pyramid stream (0: one stream)
    int march = 1;           // marching blur for large planes (0: tiled blur everywhere)
    int march_wgs = 0;       // workgroups wanted by the marching blur (0: 1024, 768 for 27 taps)
    int xcd_map = 1;         // marching blur: whole segment rows per XCD (k_pyramid.hpp: the strips' halo columns become L2 hits)
    int march_prio = 1;      // marching blur: wave priority falls by one level per quarter of a workgroup's march, so that the workgroups of a CU
                             // advance together instead of oldest first (k_pyramid.hpp: set_prio): 0 never, 1 by launch_team's rule, 2 every launch
    int ori_blocks = 4096, ori_pad = 0;      // orientation launch: workgroups (upper bound; the kernel cuts it down by the group's count)
    // descriptor launch: workgroups (keypoints are handed out dynamically, so a workgroup stays until the group is done:
    // 1024 = every wave slot of the chip, which starves the other stream's kernels for the whole launch -- 1024^2 smooth
    // frame 0.760 ms, with 896-960 workgroups 0.671; 2048^2 smooth 1.85 -> 1.73), bytes of dynamic LDS (residency throttle)
    int desc_blocks = 960, desc_pad = 0;
    int desc_small_blocks = 576;   // ... and for groups of fewer than 16384 (the later octaves' chain needs the room)
    int desc_dense_blocks = 960;   // ... and for groups of >= 65536 keypoints (lazy-gradient form; the MAPS form always takes desc_blocks).  Round 3: 832
                                   // (704: 5.47 ms per 154 k-keypoint call, 768-896: 5.27, 960: 5.31); round 4, with 2 KB less LDS per workgroup and the
                                   // row look-ahead: 832 / 896 / 960 -> 154 k keypoints without maps 4.56 / 4.49 / 4.42 ms, 16384^2 10.64 / 10.55 / 10.50
    int desc_stream = 0;     // 1: force the streaming form of the descriptor kernel (any window size)
    // Full gradient maps (gradient_maps_kernel) for the per-keypoint kernels: 0 never, 1 always, 2 when the previous image
    // of this plan had at least one oriented keypoint per `maps_density` pixels in the group (octave 0 / the later octaves).
    // Same records either way (tests/test_gpu_params.py); the maps pay from about one keypoint per 250 pixels (4096^2: 0.18 ms for octave 0 against 2.6 ns saved per keypoint).
    int maps = 2, maps_density = 200;
    int maps_blocks = 8192;  // workgroups of gradient_maps_kernel (grid stride over 256 x 32 pixel items)
    int mm_blocks = 256;     // few, fat workgroups: every block ends with two atomics on the same cache line
    int mm_threads = 1024;   // threads per workgroup of the f32 min/max pass: sixteen waves per workgroup = four per SIMD at the same 256 atomic pairs
                             // (256 threads: one wave per SIMD; three interleaved A/B runs 0.7924-0.7934 -> 0.7863-0.7876 ms, 512^2 0.2561 -> 0.2531)
    int desc_team = 2048;    // groups with fewer oriented keypoints than this are described by the workgroup-per-keypoint form (0: never); measured cross-over
                             // 1000-1800 alone (one workgroup slot per keypoint: 4 per CU); in a frame, round 4 (interleaved A/B, 1024 / 2048 / 3072 / 4096):
                             // 1024^2 smoothed 0.588 / 0.568 / 0.568 / 0.570 ms, 2048^2 white 0.500 / 0.499 / 0.520 / 0.521, headline and the other frames equal
    int fork = 2;            // the later octaves in two chains: octave 1 (detection to description, group 1) on `stream3`, the octaves below it
                             // (pyramids from octave 1's plane 3 on, the tail launch, group 2) on `stream2`.  0: one chain, one group; 1: always;
                             // 2: frames of at least five octaves.  Interleaved A/B, forked against one chain (white noise unless noted): 512^2
                             // 0.205 / 0.249 ms, 1024^2 0.281 / 0.328, 2048^2 0.450 / 0.508, 4096^2 with 9 / 5 / 4 / 3 octaves 0.943 / 0.956,
                             // 0.852 / 0.868, 0.830 / 0.833, 0.821 / 0.789 (three octaves: two sparse groups each pay a launch chain beside
                             // octave 0's descriptors), 16384^2 10.46 / 10.48; smoothed noise 4096^2 4.45 / 4.38, 2048^2 1.45 / 1.49
    int split = 0;           // (when the chain does not fork) the pyramids and detection of the octaves below octave 1 on `stream2` from octave 1's plane 3
                             // on, beside octave 1's last blurs and detection, into the SAME group: one orientation / descriptor launch for all later octaves
    int early_chain = 2;     // the later octaves' chain starts when plane 3 of octave 0 exists (behind its third blur: octave 1's plane 0 rides on
                             // that launch), not behind its fifth: 0 never, 1 always, 2 unless the previous image of the plan was keypoint-rich
                             // (one keypoint per `maps_density` pixels of octave 0).  Round 3 measured this slower (0.854 -> 0.894 ms): the later
                             // octaves' detection then still had to wait for octave 0's orientation pass (one shared list).  With a list per
                             // group the chain runs through: its pyramids share the chip with octave 0's last two blurs instead of with its
                             // descriptor launch, whose workgroups starve them of registers (blur_hv 21 / 27 taps on a 1024^2 plane: 83 / 115 us
                             // beside 768 descriptor workgroups, 10 / 11 alone).  Interleaved A/B, early against late: 512^2 0.194 / 0.205 ms,
                             // 1024^2 0.259 / 0.281, 2048^2 0.425 / 0.453 (3 octaves: 0.367 / 0.388), 4096^2 with 9 / 4 octaves 0.891-0.900 /
                             // 0.926, 0.817 / 0.841, with 3 (the headline) 0.785-0.799 / 0.789-0.794 (hence: not on >= 8 Mpixel frames of three
                             // octaves); smoothed noise 4096^2 4.42 / 4.26 (hence the density rule), 2048^2 1.39 / 1.38, 1024^2 0.576 / 0.596
    int desc_early_blocks = 704;   // descriptor workgroups of a small octave 0 when the later octaves started early (they are nearly done by
                                   // then: 576 leaves room nobody needs, 960 starves what is left; 640 / 704 / 768 / 832: 0.797 / 0.785 / 0.788-0.799 / 0.801)
    int fused_refine = 1;    // detection and refinement in one launch: 0 never, 1 planes below 1400^2, 2 every plane
    int fused_shrink = 1;    // octave hand-off inside the blur launch that writes plane 3 (512^2 frame -4 %, 2048^2 -4 %, 4096^2 +-0)
    int ori_team = 1024;     // groups with fewer refined keypoints than this: a workgroup per keypoint in the orientation launch (0: never)
    int desc_dynamic = 1;    // wave-per-keypoint form: keypoints beyond each wave's first are handed out through a device counter
    int tail_pixels = SIFT_TAIL_MAX_PIXELS;   // largest plane (W * H) the tail kernel takes
    int tail = 1;            // small octaves (<= 64 x 64) in one launch (octave_tail_kernel)
    int ext_rows = 0;        // rows per extrema strip: 0 by plane size (extrema_strip_rows)
    int ext_strips = 2000;   // ... which halves the 64-row strips until at least this many of them cover the plane.  Round 2 used 12288 (16-row
                             // strips on a 4096^2 plane): the march of a strip re-reads two halo rows, and many short strips are many
                             // ramp-ups; interleaved A/B, whole call: 4096^2 0.835 (12288) / 0.823 (4000) / 0.809 (2000) / 0.817 ms (1000),
                             // 2048^2 0.521 -> 0.511, 2048^2 smoothed 1.544 -> 1.517, small frames unchanged; the launch alone 94 -> 81-86 us
    int ori_small_blocks = 608;   // orientation launch: workgroups used for a group of fewer than 16384 keypoints (512 until the descriptor launch was ordered: 0.809 ms; 576-640: 0.799-0.801; 704: 0.813)
    int spin = 1;            // poll the ending streams instead of a blocking wait
    int host_timing = 0;     // print the host time of plan_enqueue
    int tail_fault = 0;      // diagnostic: the next `tail_fault` images that go through octave_tail_kernel are treated as if a
                             // workgroup of it had timed out (exercises the host's re-run path; results do not change)
};
const Options g_default_options{};

struct Event { std::string label; hipEvent_t a = nullptr, b = nullptr; bool is_blur = false; double pixels = 0; int octave = -1; int launches = 1; };

size_t dtype_size(int dt) {
    switch (dt) {
        case SIFTMI_F32: return 4; case SIFTMI_U8: return 1; case SIFTMI_U16: return 2; case SIFTMI_U32: return 4;
        case SIFTMI_U64: return 8; case SIFTMI_I32: return 4; case SIFTMI_I64: return 8; case SIFTMI_F64: return 8;
        case SIFTMI_RGB8: return 3;
    }
    return 0;
}

}  // namespace

// The lists of one group of octaves (k_keypoint.hpp: Counters): candidates, refined keypoints, oriented keypoints.  Capacities start at kpsize (plan.py:243) and grow on demand (grow_lists).
struct GroupLists {
    float4 *kp = nullptr;  int *kp_aux = nullptr;  int64_t cap_kp = 0;     // (peak, row, col, sigma), detection scale | octave << 8
    float4 *okp = nullptr; int *oaux = nullptr;    int64_t cap_out = 0;    // (x, y, scale, angle), same tag
    float4 *cand = nullptr; int64_t cap_cand = 0;                           // candidate list of the group's detection passes (one octave at a time)
};

struct siftmi_plan {
    void *chain = nullptr;        // open light-profile bracket (a Scope, profile == 1)
    int device = 0;
    hipStream_t stream = nullptr;
    int H = 0, W = 0, dtype = 0;
    siftmi_params par{};
    Options opt;
    int profile = 0;
    int n_oct = 0;
    std::vector<int> ow, oh;
    int64_t kpsize = 0;
    int64_t bytes = 0;
    float *planes = nullptr;      // all octaves' blur planes: octave o, scale s at plane(o, s)
    std::vector<size_t> oct_off;  // float offset of octave o's first plane
    float *tmp = nullptr;         // generic two-pass blur only (tap counts without a fused kernel): intermediate plane
    float *tmp_later = nullptr;   // ... of octave 1's chain, which may run beside octave 0's last two blurs (early_chain),
    float *tmp_below = nullptr;   // ... of the chain of the octaves below octave 1, which runs beside octave 1's last blurs
    hipStream_t stream2 = nullptr;            // octave 0's gradient maps; the octaves below octave 1 (pyramids, detection, the tail launch, description)
    hipStream_t stream3 = nullptr;            // octave 1 (pyramid, detection, description) -- or every later octave when they form one chain
    int64_t acc_calls = 0, acc_b0_launches = 0;   // running totals of the light profile (siftmi_plan_profile_totals)
    double acc_total_ms = 0, acc_b0_ms = 0, acc_b0_pixels = 0;
    hipEvent_t ev_early = nullptr;
    hipEvent_t ev_p3 = nullptr;               // plane 3 of octave 1 (and octave 2's plane 0) exist: the chain of the octaves below starts there
    std::vector<hipEvent_t> ev_pyr;           // pyramid of octave o complete (recorded on `stream`)
    bool overlap = true;
    float *plane(int o, int s) const { return planes + oct_off[(size_t)o] + (size_t)s * (size_t)ow[(size_t)o] * (size_t)oh[(size_t)o]; }
    void *raw = nullptr;          // host-input staging (any dtype)
    int raw_dtype = -1;           // dtype of the image currently staged in `raw` (-1: none)
    Counters *cnt_pair = nullptr;  // the two counter blocks (`cnt` points at the one of the image in flight)
    int cnt_parity = 0;
    hipStream_t fin = nullptr;    // stream on which the last enqueued image ends
    int last_group0 = 0;          // oriented keypoints of octave 0 in the previous image (occupancy heuristic)
    int last_group1 = 0;          // ... of the later octaves
    float *gmap = nullptr, *omap = nullptr;   // gradient maps (allocated on first use: half the size of `planes` each)
    size_t planes_floats = 0;
    bool maps_g0 = false, maps_g1 = false;    // the image being enqueued: MAPS forms for octave 0 / the later octaves (groups 1 and 2)
    // the image being enqueued / waited for
    bool early_cur = false;                   // the later octaves' chain started at plane 3 of octave 0
    bool split_cur = false;                   // the octaves below octave 1 are built and searched on stream2, octave 1's candidates sit in group 1's buffer (option "split")
    hipEvent_t ev_det2 = nullptr;             // ... and this is the end of their detection
    bool fork_cur = false;                    // octave 1 is a group of its own (1), the octaves below it group 2; else all later octaves in group 2
    unsigned groups_cur = 0;                  // bit g: group g has launches in this image
    int tail_first_cur = 0;                   // first octave of the tail launch (n_oct: none)
    hipEvent_t ev_maps0 = nullptr;
    bool maps_unavailable = false;   // the lazy allocation of the gradient maps failed once: dense frames keep the lazy forms
    hipEvent_t ev_join = nullptr;
    struct HostBack { Counters c[3]; } *hb = nullptr;    // pinned read-back blocks (one asynchronous D->H per ending stream)
    hipStream_t wait_s[3] = {nullptr, nullptr, nullptr}; // the streams the image enqueued last ends on (hb->c[k] was copied on wait_s[k])
    void *warp_in = nullptr, *warp_out = nullptr;   // siftmi_plan_transform staging, grown on demand
    size_t warp_in_bytes = 0, warp_out_bytes = 0;
    hipEvent_t ev_wa = nullptr, ev_wb = nullptr;
    float *conv = nullptr;        // converted f32 input when dtype != f32
    uint32_t *mm = nullptr;
    Counters *cnt = nullptr;
    float4 *tail_cand = nullptr;   // candidate lists of octave_tail_kernel: SIFT_TAIL_MAX_OCT x tail_cand_cap
    int tail_cand_cap = 0;
    GroupLists grp[SIFT_GROUPS];
    KpRecord *records = nullptr;  // the image's records: one block per group (descriptor_reserve)
    int64_t cap_rec = 0;
    bool in_flight = false;       // an image has been enqueued and not yet waited for (plan_wait) or drained
    bool records_cut = false;     // the last image hit the reference's per-octave capacity and was cut to it (cap_octaves): a pinned result array is stale
    int64_t grows = 0;            // list growths since creation (each one ran its image again)
    int64_t tail_timeouts = 0;    // images whose tail launch gave up waiting for the octave above and ran again (0 or 1: the plan then drops the tail launch)
    KpRecord *host_out = nullptr; // pinned result array of the call being enqueued (zero-copy delivery), or null
    int host_cap = 0;
    bool desc_rows = true;        // descriptor windows fit the row tables of descriptor_kernel (R <= SIFT_DESC_MAXRAD for this init_sigma)
    Taps taps[6];                 // [0..4] per-octave schedule, [5] initial blur
    bool have_init = false;
    std::vector<Event> events;
    size_t n_events = 0;
    hipEvent_t ev_first = nullptr, ev_last[3] = {nullptr, nullptr, nullptr};
    float last_min = 0, last_max = 0;
    int64_t last_count = 0;
    int last_overflow = 0;
    std::vector<void *> allocs;

    template <class T> int alloc(T **p, size_t nbytes) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, nbytes ? nbytes : 16);
        if (e != hipSuccess) return fail(SIFTMI_ENOMEM, "hipMalloc(%zu bytes): %s", nbytes, hipGetErrorString(e));
        allocs.push_back(q);
        bytes += (int64_t)nbytes;
        *p = (T *)q;
        return SIFTMI_OK;
    }
    // a buffer that may be replaced by a larger one later (not in `allocs`: freed by its owner)
    template <class T> int regrow(T **p, size_t old_bytes, size_t new_bytes) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, new_bytes ? new_bytes : 16);
        if (e != hipSuccess) return fail(SIFTMI_ENOMEM, "hipMalloc(%zu bytes): %s", new_bytes, hipGetErrorString(e));
        if (*p) { (void)hipFree(*p); bytes -= (int64_t)old_bytes; }
        bytes += (int64_t)new_bytes;
        *p = (T *)q;
        return SIFTMI_OK;
    }
};

namespace {

int compute_schedule(siftmi_plan *p) {
    // plan.py:534-539 (initial blur) and plan.py:602-618 (per-octave increments)
    const double init_sigma = p->par.init_sigma;
    p->have_init = false;
    // largest descriptor window: sigma <= init_sigma * 2^((3 + 1.5) / 3) (image.cl:354, |offset| <= 1.5), spacing = 3 sigma,
    // R = (int)(1.414 * spacing * 2.5 + 0.5) (keypoints_cpu.cl:57-62); two rows of slack for float rounding
    p->desc_rows = (int)(1.414 * 3.0 * init_sigma * std::pow(2.0, 1.5) * 2.5 + 0.5) + 2 <= SIFT_DESC_MAXRAD;
    const double cur_sigma = p->par.double_im_size ? 1.0 : 0.5;   // par.DoubleImSize (plan.py:534)
    if (init_sigma > cur_sigma) {
        const double s = std::sqrt(init_sigma * init_sigma - cur_sigma * cur_sigma);
        p->taps[5].n = kernel_size(s);
        if (p->taps[5].n > 64) return fail(SIFTMI_EINVAL, "init_sigma %g needs %d taps (> 64)", init_sigma, p->taps[5].n);
        int rc = gaussian_taps((float)s, p->taps[5].n, p->taps[5].t);
        if (rc) return rc;
        p->have_init = true;
    }
    const double ratio = std::pow(2.0, 1.0 / 3.0);   // SiftPlan.sigmaRatio, par.Scales == 3
    double prev = init_sigma;
    for (int s = 0; s < 5; s++) {
        const double inc = prev * std::sqrt(ratio * ratio - 1.0);
        p->taps[s].n = kernel_size(inc);
        if (p->taps[s].n > 64) return fail(SIFTMI_EINVAL, "sigma %g needs %d taps (> 64)", inc, p->taps[s].n);
        int rc = gaussian_taps((float)inc, p->taps[s].n, p->taps[s].t);
        if (rc) return rc;
        prev *= ratio;
    }
    for (int s = 0; s < 6; s++) {
        if (!p->taps[s].dev) { int rc = p->alloc(&p->taps[s].dev, 64 * sizeof(float)); if (rc) return rc; }
        HIPCHK(hipMemcpy(p->taps[s].dev, p->taps[s].t, 64 * sizeof(float), hipMemcpyHostToDevice));
    }
    return SIFTMI_OK;
}

// A launch whose completion IS an event (`stop` not null: hipExtLaunchKernelGGL binds the event to the dispatch's own
// completion signal) instead of a launch followed by hipEventRecord, which puts a barrier packet of its own into the queue:
// between two 20 us kernels of one stream 2.3 us instead of 3.9 (fence-free event) / 5.3-6.6 (timing event), against 1.3 us
// with no event at all; a stream waiting for the event starts 7.8 us after the producer's end instead of 9.6 / 11.7
// (tools/ubench/ext_event.hip).  hipEventElapsedTime between two bound events is end of the first launch -> end of the
// second (tools/ubench/ext_event2.hip): the light profile's bracket.
template <typename K, typename... A>
inline void launch_ev(hipEvent_t stop, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
    if (stop) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, nullptr, stop, 0u, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

template <int N, bool NORM, int DT, int TX, int TY, int VR>
void launch_blur_geom(hipStream_t st, const void *in, float *out, int W, int H, const TapsArg<N> &ta, const uint32_t *mm, float *half, hipEvent_t stop = nullptr) {
    using G = BlurGeom<N, TX, TY>;
    dim3 grid((unsigned)((W + TX - 1) / TX), (unsigned)((H + TY - 1) / TY));
    launch_ev(stop, blur_hv_kernel<N, NORM, DT, TX, TY, VR>, grid, dim3(256), (size_t)G::LDS_BYTES, st, in, out, W, H, ta, mm, half);
}

// Tile shape: planes that reach this kernel are narrower than 1024 columns or shorter than 512 rows (larger ones take the
// marching kernel); on all of them the 32 x 16 tile measured fastest (whole call, MI355X, round 2: 512^2 0.75 / 0.52 / 0.44 ms
// and 1020^2 0.90 / 0.63 / 0.53 ms for 128x64 / 64x32 / 32x16 tiles).
template <int N, bool NORM, int DT = 0>
void launch_blur_t(const Options &opt, hipStream_t st, const void *in, float *out, int W, int H, const float *taps, const uint32_t *mm, float *half = nullptr, hipEvent_t stop = nullptr) {
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = taps[i];
    (void)opt;
    launch_blur_geom<N, NORM, DT, 32, 16, 4>(st, in, out, W, H, ta, mm, half, stop);
}

// team form of the marching blur (blur_team_kernel): S sub-blocks per accumulator period, `wgs` workgroups wanted.
// A segment (grid row) outputs rows_out = ceil(H / segments) rows and marches rows_out + N - 1 rows rounded up to whole
// sub-blocks (nblocks - 1 full periods + last_subs sub-blocks), so the workgroup count is the one asked for -- 768 for
// 27 taps on a 4096^2 plane, 3 per CU -- instead of one quantised by segment heights in multiples of N rows (608: 2.4 per CU).
template <int N, bool NORM, int S, int DT = 0>
void launch_team(const Options &opt, hipStream_t st, const void *in, float *out, int W, int H, const float *taps, const uint32_t *mm, int wgs, float *half, hipEvent_t stop) {
    using G = March2Geom<N, 128, S>;
    using SS = SubSplit<N, S>;
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = taps[i];
    if (opt.march_wgs > 0) wgs = opt.march_wgs;
    const int gx = (W + G::TX - 1) / G::TX;
    int gy = wgs / gx;
    if (gy < 1) gy = 1;
    if (gy > H) gy = H;
    int rows_out = (H + gy - 1) / gy;
    if (rows_out < 2 * N + 1) rows_out = 2 * N + 1;     // small planes: at most one third of the marched rows is warm-up
    gy = (H + rows_out - 1) / rows_out;
    // rows marched by b full periods + m sub-blocks of the next one
    auto covered = [](int b, int m) { int r = b * N; for (int q = 0; q < m; q++) r += SS::rows(q); return r; };
    const int need = rows_out + N - 1;
    const int b = need / N;
    int m = 0;
    while (covered(b, m) < need) m++;                // m <= S
    const int nblocks = b + (m > 0 ? 1 : 0), last_subs = m > 0 ? m : S;
    dim3 grid((unsigned)gx, (unsigned)gy);
    // priority feedback (k_pyramid.hpp: set_prio), option march_prio: 1 = where a CU holds about three or more workgroups of the
    // launch (704 of them: the workgroups of a CU then advance together instead of oldest first), 2 = every launch (launch_blur
    // asks for it on the later octaves' chains, whose launches run beside octave 0's per-keypoint kernels and end the frame:
    // their waves outrank those), 0 = never.  Interleaved A/B, whole calls, off -> on: 4096^2 with 3 / 9 octaves 0.800 -> 0.766 /
    // 0.903 -> 0.865 ms, 3000^2 0.624 -> 0.610, 16384^2 10.47 -> 10.2 (its full-resolution launches 597 -> 530 us), smoothed
    // 4096^2 4.28 -> 4.22.  Octave 0 of a 2048^2 frame (536 workgroups) loses 3 % with it and keeps the arbiter's order.
    const int prio = (opt.march_prio == 2 || (opt.march_prio == 1 && gx * gy >= 704)) ? 1 : 0;
    launch_ev(stop, blur_team_kernel<N, NORM, S, DT>, grid, dim3(256), (size_t)3 * G::LDS_BYTES, st, in, out, W, H, nblocks, last_subs,
              rows_out, ta, mm, half, opt.xcd_map, prio);
}

// Large planes: the team form, with the sub-block count and workgroup count that measured best per tap count on a 4096^2
// plane (tools/ubench/blur_team.hip: 31 / 38 / 41 / 47 / 65 us against 33 / 43 / 45 / 53 / 73 us for the one-block form of round 1).
// returns whether `half` (the fused octave hand-off) was written
template <int N, bool NORM, int DT = 0>
bool launch_march_t(const Options &opt, hipStream_t st, const void *in, float *out, int W, int H, const float *taps, const uint32_t *mm, float *half = nullptr, hipEvent_t stop = nullptr) {
    constexpr int S = (N <= 15) ? 2 : (N <= 21 ? 3 : 4);
    launch_team<N, NORM, S, DT>(opt, st, in, out, W, H, taps, mm, N >= 27 ? 768 : 1024, half, stop);
    return half != nullptr;
}

// returns 0 when no tiled instantiation exists for this tap count, 1 when launched, 2 when launched and `half` written
// The marching kernels amortise their prologue over long strips; measured cross-over with the 32 x 16 tile kernel
// is near 1400^2 (whole call 0.607 vs 0.609 ms; 1024^2 0.555 vs 0.529, 2048^2 0.732 vs 0.779).
inline bool march_plane(int W, int H) { return W >= 1024 && H >= 512 && (int64_t)W * H >= 1400 * 1400; }

template <bool NORM>
int launch_blur_tiled(const Options &opt, hipStream_t st, const float *in, float *out, int W, int H, const Taps &t, const uint32_t *mm, float *half = nullptr, hipEvent_t stop = nullptr) {
    bool symmetric = true;
    for (int i = 0; i < t.n / 2; i++) symmetric = symmetric && (memcmp(&t.t[i], &t.t[t.n - 1 - i], 4) == 0);
    if constexpr (NORM) {
        // The normalising form exists for the default initial kernel only (15 taps, init_sigma = 1.6); any other
        // InitSigma sends its ONE initial blur through the generic two-pass path (return 0).
        if (t.n != 15) return 0;
        if (march_plane(W, H) && symmetric && opt.march) return launch_march_t<15, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop) ? 2 : 1;
        launch_blur_t<15, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop);
        return half ? 2 : 1;
    } else {
    if (march_plane(W, H) && symmetric && opt.march) {
        switch (t.n) {
            case 11: return launch_march_t<11, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop) ? 2 : 1;
            case 15: return launch_march_t<15, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop) ? 2 : 1;
            case 17: return launch_march_t<17, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop) ? 2 : 1;
            case 21: return launch_march_t<21, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop) ? 2 : 1;
            case 27: return launch_march_t<27, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop) ? 2 : 1;
            default: break;
        }
    }
    switch (t.n) {
        case 11: launch_blur_t<11, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop); return half ? 2 : 1;
        case 15: launch_blur_t<15, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop); return half ? 2 : 1;
        case 17: launch_blur_t<17, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop); return half ? 2 : 1;
        case 21: launch_blur_t<21, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop); return half ? 2 : 1;
        case 27: launch_blur_t<27, NORM>(opt, st, in, out, W, H, t.t, mm, half, stop); return half ? 2 : 1;
        default: return 0;
    }
    }
}

void launch_blur_generic(hipStream_t st, const float *in, float *out, float *tmp, int W, int H, const Taps &t,
                         const uint32_t *mm, bool norm) {
    dim3 grid((unsigned)((W + 255) / 256), (unsigned)H);
    hipLaunchKernelGGL(blur_generic_pass, grid, dim3(256), 0, st, in, tmp, W, H, t.dev, t.n, 0, mm, norm ? 1 : 0);
    hipLaunchKernelGGL(blur_generic_pass, grid, dim3(256), 0, st, (const float *)tmp, out, W, H, t.dev, t.n, 1, mm, 0);
}

// `half` not null: the launch may also write out[2y][2x] there (the next octave's plane 0); returns whether it did
// `stop` not null: the event is complete when the blur is (bound to the launch, launch_ev; recorded behind the two-pass form)
bool launch_blur(siftmi_plan *p, const float *in, float *out, int W, int H, const Taps &t, bool norm, hipStream_t st = nullptr, float *half = nullptr,
                 hipEvent_t stop = nullptr) {
    if (!st) st = p->stream;
    Options opt = p->opt;
    if (opt.march_prio == 1 && st != p->stream) opt.march_prio = 2;      // a later octave's chain (launch_team)
    const int r = norm ? launch_blur_tiled<true>(opt, st, in, out, W, H, t, p->mm, half, stop)
                       : launch_blur_tiled<false>(opt, st, in, out, W, H, t, p->mm, half, stop);
    if (!r) {       // one intermediate plane per stream that builds pyramids: the chains run beside one another
        float *tmp = (st == p->stream2 && p->tmp_below) ? p->tmp_below : ((st == p->stream3 && p->tmp_later) ? p->tmp_later : p->tmp);
        launch_blur_generic(st, in, out, tmp, W, H, t, p->mm, norm);
        if (stop) (void)hipEventRecord(stop, st);
    }
    return r == 2;
}

bool taps_symmetric(const Taps &t) {
    bool symmetric = true;
    for (int i = 0; i < t.n / 2; i++) symmetric = symmetric && (memcmp(&t.t[i], &t.t[t.n - 1 - i], 4) == 0);
    return symmetric;
}

// Initial blur reading a typed (integer / RGB) frame directly: instantiated for the default 15-tap initial
// kernel (init_sigma = 1.6); any other tap count goes through the convert pass.
template <int DT>
bool launch_init_blur_dt(const Options &opt, hipStream_t st, const void *in, float *out, int W, int H, const Taps &t, const uint32_t *mm) {
    if (t.n != 15) return false;
    if (march_plane(W, H) && taps_symmetric(t) && opt.march)
        launch_march_t<15, true, DT>(opt, st, in, out, W, H, t.t, mm);
    else
        launch_blur_t<15, true, DT>(opt, st, in, out, W, H, t.t, mm);
    return true;
}

// dispatch F(DT) over the typed-frame codes that have a fused path
#define SIFTMI_TYPED_DISPATCH(dt, CALL)                       \
    switch (dt) {                                             \
        case SIFTMI_U8: { constexpr int DT = 1; CALL; } break;   \
        case SIFTMI_U16: { constexpr int DT = 2; CALL; } break;  \
        case SIFTMI_U32: { constexpr int DT = 3; CALL; } break;  \
        case SIFTMI_U64: { constexpr int DT = 4; CALL; } break;  \
        case SIFTMI_I32: { constexpr int DT = 5; CALL; } break;  \
        case SIFTMI_I64: { constexpr int DT = 6; CALL; } break;  \
        case SIFTMI_RGB8: { constexpr int DT = 8; CALL; } break; \
        default: break;                                       \
    }

struct Scope {   // optional hipEvent bracket around one launch (profile=1: blur launches only; 2: every stage)
    siftmi_plan *p; size_t idx = (size_t)-1; hipStream_t st;
    bool stop_bound = false;      // the closing event rides on the bracket's last launch (launch_ev): nothing to record at the end
    // record_start false: the caller binds the opening event to the launch in FRONT of the bracket (its end opens it)
    Scope(siftmi_plan *pl, const char *label, bool is_blur = false, double pixels = 0, hipStream_t s = nullptr, int octave = -1, bool record_start = true) : p(pl) {
        st = s ? s : p->stream;
        if (!p->profile || (p->profile == 1 && !(is_blur && octave == 0))) return;
        if (p->n_events == p->events.size()) {
            Event e;
            hipEventCreate(&e.a); hipEventCreate(&e.b);
            p->events.push_back(e);
        }
        idx = p->n_events++;
        Event &e = p->events[idx];
        e.label = label; e.is_blur = is_blur; e.pixels = pixels; e.octave = octave; e.launches = 1;
        if (record_start) hipEventRecord(e.a, st);
    }
    ~Scope() { if (idx != (size_t)-1 && !stop_bound) hipEventRecord(p->events[idx].b, st); }
};

int grid_for(int64_t n, int block, int max_blocks) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

double contrast_threshold(const siftmi_params &par) { return 0.8 * (double)par.peak_thresh; }   // image.cl:152

OctaveTable octave_table(const siftmi_plan *p) {
    OctaveTable tab;
    memset(&tab, 0, sizeof tab);
    tab.base = p->planes;
    tab.gmap = p->gmap; tab.omap = p->omap;
    for (int o = 0; o < p->n_oct && o < SIFT_MAX_OCTAVES; o++) {
        tab.off[o] = (long long)p->oct_off[(size_t)o];
        tab.W[o] = p->ow[(size_t)o];
        tab.H[o] = p->oh[(size_t)o];
    }
    return tab;
}

// ---- list capacities -----------------------------------------------------------------------------------------------
// The reference gives EVERY OCTAVE a keypoint buffer of kpsize = H*W / PIX_PER_KP entries and one counter, reset per octave
// (plan.py:243, 797-804): the candidates of a detection scale are appended behind the octave's oriented keypoints so far,
// and the oriented keypoints of the scale behind those.  An image may therefore return up to kpsize records per octave.
// Here the lists of a group start at kpsize entries and grow when an image needs more (the image is then run again), up
// to what that rule can ever admit; `reference_overflow` evaluates the rule itself from the per-scale counts.
int group_of(const siftmi_plan *p, int oct);
int cand_group_of(const siftmi_plan *p, int oct);
int64_t list_limit(const siftmi_plan *p, int what, int g) {      // what: 0 candidates, 1 refined, 2 oriented, 3 records
    const int64_t K = p->kpsize, O = std::max(1, p->n_oct);
    int64_t lim = what == 0 ? 3 * K : (what == 1 ? 3 * K * (g == 2 ? O : 1) : (what == 2 ? K * (g == 2 ? O : 1) : K * O));
    return std::min<int64_t>(lim, 0x7fffff00);
}

// c == null: the initial allocation.  Else: grow every list the image behind `c` (its counters) has outrun.  *grown says
// whether anything changed (the image has to run again: what lies behind a cut list was never produced).
void drain_streams(siftmi_plan *p);
int grow_lists(siftmi_plan *p, const Counters *c, bool *grown = nullptr) {
    if (grown) *grown = false;
    bool drained = c == nullptr;
    // nothing of the plan may be in flight while a buffer is replaced (another ending stream may still be copying its counters)
    auto quiesce = [&]() { if (!drained) { drain_streams(p); drained = true; } };
    const bool forks = p->n_oct > 2;                    // (group 1 exists only beside a group 2)
    auto want = [&](int64_t cap, int64_t need, int64_t limit) {
        if (need <= cap) return cap;
        if (!c) return std::min<int64_t>(limit, need);                        // creation: exactly the reference's kpsize
        return std::min<int64_t>(limit, std::max<int64_t>(need + need / 4 + 64, cap));
    };
    for (int g = 0; g < SIFT_GROUPS; g++) {
        if (g == 1 && !forks) continue;
        GroupLists &G = p->grp[g];
        int64_t need_cand = p->kpsize, need_kp = p->kpsize, need_out = p->kpsize;
        if (c) {
            need_cand = 0;
            for (int o = 0; o < p->n_oct && o < SIFT_MAX_OCTAVES; o++) if (cand_group_of(p, o) == g) need_cand = std::max<int64_t>(need_cand, c->n_cand[o]);
            need_kp = c->g_kp[g]; need_out = c->g_out[g];
        }
        const int64_t cc = want(G.cap_cand, need_cand, list_limit(p, 0, g)), ck = want(G.cap_kp, need_kp, list_limit(p, 1, g)),
                      co = want(G.cap_out, need_out, list_limit(p, 2, g));
        int rc = SIFTMI_OK;
        if (cc != G.cap_cand || ck != G.cap_kp || co != G.cap_out) quiesce();
        if (cc != G.cap_cand) { if ((rc = p->regrow(&G.cand, (size_t)G.cap_cand * 16, (size_t)cc * 16))) return rc; G.cap_cand = cc; if (grown) *grown = true; }
        if (ck != G.cap_kp) {
            if ((rc = p->regrow(&G.kp, (size_t)G.cap_kp * 16, (size_t)ck * 16)) || (rc = p->regrow(&G.kp_aux, (size_t)G.cap_kp * 4, (size_t)ck * 4))) return rc;
            G.cap_kp = ck; if (grown) *grown = true;
        }
        if (co != G.cap_out) {
            if ((rc = p->regrow(&G.okp, (size_t)G.cap_out * 16, (size_t)co * 16)) || (rc = p->regrow(&G.oaux, (size_t)G.cap_out * 4, (size_t)co * 4))) return rc;
            G.cap_out = co; if (grown) *grown = true;
        }
    }
    const int64_t cr = want(p->cap_rec, c ? (int64_t)c->n_rec : p->kpsize, list_limit(p, 3, 0));
    if (cr != p->cap_rec) {
        quiesce();
        int rc = p->regrow(&p->records, (size_t)p->cap_rec * sizeof(KpRecord), (size_t)cr * sizeof(KpRecord));
        if (rc) return rc;
        p->cap_rec = cr; if (grown) *grown = true;
    }
    if (grown && *grown) p->grows++;
    return SIFTMI_OK;
}

// The reference's capacity rule on the counts of an image (see Counters::c_scale): per octave, with `last` the oriented
// keypoints of the scales before, local_maxmin of scale s leaves the counter at last + candidates(s) and the orientation
// pass at last + oriented(s); either beyond kpsize means the reference dropped entries (image.cl:203-205,
// orientation_cpu.cl:150-172, plan.py:771 only warns).  oracle/sift_oracle.c: so_keypoints applies the same rule.
bool reference_overflow(const siftmi_plan *p, const Counters &c) {
    for (int o = 0; o < p->n_oct && o < SIFT_MAX_OCTAVES; o++) {
        int64_t last = 0;
        for (int s = 0; s < 3; s++) {
            if (last + c.c_scale[o][s] > p->kpsize) return true;
            last += c.o_scale[o][s];
            if (last > p->kpsize) return true;
        }
    }
    return false;
}

// Group of an octave in the image being enqueued: octave 0, octave 1, everything below -- or, when the later octaves form
// one chain (single stream, option "fork" = 0, octave 1 inside the tail launch), octave 0 and everything below it.
int group_of(const siftmi_plan *p, int oct) { return oct == 0 ? 0 : ((oct == 1 && p->fork_cur) ? 1 : 2); }
// ... and the group whose CANDIDATE buffer the octave's detection uses (a buffer holds one octave at a time): with option
// "split" octave 1 is searched beside the octaves below it, so its candidates take group 1's otherwise unused buffer.
int cand_group_of(const siftmi_plan *p, int oct) { return (oct == 1 && p->split_cur) ? 1 : group_of(p, oct); }

// Extrema of the three detection scales + sub-pixel refinement of one octave; survivors are appended to the refined list
// of the octave's group (tagged with the octave).
void launch_detect_octave(siftmi_plan *p, int oct, hipStream_t st) {
    const int W = p->ow[(size_t)oct], H = p->oh[(size_t)oct];
    const int octsize = 1 << oct;
    const int g = group_of(p, oct);
    GroupLists &G = p->grp[g];
    p->groups_cur |= 1u << g;
    char lab[96];
    BlurPlanes bp;
    for (int s = 0; s < 6; s++) bp.p[s] = p->plane(oct, s);
    const int border = p->par.border_dist;
    GroupLists &GC = p->grp[cand_group_of(p, oct)];
    const int ccap = (int)GC.cap_cand, kcap = (int)G.cap_kp;
    int *n_cand = &p->cnt->n_cand[oct];
    if (!(W > 2 * border && H > 2 * border)) return;
    const int rows = p->opt.ext_rows > 0 ? p->opt.ext_rows : extrema_strip_rows(W, H, border, p->opt.ext_strips);
    const int nx = (W - 2 * border + 61) / 62, ny = (H - 2 * border + rows - 1) / rows;
    const int blocks = std::max(1, (nx * ny + 3) / 4);
    const float edth = (octsize <= 1) ? p->par.edge_thresh0 : p->par.edge_thresh;   // image.cl:193, plan.py:633-634
    const RefineArgs ra = {p->par.peak_thresh, (float)p->par.init_sigma, G.kp, G.kp_aux, &p->cnt->g_kp[g], kcap, oct, &p->cnt->c_scale[oct][0]};
    // One launch detects and refines (the survivors of the edge test are refined by the wave that parked them: no
    // candidate list, no second launch) unless every stage is bracketed on its own (full profile) or option
    // "fused_refine" says otherwise (0: never, 1: planes below 1400^2, 2: every plane).
    const bool fused = p->profile <= 1 && (p->opt.fused_refine == 2 || (p->opt.fused_refine == 1 && !march_plane(W, H)));
    if (fused) {
        snprintf(lab, sizeof lab, "local_maxmin+interp_keypoint %d", oct);
        Scope sc(p, lab, false, 0, st);
        hipLaunchKernelGGL(extrema_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, bp, W, H, border, rows,
                           contrast_threshold(p->par), edth, GC.cand, n_cand, ccap, ra, -1, -1, p->opt.xcd_map);
        return;
    }
    {
        snprintf(lab, sizeof lab, "local_maxmin %d", oct);
        Scope sc(p, lab, false, 0, st);
        hipLaunchKernelGGL(extrema_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, bp, W, H, border, rows,
                           contrast_threshold(p->par), edth, GC.cand, n_cand, ccap, ra, -1, -1, p->opt.xcd_map);
    }
    {
        snprintf(lab, sizeof lab, "interp_keypoint+compact %d", oct);
        Scope sc(p, lab, false, 0, st);
        hipLaunchKernelGGL(refine_kernel, dim3(512), dim3(256), 0, st, bp, W, H, (const float4 *)GC.cand,
                           (const int *)n_cand, ccap, p->par.peak_thresh, (float)p->par.init_sigma, G.kp,
                           G.kp_aux, &p->cnt->g_kp[g], kcap, oct, &p->cnt->c_scale[oct][0]);
    }
}

// First octave of the run that octave_tail_kernel takes (k_tail.hpp), or n_oct when it takes none: octaves >= 1 whose
// planes are small enough to sit in LDS, all the way down to the last one.
int tail_first_octave(const siftmi_plan *p) {
    if (!p->opt.tail || p->profile > 1 || p->n_oct < 2) return p->n_oct;   // full profile: every stage keeps its own launch and label
    for (int s = 0; s < 5; s++) {
        const int n = p->taps[s].n;
        if (n != 11 && n != 15 && n != 17 && n != 21 && n != 27) return p->n_oct;
    }
    const int last = p->n_oct - 1;
    if (p->ow[(size_t)last] < 14 || p->oh[(size_t)last] < 14) return p->n_oct;
    int first = p->n_oct;
    for (int o = last; o >= 1; o--) {
        const int W = p->ow[(size_t)o], H = p->oh[(size_t)o];
        if ((int64_t)W * H > p->opt.tail_pixels || W > 128 || H > 128 || last - o + 1 > SIFT_TAIL_MAX_OCT) break;
        first = o;
    }
    return first;
}

// octaves [first, n_oct): pyramid + detection in one launch
int launch_tail(siftmi_plan *p, int first, hipStream_t st) {
    TailArgs a;
    memset(&a, 0, sizeof a);
    a.src = p->plane(first - 1, 3);
    a.src_w = p->ow[(size_t)first - 1];
    a.n = p->n_oct - first;
    if (a.n < 1 || a.n > SIFT_TAIL_MAX_OCT) return fail(SIFTMI_EINVAL, "octave_tail_kernel walks 1..%d octaves, not %d", SIFT_TAIL_MAX_OCT, a.n);
    for (int k = 0; k < a.n; k++) {
        const int oct = first + k;
        TailOctave &o = a.o[k];
        for (int s = 0; s < 6; s++) o.plane[s] = p->plane(oct, s);
        o.W = p->ow[(size_t)oct]; o.H = p->oh[(size_t)oct]; o.oct = oct;
        o.edth = ((1 << oct) <= 1) ? p->par.edge_thresh0 : p->par.edge_thresh;   // image.cl:193, plan.py:633-634
    }
    for (int s = 0; s < 5; s++) { a.taps[s] = p->taps[s].dev; a.ntaps[s] = p->taps[s].n; }
    const size_t lds = tail_lds_bytes(a.o[0].W, a.o[0].H);
    {
        // The dynamic-LDS limit of a kernel is a property of the function on a device, not of a plan: keep the largest
        // value any plan has asked for (a plan of smaller frames must not lower it under a plan of larger ones).
        static std::mutex mu;
        static size_t granted[64] = {0};
        std::lock_guard<std::mutex> g(mu);
        size_t &have = granted[p->device & 63];
        if (have < 64 * 1024) have = 64 * 1024;
        if (lds > have) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&octave_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            have = lds;
        }
    }
    GroupLists &G = p->grp[2];                 // the tail's octaves are the last ones: group 2
    p->groups_cur |= 1u << 2;
    hipLaunchKernelGGL(octave_tail_kernel, dim3((unsigned)a.n), dim3(SIFT_TAIL_THREADS), lds, st, a, p->par.border_dist,
                       contrast_threshold(p->par), p->par.peak_thresh, (float)p->par.init_sigma, p->tail_cand, p->tail_cand_cap,
                       p->cnt->n_cand, p->cnt->tail_ready, G.kp, G.kp_aux, &p->cnt->g_kp[2], (int)G.cap_kp, &p->cnt->c_scale[0][0], &p->cnt->tail_timeout);
    return SIFTMI_OK;
}

// orientation of every refined keypoint of one group (its refinement launches precede this one on `st`)
void launch_orient_group(siftmi_plan *p, int group, hipStream_t st) {
    const OctaveTable tab = octave_table(p);
    GroupLists &G = p->grp[group];
    char lab[96];
    snprintf(lab, sizeof lab, "orientation_assignment group %d", group);
    Scope sc(p, lab, false, 0, st);
    const int ori_blocks = p->opt.ori_blocks, ori_pad = p->opt.ori_pad;
    const bool maps = group == 0 ? p->maps_g0 : p->maps_g1;
    if (maps)
        hipLaunchKernelGGL(orientation_kernel<true>, dim3((unsigned)ori_blocks), dim3(256), (size_t)ori_pad, st, tab, p->par.ori_sigma,
                           (const float4 *)G.kp, (const int *)G.kp_aux, p->cnt, group, (int)G.cap_kp, G.okp, G.oaux, (int)G.cap_out,
                           p->opt.ori_team, p->opt.ori_small_blocks);
    else
        hipLaunchKernelGGL(orientation_kernel<false>, dim3((unsigned)ori_blocks), dim3(256), (size_t)ori_pad, st, tab, p->par.ori_sigma,
                           (const float4 *)G.kp, (const int *)G.kp_aux, p->cnt, group, (int)G.cap_kp, G.okp, G.oaux, (int)G.cap_out,
                           p->opt.ori_team, p->opt.ori_small_blocks);
}

// descriptors of one group's oriented keypoints
void launch_descriptor_group(siftmi_plan *p, int group, hipStream_t st) {
    const OctaveTable tab = octave_table(p);
    GroupLists &G = p->grp[group];
    char lab[96];
    snprintf(lab, sizeof lab, "descriptors group %d", group);
    Scope sc(p, lab, false, 0, st);
    const int desc_blocks = p->opt.desc_blocks;
    // Optional residency throttle (option "desc_pad": bytes of unused dynamic LDS per workgroup).  Round 1 capped the
    // octave-0 launch at 3 workgroups per CU so that the later octaves' kernels found registers; with the round-2
    // kernels (shorter detection chain, 4 workgroups per CU by registers) the unthrottled launch is faster
    // (0.96 against 1.01 ms per 4096^2 frame), so the default is 0.
    const int desc_pad = p->opt.desc_pad > 0 ? p->opt.desc_pad : 0;
    // a small group of octave 0 of a LARGE frame leaves room for the later octaves' chain, which ends such an image (4096^2
    // headline -2.2 %, 4096^2 with every octave -2.5 %); on a 1024^2 frame that chain is short and the same cut costs 2 %
    const int small_blocks = (group == 0 && p->n_oct > 1 && march_plane(p->ow[0], p->oh[0])) ? (p->early_cur ? p->opt.desc_early_blocks : p->opt.desc_small_blocks) : desc_blocks;
    const int ocap = (int)G.cap_out, rcap = (int)std::min<int64_t>(p->cap_rec, 0x7fffffff);
    if (p->desc_rows && !p->opt.desc_stream) {
        // one launch, two forms: the count of the group (known on the device only) picks the wave-per-keypoint form
        // (throughput) or the workgroup-per-keypoint form (latency of a sparse group)
        const bool maps = group == 0 ? p->maps_g0 : p->maps_g1;
        if (maps)
            // (the MAPS form of a dense group wants every workgroup of the launch: 154 k keypoints 4.68 ms at 832, 4.48 at 960)
            hipLaunchKernelGGL(descriptor_kernel<true>, dim3((unsigned)desc_blocks), dim3(256), (size_t)desc_pad, st, tab,
                               (const float4 *)G.okp, (const int *)G.oaux, p->cnt, group, 0, 0, ocap, p->records, rcap, p->host_out, p->host_cap,
                               p->opt.desc_team, p->opt.desc_dynamic, desc_blocks, small_blocks);
        else
            hipLaunchKernelGGL(descriptor_kernel<false>, dim3((unsigned)desc_blocks), dim3(256), (size_t)desc_pad, st, tab,
                               (const float4 *)G.okp, (const int *)G.oaux, p->cnt, group, 0, 0, ocap, p->records, rcap, p->host_out, p->host_cap,
                               p->opt.desc_team, p->opt.desc_dynamic, p->opt.desc_dense_blocks, small_blocks);
    } else
        hipLaunchKernelGGL(descriptor_stream_kernel, dim3((unsigned)desc_blocks), dim3(256), (size_t)desc_pad, st, tab,
                           (const float4 *)G.okp, (const int *)G.oaux, p->cnt, group, 0, 0, ocap, p->records, rcap, p->host_out, p->host_cap);
}

// gradient maps of the octaves [oct_lo, oct_hi) (their pyramids exist on `st`)
void launch_gradient_maps(siftmi_plan *p, int oct_lo, int oct_hi, hipStream_t st) {
    if (oct_lo >= oct_hi) return;
    const OctaveTable tab = octave_table(p);
    long long total = 0;
    for (int o = oct_lo; o < oct_hi; o++) total += gradient_map_items_of(p->ow[(size_t)o], p->oh[(size_t)o]);
    char lab[96];
    snprintf(lab, sizeof lab, "gradient maps octaves %d-%d", oct_lo, oct_hi - 1);
    Scope sc(p, lab, false, 0, st);
    const unsigned blocks = (unsigned)std::min<long long>(total, p->opt.maps_blocks);
    hipLaunchKernelGGL(gradient_maps_kernel, dim3(blocks), dim3(256), 0, st, tab, oct_lo, oct_hi, total, p->gmap, p->omap);
}

// orientation + descriptors of one group, on one stream, ending with the read-back of the counters into pinned block `slot`
int launch_describe_group(siftmi_plan *p, int group, hipStream_t st, int slot) {
    p->groups_cur |= 1u << group;
    launch_orient_group(p, group, st);
    launch_descriptor_group(p, group, st);
    if (p->profile > 1) hipEventRecord(p->ev_last[slot], st);
    HIPCHK(hipMemcpyAsync(&p->hb->c[slot], p->cnt, sizeof(Counters), hipMemcpyDeviceToHost, st));
    p->wait_s[slot] = st;
    return SIFTMI_OK;
}

}  // namespace

// ============================================================================================
extern "C" {

const char *siftmi_last_error(void) { return g_err.c_str(); }
#ifdef SIFT_PHASE_CLOCK
// development instrument (k_keypoint.hpp: PhaseClock): the 32 accumulators of the per-keypoint kernels; reset != 0 clears them
int siftmi_dev_phase(uint64_t *out32, int32_t reset) {
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(siftk::g_phase), sizeof h) != hipSuccess) return fail(SIFTMI_EDEVICE, "phase clock read failed");
    if (out32) for (int i = 0; i < 32; i++) out32[i] = h[i];
    if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(siftk::g_phase), h, sizeof h) != hipSuccess) return fail(SIFTMI_EDEVICE, "phase clock reset failed"); }
    return SIFTMI_OK;
}
#endif
const char *siftmi_version(void) { return "sift_pyocl_amd 0.1 (gfx950)"; }


int siftmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int siftmi_device_name(int device_id, char *buf, int64_t buflen) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    snprintf(buf, (size_t)buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return SIFTMI_OK;
}

static thread_local int g_lane_mode = 0;   // set by siftmi_batch_create around its plan constructions: 1 single-stream lane, 2 multi-stream lane

int siftmi_plan_create(int32_t height, int32_t width, int32_t in_dtype, int32_t device_id,
                       const siftmi_params *params, int32_t profile, siftmi_plan **out) {
    if (!out || !params) return fail(SIFTMI_EINVAL, "null argument");
    *out = nullptr;
    if (height < 1 || width < 1) return fail(SIFTMI_EINVAL, "bad shape %dx%d", height, width);
    // planes are addressed with 32-bit byte offsets in the extrema kernel and 32-bit sample indices in the window kernels
    if ((int64_t)height * width > (int64_t)1 << 30) return fail(SIFTMI_EINVAL, "image too large: at most 2^30 pixels (32768 x 32768)");
    if (dtype_size(in_dtype) == 0) return fail(SIFTMI_EINVAL, "invalid input format (%d)", in_dtype);
    if (params->pix_per_kp < 1) return fail(SIFTMI_EINVAL, "pix_per_kp must be >= 1");
    if (params->border_dist < 1) return fail(SIFTMI_EINVAL, "border_dist must be >= 1");
    int ndev = siftmi_device_count();
    if (ndev < 1) return fail(SIFTMI_EDEVICE, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(SIFTMI_EINVAL, "device %d out of range (%d devices)", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));
    siftmi_plan *p = new (std::nothrow) siftmi_plan();
    if (!p) return fail(SIFTMI_ENOMEM, "host allocation failed");
    p->device = device_id; p->H = height; p->W = width; p->dtype = in_dtype; p->par = *params; p->profile = profile;
    // octave shapes, plan.py:213-224
    {
        int h = height, w = width;
        std::vector<int> hh{h}, ww{w};
        while ((h < w ? h : w) > 2 * params->border_dist + 2) { h /= 2; w /= 2; hh.push_back(h); ww.push_back(w); }
        hh.pop_back(); ww.pop_back();
        p->oh = hh; p->ow = ww;
        p->n_oct = (int)hh.size();
        if (params->octave_max > 0 && params->octave_max < p->n_oct) p->n_oct = params->octave_max;
        if (p->n_oct > SIFT_MAX_OCTAVES) p->n_oct = SIFT_MAX_OCTAVES;
    }
    const size_t N = (size_t)height * width;
    p->kpsize = (int64_t)(N / (size_t)params->pix_per_kp);   // plan.py:243
    if (p->kpsize < 1) p->kpsize = 1;
    int rc = SIFTMI_OK;
    // Stream priorities (pyramid and later-octave streams above the octave-0 detection stream, whose long orientation /
    // descriptor kernels would otherwise win every dispatch slot) pay off where two multi-stream plans share the GPU
    // (BatchPlan with 2 lanes of large frames: 1.13 instead of 1.33 ms per 4096^2 frame); for a single plan they are
    // within noise (round 2, later-octave chain above the octave-0 chain: 0.962 against 0.972 ms), and once a process has created prioritised streams its normal-priority streams get fewer hardware
    // queues (8 single-stream lanes of 512^2 frames: 0.72 instead of 0.41 ms per frame).  Hence: only for the
    // multi-stream lanes of a batch.
    const bool prio = g_lane_mode == 2;
    int prio_lo = 0, prio_hi = 0;
    if (prio) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = prio ? hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete p; return fail(SIFTMI_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
    {
        size_t off = 0;
        for (int o = 0; o < p->n_oct; o++) { p->oct_off.push_back(off); off += 6 * (size_t)p->ow[(size_t)o] * p->oh[(size_t)o]; }
        if (p->n_oct == 0) { p->oct_off.push_back(0); off = 6 * N; p->ow.assign(1, width); p->oh.assign(1, height); }
        p->planes_floats = off;
        rc = p->alloc(&p->planes, off * sizeof(float));
    }
    if (!rc && (prio ? hipStreamCreateWithPriority(&p->stream2, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking)) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "hipStreamCreate failed");
    if (!rc && (prio ? hipStreamCreateWithPriority(&p->stream3, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&p->stream3, hipStreamNonBlocking)) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "hipStreamCreate failed");
    if (!rc && hipEventCreateWithFlags(&p->ev_p3, SIFT_SYNC_EVENT) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "hipEventCreate failed");
    p->overlap = true;
    for (int o = 0; o < p->n_oct && !rc; o++) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, SIFT_SYNC_EVENT) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "hipEventCreate failed");
        else p->ev_pyr.push_back(e);
    }
    if (!rc) rc = p->alloc(&p->tmp, N * sizeof(float));
    if (!rc && p->n_oct > 1) rc = p->alloc(&p->tmp_later, (size_t)p->ow[1] * p->oh[1] * sizeof(float));
    if (!rc && p->n_oct > 2) rc = p->alloc(&p->tmp_below, (size_t)p->ow[2] * p->oh[2] * sizeof(float));
    if (!rc) rc = p->alloc(&p->raw, N * (dtype_size(in_dtype) > 4 ? dtype_size(in_dtype) : 4));
    if (!rc && in_dtype != SIFTMI_F32) rc = p->alloc(&p->conv, N * sizeof(float));

    // two counter blocks, used by alternate images: the min/max pass of an image resets the other one for its successor
    if (!rc) rc = p->alloc(&p->cnt_pair, 2 * sizeof(Counters));
    if (!rc) { p->cnt = p->cnt_pair; p->mm = p->cnt->mm; }   // (mm: device address of the min/max slots inside the counter block)
    // a tail octave (<= SIFT_TAIL_MAX_PIXELS samples, 3 scales) cannot hold more candidates than this
    p->tail_cand_cap = 3 * SIFT_TAIL_MAX_PIXELS;
    if (!rc) rc = p->alloc(&p->tail_cand, (size_t)SIFT_TAIL_MAX_OCT * p->tail_cand_cap * sizeof(float4));
    // The lists start at the reference's kpsize (plan.py:243: its capacity PER OCTAVE) and grow when an image needs more
    // (grow_lists): a frame of the default PIX_PER_KP never does.  Octave 0 in two groups needs group 1's lists too:
    // large planes only (the planes the marching blur takes).
    if (!rc) rc = grow_lists(p, nullptr);
    if (!rc) rc = compute_schedule(p);
    if (!rc) {
        // both blocks all zero, min / max slots at their identities (minmax_reset_next does the same for every later image)
        Counters zero;
        memset(&zero, 0, sizeof zero);
        zero.mm[0] = 0xffffffffu;
        if (hipMemcpy(p->cnt_pair, &zero, sizeof zero, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(p->cnt_pair + 1, &zero, sizeof zero, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "counter initialisation failed");
    }
    if (!rc) { hipEventCreate(&p->ev_first); for (hipEvent_t &e : p->ev_last) hipEventCreate(&e); }
    if (rc) { std::string keep = g_err; siftmi_plan_destroy(p); g_err = keep; return rc; }
    *out = p;
    return SIFTMI_OK;
}

int siftmi_plan_destroy(siftmi_plan *p) {
    if (!p) return SIFTMI_OK;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    if (p->stream2) { hipStreamSynchronize(p->stream2); hipStreamDestroy(p->stream2); }
    if (p->stream3) { hipStreamSynchronize(p->stream3); hipStreamDestroy(p->stream3); }
    if (p->ev_p3) hipEventDestroy(p->ev_p3);
    if (p->ev_det2) hipEventDestroy(p->ev_det2);
    if (p->ev_early) hipEventDestroy(p->ev_early);
    if (p->ev_maps0) hipEventDestroy(p->ev_maps0);
    for (hipEvent_t e : p->ev_pyr) hipEventDestroy(e);
    for (void *q : p->allocs) hipFree(q);
    for (GroupLists &G : p->grp)
        for (void *q : {(void *)G.kp, (void *)G.kp_aux, (void *)G.okp, (void *)G.oaux, (void *)G.cand}) if (q) hipFree(q);
    if (p->records) hipFree(p->records);
    if (p->hb) hipHostFree(p->hb);
    if (p->ev_join) hipEventDestroy(p->ev_join);
    if (p->warp_in) hipFree(p->warp_in);
    if (p->warp_out) hipFree(p->warp_out);
    if (p->ev_wa) hipEventDestroy(p->ev_wa);
    if (p->ev_wb) hipEventDestroy(p->ev_wb);
    for (Event &e : p->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    if (p->ev_first) hipEventDestroy(p->ev_first);
    for (hipEvent_t e : p->ev_last) if (e) hipEventDestroy(e);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return SIFTMI_OK;
}

int siftmi_plan_info(const siftmi_plan *p, int32_t *n_octaves, int64_t *kpsize, int64_t *bytes_allocated) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (n_octaves) *n_octaves = p->n_oct;
    if (kpsize) *kpsize = p->kpsize;
    if (bytes_allocated) *bytes_allocated = p->bytes;
    return SIFTMI_OK;
}

int siftmi_plan_capacity(const siftmi_plan *p, int64_t *records, int64_t *growths) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (records) *records = p->cap_rec;
    if (growths) *growths = p->grows;
    return SIFTMI_OK;
}

int siftmi_plan_tail_timeouts(const siftmi_plan *p, int64_t *timeouts, int32_t *tail_enabled) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (timeouts) *timeouts = p->tail_timeouts;
    if (tail_enabled) *tail_enabled = p->opt.tail ? 1 : 0;
    return SIFTMI_OK;
}

int siftmi_plan_set_params(siftmi_plan *p, const siftmi_params *params) {
    if (!p || !params) return fail(SIFTMI_EINVAL, "null argument");
    if (params->pix_per_kp != p->par.pix_per_kp || params->octave_max != p->par.octave_max)
        return fail(SIFTMI_EINVAL, "pix_per_kp / octave_max are fixed at plan creation");
    // border_dist may change between calls: like the reference, which sizes its octave list from par.BorderDist in the
    // constructor (plan.py:213-224) but hands the CURRENT par.BorderDist to local_maxmin (plan.py:631-641), the octave
    // shapes stay those of creation and only the detection border follows.
    if (params->border_dist < 1) return fail(SIFTMI_EINVAL, "border_dist must be >= 1");
    HIPCHK(hipSetDevice(p->device));
    const bool resched = params->init_sigma != p->par.init_sigma || (params->double_im_size != 0) != (p->par.double_im_size != 0);
    p->par = *params;
    return resched ? compute_schedule(p) : SIFTMI_OK;
}

// Tuning / diagnostic options of one plan, by name.  Results never depend on them (every path is bit-identical);
// tests use "fused_convert" = 0 to compare the fused typed-frame path with the separate convert pass.
int siftmi_plan_set_option(siftmi_plan *p, const char *name, int64_t value) {
    if (!p || !name) return fail(SIFTMI_EINVAL, "null argument");
    const std::string n(name);
    const int v = (int)value;
    Options &o = p->opt;
    if (n == "fused_convert") o.fused_convert = v != 0;
    else if (n == "overlap") { o.overlap = v != 0; p->overlap = o.overlap; }
    else if (n == "march") o.march = v != 0;
    else if (n == "march_wgs") o.march_wgs = v > 0 ? v : 0;
    else if (n == "xcd_map") o.xcd_map = v != 0;
    else if (n == "march_prio") { if (v < 0 || v > 2) return fail(SIFTMI_EINVAL, "march_prio must be 0, 1 or 2"); o.march_prio = (int)v; }
    else if (n == "ori_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "ori_blocks must be >= 1"); o.ori_blocks = v; }
    else if (n == "ori_pad") o.ori_pad = v > 0 ? v : 0;
    else if (n == "desc_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "desc_blocks must be >= 1"); o.desc_blocks = v; }
    else if (n == "desc_pad") o.desc_pad = v;
    else if (n == "desc_stream") o.desc_stream = v != 0;
    else if (n == "mm_threads") { if (v != 256 && v != 512 && v != 1024) return fail(SIFTMI_EINVAL, "mm_threads must be 256, 512 or 1024"); o.mm_threads = v; }
    else if (n == "mm_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "mm_blocks must be >= 1"); o.mm_blocks = v; }
    else if (n == "ori_small_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "ori_small_blocks must be >= 1"); o.ori_small_blocks = v; }
    else if (n == "ext_rows") o.ext_rows = (int)v;
    else if (n == "ext_strips") { if (v < 1) return fail(SIFTMI_EINVAL, "ext_strips must be >= 1"); o.ext_strips = (int)v; }
    else if (n == "tail") o.tail = v != 0;
    else if (n == "tail_pixels") { if (v < 1 || v > SIFT_TAIL_MAX_PIXELS) return fail(SIFTMI_EINVAL, "tail_pixels must be in 1..%d", SIFT_TAIL_MAX_PIXELS); o.tail_pixels = (int)v; }
    else if (n == "desc_team") o.desc_team = v > 0 ? v : 0;
    else if (n == "desc_dynamic") o.desc_dynamic = v != 0;
    else if (n == "ori_team") o.ori_team = v > 0 ? v : 0;
    else if (n == "fused_shrink") o.fused_shrink = v != 0;
    else if (n == "fused_refine") { if (v < 0 || v > 2) return fail(SIFTMI_EINVAL, "fused_refine must be 0, 1 or 2"); o.fused_refine = (int)v; }
    else if (n == "maps") { if (v < 0 || v > 2) return fail(SIFTMI_EINVAL, "maps must be 0 (never), 1 (always) or 2 (by the previous image)"); o.maps = v; }
    else if (n == "maps_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "maps_blocks must be >= 1"); o.maps_blocks = v; }
    else if (n == "maps_density") { if (v < 1) return fail(SIFTMI_EINVAL, "maps_density must be >= 1"); o.maps_density = v; }
    else if (n == "early_chain") { if (v < 0 || v > 2) return fail(SIFTMI_EINVAL, "early_chain must be 0 (never), 1 (always) or 2 (unless the previous image was keypoint-rich)"); o.early_chain = v; }
    else if (n == "desc_early_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "desc_early_blocks must be >= 1"); o.desc_early_blocks = v; }
    else if (n == "split") o.split = v != 0;
    else if (n == "fork") { if (v < 0 || v > 2) return fail(SIFTMI_EINVAL, "fork must be 0 (never), 1 (always) or 2 (frames of five octaves and more)"); o.fork = v; }
    else if (n == "desc_small_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "desc_small_blocks must be >= 1"); o.desc_small_blocks = v; }
    else if (n == "desc_dense_blocks") { if (v < 1) return fail(SIFTMI_EINVAL, "desc_dense_blocks must be >= 1"); o.desc_dense_blocks = v; }
    else if (n == "spin") o.spin = v != 0;
    else if (n == "host_timing") o.host_timing = v != 0;
    else if (n == "tail_fault") o.tail_fault = v > 0 ? v : 0;
    else return fail(SIFTMI_EINVAL, "unknown option '%s'", name);
    return SIFTMI_OK;
}

}  // extern "C"

namespace {
int enqueue_body(siftmi_plan *p);

// Enqueue the whole pipeline of one image on the plan's streams, ending with the asynchronous read-back of the
// counters into the plan's pinned block.  No host wait.  `sync_device`: order after the caller's other streams.
int plan_enqueue(siftmi_plan *p, const void *image, int32_t image_dtype, int32_t image_is_device, bool sync_device) {
    if (image_dtype != p->dtype && image_dtype != SIFTMI_F32)
        return fail(SIFTMI_EINVAL, "image dtype %d is neither the plan's (%d) nor float32", image_dtype, p->dtype);
    HIPCHK(hipSetDevice(p->device));
    if (!p->hb) HIPCHK(hipHostMalloc((void **)&p->hb, sizeof(*p->hb), hipHostMallocDefault));
    if (!p->ev_join) HIPCHK(hipEventCreate(&p->ev_join));
    const size_t N = (size_t)p->H * p->W;
    const void *src = image;
    if (image_is_device) {
        if (sync_device) HIPCHK(hipDeviceSynchronize());   // order after the caller's work on other streams
    } else {
        HIPCHK(hipMemcpyAsync(p->raw, image, N * dtype_size(image_dtype), hipMemcpyHostToDevice, p->stream));
        src = p->raw;
        p->raw_dtype = image_dtype;
    }
    const bool htime = p->opt.host_timing != 0;
    auto tnow = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = tnow();
    p->n_events = 0;
    if (p->profile > 1) hipEventRecord(p->ev_first, p->stream);     // (light profile: only the blur bracket -- every event record between kernels is a bubble)
    // this image's counter block was reset by the min/max pass of the previous image (or at plan creation); the other block
    // is reset by this image's min/max pass (k_pyramid.hpp: minmax_reset_next)
    // (the two blocks alternate only if an image never overlaps its predecessor on the same plan: plan_wait -- or a drain
    // on an error path -- always comes between two enqueues; `in_flight` enforces it)
    if (p->in_flight) return fail(SIFTMI_EINVAL, "the plan still has an image in flight");
    p->cnt = p->cnt_pair + p->cnt_parity;
    p->mm = p->cnt->mm;
    Counters *next_cnt = p->cnt_pair + (p->cnt_parity ^ 1);
    constexpr int kCntWords = (int)(sizeof(Counters) / 4), kCntOnes = (int)(offsetof(Counters, mm) / 4);
    static_assert(sizeof(Counters) % 4 == 0, "Counters is a block of 32-bit words");
    const float *f32src = (const float *)src;
    // Typed frames (u8 / u16 / ... / RGB8) are converted at the point of use by the min/max and the initial blur
    // (or the plain normalise) -- no f32 copy of the frame in HBM.  Fallback to the convert pass: float64 frames,
    // a non-default initial tap count, a frame that is not 16-byte aligned.
    const bool fused_in = image_dtype != SIFTMI_F32 && image_dtype != SIFTMI_F64 && (((uintptr_t)src) & 15) == 0 &&
                          (!p->have_init || p->taps[5].n == 15) && p->opt.fused_convert;
    if (image_dtype != SIFTMI_F32 && !fused_in) {
        Scope sc(p, "convert -> float");
        const int g = grid_for((int64_t)N, 256, 4096);
        switch (image_dtype) {
            case SIFTMI_U8: hipLaunchKernelGGL(convert_kernel<uint8_t>, dim3(g), dim3(256), 0, p->stream, (const uint8_t *)src, p->conv, (int64_t)N); break;
            case SIFTMI_U16: hipLaunchKernelGGL(convert_kernel<uint16_t>, dim3(g), dim3(256), 0, p->stream, (const uint16_t *)src, p->conv, (int64_t)N); break;
            case SIFTMI_U32: hipLaunchKernelGGL(convert_kernel<uint32_t>, dim3(g), dim3(256), 0, p->stream, (const uint32_t *)src, p->conv, (int64_t)N); break;
            case SIFTMI_U64: hipLaunchKernelGGL(convert_kernel<uint64_t>, dim3(g), dim3(256), 0, p->stream, (const uint64_t *)src, p->conv, (int64_t)N); break;
            case SIFTMI_I32: hipLaunchKernelGGL(convert_kernel<int32_t>, dim3(g), dim3(256), 0, p->stream, (const int32_t *)src, p->conv, (int64_t)N); break;
            case SIFTMI_I64: hipLaunchKernelGGL(convert_kernel<int64_t>, dim3(g), dim3(256), 0, p->stream, (const int64_t *)src, p->conv, (int64_t)N); break;
            case SIFTMI_F64: hipLaunchKernelGGL(convert_kernel<double>, dim3(g), dim3(256), 0, p->stream, (const double *)src, p->conv, (int64_t)N); break;
            case SIFTMI_RGB8: hipLaunchKernelGGL(convert_rgb_kernel, dim3(g), dim3(256), 0, p->stream, (const uint8_t *)src, p->conv, (int64_t)N); break;
            default: return fail(SIFTMI_EINVAL, "invalid input format");
        }
        f32src = p->conv;
    }
    const int mm_blocks = p->opt.mm_blocks;
    // light profiling: ONE event pair around the six full-resolution blur launches (initial + five scales of octave 0),
    // opened by the END of the min/max launch and closed by the end of octave 0's last blur (enqueue_body) -- both events
    // ride on those launches (launch_ev), no packet of their own sits between two kernels
    hipEvent_t open_ev = nullptr;
    if (p->profile == 1 && p->have_init) {
        Scope *ch = new Scope(p, "Blur octave 0: initial + scales 0-4 (one bracket)", true, 6.0 * (double)N, nullptr, 0, false);
        if (ch->idx != (size_t)-1) { p->events[ch->idx].launches = 6; open_ev = p->events[ch->idx].a; }
        p->chain = ch;
    }
    {
        Scope sc(p, "max_min");
        if (fused_in) {
            SIFTMI_TYPED_DISPATCH(image_dtype, launch_ev(open_ev, minmax_typed_kernel<DT>, dim3(grid_for((int64_t)N / TypedChunk<DT>::PX, 256, mm_blocks)),
                                                         dim3(256), 0, p->stream, src, (int64_t)N, p->mm, (uint32_t *)next_cnt, kCntWords, kCntOnes));
        } else {
            const int mm_threads = p->opt.mm_threads;
            launch_ev(open_ev, minmax_kernel, dim3(grid_for((int64_t)N / 4, mm_threads, mm_blocks)), dim3(mm_threads), 0, p->stream, f32src,
                      (int64_t)N, p->mm, (uint32_t *)next_cnt, kCntWords, kCntOnes);
        }
    }
    p->cnt_parity ^= 1;          // the pass that resets the other block is in the queue: the next image takes that block
    p->in_flight = true;
    float *base0 = p->plane(0, 0);
    if (p->have_init) {
        Scope *sc = nullptr;
        if (p->profile != 1) sc = new Scope(p, "normalize + initial blur", true, (double)N, nullptr, 0);
        if (fused_in) {
            SIFTMI_TYPED_DISPATCH(image_dtype, launch_init_blur_dt<DT>(p->opt, p->stream, src, base0, p->W, p->H, p->taps[5], p->mm));
        } else {
            launch_blur(p, f32src, base0, p->W, p->H, p->taps[5], true);
        }
        delete sc;
    } else {
        Scope sc(p, "normalize");
        const dim3 g(grid_for((int64_t)N, 256, 4096));
        if (fused_in) {
            SIFTMI_TYPED_DISPATCH(image_dtype, hipLaunchKernelGGL(normalize_kernel<DT>, g, dim3(256), 0, p->stream, src, base0, (int64_t)N,
                                                                   (const uint32_t *)p->mm));
        } else {
            hipLaunchKernelGGL(normalize_kernel<0>, g, dim3(256), 0, p->stream, (const void *)f32src, base0, (int64_t)N,
                               (const uint32_t *)p->mm);
        }
    }
    // ---- everything below depends on plan-owned buffers only
    int rc;
    rc = enqueue_body(p);
    if (htime) fprintf(stderr, "[siftmi] enqueue %.0f us\n", tnow() - t_enter);
    return rc;
}

// Pyramid of every octave, detection, orientation, description, read-back of the counters: the part of one image's
// work that touches plan-owned buffers only.  Sets p->fin and p->wait_s.
int enqueue_body(siftmi_plan *p) {
    char lab[96];
    // Streams (option "overlap"; a single-stream plan runs the same launches in this order on `stream`):
    //   stream  : octave 0 end to end -- pyramid, detection, orientation, descriptors (group 0), read-back of the counters;
    //   stream3 : octave 1 from octave 0's pyramid on -- its pyramid, detection, orientation and descriptors (group 1);
    //   stream2 : the octaves below (option "fork"), from the moment octave 1's plane 3 exists -- pyramids, detection, the tail
    //             launch, orientation and descriptors (group 2); before that, octave 0's gradient maps on keypoint-rich frames.
    // The groups have their own lists and counters and meet only in the record list (descriptor_open): nothing orders them
    // but the two pyramid events.  Octave planes are never rewritten.  (Round 4 had one chain for all later octaves, whose
    // detection also waited for the end of octave 0's orientation pass -- the lists were shared; that chain ended the frame,
    // ~75 us after octave 0's descriptors, on a nearly idle chip.  Round 5 also tried octave 0 in two groups -- scale 1 from
    // plane 3 on, under the last two blurs: 0.86 against 0.80 ms, the blurs beside it slow down by more than the chain gains.)
    const bool two = p->overlap && p->n_oct > 0;
    const int tail_first = tail_first_octave(p);
    p->tail_first_cur = tail_first;
    p->groups_cur = 0;
    for (hipStream_t &w : p->wait_s) w = nullptr;
    // Gradient maps for the per-keypoint kernels of this image (option "maps"): large frames, by the previous image's counts
    {
        const int mode = p->opt.maps;
        long long px0 = 0, px1 = 0;
        for (int o = 0; o < p->n_oct; o++) {
            const long long px = (long long)p->ow[(size_t)o] * p->oh[(size_t)o];
            if (o == 0) px0 = px; else px1 += px;
        }
        auto dense = [&](int count, long long pixels) { return mode == 1 || (mode == 2 && (long long)count * p->opt.maps_density >= pixels); };
        const bool layout_ok = mode != 0 && p->n_oct > 0 && (mode == 1 || march_plane(p->ow[0], p->oh[0]));
        bool want0 = layout_ok && dense(p->last_group0, px0);
        bool want1 = layout_ok && p->n_oct > 1 && dense(p->last_group1, px1);
        if ((want0 || want1) && !p->gmap) {
            // first keypoint-rich image of this plan: two maps of half the pyramid's size each (three planes per octave)
            float *g = nullptr, *o = nullptr;
            const size_t bytes = (p->planes_floats / 2 + 16) * sizeof(float);
            if (!p->maps_unavailable && p->alloc(&g, bytes) == SIFTMI_OK) {          // (alloc() keeps the books: allocs, bytes)
                if (p->alloc(&o, bytes) == SIFTMI_OK) { p->gmap = g; p->omap = o; }
                else {                       // half a pair is of no use: give it back instead of holding it until the plan dies
                    (void)hipFree(g);
                    p->allocs.pop_back();
                    p->bytes -= (int64_t)bytes;
                }
            }
            if (!p->gmap) {                  // no room: the lazy forms, and no further attempt (a hipMalloc synchronises the device)
                p->maps_unavailable = true;
                want0 = want1 = false;
                (void)hipGetLastError();
            }
        }
        p->maps_g0 = want0; p->maps_g1 = want1;
        if (want0 && two && !p->ev_maps0) HIPCHK(hipEventCreateWithFlags(&p->ev_maps0, SIFT_SYNC_EVENT));
    }
    // the later octaves in two chains: two streams, an octave 1 outside the tail launch, at least one octave below it
    const bool fork = two && (p->opt.fork == 1 || (p->opt.fork == 2 && p->n_oct >= 5)) && p->n_oct > 2 && tail_first != 1 && p->grp[1].okp;
    p->fork_cur = fork;
    // ... or ONE group whose octaves are built and searched on two streams (option "split"; not beside octave 0's gradient maps,
    // which use stream2)
    const bool split = two && !fork && p->opt.split && p->n_oct > 2 && tail_first != 1 && p->grp[1].cand && !p->maps_g0;
    p->split_cur = split;
    if (split && !p->ev_det2) HIPCHK(hipEventCreateWithFlags(&p->ev_det2, SIFT_SYNC_EVENT));
    hipStream_t later = two ? p->stream3 : p->stream;       // octave 1's chain (every later octave's without the fork)
    hipStream_t below = (fork || split) ? p->stream2 : later;   // the chain of the octaves below octave 1
    // ... starts at plane 3 of octave 0 (option "early_chain"): needs the fused hand-off (octave 1's plane 0 written by octave 0's third blur)
    bool early = two && p->opt.early_chain && p->n_oct > 1 && p->profile <= 1 && p->opt.fused_shrink && tail_first != 1;
    if (early && p->opt.early_chain == 2) {
        const long long px0 = (long long)p->ow[0] * p->oh[0];
        // not on keypoint-rich frames (+4 % at 4096^2), and not where it brings nothing: a frame of at least 8 Mpixel with three
        // octaves (the later octaves' chain is short there: +-1 %, and octave 0's last two blurs would share the chip for nothing)
        early = (long long)p->last_group0 * p->opt.maps_density < px0 && (p->n_oct >= 4 || px0 < (8ll << 20));
    }
    p->early_cur = early;
    bool handed[SIFT_MAX_OCTAVES + 1] = {false};   // plane 0 of the octave was written by the blur launch of the octave above
    hipEvent_t pyr0_done = nullptr;            // light profile: the blur bracket's closing event stands in for ev_pyr[0]
    int slot = 0;                              // read-back blocks used so far (one per ending stream)
    // shrink + five blurs of one octave on stream `pyr`
    auto build_pyramid = [&](int oct, hipStream_t pyr) -> int {
        const int W = p->ow[(size_t)oct], H = p->oh[(size_t)oct];
        if (oct > 0 && !handed[oct]) {
            const int LW = p->ow[(size_t)oct - 1];
            snprintf(lab, sizeof lab, "shrink %d", oct - 1);
            Scope sc(p, lab, false, 0, pyr);
            hipLaunchKernelGGL(shrink_kernel, shrink_grid(W, H), dim3(256), 0, pyr,
                               (const float *)p->plane(oct - 1, 3), p->plane(oct, 0), LW, W, H);
        }
        // The octave hand-off (next[y][x] = plane 3 [2y][2x]) rides on the launch that writes plane 3, unless the next octave
        // is built by the tail kernel (which shrinks for itself) or every stage is bracketed on its own (full profile).
        float *half = nullptr;
        if (p->opt.fused_shrink && p->profile <= 1 && oct + 1 < p->n_oct && oct + 1 != tail_first)
            half = p->plane(oct + 1, 0);
        if (p->profile == 1 && oct == 0 && !p->chain) {          // no initial blur: the bracket opens here, five launches
            Scope *ch = new Scope(p, "Blur octave 0, scales 0-4 (one bracket)", true, 5.0 * W * H, nullptr, 0);
            if (ch->idx != (size_t)-1) p->events[ch->idx].launches = 5;
            p->chain = ch;
        }
        bool pyr0_bound = false;
        for (int s = 0; s < 5; s++) {
            if (p->profile > 1) snprintf(lab, sizeof lab, "Blur octave %d scale %d (%d taps)", oct, s, p->taps[s].n);
            // the events other streams (or the light profile) wait for are the END of a blur launch: they ride on it (launch_ev)
            hipEvent_t done = nullptr;
            if (s == 2 && oct == 0 && early) {
                if (!p->ev_early) HIPCHK(hipEventCreateWithFlags(&p->ev_early, SIFT_SYNC_EVENT));
                done = p->ev_early;
            } else if (s == 2 && oct == 1 && (fork || split)) {
                done = p->ev_p3;       // plane 3 of octave 1 and plane 0 of octave 2 exist (or will be shrunk from it): the chain below starts here
            } else if (s == 4 && oct == 0 && p->profile == 1 && p->chain) {
                Scope *ch = static_cast<Scope *>(p->chain);
                if (ch->idx != (size_t)-1) { done = p->events[ch->idx].b; ch->stop_bound = true; }     // closes the light profile's bracket
            } else if (s == 4 && oct == 0 && p->profile == 0 && two) {
                done = p->ev_pyr[0]; pyr0_bound = true;
            }
            {
                Scope sc(p, lab, true, (double)W * H, pyr, p->profile > 1 ? oct : -2);      // (light profile: octave 0 is inside the open bracket, the others are not timed)
                if (launch_blur(p, p->plane(oct, s), p->plane(oct, s + 1), W, H, p->taps[s], false, pyr, s == 2 ? half : nullptr, done)) handed[oct + 1] = true;
            }
            if (s == 2 && oct == 1 && (fork || split)) HIPCHK(hipStreamWaitEvent(below, p->ev_p3, 0));
        }
        if (oct == 0 && p->profile == 1) {
            Scope *ch = static_cast<Scope *>(p->chain);
            const size_t idx = ch ? ch->idx : (size_t)-1;
            delete ch; p->chain = nullptr;                  // (the closing event rode on the last blur launch)
            if (idx != (size_t)-1 && two) pyr0_done = p->events[idx].b;   // ... which is also "octave 0's pyramid exists"
        }
        if (two && oct == 0 && !pyr0_done && !pyr0_bound) HIPCHK(hipEventRecord(p->ev_pyr[0], pyr));
        return SIFTMI_OK;
    };
    auto wait_pyr0 = [&](hipStream_t st) { return hipStreamWaitEvent(st, pyr0_done ? pyr0_done : p->ev_pyr[0], 0); };
    auto wait_later = [&](hipStream_t st) { return early ? hipStreamWaitEvent(st, p->ev_early, 0) : wait_pyr0(st); };
    int rc = SIFTMI_OK;
    if (p->n_oct > 0) {
        // ---- octave 0 (tail_first_octave never takes it: the tail launch starts from plane 3 of an octave above)
        if ((rc = build_pyramid(0, p->stream))) return rc;
        if (p->maps_g0) {
            // octave 0's maps: on the idle stream beside detection and refinement (two streams), else in line
            if (two) {
                HIPCHK(wait_pyr0(p->stream2));
                launch_gradient_maps(p, 0, 1, p->stream2);
                HIPCHK(hipEventRecord(p->ev_maps0, p->stream2));
            } else launch_gradient_maps(p, 0, 1, p->stream);
        }
        launch_detect_octave(p, 0, p->stream);
        // octave 0's gradient maps ran beside detection and refinement: only orientation and description read them
        if (p->maps_g0 && two) HIPCHK(hipStreamWaitEvent(p->stream, p->ev_maps0, 0));
        if ((rc = launch_describe_group(p, 0, p->stream, slot++))) return rc;
    }
    if (p->n_oct > 1) {
        // ---- the later octaves: octave 1 (group 1 when the chain forks, else part of group 2), then the octaves below (group 2)
        if (two) HIPCHK(wait_later(later));
        for (int oct = 1; oct < p->n_oct; oct++) {
            hipStream_t st = (oct == 1) ? later : below;
            if (oct == tail_first) {   // this octave and every later one: one launch (k_tail.hpp)
                if ((rc = launch_tail(p, oct, st))) return rc;
                break;
            }
            if ((rc = build_pyramid(oct, st))) return rc;
            launch_detect_octave(p, oct, st);
            if (oct == 1 && fork) {
                if (p->maps_g1) launch_gradient_maps(p, 1, 2, later);
                if ((rc = launch_describe_group(p, 1, later, slot++))) return rc;
            }
        }
        hipStream_t desc2 = below;
        if (split) {              // the two detection chains meet: orientation and description of the group follow octave 1's chain
            HIPCHK(hipEventRecord(p->ev_det2, below));
            HIPCHK(hipStreamWaitEvent(later, p->ev_det2, 0));
            desc2 = later;
        }
        if (p->maps_g1) launch_gradient_maps(p, fork ? 2 : 1, p->n_oct, desc2);
        if ((rc = launch_describe_group(p, 2, desc2, slot++))) return rc;
    }
    if (p->chain) { delete static_cast<Scope *>(p->chain); p->chain = nullptr; }   // no octave closed the light-profile bracket (n_oct == 0)
    if (slot == 0) {                           // no octave at all: the counters (min / max) still come back
        HIPCHK(hipMemcpyAsync(&p->hb->c[0], p->cnt, sizeof(Counters), hipMemcpyDeviceToHost, p->stream));
        p->wait_s[0] = p->stream;
    }
    // (Capturing this fork / join into a hipGraph was tried in round 2: it replays correctly, but a graph launch is no faster
    // than the plain launches: small images are bound by the GPU-side latency of dependent kernels, not by host launch cost.)
    p->fin = p->stream;       // copies of the records are issued here after the wait: ordered before the next image's kernels
    return SIFTMI_OK;
}

// wait for everything in flight on the plan's streams, ignoring errors (error paths only)
void drain_streams(siftmi_plan *p) {
    for (hipStream_t s : {p->stream, p->stream2, p->stream3})
        if (s) (void)hipStreamSynchronize(s);
    p->in_flight = false;
}

// The reference's capacity per octave, enforced after the fact on an image that broke it (rare: reference_overflow): at
// most kpsize records of each octave stay -- which ones is as arbitrary as in the reference, whose atomic counter decides.
// Device list compacted in place through a gather into a temporary; a pinned result array of the call is stale afterwards
// (records_cut: siftmi_plan_keypoints copies the list out again).
int cap_octaves(siftmi_plan *p, const Counters &c, int64_t *n_io) {
    std::vector<int32_t> keep;
    std::vector<int64_t> kept((size_t)SIFT_MAX_OCTAVES, 0);
    std::vector<int32_t> aux;
    bool cut = false;
    // the record blocks in list order
    std::vector<std::pair<int64_t, int>> blocks;            // (base, group)
    for (int g = 0; g < SIFT_GROUPS; g++)
        if ((p->groups_cur >> g) & 1u) blocks.push_back({(int64_t)(c.rec_word[g] & 0x7fffffffu), g});
    std::sort(blocks.begin(), blocks.end());
    for (auto &bg : blocks) {
        const int g = bg.second;
        const int64_t n = std::min<int64_t>(std::min<int64_t>(c.g_out[g], p->grp[g].cap_out), std::max<int64_t>(0, *n_io - bg.first));
        if (n <= 0) continue;
        aux.resize((size_t)n);
        HIPCHK(hipMemcpy(aux.data(), p->grp[g].oaux, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            const int oct = std::min(std::max(aux[(size_t)i] >> 8, 0), SIFT_MAX_OCTAVES - 1);
            if (kept[(size_t)oct] < p->kpsize) { kept[(size_t)oct]++; keep.push_back((int32_t)(bg.first + i)); }
            else cut = true;
        }
    }
    if (!cut) return SIFTMI_OK;
    const size_t m = keep.size();
    int32_t *didx = nullptr; KpRecord *tmp = nullptr;
    if (hipMalloc((void **)&didx, std::max<size_t>(m, 1) * 4) != hipSuccess || hipMalloc((void **)&tmp, std::max<size_t>(m, 1) * sizeof(KpRecord)) != hipSuccess) {
        if (didx) (void)hipFree(didx);
        return fail(SIFTMI_ENOMEM, "hipMalloc failed while cutting the record list to the per-octave capacity");
    }
    hipError_t e = hipMemcpy(didx, keep.data(), m * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && m) {
        hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)std::min<size_t>((m * 36 + 255) / 256, 65535)), dim3(256), 0, p->stream,
                           (const KpRecord *)p->records, (const int *)didx, tmp, (int)m);
        e = hipMemcpyAsync(p->records, tmp, m * sizeof(KpRecord), hipMemcpyDeviceToDevice, p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    }
    (void)hipFree(didx); (void)hipFree(tmp);
    if (e != hipSuccess) return fail(SIFTMI_EDEVICE, "cutting the record list: %s", hipGetErrorString(e));
    *n_io = (int64_t)m;
    p->records_cut = true;
    return SIFTMI_OK;
}

// Wait for the image enqueued last on this plan; returns its record count (records stay on the device).
// SIFTMI_ETAILRETRY / SIFTMI_EGROW (internal): the image has to be enqueued again (the plan has been adjusted).
int plan_wait(siftmi_plan *p, int64_t *n_out, int32_t *overflow) {
    HIPCHK(hipSetDevice(p->device));
    // wait by polling: a blocking hipStreamSynchronize can add wake-up latency to a ~1 ms call
    const bool spin = p->opt.spin != 0;
    for (hipStream_t w : p->wait_s) {
        if (!w) continue;
        if (spin) {
            hipError_t q;
            while ((q = hipStreamQuery(w)) == hipErrorNotReady) __builtin_ia32_pause();
            if (q != hipSuccess) HIPCHK(q);
        }
        HIPCHK(hipStreamSynchronize(w));
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    p->in_flight = false;
    HIPCHK(hipGetLastError());
    // The complete copy of the counters: the one taken after every group of the image had reserved its record block
    // (a group reserves when its descriptor launch starts, i.e. after its orientation launch; the stream that ends last
    // copied after all of them).
    const Counters *hcp = nullptr;
    bool tail_timed_out = false;
    for (int k = 0; k < 3; k++) {
        if (!p->wait_s[k]) continue;
        const Counters &c = p->hb->c[k];
        tail_timed_out = tail_timed_out || c.tail_timeout;
        bool complete = true;
        for (int g = 0; g < SIFT_GROUPS; g++) if (((p->groups_cur >> g) & 1u) && !(c.rec_word[g] & 0x80000000u)) complete = false;
        if (complete && (!hcp || c.n_rec >= hcp->n_rec)) hcp = &c;
    }
    if (!hcp) { drain_streams(p); return fail(SIFTMI_EDEVICE, "no complete copy of the image's counters came back"); }
    const Counters &hc = *hcp;
    {
        auto dec = [](uint32_t u) { uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &v, 4); return f; };
        p->last_min = dec(hc.mm[0]); p->last_max = dec(hc.mm[1]);
    }
    if (!tail_timed_out && p->opt.tail_fault > 0 && p->opt.tail) { p->opt.tail_fault--; tail_timed_out = true; }   // injected (option "tail_fault")
    if (tail_timed_out) {
        // a workgroup of octave_tail_kernel stopped waiting for the octave above (k_tail.hpp): this image is incomplete.
        // From now on the plan walks the small octaves with the per-octave launches; the caller runs the image again.
        p->opt.tail = 0;
        p->tail_timeouts++;
        drain_streams(p);
        return fail(SIFTMI_ETAILRETRY, "octave_tail_kernel timed out waiting for the previous octave; tail launches disabled for this plan");
    }
    {
        // a list that was cut at its buffer's size: grow it and run the image again (what lies behind the cut was never made)
        bool grown = false;
        int rc = grow_lists(p, &hc, &grown);
        if (rc) return rc;
        if (grown) return fail(SIFTMI_EGROW, "a keypoint list was grown; the image runs again");
    }
    int64_t n = 0;
    for (int g = 0; g < SIFT_GROUPS; g++) if ((p->groups_cur >> g) & 1u) n += std::min<int64_t>(hc.g_out[g], p->grp[g].cap_out);
    n = std::min<int64_t>(std::min<int64_t>(n, hc.n_rec), p->cap_rec);
    // the reference's rule: kpsize entries per octave (see reference_overflow); an image that breaks it is flagged and cut to it
    int ovf = reference_overflow(p, hc) ? 1 : 0;
    // (a list that cannot grow any further has been cut: only an image beyond that rule gets there)
    for (int g = 0; g < SIFT_GROUPS; g++)
        if (((p->groups_cur >> g) & 1u) && (hc.g_kp[g] > p->grp[g].cap_kp || hc.g_out[g] > p->grp[g].cap_out)) ovf = 1;
    if (hc.n_rec > p->cap_rec) ovf = 1;
    // ... and so has a candidate list that stayed cut (grow_lists stops at what the reference's rule can admit): the per-scale
    // counts reference_overflow sees were taken from the cut list, so the cut itself is the evidence
    for (int g = 0; g < SIFT_GROUPS; g++) {
        if (!((p->groups_cur >> g) & 1u) || p->grp[g].cap_cand <= 0) continue;
        for (int o = 0; o < p->n_oct; o++)
            if (o < SIFT_MAX_OCTAVES && cand_group_of(p, o) == g && hc.n_cand[o] > p->grp[g].cap_cand) ovf = 1;
    }
    p->records_cut = false;
    if (ovf) { int rc = cap_octaves(p, hc, &n); if (rc) return rc; }
    if (p->profile == 1) {                   // light profile: running totals, read once by the caller's benchmark loop
        float tot = 0, bms = 0; int32_t bl = 0; double px = 0;
        if (siftmi_plan_last_kernel_ms(p, &tot, nullptr, nullptr, nullptr) == SIFTMI_OK && siftmi_plan_blur_ms(p, 0, &bms, &bl, &px) == SIFTMI_OK) {
            p->acc_calls++; p->acc_total_ms += tot; p->acc_b0_ms += bms; p->acc_b0_launches += bl; p->acc_b0_pixels += px;
        }
    }
    p->last_count = n;
    p->last_overflow = ovf;
    p->last_group0 = (int)std::min<int64_t>(hc.g_out[0], n);
    p->last_group1 = (int)n - p->last_group0;
    *n_out = n;
    if (overflow) *overflow = ovf;
    return SIFTMI_OK;
}

// One image, start to finish: enqueue + wait, run again when the plan had to be adjusted under it -- once after a tail
// time-out (plan_wait has switched to the per-octave launches), and after every growth of a list (at most one per list
// stage: candidates, refined, oriented, records; what lies behind a cut list only shows once the list is whole).
int plan_run(siftmi_plan *p, const void *image, int32_t image_dtype, int32_t image_is_device, bool sync_device, int64_t *n, int32_t *ovf) {
    int rc = SIFTMI_OK, tail_retries = 0;
    for (int attempt = 0; attempt < 12; attempt++) {
        rc = plan_enqueue(p, image, image_dtype, image_is_device, sync_device);
        if (!rc) rc = plan_wait(p, n, ovf);
        if (rc == SIFTMI_ETAILRETRY && tail_retries++ == 0) continue;
        if (rc == SIFTMI_EGROW) continue;
        break;
    }
    if (rc == SIFTMI_ETAILRETRY || rc == SIFTMI_EGROW) rc = SIFTMI_EDEVICE;
    // A failed enqueue or wait must not leave the plan marked busy: the next call would then report "still has an image in
    // flight" instead of this error.  Whatever was launched is drained here (the message of `rc` stays the last error).
    if (rc && p->in_flight) { const std::string keep = g_err; drain_streams(p); g_err = keep; }
    return rc;
}
}  // namespace

extern "C" {

int siftmi_plan_keypoints(siftmi_plan *p, const void *image, int32_t image_dtype, int32_t image_is_device, siftmi_keypoint *out,
                          int32_t out_is_device, int64_t capacity, int64_t *n_out, int32_t *overflow) {
    if (!p || !image || !n_out) return fail(SIFTMI_EINVAL, "null argument");
    if (capacity > 0 && !out) return fail(SIFTMI_EINVAL, "null output with capacity > 0");
    if (out_is_device < 0 || out_is_device > SIFTMI_OUT_PINNED) return fail(SIFTMI_EINVAL, "out_is_device must be 0 (host), 1 (device) or 2 (pinned host)");
    *n_out = 0;
    if (overflow) *overflow = 0;
    // Pinned host result array (siftmi_host_alloc): the descriptor kernels write every record to it as well (zero-copy over
    // PCIe while they run), so nothing is left to copy once the last kernel has ended.
    const bool pinned = out_is_device == SIFTMI_OUT_PINNED && out && capacity > 0;
    p->host_out = pinned ? reinterpret_cast<KpRecord *>(out) : nullptr;
    p->host_cap = pinned ? (int)(capacity < 0x7fffffff ? capacity : 0x7fffffff) : 0;
    int64_t n = 0;
    int32_t ovf = 0;
    int rc = plan_run(p, image, image_dtype, image_is_device, true, &n, &ovf);
    p->host_out = nullptr; p->host_cap = 0;
    if (rc) {
        // descriptor kernels that were already launched may still be writing to the caller's pinned block: nothing
        // returns before every stream of the plan has drained (the caller recycles the block on an error)
        if (pinned) drain_streams(p);
        return rc;
    }
    hipStream_t fin = p->fin;
    if (out == nullptr && capacity == 0) {
        // count-only call: the records stay on the device until siftmi_plan_fetch()
    } else {
        if (n > capacity) { n = capacity; rc = SIFTMI_ECAPACITY; g_err = "output capacity too small; result truncated"; }
        // (pinned: the kernels have written the records already -- unless the list was cut to the per-octave capacity afterwards)
        if (n > 0 && (!pinned || p->records_cut)) {
            HIPCHK(hipMemcpyAsync(out, p->records, (size_t)n * sizeof(KpRecord),
                                  out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, fin));
            HIPCHK(hipStreamSynchronize(fin));
        }
    }
    *n_out = n;
    if (overflow) *overflow = ovf;
    return rc;
}

// ---- pinned host blocks for result arrays (size-bucketed pool: hipHostMalloc / hipHostFree cost 0.1-1 ms each)
}  // extern "C"
namespace {
struct HostPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;                 // block -> bucket size
    std::unordered_map<size_t, std::vector<void *>> spare;   // bucket size -> free blocks
    std::vector<void *> trash;                               // blocks to hand back to the driver at the next allocation
    size_t live_bytes = 0, spare_bytes = 0;                  // handed out / parked in `spare`
    size_t limit = (size_t)2 << 30;                          // page-locked bytes the pool may hold in all (siftmi_host_pool_limit)
    static constexpr size_t kKeep = 8;                        // spare blocks kept per bucket
};
HostPool &host_pool() { static HostPool *hp = new HostPool(); return *hp; }   // leaked on purpose: no teardown order issues
// Sizes round up to a power of two from 64 KiB to 1 MiB and to a multiple of 2 MiB above (a 3000 x 3000 float32 frame takes
// 36 MiB, not 64: a stack-alignment loop keeps every aligned frame alive, and page-locked memory cannot be swapped).
size_t host_bucket(size_t bytes) {
    size_t bucket = (size_t)1 << 16;
    while (bucket < bytes && bucket < ((size_t)1 << 20)) bucket <<= 1;
    if (bucket < bytes) bucket = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
    return bucket;
}
}  // namespace
extern "C" {

// The pool never holds more than `limit` bytes of page-locked memory (live + spare): beyond it siftmi_host_alloc returns
// SIFTMI_ENOMEM and the Python layer hands out an ordinary array instead (one copy after the last kernel).
int siftmi_host_pool_limit(int64_t limit_bytes, int64_t *live_bytes, int64_t *spare_bytes) {
    HostPool &hp = host_pool();
    std::lock_guard<std::mutex> g(hp.mu);
    if (limit_bytes >= 0) hp.limit = (size_t)limit_bytes;
    if (live_bytes) *live_bytes = (int64_t)hp.live_bytes;
    if (spare_bytes) *spare_bytes = (int64_t)hp.spare_bytes;
    return SIFTMI_OK;
}

int siftmi_host_alloc(int64_t bytes, void **out) {
    if (!out || bytes < 0) return fail(SIFTMI_EINVAL, "bad argument");
    *out = nullptr;
    const size_t bucket = host_bucket((size_t)bytes);
    HostPool &hp = host_pool();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(hp.mu);
        drop.swap(hp.trash);
    }
    for (void *q : drop) (void)hipHostFree(q);           // (what siftmi_host_free set aside: released here, in a caller's context)
    drop.clear();
    {
        std::lock_guard<std::mutex> g(hp.mu);
        auto it = hp.spare.find(bucket);
        if (it != hp.spare.end() && !it->second.empty()) {
            void *q = it->second.back();
            it->second.pop_back();
            hp.spare_bytes -= bucket;
            hp.live[q] = bucket; hp.live_bytes += bucket;
            *out = q;
            return SIFTMI_OK;
        }
        if (hp.live_bytes + bucket > hp.limit)
            return fail(SIFTMI_ENOMEM, "pinned result pool: %zu bytes live, %zu more would pass the limit of %zu (siftmi_host_pool_limit)",
                        hp.live_bytes, bucket, hp.limit);
        // room for the new block: spare blocks of other sizes go first
        for (auto &kv : hp.spare) {
            while (hp.live_bytes + hp.spare_bytes + bucket > hp.limit && !kv.second.empty()) {
                drop.push_back(kv.second.back()); kv.second.pop_back(); hp.spare_bytes -= kv.first;
            }
        }
    }
    for (void *q : drop) (void)hipHostFree(q);
    void *q = nullptr;
    hipError_t e = hipHostMalloc(&q, bucket, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SIFTMI_ENOMEM, "hipHostMalloc(%zu): %s", bucket, hipGetErrorString(e)); }
    std::lock_guard<std::mutex> g(hp.mu);
    hp.live[q] = bucket; hp.live_bytes += bucket;
    *out = q;
    return SIFTMI_OK;
}

// A block returns to the pool; nothing is handed back to the driver here (hipHostFree synchronises the device, and this runs
// from a Python destructor): a block beyond what its bucket keeps (8 blocks, 3 from 16 MiB on) is set aside and released by
// the next siftmi_host_alloc, or by siftmi_host_pool_trim.
int siftmi_host_free(void *ptr) {
    if (!ptr) return SIFTMI_OK;
    HostPool &hp = host_pool();
    std::lock_guard<std::mutex> g(hp.mu);
    auto it = hp.live.find(ptr);
    if (it == hp.live.end()) return fail(SIFTMI_EINVAL, "not a siftmi_host_alloc block");
    const size_t bucket = it->second;
    hp.live.erase(it);
    hp.live_bytes -= bucket;
    std::vector<void *> &v = hp.spare[bucket];
    if (v.size() < (bucket >= ((size_t)16 << 20) ? (size_t)3 : HostPool::kKeep)) { v.push_back(ptr); hp.spare_bytes += bucket; }
    else hp.trash.push_back(ptr);
    return SIFTMI_OK;
}

// Release spare blocks until at most `keep_bytes` of them remain (0: all).  Synchronises the device (hipHostFree).
int siftmi_host_pool_trim(int64_t keep_bytes) {
    HostPool &hp = host_pool();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(hp.mu);
        drop.swap(hp.trash);
        for (auto &kv : hp.spare)
            while ((int64_t)hp.spare_bytes > keep_bytes && !kv.second.empty()) {
                drop.push_back(kv.second.back()); kv.second.pop_back(); hp.spare_bytes -= kv.first;
            }
    }
    for (void *q : drop) (void)hipHostFree(q);
    return SIFTMI_OK;
}

int siftmi_plan_fetch(siftmi_plan *p, siftmi_keypoint *out, int32_t out_is_device, int64_t first, int64_t count) {
    if (!p || (count > 0 && !out)) return fail(SIFTMI_EINVAL, "null argument");
    if (first < 0 || count < 0 || first + count > p->last_count) return fail(SIFTMI_EINVAL, "record range out of bounds");
    if (count == 0) return SIFTMI_OK;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(out, p->records + first, (size_t)count * sizeof(KpRecord),
                          out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return SIFTMI_OK;
}

// ---- batched, pipelined keypoints --------------------------------------------------------
// `lanes` independent plans (own planes, lists and streams) take the images round-robin.  Everything is enqueued
// without waiting; a lane is only waited for when it is about to be reused, and its records are then parked in a
// device arena by a device-to-device copy on the lane's own stream.  One device-to-host copy hands the whole batch back.
}  // extern "C"
struct siftmi_batch {
    int device = 0;
    std::vector<siftmi_plan *> lanes;
    std::vector<int> lane_image;          // image index in flight on each lane (-1: idle)
    uint8_t *arena = nullptr;             // parked records of the current batch, image after image
    size_t arena_cap = 0, arena_used = 0;
    std::vector<int64_t> counts, offsets;
    int64_t retired = 0, batch_size = 0;  // frames retired so far / frames in the current batch
    double blur0_ms = 0, blur0_pixels = 0;         // light profiling: octave-0 blur brackets summed over the frames of the batch
    int64_t blur0_launches = 0;
    siftmi_keypoint *const *host_outs = nullptr;   // optional: one caller-owned host array per frame, filled while the batch runs
    const int64_t *host_caps = nullptr;            // their capacities in records (a frame that does not fit stays parked in the arena)
    // host frames: uploads run on ONE copy stream, in frame order, one frame ahead of the kernel enqueue, into a ring of
    // lanes + 1 staging buffers (a lane's own staging buffer would hold its next upload back until its frame has retired:
    // with two lanes both uploads then share the link, both pyramids start together and the link idles while they run --
    // 4096^2 frames: 1.53 ms per frame against 1.16 ms of PCIe time)
    std::vector<void *> ring;
    std::vector<hipEvent_t> ring_ev;
    size_t ring_bytes = 0;
    hipStream_t copy_stream = nullptr;             // (the two halves of a frame on two copy streams -- two SDMA engines -- measured
                                                   // no better: 1.27-1.45 against 1.30-1.31 ms per 4096^2 frame on one box)
    const void *const *cur_images = nullptr;       // the frames of the call in progress (a lane re-runs its frame after a tail time-out)
    int32_t cur_dtype = 0, cur_is_device = 0;
    int64_t tail_retries = 0;                      // frames re-run since the batch was created
};
extern "C" {

int siftmi_batch_destroy(siftmi_batch *b) {
    if (!b) return SIFTMI_OK;
    for (siftmi_plan *p : b->lanes) siftmi_plan_destroy(p);
    hipSetDevice(b->device);
    if (b->arena) hipFree(b->arena);
    for (void *q : b->ring) hipFree(q);
    for (hipEvent_t e : b->ring_ev) hipEventDestroy(e);
    if (b->copy_stream) hipStreamDestroy(b->copy_stream);
    delete b;
    return SIFTMI_OK;
}

int siftmi_batch_create(int32_t height, int32_t width, int32_t in_dtype, int32_t device_id, const siftmi_params *params,
                        int32_t lanes, siftmi_batch **out) {
    if (!out) return fail(SIFTMI_EINVAL, "null argument");
    *out = nullptr;
    if (lanes < 1 || lanes > 64) return fail(SIFTMI_EINVAL, "lanes must be in 1..64, got %d", lanes);
    siftmi_batch *b = new (std::nothrow) siftmi_batch();
    if (!b) return fail(SIFTMI_ENOMEM, "host allocation failed");
    b->device = device_id;
    for (int l = 0; l < lanes; l++) {
        siftmi_plan *p = nullptr;
        // Small frames with many lanes: one stream per lane, all at the same priority.  Lanes already overlap each other,
        // and three streams per lane oversubscribe the hardware queues (512^2, 8 lanes: 0.70 ms per frame with three
        // streams, 0.43 ms with one; priorities between peer lanes cost another 20 %).  Few lanes of large frames keep the
        // three prioritised streams of a plain plan (4096^2, 2 lanes: 1.05 ms per frame with, 1.40 ms without priorities).
        const bool single_stream = lanes >= 4 && (int64_t)height * width <= (int64_t)2048 * 2048;
        g_lane_mode = single_stream ? 1 : 2;
        int rc = siftmi_plan_create(height, width, in_dtype, device_id, params, 0, &p);
        g_lane_mode = 0;
        if (rc) { std::string keep = g_err; siftmi_batch_destroy(b); g_err = keep; return rc; }
        if (single_stream) p->overlap = false;
        b->lanes.push_back(p);
        b->lane_image.push_back(-1);
    }
    *out = b;
    return SIFTMI_OK;
}

int siftmi_batch_set_params(siftmi_batch *b, const siftmi_params *params) {
    if (!b) return fail(SIFTMI_EINVAL, "null batch");
    for (siftmi_plan *p : b->lanes) { int rc = siftmi_plan_set_params(p, params); if (rc) return rc; }
    return SIFTMI_OK;
}

int siftmi_batch_set_option(siftmi_batch *b, const char *name, int64_t value) {
    if (!b) return fail(SIFTMI_EINVAL, "null batch");
    for (siftmi_plan *p : b->lanes) { int rc = siftmi_plan_set_option(p, name, value); if (rc) return rc; }
    return SIFTMI_OK;
}

int siftmi_batch_set_profile(siftmi_batch *b, int32_t level) {
    if (!b) return fail(SIFTMI_EINVAL, "null batch");
    if (level < 0 || level > 1) return fail(SIFTMI_EINVAL, "batch lanes support profile 0 or 1 (light)");
    for (siftmi_plan *p : b->lanes) p->profile = level;
    return SIFTMI_OK;
}

int siftmi_batch_blur_ms(const siftmi_batch *b, double *blur_ms, int64_t *blur_launches, double *blur_pixels) {
    if (!b) return fail(SIFTMI_EINVAL, "null batch");
    if (blur_ms) *blur_ms = b->blur0_ms;
    if (blur_launches) *blur_launches = b->blur0_launches;
    if (blur_pixels) *blur_pixels = b->blur0_pixels;
    return SIFTMI_OK;
}

int siftmi_batch_info(const siftmi_batch *b, int32_t *lanes, int64_t *bytes_allocated) {
    if (!b) return fail(SIFTMI_EINVAL, "null batch");
    if (lanes) *lanes = (int32_t)b->lanes.size();
    if (bytes_allocated) {
        int64_t t = (int64_t)b->arena_cap + (int64_t)(b->ring.size() * b->ring_bytes);
        for (const siftmi_plan *p : b->lanes) t += p->bytes;
        *bytes_allocated = t;
    }
    return SIFTMI_OK;
}

int siftmi_batch_tail_timeouts(const siftmi_batch *b, int64_t *timeouts, int32_t *lanes_with_tail) {
    if (!b) return fail(SIFTMI_EINVAL, "null batch");
    int64_t t = 0; int32_t on = 0;
    for (const siftmi_plan *p : b->lanes) { t += p->tail_timeouts; on += p->opt.tail ? 1 : 0; }
    if (timeouts) *timeouts = t;
    if (lanes_with_tail) *lanes_with_tail = on;
    return SIFTMI_OK;
}

}  // extern "C"
namespace {
// lane `l` has finished its image: learn the count and park the records in the arena (async D->D on the lane's stream)
int batch_retire(siftmi_batch *b, size_t l, int32_t *overflow) {
    siftmi_plan *p = b->lanes[l];
    const int img = b->lane_image[l];
    if (img < 0) return SIFTMI_OK;
    if (img >= b->batch_size) { b->lane_image[l] = -1; return fail(SIFTMI_EINVAL, "stale frame index %d on lane %zu", img, l); }
    int64_t n = 0; int32_t ovf = 0;
    int rc = plan_wait(p, &n, &ovf);
    if ((rc == SIFTMI_ETAILRETRY || rc == SIFTMI_EGROW) && b->cur_images) {
        // A workgroup of octave_tail_kernel gave up waiting (k_tail.hpp: most likely queued behind another lane's persistent
        // descriptor workgroups -- a batch-only condition) and plan_wait has switched the lane to the per-octave launches, or
        // the frame outran a list of the lane and plan_wait has grown it: the frame runs again on the same lane, as
        // siftmi_plan_keypoints does for a single plan (plan_run).
        if (rc == SIFTMI_ETAILRETRY) b->tail_retries++;
        // (a host frame is still in its ring slot: the slot is not reused before its frame has retired)
        const bool staged = !b->cur_is_device && !b->ring.empty();
        rc = plan_run(p, staged ? b->ring[(size_t)img % b->ring.size()] : b->cur_images[img], b->cur_dtype, staged ? 1 : b->cur_is_device, false, &n, &ovf);
    }
    if (rc == SIFTMI_ETAILRETRY || rc == SIFTMI_EGROW) rc = fail(SIFTMI_EDEVICE, "frame %d could not be completed after its plan was adjusted", img);
    if (rc) return rc;
    if (ovf && overflow) *overflow = 1;
    if (p->profile) {
        float ms = 0; int32_t nl = 0; double px = 0;
        if (siftmi_plan_blur_ms(p, 0, &ms, &nl, &px) == SIFTMI_OK) { b->blur0_ms += ms; b->blur0_launches += nl; b->blur0_pixels += px; }
    }
    const size_t bytes = (size_t)n * sizeof(KpRecord);
    if (b->host_outs && b->host_outs[img] && n <= b->host_caps[img]) {
        // straight into the caller's array for this frame: a blocking copy of one frame's records (~1.5 MB) costs the
        // host ~0.1 ms while the other lanes keep the GPU busy -- cheaper than parking everything and copying it at the end
        if (n > 0) {
            HIPCHK(hipMemcpyAsync(b->host_outs[img], p->records, bytes, hipMemcpyDeviceToHost, p->fin));
            HIPCHK(hipStreamSynchronize(p->fin));
        }
        b->counts[(size_t)img] = n;
        b->offsets[(size_t)img] = -1;                 // delivered
        b->retired++;
        b->lane_image[l] = -1;
        return SIFTMI_OK;
    }
    const size_t need = b->arena_used + bytes;
    if (need > b->arena_cap) {
        // grow to the projected size of the whole batch (records so far / frames retired x frames of the batch,
        // +25 %), so that a batch regrows the arena at most a couple of times: each regrow drains the device
        size_t cap = b->arena_cap ? b->arena_cap : ((size_t)1 << 22);
        const size_t projected = (size_t)((double)need / (double)(b->retired + 1) * (double)b->batch_size * 1.25);
        while (cap < need || cap < projected) cap *= 2;
        uint8_t *bigger = nullptr;
        hipError_t e = hipMalloc((void **)&bigger, cap);
        if (e != hipSuccess) return fail(SIFTMI_ENOMEM, "hipMalloc(%zu): %s", cap, hipGetErrorString(e));
        if (b->arena_used) {
            HIPCHK(hipDeviceSynchronize());               // earlier parking copies may still be in flight
            HIPCHK(hipMemcpy(bigger, b->arena, b->arena_used, hipMemcpyDeviceToDevice));
        }
        if (b->arena) { HIPCHK(hipDeviceSynchronize()); hipFree(b->arena); }
        b->arena = bigger; b->arena_cap = cap;
    }
    if (n > 0)
        HIPCHK(hipMemcpyAsync(b->arena + b->arena_used, p->records, bytes, hipMemcpyDeviceToDevice, p->fin));
    b->counts[(size_t)img] = n;
    b->offsets[(size_t)img] = (int64_t)(b->arena_used / sizeof(KpRecord));
    b->arena_used = need;
    b->retired++;
    b->lane_image[l] = -1;
    return SIFTMI_OK;
}

// Forget whatever is still in flight on the lanes (an earlier call that ended with an error): wait for each lane's
// streams, ignore the results, mark every lane idle.  Without this the next call would retire stale image indices
// into arrays sized for ITS batch.
void batch_drain(siftmi_batch *b) {
    for (size_t l = 0; l < b->lanes.size(); l++) {
        if (b->lane_image[l] < 0) continue;
        siftmi_plan *p = b->lanes[l];
        if (hipSetDevice(p->device) == hipSuccess) {
            drain_streams(p);
        }
        b->lane_image[l] = -1;
    }
}
}  // namespace
extern "C" {

int siftmi_batch_keypoints_into(siftmi_batch *b, const void *const *images, int32_t n_images, int32_t image_dtype,
                                int32_t images_are_device, siftmi_keypoint *const *host_outs, const int64_t *host_caps,
                                int64_t *counts, int64_t *offsets, int64_t *total_parked, int32_t *overflow) {
    if (!b || (n_images > 0 && !images) || !counts || !offsets || !total_parked) return fail(SIFTMI_EINVAL, "null argument");
    if (n_images < 0) return fail(SIFTMI_EINVAL, "negative image count");
    if ((host_outs == nullptr) != (host_caps == nullptr)) return fail(SIFTMI_EINVAL, "host_outs and host_caps go together");
    if (dtype_size(image_dtype) == 0) return fail(SIFTMI_EINVAL, "invalid input format (%d)", image_dtype);
    if (!b->lanes.empty() && image_dtype != b->lanes[0]->dtype && image_dtype != SIFTMI_F32)
        return fail(SIFTMI_EINVAL, "image dtype %d is neither the plan's (%d) nor float32", image_dtype, b->lanes[0]->dtype);
    HIPCHK(hipSetDevice(b->device));
    batch_drain(b);                      // no-op unless the previous call failed half way
    if (overflow) *overflow = 0;
    *total_parked = 0;
    b->arena_used = 0;
    b->retired = 0; b->batch_size = n_images;
    b->blur0_ms = 0; b->blur0_pixels = 0; b->blur0_launches = 0;
    b->host_outs = host_outs; b->host_caps = host_caps;
    b->cur_images = images; b->cur_dtype = image_dtype; b->cur_is_device = images_are_device;
    b->counts.assign((size_t)n_images, 0);
    b->offsets.assign((size_t)n_images, 0);
    const size_t L = b->lanes.size();
    int rc = SIFTMI_OK;
    // (from here on an error must not return at once: the lanes and the copy stream may be reading the caller's frames or
    // writing its arrays -- every failure goes through the drain at the end)
#define BATCHCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess && !rc) rc = fail(e_ == hipErrorOutOfMemory ? SIFTMI_ENOMEM : SIFTMI_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
    if (images_are_device) BATCHCHK(hipDeviceSynchronize());   // once per batch: order after the caller's streams
    const bool htime = L > 0 && b->lanes[0]->opt.host_timing;   // diagnostic: where the host thread spends the batch
    auto tnow = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_retire = 0, t_enqueue = 0;
    // host frames go through the upload ring (see siftmi_batch)
    const bool staged = !images_are_device && n_images > 0 && L > 0;
    const size_t frame_bytes = staged ? (size_t)b->lanes[0]->H * b->lanes[0]->W * dtype_size(image_dtype) : 0;
    if (staged && !rc) {
        if (b->ring_bytes < frame_bytes || b->ring.size() != L + 1) {
            BATCHCHK(hipDeviceSynchronize());
            for (void *q : b->ring) hipFree(q);
            b->ring.clear(); b->ring_bytes = 0;
            for (size_t k = 0; k < L + 1 && !rc; k++) {
                void *q = nullptr;
                BATCHCHK(hipMalloc(&q, frame_bytes));
                if (!rc) b->ring.push_back(q);
            }
            if (!rc) b->ring_bytes = frame_bytes;          // (counted by siftmi_batch_info: L + 1 staged frames in HBM)
            else { for (void *q : b->ring) hipFree(q); b->ring.clear(); }
        }
        // (the ring's events order a DMA copy before kernels of another stream: plain events, WITH their system-scope fence --
        // unlike the plan's kernel-to-kernel events, SIFT_SYNC_EVENT)
        while (b->ring_ev.size() < L + 1 && !rc) { hipEvent_t e = nullptr; BATCHCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); if (!rc) b->ring_ev.push_back(e); }
        if (!b->copy_stream && !rc) BATCHCHK(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    }
    auto upload = [&](int i) -> int {             // frame i -> ring slot i % (L + 1); the slot's previous frame (i - L - 1) has retired
        if (!images[i]) return fail(SIFTMI_EINVAL, "null image %d", i);
        const size_t k = (size_t)i % (L + 1);
        HIPCHK(hipMemcpyAsync(b->ring[k], images[i], frame_bytes, hipMemcpyHostToDevice, b->copy_stream));
        HIPCHK(hipEventRecord(b->ring_ev[k], b->copy_stream));
        return SIFTMI_OK;
    };
    if (staged && !rc) rc = upload(0);
    for (int i = 0; i < n_images && !rc; i++) {
        if (!images[i]) { rc = fail(SIFTMI_EINVAL, "null image %d", i); break; }
        const size_t l = (size_t)i % L;
        const double ta = htime ? tnow() : 0;
        if ((rc = batch_retire(b, l, overflow))) break;
        const double tb = htime ? tnow() : 0;
        if (htime) b->lanes[l]->opt.host_timing = 0;             // the per-call line would flood
        if (staged) {
            // frame i - L has just retired, so has every earlier one: slot (i + 1) % (L + 1), last used by frame i - L, is free
            if (i + 1 < n_images && (rc = upload(i + 1))) break;
            const size_t k = (size_t)i % (L + 1);
            BATCHCHK(hipStreamWaitEvent(b->lanes[l]->stream, b->ring_ev[k], 0));
            if (!rc) rc = plan_enqueue(b->lanes[l], b->ring[k], image_dtype, 1, false);
        } else
        rc = plan_enqueue(b->lanes[l], images[i], image_dtype, images_are_device, false);
        if (htime) { b->lanes[l]->opt.host_timing = 1; t_retire += tb - ta; t_enqueue += tnow() - tb; }
        if (rc) break;
        b->lane_image[l] = i;
    }
    if (htime) fprintf(stderr, "[siftmi] batch of %d: waiting for lanes %.0f us, enqueueing %.0f us\n", n_images, t_retire, t_enqueue);
    for (int i = n_images > (int)L ? n_images - (int)L : 0; i < n_images && !rc; i++) rc = batch_retire(b, (size_t)i % L, overflow);
    b->host_outs = nullptr; b->host_caps = nullptr; b->cur_images = nullptr;
    if (rc) {                                // nothing of this call may still read the caller's frames or write its arrays
        std::string keep = g_err;
        if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
        batch_drain(b);
        g_err = keep;
        return rc;
    }
#undef BATCHCHK
    HIPCHK(hipDeviceSynchronize());                            // the parking copies
    for (int i = 0; i < n_images; i++) { counts[i] = b->counts[(size_t)i]; offsets[i] = b->offsets[(size_t)i]; }
    *total_parked = (int64_t)(b->arena_used / sizeof(KpRecord));
    return SIFTMI_OK;
}

int siftmi_batch_keypoints(siftmi_batch *b, const void *const *images, int32_t n_images, int32_t image_dtype,
                           int32_t images_are_device, int64_t *counts, int64_t *offsets, int64_t *total, int32_t *overflow) {
    return siftmi_batch_keypoints_into(b, images, n_images, image_dtype, images_are_device, nullptr, nullptr, counts, offsets, total,
                                       overflow);
}

int siftmi_batch_fetch(siftmi_batch *b, siftmi_keypoint *out, int32_t out_is_device, int64_t first, int64_t count) {
    if (!b || (count > 0 && !out)) return fail(SIFTMI_EINVAL, "null argument");
    if (first < 0 || count < 0 || (size_t)(first + count) * sizeof(KpRecord) > b->arena_used) return fail(SIFTMI_EINVAL, "record range out of bounds");
    if (count == 0) return SIFTMI_OK;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipMemcpy(out, b->arena + (size_t)first * sizeof(KpRecord), (size_t)count * sizeof(KpRecord),
                     out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_plan_records_device(const siftmi_plan *p, const siftmi_keypoint **records, int64_t *count) {
    if (!p || !records) return fail(SIFTMI_EINVAL, "null argument");
    *records = reinterpret_cast<const siftmi_keypoint *>(p->records);
    if (count) *count = p->last_count;
    return SIFTMI_OK;
}

// Affine warp of an image of the plan's shape (transform.cl:22,116; host side alignment.py:325-348).
int siftmi_plan_transform(siftmi_plan *p, const void *image, int32_t image_is_device, int32_t channels, void *out,
                          int32_t out_is_device, int32_t OW, int32_t OH, const float *matrix, const float *offset, float fill,
                          int32_t mode, double *kernel_ms) {
    if (!p || !out || !matrix || !offset) return fail(SIFTMI_EINVAL, "null argument");
    if (channels != 1 && channels != 3) return fail(SIFTMI_EINVAL, "channels must be 1 (float32) or 3 (RGB8), got %d", channels);
    if (OW < 0 || OH < 0) return fail(SIFTMI_EINVAL, "negative output shape");
    HIPCHK(hipSetDevice(p->device));
    const size_t px = channels == 1 ? 4 : 3;
    const size_t in_bytes = (size_t)p->W * p->H * px, out_bytes = (size_t)OW * OH * px;
    const void *src = image;
    if (!image) {
        const int want = channels == 1 ? SIFTMI_F32 : SIFTMI_RGB8;
        if (p->raw_dtype != want)
            return fail(SIFTMI_EINVAL, "no staged input of the requested format (staged dtype %d, wanted %d)", p->raw_dtype, want);
        src = p->raw;
    } else if (image_is_device) {
        HIPCHK(hipDeviceSynchronize());
    } else {
        if (p->warp_in_bytes < in_bytes) {
            if (p->warp_in) hipFree(p->warp_in);
            p->warp_in = nullptr; p->warp_in_bytes = 0;
            hipError_t e = hipMalloc(&p->warp_in, in_bytes ? in_bytes : 16);
            if (e != hipSuccess) return fail(SIFTMI_ENOMEM, "hipMalloc(%zu): %s", in_bytes, hipGetErrorString(e));
            p->warp_in_bytes = in_bytes;
        }
        HIPCHK(hipMemcpyAsync(p->warp_in, image, in_bytes, hipMemcpyHostToDevice, p->stream));
        src = p->warp_in;
    }
    void *dst = out;
    if (!out_is_device) {
        if (p->warp_out_bytes < out_bytes) {
            if (p->warp_out) hipFree(p->warp_out);
            p->warp_out = nullptr; p->warp_out_bytes = 0;
            hipError_t e = hipMalloc(&p->warp_out, out_bytes ? out_bytes : 16);
            if (e != hipSuccess) return fail(SIFTMI_ENOMEM, "hipMalloc(%zu): %s", out_bytes, hipGetErrorString(e));
            p->warp_out_bytes = out_bytes;
        }
        dst = p->warp_out;
    }
    if (!p->ev_wa) { HIPCHK(hipEventCreate(&p->ev_wa)); HIPCHK(hipEventCreate(&p->ev_wb)); }
    AffineArgs a{matrix[0], matrix[1], matrix[2], matrix[3], offset[0], offset[1], fill, mode};
    if (OW > 0 && OH > 0) {
        const int xcols = channels == 1 ? 64 : 256;   // RGB: 4 pixels per thread
        const dim3 grid((unsigned)((OW + xcols - 1) / xcols), (unsigned)((OH + 3) / 4)), block(64, 4);
        hipEventRecord(p->ev_wa, p->stream);
        if (channels == 1)
            hipLaunchKernelGGL(transform_kernel, grid, block, 0, p->stream, (const float *)src, (float *)dst, a, p->W, p->H, OW, OH);
        else
            hipLaunchKernelGGL(transform_rgb_kernel, grid, block, 0, p->stream, (const uint8_t *)src, (uint8_t *)dst, a, p->W, p->H, OW, OH);
        hipEventRecord(p->ev_wb, p->stream);
        if (!out_is_device) HIPCHK(hipMemcpyAsync(out, dst, out_bytes, hipMemcpyDeviceToHost, p->stream));
    }
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    if (kernel_ms) {
        float ms = 0;
        if (OW > 0 && OH > 0) hipEventElapsedTime(&ms, p->ev_wa, p->ev_wb);
        *kernel_ms = ms;
    }
    return SIFTMI_OK;
}

int siftmi_plan_get_minmax(const siftmi_plan *p, float *mn, float *mx) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (mn) *mn = p->last_min;
    if (mx) *mx = p->last_max;
    return SIFTMI_OK;
}

int siftmi_plan_profile(const siftmi_plan *p, char *buf, int64_t buflen) {
    if (!p || !buf || buflen < 1) return fail(SIFTMI_EINVAL, "bad argument");
    std::string s;
    char line[160];
    for (size_t i = 0; i < p->n_events; i++) {
        float ms = 0;
        hipEventElapsedTime(&ms, p->events[i].a, p->events[i].b);
        snprintf(line, sizeof line, "%s\t%.6f\n", p->events[i].label.c_str(), ms);
        s += line;
    }
    snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return SIFTMI_OK;
}

int siftmi_plan_last_kernel_ms(const siftmi_plan *p, float *total_ms, float *blur_ms, int32_t *blur_launches,
                               double *blur_pixels) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (!p->profile) return fail(SIFTMI_EINVAL, "plan was created with profile=0");
    float tot = 0;                            // light profile: not measured (0)
    if (p->profile > 1) {
        for (int k = 0; k < 3; k++) {
            float tb = 0;
            if (p->wait_s[k] && hipEventElapsedTime(&tb, p->ev_first, p->ev_last[k]) == hipSuccess && tb > tot) tot = tb;
        }
    }
    if (total_ms) *total_ms = tot;
    return siftmi_plan_blur_ms(p, -1, blur_ms, blur_launches, blur_pixels);
}

int siftmi_plan_profile_totals(siftmi_plan *p, int32_t reset, int64_t *calls, double *total_ms, double *blur0_ms,
                               int64_t *blur0_launches, double *blur0_pixels) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (calls) *calls = p->acc_calls;
    if (total_ms) *total_ms = p->acc_total_ms;
    if (blur0_ms) *blur0_ms = p->acc_b0_ms;
    if (blur0_launches) *blur0_launches = p->acc_b0_launches;
    if (blur0_pixels) *blur0_pixels = p->acc_b0_pixels;
    if (reset) { p->acc_calls = 0; p->acc_b0_launches = 0; p->acc_total_ms = p->acc_b0_ms = p->acc_b0_pixels = 0; }
    return SIFTMI_OK;
}

int siftmi_plan_blur_ms(const siftmi_plan *p, int32_t octave, float *blur_ms, int32_t *blur_launches, double *blur_pixels) {
    if (!p) return fail(SIFTMI_EINVAL, "null plan");
    if (!p->profile) return fail(SIFTMI_EINVAL, "plan was created with profile=0");
    float bms = 0; int bl = 0; double px = 0;
    for (size_t i = 0; i < p->n_events; i++)
        if (p->events[i].is_blur && (octave < 0 || p->events[i].octave == octave)) {
            float ms = 0;
            hipEventElapsedTime(&ms, p->events[i].a, p->events[i].b);
            bms += ms; bl += p->events[i].launches; px += p->events[i].pixels;
        }
    if (blur_ms) *blur_ms = bms;
    if (blur_launches) *blur_launches = bl;
    if (blur_pixels) *blur_pixels = px;
    return SIFTMI_OK;
}

// ============================================================================================
// MatchPlan
}  // extern "C"

struct siftmi_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t size = 0;
    int profile = 0;
    uint8_t *kp1 = nullptr, *kp2 = nullptr;
    int64_t cap1 = 0, cap2 = 0;
    int2 *pairs = nullptr;
    int64_t cap_pairs = 0;
    MatchPartial *partial = nullptr;
    int64_t cap_partial = 0;
    int *counter = nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    float last_ms = 0;
    // profile != 0: the events of match.py:226-263 -- "copy H->D KP_1", "copy H->D KP_2", "matching", "copy D->H match" -- as
    // device times of the last call in ms (-1: the stage did not run: a device-resident list, no pair to copy)
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float stage_ms[4] = {-1.f, -1.f, -1.f, -1.f};
    // ROI mask (MatchPlan.set_roi, match.py:312-320) and the scratch of the masked / mutual variants
    int8_t *roi = nullptr;
    int64_t cap_roi = 0;
    int roi_w = 0, roi_h = 0;
    uint8_t *q1 = nullptr, *l1 = nullptr, *q2 = nullptr, *l2 = nullptr;   // per-keypoint flags (as query / as list element)
    int64_t cap_q1 = 0, cap_l1 = 0, cap_q2 = 0, cap_l2 = 0;
    int *nearest = nullptr;
    int64_t cap_nearest = 0;
    int2 *pairs2 = nullptr;
    int64_t cap_pairs2 = 0;
};

namespace {
int ensure(void **ptr, int64_t *cap, int64_t need, size_t elem) {
    if (need <= *cap && *ptr) return SIFTMI_OK;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    hipError_t e = hipMalloc(ptr, (size_t)(need > 0 ? need : 1) * elem);
    if (e != hipSuccess) return fail(SIFTMI_ENOMEM, "hipMalloc(%lld x %zu): %s", (long long)need, elem, hipGetErrorString(e));
    *cap = need;
    return SIFTMI_OK;
}
}  // namespace

extern "C" {

int siftmi_match_create(int64_t size, int32_t device_id, int32_t profile, siftmi_matcher **out) {
    if (!out) return fail(SIFTMI_EINVAL, "null argument");
    *out = nullptr;
    if (size < 1) return fail(SIFTMI_EINVAL, "size must be >= 1");
    int ndev = siftmi_device_count();
    if (ndev < 1) return fail(SIFTMI_EDEVICE, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(SIFTMI_EINVAL, "device %d out of range", device_id);
    HIPCHK(hipSetDevice(device_id));
    siftmi_matcher *m = new (std::nothrow) siftmi_matcher();
    if (!m) return fail(SIFTMI_ENOMEM, "host allocation failed");
    m->device = device_id; m->size = size; m->profile = profile;
    int rc = SIFTMI_OK;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "hipStreamCreate failed");
    if (!rc) rc = ensure((void **)&m->kp1, &m->cap1, size, 144);
    if (!rc) rc = ensure((void **)&m->kp2, &m->cap2, size, 144);
    if (!rc) rc = ensure((void **)&m->pairs, &m->cap_pairs, size, sizeof(int2));
    if (!rc && hipMalloc((void **)&m->counter, 16) != hipSuccess) rc = fail(SIFTMI_ENOMEM, "hipMalloc failed");
    if (!rc) { hipEventCreate(&m->ea); hipEventCreate(&m->eb); }
    if (!rc && profile) for (hipEvent_t &e : m->ev) if (hipEventCreate(&e) != hipSuccess) rc = fail(SIFTMI_EDEVICE, "hipEventCreate failed");
    if (rc) { std::string keep = g_err; siftmi_match_destroy(m); g_err = keep; return rc; }
    *out = m;
    return SIFTMI_OK;
}

int siftmi_match_destroy(siftmi_matcher *m) {
    if (!m) return SIFTMI_OK;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->kp1) hipFree(m->kp1);
    if (m->kp2) hipFree(m->kp2);
    if (m->pairs) hipFree(m->pairs);
    if (m->partial) hipFree(m->partial);
    for (void *q : {(void *)m->roi, (void *)m->q1, (void *)m->l1, (void *)m->q2, (void *)m->l2, (void *)m->nearest, (void *)m->pairs2})
        if (q) hipFree(q);
    if (m->counter) hipFree(m->counter);
    if (m->ea) hipEventDestroy(m->ea);
    if (m->eb) hipEventDestroy(m->eb);
    for (hipEvent_t e : m->ev) if (e) hipEventDestroy(e);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    return SIFTMI_OK;
}

int siftmi_match_set_roi(siftmi_matcher *m, const int8_t *roi, int32_t roi_width, int32_t roi_height) {
    if (!m) return fail(SIFTMI_EINVAL, "null matcher");
    HIPCHK(hipSetDevice(m->device));
    if (!roi) { m->roi_w = m->roi_h = 0; return SIFTMI_OK; }       // unset_roi
    if (roi_width < 1 || roi_height < 1) return fail(SIFTMI_EINVAL, "bad ROI shape %d x %d", roi_width, roi_height);
    int rc = ensure((void **)&m->roi, &m->cap_roi, (int64_t)roi_width * roi_height, 1);
    if (rc) return rc;
    HIPCHK(hipMemcpy(m->roi, roi, (size_t)roi_width * roi_height, hipMemcpyHostToDevice));
    m->roi_w = roi_width; m->roi_h = roi_height;
    return SIFTMI_OK;
}

namespace {
// one direction of the brute-force scan: partials of `nq` queries against `nl` list elements, folded by the merge kernel
int match_direction(siftmi_matcher *m, const uint8_t *dq, int64_t nq, const uint8_t *dl, int64_t nl, const uint8_t *qflag,
                    const uint8_t *lflag, float ratio_th, int2 *pairs, int cap, int *nearest) {
    // 2-D decomposition: query blocks x partitions of the list, enough workgroups to fill 256 CUs
    const int qblocks = (int)((nq + 256 * SIFT_MATCH_QPT - 1) / (256 * SIFT_MATCH_QPT));
    int nparts = (2048 + qblocks - 1) / qblocks;
    const int max_parts = (int)((nl + 4 * SIFT_MATCH_TILE - 1) / (4 * SIFT_MATCH_TILE));
    if (nparts > max_parts) nparts = max_parts;
    const int min_parts = (int)((nl + SIFT_MATCH_MAX_PART - 1) / SIFT_MATCH_MAX_PART);     // 16-bit index inside a partition
    if (nparts < min_parts) nparts = min_parts;
    if (nparts < 1) nparts = 1;
    int part_len = (int)((nl + nparts - 1) / nparts);
    part_len = (part_len + SIFT_MATCH_TILE - 1) / SIFT_MATCH_TILE * SIFT_MATCH_TILE;
    nparts = (int)((nl + part_len - 1) / part_len);
    int rc;
    if ((rc = ensure((void **)&m->partial, &m->cap_partial, (int64_t)nparts * nq, sizeof(MatchPartial)))) return rc;
    const dim3 grid((unsigned)qblocks, (unsigned)nparts);
    if (lflag)
        hipLaunchKernelGGL(match_partial_kernel<true>, grid, dim3(256), 0, m->stream, dq, (int)nq, dl, (int)nl, part_len, m->partial, qflag, lflag);
    else
        hipLaunchKernelGGL(match_partial_kernel<false>, grid, dim3(256), 0, m->stream, dq, (int)nq, dl, (int)nl, part_len, m->partial,
                           (const uint8_t *)nullptr, (const uint8_t *)nullptr);
    hipLaunchKernelGGL(match_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, m->stream,
                       (const MatchPartial *)m->partial, (int)nq, nparts, ratio_th, pairs, m->counter, cap, qflag, nearest);
    return SIFTMI_OK;
}
}  // namespace

int siftmi_match_ex(siftmi_matcher *m, const siftmi_keypoint *kp1, int64_t n1, int32_t kp1_is_device,
                    const siftmi_keypoint *kp2, int64_t n2, int32_t kp2_is_device, float ratio_th, int32_t roi_mode,
                    int32_t mutual, int32_t *pairs, int64_t capacity, int64_t *n_out, int64_t *n_total) {
    if (!m || !n_out) return fail(SIFTMI_EINVAL, "null argument");
    if (n1 < 0 || n2 < 0 || n1 > 0x7fffffff || n2 > 0x7fffffff) return fail(SIFTMI_EINVAL, "bad list size");
    if ((n1 > 0 && !kp1) || (n2 > 0 && !kp2)) return fail(SIFTMI_EINVAL, "null keypoint list");
    if (roi_mode < 0 || roi_mode > 2) return fail(SIFTMI_EINVAL, "roi_mode must be 0 (off), 1 (matching_valid) or 2 (strict)");
    if (roi_mode && !(m->roi && m->roi_w > 0)) return fail(SIFTMI_EINVAL, "roi_mode %d without a region of interest (siftmi_match_set_roi)", roi_mode);
    HIPCHK(hipSetDevice(m->device));
    *n_out = 0;
    if (n_total) *n_total = 0;
    for (float &v : m->stage_ms) v = -1.f;
    if (n1 == 0 || n2 == 0) return SIFTMI_OK;   // dist1 == dist2 == 1e12 -> ratio 1, never < ratio_th
    if (kp1_is_device || kp2_is_device) HIPCHK(hipDeviceSynchronize());
    const uint8_t *d1 = (const uint8_t *)kp1, *d2 = (const uint8_t *)kp2;
    int rc;
    const bool prof = m->profile && m->ev[0];
    if (prof) hipEventRecord(m->ev[0], m->stream);
    if (!kp1_is_device) {
        if ((rc = ensure((void **)&m->kp1, &m->cap1, n1, 144))) return rc;
        HIPCHK(hipMemcpyAsync(m->kp1, kp1, (size_t)n1 * 144, hipMemcpyHostToDevice, m->stream));
        d1 = m->kp1;
    }
    if (prof) hipEventRecord(m->ev[1], m->stream);
    if (!kp2_is_device) {
        if ((rc = ensure((void **)&m->kp2, &m->cap2, n2, 144))) return rc;
        HIPCHK(hipMemcpyAsync(m->kp2, kp2, (size_t)n2 * 144, hipMemcpyHostToDevice, m->stream));
        d2 = m->kp2;
    }
    if (prof) hipEventRecord(m->ev[2], m->stream);
    // match.py:241-243,252: output capacity = max(self.kpsize, min(n1, n2))
    int64_t cap = m->size;
    if ((n1 < n2 ? n1 : n2) > cap) cap = (n1 < n2 ? n1 : n2);
    if ((rc = ensure((void **)&m->pairs, &m->cap_pairs, cap, sizeof(int2)))) return rc;
    HIPCHK(hipMemsetAsync(m->counter, 0, 8, m->stream));
    hipEventRecord(m->ea, m->stream);
    const uint8_t *qf1 = nullptr, *lf2 = nullptr;
    if (roi_mode) {
        if ((rc = ensure((void **)&m->q1, &m->cap_q1, n1, 1)) || (rc = ensure((void **)&m->l1, &m->cap_l1, n1, 1)) ||
            (rc = ensure((void **)&m->q2, &m->cap_q2, n2, 1)) || (rc = ensure((void **)&m->l2, &m->cap_l2, n2, 1))) return rc;
        hipLaunchKernelGGL(match_roi_flags_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, m->stream, d1, (int)n1,
                           (const int8_t *)m->roi, m->roi_w, m->roi_h, roi_mode, m->q1, m->l1);
        hipLaunchKernelGGL(match_roi_flags_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, m->stream, d2, (int)n2,
                           (const int8_t *)m->roi, m->roi_w, m->roi_h, roi_mode, m->q2, m->l2);
        qf1 = m->q1; lf2 = m->l2;
    }
    if ((rc = match_direction(m, d1, n1, d2, n2, qf1, lf2, ratio_th, m->pairs, (int)cap, nullptr))) return rc;
    int2 *result = m->pairs;
    int *result_counter = m->counter;
    int count = 0;
    if (mutual) {
        // reverse scan: nearest list-1 keypoint of every list-2 keypoint over the same masked distances
        if ((rc = ensure((void **)&m->nearest, &m->cap_nearest, n2, sizeof(int))) ||
            (rc = ensure((void **)&m->pairs2, &m->cap_pairs2, cap, sizeof(int2)))) return rc;
        const uint8_t *qf2 = nullptr, *lf1 = nullptr;
        if (roi_mode) {
            hipLaunchKernelGGL(match_reverse_flags_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, m->stream, (const uint8_t *)m->l2, (int)n2, m->q2);
            hipLaunchKernelGGL(match_reverse_list_flags_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, m->stream, (const uint8_t *)m->q1, (int)n1, m->l1);
            qf2 = m->q2; lf1 = m->l1;
        }
        HIPCHK(hipMemcpyAsync(&count, m->counter, 4, hipMemcpyDeviceToHost, m->stream));   // forward count (the partial buffer is reused below)
        if ((rc = match_direction(m, d2, n2, d1, n1, qf2, lf1, ratio_th, nullptr, 0, m->nearest))) return rc;
        HIPCHK(hipStreamSynchronize(m->stream));
        const int nfwd = count < cap ? count : (int)cap;
        if (nfwd > 0)
            hipLaunchKernelGGL(match_mutual_filter_kernel, dim3((unsigned)((nfwd + 255) / 256)), dim3(256), 0, m->stream,
                               (const int2 *)m->pairs, nfwd, (const int *)m->nearest, m->pairs2, m->counter + 1);
        result = m->pairs2; result_counter = m->counter + 1;
    }
    hipEventRecord(m->eb, m->stream);
    HIPCHK(hipMemcpyAsync(&count, result_counter, 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipGetLastError());
    hipEventElapsedTime(&m->last_ms, m->ea, m->eb);
    if (n_total) *n_total = count;
    int64_t n = count < cap ? count : cap;
    rc = SIFTMI_OK;
    if (n > capacity) { n = capacity; rc = SIFTMI_ECAPACITY; g_err = "pair capacity too small; result truncated"; }
    if (prof) {
        m->stage_ms[2] = m->last_ms;
        if (!kp1_is_device) hipEventElapsedTime(&m->stage_ms[0], m->ev[0], m->ev[1]);
        if (!kp2_is_device) hipEventElapsedTime(&m->stage_ms[1], m->ev[1], m->ev[2]);
    }
    if (n > 0) {
        if (!pairs) return fail(SIFTMI_EINVAL, "null pairs buffer");
        if (prof) {
            hipEventRecord(m->ev[3], m->stream);
            HIPCHK(hipMemcpyAsync(pairs, result, (size_t)n * sizeof(int2), hipMemcpyDeviceToHost, m->stream));
            hipEventRecord(m->ev[4], m->stream);
            HIPCHK(hipStreamSynchronize(m->stream));
            hipEventElapsedTime(&m->stage_ms[3], m->ev[3], m->ev[4]);
        } else {
            HIPCHK(hipMemcpy(pairs, result, (size_t)n * sizeof(int2), hipMemcpyDeviceToHost));
        }
    }
    *n_out = n;
    return rc;
}

int siftmi_match(siftmi_matcher *m, const siftmi_keypoint *kp1, int64_t n1, int32_t kp1_is_device,
                 const siftmi_keypoint *kp2, int64_t n2, int32_t kp2_is_device, float ratio_th, int32_t *pairs,
                 int64_t capacity, int64_t *n_out, int64_t *n_total) {
    return siftmi_match_ex(m, kp1, n1, kp1_is_device, kp2, n2, kp2_is_device, ratio_th, 0, 0, pairs, capacity, n_out, n_total);
}

int siftmi_match_last_kernel_ms(const siftmi_matcher *m, float *ms) {
    if (!m || !ms) return fail(SIFTMI_EINVAL, "null argument");
    *ms = m->last_ms;
    return SIFTMI_OK;
}

int siftmi_match_last_stage_ms(const siftmi_matcher *m, float *ms4) {
    if (!m || !ms4) return fail(SIFTMI_EINVAL, "null argument");
    if (!m->profile) return fail(SIFTMI_EINVAL, "the matcher was created without profiling");
    for (int i = 0; i < 4; i++) ms4[i] = m->stage_ms[i];
    return SIFTMI_OK;
}

}  // extern "C"

// ============================================================================================
// Per-stage entry points: host arrays in, one kernel, host arrays out.
namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t n) {
        hipError_t e = hipMalloc(&p, n ? n : 16);
        return e == hipSuccess ? SIFTMI_OK : fail(SIFTMI_ENOMEM, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
    }
    int upload(const void *h, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        HIPCHK(hipMemcpy(p, h, n, hipMemcpyHostToDevice));
        return SIFTMI_OK;
    }
    template <class T> T *as() { return (T *)p; }
};

int stage_begin(int device_id) {
    int ndev = siftmi_device_count();
    if (ndev < 1) return fail(SIFTMI_EDEVICE, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(SIFTMI_EINVAL, "device %d out of range", device_id);
    HIPCHK(hipSetDevice(device_id));
    return SIFTMI_OK;
}

int stage_end() {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipGetLastError());
    return SIFTMI_OK;
}

}  // namespace

extern "C" {

int siftmi_stage_gaussian_taps(float sigma, int32_t size, float *out) {
    if (!out) return fail(SIFTMI_EINVAL, "null argument");
    return gaussian_taps(sigma, size, out);
}

int32_t siftmi_stage_xcd_order(int32_t id, int32_t n) { return (n > 0 && id >= 0 && id < n) ? siftk::xcd_contiguous(id, n) : -1; }

int siftmi_stage_minmax_normalize(int32_t dev, const float *in, float *out, int32_t W, int32_t H, float *mn, float *mx) {
    int rc = stage_begin(dev); if (rc) return rc;
    const size_t N = (size_t)W * H;
    DevBuf a, b, mm;
    if ((rc = a.upload(in, N * 4)) || (rc = b.alloc(N * 4)) || (rc = mm.alloc(8))) return rc;
    hipLaunchKernelGGL(minmax_init, dim3(1), dim3(1), 0, 0, mm.as<uint32_t>());
    hipLaunchKernelGGL(minmax_kernel, dim3(grid_for((int64_t)N / 4, 256, 2048)), dim3(256), 0, 0, a.as<float>(), (int64_t)N, mm.as<uint32_t>());
    hipLaunchKernelGGL(normalize_kernel<0>, dim3(grid_for((int64_t)N, 256, 4096)), dim3(256), 0, 0, (const void *)a.as<float>(), b.as<float>(), (int64_t)N, (const uint32_t *)mm.as<uint32_t>());
    if ((rc = stage_end())) return rc;
    uint32_t h[2];
    HIPCHK(hipMemcpy(h, mm.p, 8, hipMemcpyDeviceToHost));
    auto dec = [](uint32_t u) { uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &v, 4); return f; };
    if (mn) *mn = dec(h[0]);
    if (mx) *mx = dec(h[1]);
    if (out) HIPCHK(hipMemcpy(out, b.p, N * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_stage_blur(int32_t dev, const float *in, float *out, int32_t W, int32_t H, const float *taps, int32_t ntaps) {
    int rc = stage_begin(dev); if (rc) return rc;
    if (ntaps < 1 || ntaps > 64) return fail(SIFTMI_EINVAL, "ntaps must be in 1..64");
    const size_t N = (size_t)W * H;
    DevBuf a, b, t, dt;
    if ((rc = a.upload(in, N * 4)) || (rc = b.alloc(N * 4)) || (rc = t.alloc(N * 4))) return rc;
    Taps tp; tp.n = ntaps;
    for (int i = 0; i < ntaps; i++) tp.t[i] = taps[i];
    if ((rc = dt.upload(tp.t, sizeof tp.t))) return rc;
    tp.dev = dt.as<float>();
    if (!launch_blur_tiled<false>(g_default_options, 0, a.as<float>(), b.as<float>(), W, H, tp, nullptr))
        launch_blur_generic(0, a.as<float>(), b.as<float>(), t.as<float>(), W, H, tp, nullptr, false);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(out, b.p, N * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

// The blur stage with a plan's launch choices exposed, so that a stage test reaches every instance of the marching team
// kernel (the plain entry point above takes planes below 1400^2 pixels only through the tiled kernel): the normalising
// instance behind a min/max pass, the typed-frame instances, both workgroup orders, any workgroup count.
int siftmi_stage_blur_ex(int32_t dev, const void *in, int32_t in_dtype, float *out, int32_t W, int32_t H, const float *taps, int32_t ntaps,
                         int32_t norm, int32_t xcd_map, int32_t march_wgs, int32_t *kernel_used) {
    int rc = stage_begin(dev); if (rc) return rc;
    if (!in || !out || !taps || W < 1 || H < 1) return fail(SIFTMI_EINVAL, "null argument or empty plane");
    if (ntaps < 1 || ntaps > 64) return fail(SIFTMI_EINVAL, "ntaps must be in 1..64");
    const size_t esz = dtype_size(in_dtype);
    if (!esz || in_dtype == SIFTMI_F64) return fail(SIFTMI_EINVAL, "input format %d has no fused blur", in_dtype);
    if (in_dtype != SIFTMI_F32 && !(norm && ntaps == 15)) return fail(SIFTMI_EINVAL, "typed frames enter through the normalising 15-tap blur only");
    const size_t N = (size_t)W * H;
    DevBuf a, b, t, dt, mm;
    if ((rc = a.upload(in, N * esz)) || (rc = b.alloc(N * 4)) || (rc = t.alloc(N * 4)) || (rc = mm.alloc(8))) return rc;
    Taps tp; tp.n = ntaps;
    for (int i = 0; i < ntaps; i++) tp.t[i] = taps[i];
    if ((rc = dt.upload(tp.t, sizeof tp.t))) return rc;
    tp.dev = dt.as<float>();
    Options opt = g_default_options;
    opt.xcd_map = (xcd_map & 1) ? 1 : 0;
    opt.march_prio = (xcd_map & 2) ? 0 : 2;       // (forced where not off: the stage planes are smaller than the rule's)
    opt.march_wgs = march_wgs > 0 ? march_wgs : 0;
    uint32_t *mmp = mm.as<uint32_t>();
    if (norm) {
        hipLaunchKernelGGL(minmax_init, dim3(1), dim3(1), 0, 0, mmp);
        if (in_dtype == SIFTMI_F32) {
            hipLaunchKernelGGL(minmax_kernel, dim3(grid_for((int64_t)N / 4, 256, 2048)), dim3(256), 0, 0, a.as<float>(), (int64_t)N, mmp);
        } else {
            SIFTMI_TYPED_DISPATCH(in_dtype, hipLaunchKernelGGL(minmax_typed_kernel<DT>, dim3(grid_for((int64_t)N / TypedChunk<DT>::PX, 256, 256)), dim3(256), 0, 0,
                                                               (const void *)a.p, (int64_t)N, mmp));
        }
    }
    int used = 0;
    if (in_dtype != SIFTMI_F32) {
        bool ok = false;
        SIFTMI_TYPED_DISPATCH(in_dtype, ok = launch_init_blur_dt<DT>(opt, 0, (const void *)a.p, b.as<float>(), W, H, tp, mmp));
        if (!ok) return fail(SIFTMI_EINVAL, "no fused blur for input format %d", in_dtype);
        used = (march_plane(W, H) && taps_symmetric(tp) && opt.march) ? 2 : 1;
    } else {
        const int r = norm ? launch_blur_tiled<true>(opt, 0, a.as<float>(), b.as<float>(), W, H, tp, mmp)
                           : launch_blur_tiled<false>(opt, 0, a.as<float>(), b.as<float>(), W, H, tp, mmp);
        if (!r) launch_blur_generic(0, a.as<float>(), b.as<float>(), t.as<float>(), W, H, tp, mmp, norm != 0);
        else {
            bool marchable = false;
            for (int n : {11, 15, 17, 21, 27}) marchable = marchable || n == ntaps;
            used = (march_plane(W, H) && taps_symmetric(tp) && opt.march && marchable) ? 2 : 1;
        }
    }
    if ((rc = stage_end())) return rc;
    if (kernel_used) *kernel_used = used;
    HIPCHK(hipMemcpy(out, b.p, N * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_stage_dog(int32_t dev, const float *blur_a, const float *blur_b, float *out, int64_t n) {
    int rc = stage_begin(dev); if (rc) return rc;
    DevBuf a, b, o;
    if ((rc = a.upload(blur_a, (size_t)n * 4)) || (rc = b.upload(blur_b, (size_t)n * 4)) || (rc = o.alloc((size_t)n * 4))) return rc;
    hipLaunchKernelGGL(dog_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, 0, a.as<float>(), b.as<float>(), o.as<float>(), n);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(out, o.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_stage_local_maxmin(int32_t dev, const float *blurs, int32_t W, int32_t H, int32_t octsize,
                              const siftmi_params *par, float *out, int64_t capacity, int64_t *n_out) {
    int rc = stage_begin(dev); if (rc) return rc;
    if (!par || !n_out) return fail(SIFTMI_EINVAL, "null argument");
    const size_t N = (size_t)W * H;
    DevBuf b, c, cnt;
    if ((rc = b.upload(blurs, 6 * N * 4)) || (rc = c.alloc((size_t)capacity * 16)) || (rc = cnt.alloc(sizeof(Counters)))) return rc;
    HIPCHK(hipMemset(cnt.p, 0, sizeof(Counters)));
    BlurPlanes bp;
    for (int s = 0; s < 6; s++) bp.p[s] = b.as<float>() + (size_t)s * N;
    const int border = par->border_dist;
    if (W > 2 * border && H > 2 * border) {
        const int rows = extrema_strip_rows(W, H, border);
        const int nx = (W - 2 * border + 61) / 62, ny = (H - 2 * border + rows - 1) / rows;
        const float edth = (octsize <= 1) ? par->edge_thresh0 : par->edge_thresh;
        hipLaunchKernelGGL(extrema_kernel<false>, dim3((unsigned)((nx * ny + 3) / 4)), dim3(256), 0, 0, bp, W, H, border, rows,
                           contrast_threshold(*par), edth, c.as<float4>(), &cnt.as<Counters>()->n_cand[0], (int)capacity, RefineArgs{}, -1, -1, 1);
    }
    if ((rc = stage_end())) return rc;
    Counters hc;
    HIPCHK(hipMemcpy(&hc, cnt.p, sizeof hc, hipMemcpyDeviceToHost));
    int64_t n = hc.n_cand[0] < capacity ? hc.n_cand[0] : capacity;
    if (n > 0) HIPCHK(hipMemcpy(out, c.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    *n_out = n;
    return SIFTMI_OK;
}

int siftmi_stage_interp(int32_t dev, const float *blurs, int32_t W, int32_t H, const float *cand, int64_t n,
                        const siftmi_params *par, float *out, int32_t *out_scale, int64_t *n_out) {
    int rc = stage_begin(dev); if (rc) return rc;
    if (!par || !n_out) return fail(SIFTMI_EINVAL, "null argument");
    const size_t N = (size_t)W * H;
    DevBuf b, c, k, ks, cnt;
    if ((rc = b.upload(blurs, 6 * N * 4)) || (rc = c.upload(cand, (size_t)n * 16)) || (rc = k.alloc((size_t)n * 16)) ||
        (rc = ks.alloc((size_t)n * 4)) || (rc = cnt.alloc(sizeof(Counters)))) return rc;
    Counters hc{};
    hc.n_cand[0] = (int)n;
    HIPCHK(hipMemcpy(cnt.p, &hc, sizeof hc, hipMemcpyHostToDevice));
    BlurPlanes bp;
    for (int s = 0; s < 6; s++) bp.p[s] = b.as<float>() + (size_t)s * N;
    Counters *dc = cnt.as<Counters>();
    hipLaunchKernelGGL(refine_kernel, dim3(grid_for(n, 256, 512)), dim3(256), 0, 0, bp, W, H, (const float4 *)c.as<float4>(),
                       (const int *)&dc->n_cand[0], (int)n, par->peak_thresh, (float)par->init_sigma, k.as<float4>(),
                       ks.as<int>(), &dc->g_kp[0], (int)n, 0, &dc->c_scale[0][0]);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(&hc, cnt.p, sizeof hc, hipMemcpyDeviceToHost));
    const int64_t m = hc.g_kp[0];
    if (m > 0) {
        HIPCHK(hipMemcpy(out, k.p, (size_t)m * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out_scale, ks.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    *n_out = m;
    return SIFTMI_OK;
}

int siftmi_stage_compact(int32_t dev, const float *kps, int64_t n, int64_t start, int64_t end, float *out, int64_t *n_out) {
    int rc = stage_begin(dev); if (rc) return rc;
    if (!n_out || start < 0 || end < start || end > n) return fail(SIFTMI_EINVAL, "bad range");
    DevBuf a, o, cnt;
    if ((rc = a.upload(kps, (size_t)n * 16)) || (rc = o.alloc((size_t)n * 16)) || (rc = cnt.alloc(16))) return rc;
    // the reference compacts in place of a second buffer that starts as a copy of the head: entries below `start` stay
    HIPCHK(hipMemcpy(o.p, a.p, (size_t)start * 16, hipMemcpyDeviceToDevice));
    const int first = (int)start;
    HIPCHK(hipMemcpy(cnt.p, &first, 4, hipMemcpyHostToDevice));
    if (end > start)
        hipLaunchKernelGGL(compact_kernel, dim3(grid_for(end - start, 256, 1024)), dim3(256), 0, 0, (const float4 *)a.as<float4>(),
                           o.as<float4>(), cnt.as<int>(), (int)start, (int)end, (int)n);
    if ((rc = stage_end())) return rc;
    int count = 0;
    HIPCHK(hipMemcpy(&count, cnt.p, 4, hipMemcpyDeviceToHost));
    if (count > 0) HIPCHK(hipMemcpy(out, o.p, (size_t)count * 16, hipMemcpyDeviceToHost));
    *n_out = count;
    return SIFTMI_OK;
}

int siftmi_stage_gradient(int32_t dev, const float *img, float *grad, float *ori, int32_t W, int32_t H) {
    int rc = stage_begin(dev); if (rc) return rc;
    const size_t N = (size_t)W * H;
    DevBuf a, g, o;
    if ((rc = a.upload(img, N * 4)) || (rc = g.alloc(N * 4)) || (rc = o.alloc(N * 4))) return rc;
    hipLaunchKernelGGL(gradient_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, 0, a.as<float>(),
                       g.as<float>(), o.as<float>(), W, H);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(grad, g.p, N * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ori, o.p, N * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_stage_orientation(int32_t dev, const float *blurs, int32_t W, int32_t H, int32_t octsize, const float *kps,
                             const int32_t *kp_scale, int64_t n, const siftmi_params *par, float *out,
                             int32_t *out_scale, int64_t capacity, int64_t *n_out) {
    int rc = stage_begin(dev); if (rc) return rc;
    if (!par || !n_out) return fail(SIFTMI_EINVAL, "null argument");
    const size_t N = (size_t)W * H;
    DevBuf b, k, ks, o, oa, cnt;
    if ((rc = b.upload(blurs, 6 * N * 4)) || (rc = k.upload(kps, (size_t)n * 16)) || (rc = ks.upload(kp_scale, (size_t)n * 4)) ||
        (rc = o.alloc((size_t)capacity * 16)) || (rc = oa.alloc((size_t)capacity * 4)) || (rc = cnt.alloc(sizeof(Counters)))) return rc;
    Counters hc{};
    hc.g_kp[0] = (int)n;
    HIPCHK(hipMemcpy(cnt.p, &hc, sizeof hc, hipMemcpyHostToDevice));
    int oct = 0;
    while ((1 << oct) < octsize && oct < SIFT_MAX_OCTAVES - 1) oct++;
    if ((1 << oct) != octsize) return fail(SIFTMI_EINVAL, "octsize must be a power of two");
    OctaveTable tab;
    memset(&tab, 0, sizeof tab);
    tab.base = b.as<float>(); tab.off[oct] = 0; tab.W[oct] = W; tab.H[oct] = H;
    std::vector<int32_t> aux((size_t)n);
    for (int64_t i = 0; i < n; i++) aux[(size_t)i] = kp_scale[i] | (oct << 8);
    HIPCHK(hipMemcpy(ks.p, aux.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(orientation_kernel<false>, dim3(grid_for(n * 64, 256, 1024)), dim3(256), 0, 0, tab,
                       par->ori_sigma, (const float4 *)k.as<float4>(), (const int *)ks.as<int>(), cnt.as<Counters>(), 0, (int)n,
                       o.as<float4>(), oa.as<int>(), (int)capacity, 0, 512);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(&hc, cnt.p, sizeof hc, hipMemcpyDeviceToHost));
    int64_t m = hc.g_out[0] < capacity ? hc.g_out[0] : capacity;
    if (m > 0) {
        HIPCHK(hipMemcpy(out, o.p, (size_t)m * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out_scale, oa.p, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < m; i++) out_scale[i] &= 0xff;
    }
    *n_out = m;
    return SIFTMI_OK;
}

int siftmi_stage_descriptor(int32_t dev, const float *blurs, int32_t W, int32_t H, int32_t octsize, const float *kps,
                            const int32_t *kp_scale, int64_t n, uint8_t *desc) {
    int rc = stage_begin(dev); if (rc) return rc;
    const size_t N = (size_t)W * H;
    DevBuf b, k, ks, r;
    if ((rc = b.upload(blurs, 6 * N * 4)) || (rc = k.upload(kps, (size_t)n * 16)) || (rc = ks.upload(kp_scale, (size_t)n * 4)) ||
        (rc = r.alloc((size_t)n * sizeof(KpRecord)))) return rc;
    int oct = 0;
    while ((1 << oct) < octsize && oct < SIFT_MAX_OCTAVES - 1) oct++;
    if ((1 << oct) != octsize) return fail(SIFTMI_EINVAL, "octsize must be a power of two");
    OctaveTable tab;
    memset(&tab, 0, sizeof tab);
    tab.base = b.as<float>(); tab.off[oct] = 0; tab.W[oct] = W; tab.H[oct] = H;
    std::vector<int32_t> aux((size_t)n);
    for (int64_t i = 0; i < n; i++) aux[(size_t)i] = kp_scale[i] | (oct << 8);
    if (n > 0) {
        HIPCHK(hipMemcpy(ks.p, aux.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        // row-interval form unless a window is too large for its row tables (same expressions as the kernel: keypoints_cpu.cl:57-62)
        bool block_ok = true;
        for (int64_t i = 0; i < n; i++) {
            const float spacing = kps[4 * i + 2] / (float)octsize * 3.0f;
            if (!((int)((1.414f * spacing * 2.5f) + 0.5f) <= SIFT_DESC_MAXRAD)) block_ok = false;
        }
        if (block_ok)
            hipLaunchKernelGGL(descriptor_kernel<false>, dim3(grid_for(n, 4, 2048)), dim3(256), 0, 0, tab,
                               (const float4 *)k.as<float4>(), (const int *)ks.as<int>(), (Counters *)nullptr, 0, 0, (int)n,
                               (int)n, r.as<KpRecord>(), (int)n, (KpRecord *)nullptr, 0, 0, 0, 1 << 30, 1 << 30);
        else
            hipLaunchKernelGGL(descriptor_stream_kernel, dim3(grid_for(n, 4, 2048)), dim3(256), 0, 0, tab,
                               (const float4 *)k.as<float4>(), (const int *)ks.as<int>(), (Counters *)nullptr, 0, 0, (int)n,
                               (int)n, r.as<KpRecord>(), (int)n, (KpRecord *)nullptr, 0);
    }
    if ((rc = stage_end())) return rc;
    std::vector<KpRecord> h((size_t)n);
    if (n > 0) HIPCHK(hipMemcpy(h.data(), r.p, (size_t)n * sizeof(KpRecord), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) memcpy(desc + (size_t)i * 128, h[(size_t)i].desc, 128);
    return SIFTMI_OK;
}

int siftmi_stage_shrink(int32_t dev, const float *in, float *out, int32_t W, int32_t H) {
    int rc = stage_begin(dev); if (rc) return rc;
    const int SW = W / 2, SH = H / 2;
    DevBuf a, o;
    if ((rc = a.upload(in, (size_t)W * H * 4)) || (rc = o.alloc((size_t)SW * SH * 4))) return rc;
    if (SW > 0 && SH > 0)
        hipLaunchKernelGGL(shrink_kernel, shrink_grid(SW, SH), dim3(256), 0, 0, a.as<float>(), o.as<float>(), W, SW, SH);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(out, o.p, (size_t)SW * SH * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_stage_convert(int32_t dev, const void *in, int32_t dt, float *out, int32_t W, int32_t H) {
    int rc = stage_begin(dev); if (rc) return rc;
    const size_t N = (size_t)W * H;
    if (dtype_size(dt) == 0) return fail(SIFTMI_EINVAL, "invalid input format (%d)", dt);
    DevBuf a, o;
    if ((rc = a.upload(in, N * dtype_size(dt))) || (rc = o.alloc(N * 4))) return rc;
    const int g = grid_for((int64_t)N, 256, 4096);
    switch (dt) {
        case SIFTMI_F32: HIPCHK(hipMemcpy(o.p, a.p, N * 4, hipMemcpyDeviceToDevice)); break;
        case SIFTMI_U8: hipLaunchKernelGGL(convert_kernel<uint8_t>, dim3(g), dim3(256), 0, 0, a.as<uint8_t>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_U16: hipLaunchKernelGGL(convert_kernel<uint16_t>, dim3(g), dim3(256), 0, 0, a.as<uint16_t>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_U32: hipLaunchKernelGGL(convert_kernel<uint32_t>, dim3(g), dim3(256), 0, 0, a.as<uint32_t>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_U64: hipLaunchKernelGGL(convert_kernel<uint64_t>, dim3(g), dim3(256), 0, 0, a.as<uint64_t>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_I32: hipLaunchKernelGGL(convert_kernel<int32_t>, dim3(g), dim3(256), 0, 0, a.as<int32_t>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_I64: hipLaunchKernelGGL(convert_kernel<int64_t>, dim3(g), dim3(256), 0, 0, a.as<int64_t>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_F64: hipLaunchKernelGGL(convert_kernel<double>, dim3(g), dim3(256), 0, 0, a.as<double>(), o.as<float>(), (int64_t)N); break;
        case SIFTMI_RGB8: hipLaunchKernelGGL(convert_rgb_kernel, dim3(g), dim3(256), 0, 0, a.as<uint8_t>(), o.as<float>(), (int64_t)N); break;
    }
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(out, o.p, N * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

int siftmi_stage_math(int32_t dev, int32_t fn, const float *a, const float *b, float *out, int64_t n) {
    int rc = stage_begin(dev); if (rc) return rc;
    DevBuf da, db, o;
    if ((rc = da.upload(a, (size_t)n * 4)) || (rc = db.upload(b ? b : a, (size_t)n * 4)) || (rc = o.alloc((size_t)n * 4))) return rc;
    hipLaunchKernelGGL(math_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, 0, fn, da.as<float>(), db.as<float>(), o.as<float>(), n);
    if ((rc = stage_end())) return rc;
    HIPCHK(hipMemcpy(out, o.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SIFTMI_OK;
}

}  // extern "C"
