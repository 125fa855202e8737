"""dev: tools/ubench/blur_team_x.hpp = the product's blur_team_kernel (k_pyramid.hpp), verbatim, plus a start-up stagger between
the workgroups that share a CU.  python tools/ubench/gen_blur_team_x.py"""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(R, "sift_pyocl_amd/csrc/k_pyramid.hpp")).read()
a = src.index("template <int N, bool NORM, int S, int DT = 0, int HW = 2>\n__global__ __launch_bounds__(64 * HW + 128) void blur_team_kernel")
b = src.index("// Generic (any tap count, incl. even sizes) two-pass blur")
k = src[a:b].replace("blur_team_kernel", "blur_team_x")
k = k.replace("float *__restrict__ next0,     // not null: also out[2y][2x] -> next0 (the next octave's plane 0)\n"
              "                                                          int xcd_map, int prio) {",
              "float *__restrict__ next0, int xcd_map, int prio, int stagger_mode, int stagger_units) {")
assert "stagger_mode" in k
old = "    const int x0 = bx * G::TX;\n"
new = '''    const int x0 = bx * G::TX;
    // ---- the experiment: workgroups that share a CU start out of phase.  Every workgroup of a launch does the same work at the
    // same speed and the grid is resident at once: without this the three (four) workgroups of a CU reach their load, barrier
    // and LDS phases together, and nothing runs under them.
    if (stagger_mode) {
        int s;
        if (stagger_mode == 1) s = (((int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y) >> 8) & 3;      // launch order: 256 CUs per round
        else s = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3;                            // HW_ID.wave_id: the wave slot on its SIMD
        for (int i = 0; i < s * stagger_units; i++) __builtin_amdgcn_s_sleep(16);                          // 16 x 64 cycles
    }
'''
assert old in k
k = k.replace(old, new, 1)
# ---- second experiment: the taps in VGPR pairs instead of SGPR pairs (TV bit 0: horizontal pass, bit 1: vertical march)
k = k.replace("template <int N, bool NORM, int S, int DT = 0, int HW = 2>", "template <int N, bool NORM, int S, int DT = 0, int HW = 2, int TV = 0>")
old = "    // horizontal pass of one sub-block (np row pairs) in place, by the HW waves of the H team (one wave per row pair)\n"
new = old + """    f32x2 tvh[(TV & 1) ? N : 1], tvv[(TV & 2) ? (N + 1) / 2 : 1];
"""
assert old in k
k = k.replace(old, new, 1)
old = "                const f32x2 tp2 = {tp, tp};\n"
assert old in k
k = k.replace(old, "                const f32x2 tp2 = (TV & 1) ? tvh[(TV & 1) ? N - 1 - q : 0] : (f32x2){tp, tp};\n", 1)
old = "const f32x2 t2 = {taps.t[k], taps.t[k]};"
assert old in k
k = k.replace(old, "const f32x2 t2 = (TV & 2) ? tvv[(TV & 2) ? k : 0] : (f32x2){taps.t[k], taps.t[k]};", 1)
old = "    if (role == 0) {\n        for (int blk = 0; blk < nblocks; blk++) {\n"
assert old in k
k = k.replace(old, """    if (role == 0) {
        if (TV & 1) {
#pragma unroll
            for (int q = 0; q < N; q++) { float x; asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(taps.t[q])); tvh[q] = (f32x2){x, x}; }
        }
        for (int blk = 0; blk < nblocks; blk++) {
""", 1)
old = "        return;\n    }\n    for (int blk = 0; blk < nblocks; blk++) {\n"
assert old in k
k = k.replace(old, """        return;
    }
    if (TV & 2) {
#pragma unroll
        for (int q = 0; q < (N + 1) / 2; q++) { float x; asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(taps.t[q])); tvv[q] = (f32x2){x, x}; }
    }
    for (int blk = 0; blk < nblocks; blk++) {
""", 1)
# ---- third instrument: a per-wave timeline (TR): shader-clock marks at the phase boundaries of every step, kept in LDS, dumped at the end
k = k.replace("int HW = 2, int TV = 0>", "int HW = 2, int TV = 0, int TR = 0>")
assert "int xcd_map, int prio, int stagger_mode, int stagger_units) {" in k
k = k.replace("int xcd_map, int prio, int stagger_mode, int stagger_units) {", "int xcd_map, int prio, int stagger_mode, int stagger_units, unsigned long long *trace) {")
old = "    float *sbase = reinterpret_cast<float *>(smem4);\n"
assert old in k
k = k.replace(old, old + """    constexpr int TRN = 160;                   // marks per wave
    unsigned long long *tl = reinterpret_cast<unsigned long long *>(sbase + 3 * BUF) + (threadIdx.x >> 6) * TRN;
    int ti = 2;
    if (TR && (threadIdx.x & 63) == 0) tl[TRN - 2] = __builtin_amdgcn_s_memrealtime();      // the chip-wide 100 MHz clock at the wave's start ...
#define TMARK() do { if (TR) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && ti < TRN) tl[ti] = t_; ti++; } } while (0)
#define TDUMP() do { if (TR && trace) { __builtin_amdgcn_wave_barrier(); if ((threadIdx.x & 63) == 0) { \\
        tl[0] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) | ((unsigned long long)role << 32) | ((unsigned long long)(blockIdx.x + gridDim.x * blockIdx.y) << 40); \\
        tl[1] = (unsigned long long)ti; tl[TRN - 1] = __builtin_amdgcn_s_memrealtime(); } __builtin_amdgcn_wave_barrier(); \\
        unsigned long long *dst_ = trace + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 4 + (threadIdx.x >> 6)) * TRN; \\
        for (int e_ = threadIdx.x & 63; e_ < TRN; e_ += 64) dst_[e_] = tl[e_]; } } while (0)
""", 1)
old = "                hpass(sbase + (g % 3) * BUF, SS::pairs(sub));\n                __syncthreads();\n"
assert old in k
k = k.replace(old, "                TMARK();\n                hpass(sbase + (g % 3) * BUF, SS::pairs(sub));\n                TMARK();\n                __syncthreads();\n", 1)
old = "        return;\n    }\n    if (TV & 2) {"
assert old in k
k = k.replace(old, "        TMARK();\n        TDUMP();\n        return;\n    }\n    if (TV & 2) {", 1)
old = "            // (1) vertical march of the previous step\n"
assert old in k
k = k.replace(old, "            TMARK();\n" + old, 1)
old = "            // (2) stage step g+1 (already in registers) and look ahead to g+2\n"
assert old in k
k = k.replace(old, "            TMARK();\n" + old, 1)
old = "                stage(nxt, SS::pairs(sub1));\n"
assert old in k
k = k.replace(old, old + "                TMARK();\n", 1)
old = "                if (exists(blk2, sub2)) prefetch(blk2, sub2, SS::pairs(sub2));\n            }\n            __syncthreads();\n"
assert old in k
k = k.replace(old, "                if (exists(blk2, sub2)) prefetch(blk2, sub2, SS::pairs(sub2));\n            }\n            TMARK();\n            __syncthreads();\n", 1)
old = "#undef VPASS\n"
assert old in k
k = k.replace(old, "    TMARK();\n    TDUMP();\n" + old, 1)
# ---- fourth experiment: wave priority as negative feedback on progress (prio_mode): a workgroup that is BEHIND outranks one that is
# ahead, so the three workgroups of a CU -- which the arbiter otherwise serves oldest first: they end at 54 / 76 / 100 % of the
# launch, the last one alone on its CU -- advance together
assert "unsigned long long *trace) {" in k
k = k.replace("unsigned long long *trace) {", "unsigned long long *trace, int prio_mode) {")
old = "    int g = 0;                                     // step counter: buffer of step g is g % 3\n"
assert old in k
k = k.replace(old, old + """    const int Tsteps_x = (nblocks - 1) * S + last_subs;
    int pq_x = -1;
    auto xprio = [&](int gg) {
        if (!prio_mode) return;
        int q = (gg * 4) / Tsteps_x;                 // quarter of the march this workgroup is in
        if (prio_mode == 2) q = 3 - q;
        if (prio_mode == 3) q = (gg * 2) / Tsteps_x * 3;
        if (prio_mode >= 4) {                       // thirds, and a short last phase (the final spread is what the last phase lets the arbiter serialise)
            const int last = prio_mode == 4 ? 2 : (prio_mode == 5 ? 1 : (prio_mode == 6 ? 3 : 4));
            const int body = Tsteps_x - last;
            q = gg >= body ? 3 : (gg * 3) / body;
        }
        if (prio_mode >= 11 && prio_mode <= 16) {    // quarters with a handicap: a YOUNGER workgroup (higher wave slot on its SIMD) leaves every level later,
            // i.e. is kept ahead -- inside a level the arbiter serves the older workgroups first, and the youngest ends last
            const int slot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3;
            const int den = prio_mode == 11 ? 12 : (prio_mode == 12 ? 8 : (prio_mode == 13 ? 6 : (prio_mode == 14 ? 16 : (prio_mode == 15 ? 24 : 4))));
            const int pnum = gg * den - slot * Tsteps_x;               // (g - slot * T / den) * den
            q = pnum <= 0 ? 0 : (pnum * 4) / (Tsteps_x * den);
        }
        if (prio_mode == 9) { const int q8 = (gg * 8) / Tsteps_x; q = (q8 >> 1) + ((q8 & 1) & (gg & 1)); }                   // eighths: the odd ones alternate between two levels
        if (prio_mode == 10) { const int q16 = (gg * 16) / Tsteps_x; q = (q16 >> 2) + (((gg & 3) < (q16 & 3)) ? 1 : 0); }      // sixteenths, dithered over four steps
        if (q > 3) q = 3;
        if (q == pq_x) return;
        pq_x = q;
        if (q <= 0) __builtin_amdgcn_s_setprio(3);
        else if (q == 1) __builtin_amdgcn_s_setprio(2);
        else if (q == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    };
    xprio(0);
""", 1)
assert k.count("\n                g++;\n") == 1 and k.count("\n            g++;\n") == 1
k = k.replace("\n                g++;\n", "\n                g++;\n                xprio(g);\n", 1)
k = k.replace("\n            g++;\n", "\n            g++;\n            xprio(g);\n", 1)
# ---- the prologue in the timeline: four marks in the V waves (entry of the prologue, first loads requested, sub-block 0 staged = its
# loads have arrived, second loads requested), two in the H waves; the first mark of the loops then closes the prologue barrier
old = "    if (role == 1) {\n        prefetch(0, 0, SS::pairs(0));\n        stage(sbase, SS::pairs(0));\n"
assert old in k
k = k.replace(old, "    if (role == 1) {\n        TMARK();\n        prefetch(0, 0, SS::pairs(0));\n        TMARK();\n        stage(sbase, SS::pairs(0));\n        TMARK();\n", 1)
old = "        else if (exists(1, 0)) prefetch(1, 0, SS::pairs(0));\n    }\n    __syncthreads();\n"
assert old in k
k = k.replace(old, "        else if (exists(1, 0)) prefetch(1, 0, SS::pairs(0));\n        TMARK();\n    } else { TMARK(); TMARK(); }\n    __syncthreads();\n", 1)
# ---- fifth experiment (LP): row-pair pitch of 16 bytes mod 32 and an H wave that takes TWO row pairs at once (even lanes one, odd
# lanes the other): the sixteen lanes the LDS serves together then read sixteen different 16-byte bank groups -- no 2-way conflict
k = k.replace("int TV = 0, int TR = 0>", "int TV = 0, int TR = 0, int LP = 0>")
k = k.replace("    constexpr int BUF = G::NPS * G::PITCH * 2;    // floats per LDS buffer\n", "    constexpr int XP = G::PITCH + (LP ? 2 : 0);\n    constexpr int BUF = G::NPS * XP * 2;    // floats per LDS buffer\n", 1)
assert "constexpr int XP" in k
k = k.replace("G::PITCH", "XP").replace("constexpr int XP = XP + (LP ? 2 : 0);", "constexpr int XP = G::PITCH + (LP ? 2 : 0);")
old = """        for (int task = tid; task < ((BLUR_ABL & 1) ? 0 : np * (NT / 2)); task += 64 * HW) {
            const int rp = task / (NT / 2), t4 = task % (NT / 2);
"""
assert old in k
k = k.replace(old, """        for (int task = tid; task < ((BLUR_ABL & 1) ? 0 : (LP ? 2 * NT : np * (NT / 2))); task += 64 * HW) {
            const int rp = LP ? 2 * (tid >> 6) + (tid & 1) : task / (NT / 2), t4 = LP ? ((tid & 63) >> 1) + 32 * (task / (64 * HW)) : task % (NT / 2);
            if (LP && rp >= np) continue;
""", 1)
# ---- sixth experiment (PL): the first accumulator period peeled -- march row r < N - 1 feeds only the windows that start inside the
# segment (taps j <= r): VPASSF is VPASS with those additions (and the products nobody needs) left out, used for block 0's sub-blocks
k = k.replace("int TR = 0, int LP = 0>", "int TR = 0, int LP = 0, int PL = 0>")
a = k.index("#define VPASS(sbuf, blk_, sub_)")
b = k.index("    // step (blk, sub) exists?")
macro = k[a:b]
f = macro.replace("#define VPASS(sbuf, blk_, sub_)", "#define VPASSF(sbuf, blk_, sub_)")
old1 = "                            if (k == 0) acc[slot_a] = (f32x2){0.f, 0.f} + prod;"
assert old1 in f
f = f.replace(old1, "                            if (k <= kk) { if (k == 0) acc[slot_a] = (f32x2){0.f, 0.f} + prod;")
old2 = "                            asm volatile(\"\" : \"+v\"(acc[slot_a]));"
assert old2 in f
f = f.replace(old2, "                            asm volatile(\"\" : \"+v\"(acc[slot_a])); }")
old3 = "                            if (k != N - 1 - k) { acc[slot_b]"
assert old3 in f
f = f.replace(old3, "                            if (k != N - 1 - k && (N - 1 - k) <= kk) { acc[slot_b]")
old4 = "                            const f32x2 prod = h * t2;"
assert old4 in f
f = f.replace(old4, "                            f32x2 prod = {0.f, 0.f}; if (k <= kk || (N - 1 - k) <= kk) prod = h * t2;")
k = k[:b] + f + k[b:]
old = "                if (sub == 0) { VPASS(prev, blk - 1, S - 1) } else { VPASS(prev, blk, (sub + S - 1) % S) }\n"
assert old in k
k = k.replace(old, """                if (sub == 0) { if (PL && blk == 1) { VPASSF(prev, 0, S - 1) } else { VPASS(prev, blk - 1, S - 1) } }
                else { if (PL && blk == 0) { VPASSF(prev, 0, (sub + S - 1) % S) } else { VPASS(prev, blk, (sub + S - 1) % S) } }
""", 1)
k = k.replace("#undef VPASS\n", "#undef VPASS\n#undef VPASSF\n", 1)
# ---- eighth experiment (X2): the 256 main columns of a sub-block row loaded as one 8-byte load per lane (columns 2t, 2t + 1) instead of
# two 4-byte loads (columns t, 128 + t), staged with one 16-byte LDS write per row pair instead of two 8-byte ones -- where the strip
# and the rows lie inside the plane (the reflecting path keeps the scalar loads)
k = k.replace("int LP = 0, int PL = 0>", "int LP = 0, int PL = 0, int X2 = 0>")
old = "    f32x2 pa[G::NPS], pb[G::NPS], ph[G::NB];\n"
assert old in k
k = k.replace(old, old + "    typedef float f32x2_u4 __attribute__((ext_vector_type(2), aligned(4)));\n    bool wide = false;             // the look-ahead registers hold (row 0: columns 2t, 2t+1 | row 1: the same columns)\n    const bool strip_inside = X2 && (x0 - G::C >= 0) && (x0 - G::C + G::TX <= W);\n", 1)
old = """        if (v0 >= 0 && v0 + 2 * np <= H) {
            unsigned oa = ((unsigned)v0 * (unsigned)W + (unsigned)gx_a) * 4u;"""
assert old in k
k = k.replace(old, """        wide = false;
        if (X2 && strip_inside && v0 >= 0 && v0 + 2 * np <= H) {
            wide = true;
            const char *base = reinterpret_cast<const char *>(in) + ((size_t)v0 * W + (x0 - G::C + 2 * tid)) * 4;
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    pa[rp] = *reinterpret_cast<const f32x2_u4 *>(base);
                    pb[rp] = *reinterpret_cast<const f32x2_u4 *>(base + W4);
                    base += 2u * W4;
                }
        } else if (v0 >= 0 && v0 + 2 * np <= H) {
            unsigned oa = ((unsigned)v0 * (unsigned)W + (unsigned)gx_a) * 4u;""", 1)
old = """            if (rp < np) {
                *reinterpret_cast<f32x2 *>(s + (rp * XP + tid) * 2) = norm2(pa[rp]);
                *reinterpret_cast<f32x2 *>(s + (rp * XP + NT + tid) * 2) = norm2(pb[rp]);
            }"""
assert old in k, "stage"
k = k.replace(old, """            if (rp < np) {
                if (X2 && wide) {
                    const f32x2 a = norm2(pa[rp]), b = norm2(pb[rp]);
                    *reinterpret_cast<f32x4 *>(s + (rp * XP + 2 * tid) * 2) = (f32x4){a.x, b.x, a.y, b.y};
                } else {
                    *reinterpret_cast<f32x2 *>(s + (rp * XP + tid) * 2) = norm2(pa[rp]);
                    *reinterpret_cast<f32x2 *>(s + (rp * XP + NT + tid) * 2) = norm2(pb[rp]);
                }
            }""", 1)
hdr = '''// dev: GENERATED by tools/ubench/gen_blur_team_x.py from sift_pyocl_amd/csrc/k_pyramid.hpp -- the product's blur_team_kernel,
// verbatim, plus a start-up stagger between the workgroups of a CU (stagger_mode / stagger_units).
#pragma once
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
namespace siftk {
'''
open(os.path.join(R, "tools/ubench/blur_team_x.hpp"), "w").write(hdr + k + "\n}  // namespace siftk\n")
