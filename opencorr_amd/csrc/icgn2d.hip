// icgn2d.hip -- ICGN2D1 / ICGN2D2 (inverse-compositional Gauss-Newton) on gfx950.
//
// Replaces ICGN2D1::compute(POI2D*) (src/oc_icgn.cpp:144-341) and
// ICGN2D2::compute(POI2D*) (src/oc_icgn.cpp:685-898) for a whole POI queue
// (:343-351 / :900-908).
//
// Mapping: ONE 64-lane wavefront per POI; sample s = r*W + c of the (2ry+1) x (2rx+1) subset is owned by lane s % 64,
// pass t = s / 64.  The shipped default for large queues (variant 5: G = 2, MODE 4, 80 VGPRs) packs EIGHT waves -- eight
// consecutive POIs of the visiting order -- into a workgroup, three workgroups per CU (6 waves per SIMD):
//   * per-sample state lives in LDS as [t][lane] arrays: one array per wave (the current warped-target subset; the raw
//     reference values pass through it before the first sweep) plus ONE table per workgroup of what depends on (lane, pass)
//     only -- the sample's local coordinates and its byte offset from the subset origin (MODE 3 / 4, `TAB`); the reference
//     value and the two reference gradients are re-read from the images (coalesced, L2 hits) in the passes that need them;
//   * barriers: one after the table fill, two around the cooperative inverse (ONE wave inverts the workgroup's eight 6 x 6
//     Hessians, `COOP`), and the lockstep barrier of the interpolation sweep (`SWEEP_SYNC`: every two pass groups the eight
//     waves re-align, so that neighbouring POIs ask for the same table lines while they are in the CU's L1 -- no data crosses
//     it; its deadlock-freedom rests on three invariants stated where it is defined);
//   * the only memory traffic inside the Gauss-Newton loop besides those re-reads is the 64 B / sample gather from the planar
//     bicubic table (four buffer_load_b128, one per plane) -- the "interpolation sweep".
// Other variants (oc_hip_set_tuning("icgn2d_variant")): 4-wave workgroups without table or barriers for small queues and
// large subsets (variant 2), one-wave workgroups with everything parked in LDS (variants 0, 1, 3, and the IC-LM engines).
// Reductions (mean, norms, Hessian, numerator, ZNSSD) are per-lane partial sums in increasing s followed by the xor
// butterfly of oc_device.h; the CPU oracle uses the same association (OC_ORDER_LANES) and the results are bit-identical,
// whatever the variant.  Wave-uniform state (warp matrix, norms) is held in SGPRs.
// What bounds it: no single pipe.  Inside the sweep the arithmetic hides completely under the gathers
// (tools/ubench/coissue_ubench.hip); over the whole kernel the gathers need 2.07 ms and the VALU instructions 2.25 ms of the 3.23 ms
// a launch takes on config B (0.70): the sweep and the latency-bound phases around it overlap only through the three workgroups a
// CU holds.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "dic2d_device.h"
#include "oc_kernels.h"
// OC_ABLATE2D (experiments only, tools/ablate_icgn2d.sh; the shipped library is built without it): 1 = every POI runs
// exactly three iterations (no convergence test, no abort) so that builds can be compared; 2 = from the second iteration
// on the table gathers are confined to a 16 KB window per plane (L1 hits); 4 = from the second iteration on the gathers
// are not issued at all (stale registers) -- what a register-resident coefficient cache could save at best; 8 = from the
// second iteration on the 43-operation polynomial is replaced by 6 operations on four of the fetched coefficients (the
// gathers stay): how much of the kernel's time is VALU issue; 16 = timeline: results stay valid, and exx / eyy / exy / feature of
// every POI record receive the kilocycles (s_memtime) its wave spent in  reference subset + Hessian sweep | Hessian
// reduction + inverse (incl. the barriers of the cooperative form) | interpolation sweeps, all iterations | everything
// else inside the iterations (norms, numerator pass, solve, warp update); 32 = the Hessian is not inverted (with 1: what the
// per-wave inverse costs -- 14 % of ICGN2D2 on config C.  Sharing it inside the workgroup was built and measured: two waves
// inverting four 12 x 12 matrices each in 16-lane groups, ds_bpermute broadcasts, the totals filed through the idle target
// arrays; 3.69 against 3.65 ms -- the barrier wait for ~300 dependent LDS round trips eats what the other waves save).
#ifndef OC_ABLATE2D
#define OC_ABLATE2D 0
#endif
// Lockstep sweeps (round 3).  The eight waves of a cooperative workgroup solve eight neighbouring POIs of a queue row; their
// subsets overlap in 25 of 33 columns, so in a given pass they fetch largely the SAME table lines -- if they get there at
// about the same time, the second to eighth wave find them in the CU's L1 instead of going to the L2 (which the kernel
// keeps 62 % busy).  Left alone the waves drift apart within an iteration; a barrier every OC_SWEEP_BARRIER groups of passes
// of the interpolation sweep re-aligns them.  Pure scheduling: no data crosses it, results are bit-identical.  Measured on
// config B (tools/ab_icgn2d.sh, profiles/r3h_*): K = 0: 3.43 ms, 1: 3.32, 2: 3.29, 3: 3.30, 4: 3.31, 8: 3.35; a barrier before
// the numerator pass as well: +1 %; with two instead of three workgroups per CU (variant 4) the barriers cost 6 %.
// ICGN2D2 on config C (r = 20, groups of 3 passes, two workgroups per CU): K = 0: 3.61 ms, 1: 3.61, 2: 3.54, 3: 3.51.
// All live waves of a workgroup execute the same barrier sequence (one subset size per launch in the table variants; waves
// that have finished their POI have terminated and are not waited for).
// OC_SWEEP_BARRIER: -1 = the defaults (2 for the 6-DoF, 3 for the 12-DoF kernels), 0 = off, K > 0 = every K groups.
#ifndef OC_SWEEP_BARRIER
#define OC_SWEEP_BARRIER -1
#endif
// OC_SWEEP_PRIO (experiment): 1 = a wave issues the address arithmetic and the gathers of a pass group at raised priority
// (s_setprio) and evaluates the polynomials at normal priority, 2 = the other way round
#ifndef OC_SWEEP_PRIO
#define OC_SWEEP_PRIO 0
#endif
// OC_UNIFORM_IN_VGPR (bit mask): which wave-uniform factors of the per-sample loops are read from vector registers instead of
// SGPRs (in_vgpr, dic2d_device.h): 1 = the warp coefficients and the subset centre in the interpolation sweep (6-DoF),
// 2 = the image size in the sweep's range test and address, 4 = mean / scale factors of the norm and numerator passes
// passes whose loads are issued together in the reference-subset pass, the Hessian sweep and the numerator pass (6-DoF)
#ifndef OC_SETUP_BATCH
#define OC_SETUP_BATCH 6
#endif
#ifndef OC_HESS_BATCH
#define OC_HESS_BATCH 6
#endif
#ifndef OC_NUM_BATCH
#define OC_NUM_BATCH 6
#endif
// gathers in flight and register budget (waves per SIMD) of variant 5 (experiments: tools/ab_build.py)
#ifndef OC_V5_G
#define OC_V5_G 2
#endif
#ifndef OC_V5_OCC
#define OC_V5_OCC 6
#endif
#ifndef OC_UNIFORM_IN_VGPR
#define OC_UNIFORM_IN_VGPR 0
#endif
// The experiment macros above change what the kernels compute or how they are scheduled: they exist for the A/B build of the
// library and the ablation scripts (tools/ablate_icgn2d.sh, tools/ab_icgn2d.sh: -DOC_BUILD_AB=1); the library that ships is
// built with every one of them at its default.
#if !OC_BUILD_AB && (OC_ABLATE2D != 0 || OC_SWEEP_PRIO != 0 || OC_UNIFORM_IN_VGPR != 0 || OC_SWEEP_BARRIER != -1 || OC_SETUP_BATCH != 6 || \
                     OC_HESS_BATCH != 6 || OC_NUM_BATCH != 6 || OC_V5_G != 2 || OC_V5_OCC != 6)
#error "icgn2d.hip: experiment macros are honoured in the A/B build only (-DOC_BUILD_AB=1)"
#endif

namespace ochip {

// ---------------------------------------------------------------------------
// ICGN2D1 (DOF = 6, 3x3 warp) and ICGN2D2 (DOF = 12, 6x6 warp).  One wave per POI, WPB waves (consecutive POIs) per
// workgroup; the waves share nothing but the coordinate table and the cooperative inverse of the table variants.
// Per-sample state lives in LDS as [t][lane] arrays (conflict-free ds_read/write_b32),
// NT = ceil(N/64) at run time:
//   MODE 0 (floats per wave): rs[NT*64] | ts[NT*64] | gx[NT*64] | gy[NT*64]
//   MODE 1:                   rs[NT*64] | ts[NT*64]       (gx, gy re-read from the gradient
//                             images in the numerator pass: half the LDS, twice the waves)
//   MODE 3 = MODE 1 + TAB, MODE 4 = TAB + ts[NT*64] only (the zero-mean reference value is re-formed from the image in
//                             the numerator pass as well).
//   MODE 2 (round 5):         ts[NT*64] only, NO table -- MODE 4's footprint for queues that cannot share a table, i.e.
//                             self-adaptive subsets: the arrays are sized for the LARGEST subset of the batch, so halving
//                             them is what keeps four workgroups on a CU when radii reach 20 (27 passes).
//   TAB: one table per WORKGROUP of what depends on (lane, pass) only -- the sample's local coordinates as a float
//        pair and its byte offset from the subset origin -- filled once by the workgroup's waves: the per-sample
//        walk (wrap test, selects) and the int -> float conversions of every pass become one or two LDS reads.
//        Needs one radius per launch, i.e. not available with self-adaptive subsets.
// G    = samples whose LUT gathers are issued back to back.
// PIPE = (retired) software-pipelined sweep: the gathers of the next group in flight while this group's polynomials are
//        evaluated lost in round 1 and again in round 2 (+3.5 % at G = 2 / 4 waves per SIMD, +4 % at G = 1 / 6 waves:
//        profiles/r03f_icgn2d1_variant_ab_pipelined.json), and a third time in round 3 with the lockstep sweeps (G = 1, next
//        pass's gathers in flight, loop unrolled by two: 3.28 - 3.29 against 3.25 - 3.26 ms, profiles/r3n_icgn2d1_ab_pipelined_lockstep.txt);
//        the parameter stays 0.
// OCC  = minimum waves per SIMD the register allocation must allow.
// Wave-uniform small matrices are kept one COLUMN per lane (lane j < n holds column j):
// the inverse Hessian, and for 2D2 also the 6x6 warp matrix.
// OFFS = the compute(poi_queue, center_offset_queue) overloads (src/oc_icgn.cpp:353-557,
//        910-1136): local coordinates are shifted by a per-POI offset and the target subset is
//        centred at POI + offset.
// P.self_adaptive = DIC::setSelfAdaptive(true): every POI brings its own subset radius
//        (poi->subset_radius, src/oc_icgn.cpp:152-158); L.nt then sizes the LDS arrays for the
//        largest subset of the batch.
// LM   = the inverse-compositional Levenberg-Marquardt classes ICLM2D1 / ICLM2D2 (src/oc_iclm.cpp:150-358,
//        505-731): same subset, Hessian and numerator code; the Hessian is kept (column j in lane j), damped and
//        inverted every iteration, a step is only applied when the ZNSSD went down, out-of-range samples do not
//        abort (they enter the subset as -1.f), and 2D2 weighs the convergence norm in float.
// PHASE (round 5, VERDICT r4 item 3: "give the CU two kinds of agents") = the SPLIT launch shape: 0 = everything in one kernel;
//        1 = the set-up alone (reference mean / norm, steepest-descent sweep, Hessian, inverse): no target array, no gathers ->
//        the workgroup's LDS is the coordinate table only and the register budget admits more waves; it files
//        { mean, norm, H^-1 (DOF x DOF, row-major) } per POI in P.setup; 2 = the Gauss-Newton iterations alone, which read
//        that record instead of recomputing it.  Same operations on the same operands in the same order as PHASE 0: same bits.
// Every variant performs the same floating-point operations in the same order, so all
// of them are bit-identical to the oracle in OC_ORDER_LANES.
// ---------------------------------------------------------------------------
// (the kernel and its launchers live in ochip::sep or ochip::fma -- this file is compiled once per arithmetic mode,
// oc_device.h -- so that the two builds of the same template are different functions)
namespace OC_ARITH {

struct Icgn2dLaunch {
    int stride_f;              // floats between POI records
    int nt;                    // ceil(N / 64)
    int xcd_chunk;             // > 0: workgroup b serves POI group (b % 8) * xcd_chunk + b / 8
    unsigned long long count;  // POIs
};

// what a pass of the Hessian sweep / the numerator pass fetches for one sample (passes_batched): the reference gradients,
// the reference value where it is not parked in LDS, and -- for the walking variants -- the sample's local coordinates
struct GradSample {
    float gx, gy, ref = 0.f;
    f2 xy = {0.f, 0.f};
};

// floats of a POI's set-up record (PHASE 1 -> PHASE 2): mean, norm, H^-1 row-major
constexpr int icgn2d_setup_floats(int dof) { return 2 + dof * dof; }

template <int DOF, int G, int MODE, int PIPE, int WPB, int OCC, int OFFS, int LM = 0, int PHASE = 0>
__global__ __launch_bounds__(64 * WPB, OCC) void icgn2d_kernel(Icgn2dParams P, float* __restrict__ pois,
                                                               Icgn2dLaunch L) {
    static_assert(PHASE == 0 || (LM == 0 && MODE == 4), "the split launch shape exists for the table variants of ICGN2D1 / ICGN2D2");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NH = DOF * (DOF + 1) / 2;
    constexpr bool TAB = MODE >= 3;
    constexpr bool KEEP_RS = MODE != 4 && MODE != 2;  // the zero-mean reference subset is parked in LDS
    constexpr int ARRAYS = MODE == 0 ? 4 : ((MODE == 4 || MODE == 2) ? 1 : 2);
    // ICGN2D1 in 8-wave workgroups: the eight 6 x 6 Hessians are inverted by ONE wave (coop_inverse6_x8); two more
    // barriers, which every wave passes exactly once -- also the ones that abandon their POI early (leave())
    constexpr bool COOP = MODE == 4 && DOF == 6 && LM == 0 && WPB == 8 && PHASE != 2;
    constexpr bool kSetupOnly = PHASE == 1, kIterOnly = PHASE == 2;
    constexpr int kSetupFloats = icgn2d_setup_floats(DOF);
    // lockstep sweeps (see OC_SWEEP_BARRIER above): the 8-wave table variants -- one subset size per launch, so every live
    // wave of a workgroup runs the same number of pass groups
    // (the 6-DoF kernel with only two workgroups per CU, variant 4, loses 6 % with them: it keeps them off by default)
    constexpr int SWEEP_SYNC = (MODE == 4 && LM == 0 && PHASE != 1 && (WPB == 8 || WPB == 4))
                                   ? (OC_SWEEP_BARRIER < 0 ? (DOF == 6 ? ((OCC >= 6 || WPB == 4) ? 2 : 0) : 3) : OC_SWEEP_BARRIER)
                                   : 0;
    // What keeps the sweep barriers deadlock-free although the waves of a workgroup run different iteration counts and may
    // leave at any point (guard, out-of-range sample, convergence) -- three invariants of THIS kernel shape:
    //   (a) one POI per wave and no POI loop: a wave that is done with its POI TERMINATES, and the hardware drops terminated
    //       waves from the workgroup's barrier count;
    //   (b) every wave passes the data-carrying barriers (the table fill, COOP's two) strictly BEFORE its first sweep
    //       barrier, the early leavers through leave() -- so a sweep barrier can only ever pair with sweep barriers;
    //   (c) nothing after the iteration loop waits on a barrier.
    // A persistent POI loop, a barrier behind the loop, or a table whose pass count differs between the waves of a workgroup
    // would break one of them (hang, or sweep barriers pairing with COOP's).  kOnePoiPerWave / kBarrierFreeEpilogue state
    // (a) and (c) where such a change would have to flip them; tests/test_gpu_parity_2d.py::
    // test_icgn2d_lockstep_barriers_with_mixed_wave_lifetimes runs workgroups that mix every kind of early leaver with 1-
    // and stop-iteration POIs.
    constexpr bool kOnePoiPerWave = true, kBarrierFreeEpilogue = true;
    static_assert(SWEEP_SYNC == 0 || (kOnePoiPerWave && kBarrierFreeEpilogue && TAB && LM == 0),
                  "lockstep sweep barriers need one POI per wave, one pass count per workgroup and a barrier-free epilogue");
    // passes whose global loads are issued together in the load-then-use loops outside the interpolation sweep
    // (reference subset, Hessian sweep, numerator pass): one dependent round trip per batch instead of one per pass
    constexpr int kSetupBatch = OC_SETUP_BATCH;
    // (6 DoF; the 78 running sums of the 12-DoF Hessian leave no room: passes_prefetched.  Round 3 split that sweep in two -- 45 + 33
    // sums, each reduced right away, 2 - 6 passes of loads in flight -- and measured 3.49 - 3.50 ms against 3.49 on config C
    // (profiles/r3q_icgn2d2_ab_two_hessian_sweeps.txt): the sweep is not waiting for its loads; not kept)
    constexpr int kHessBatch = OC_HESS_BATCH;
    constexpr int kNumBatch = DOF == 6 ? OC_NUM_BATCH : 4;
    __shared__ float coop_area[COOP ? WPB * 64 : 1];
    const int NTA = L.nt;  // passes the LDS arrays are sized for (>= the passes of any POI)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    [[maybe_unused]] unsigned long long tl_mark = __builtin_readcyclecounter();
    [[maybe_unused]] unsigned long long tl[4] = {0, 0, 0, 0};
    auto lap = [&](int slot) {
        if constexpr ((OC_ABLATE2D & 16) != 0) {
            __builtin_amdgcn_sched_barrier(0);  // keep the surrounding arithmetic on its side of the stamp
            const unsigned long long now = __builtin_readcyclecounter();
            tl[slot] += now - tl_mark;
            tl_mark = now;
        }
    };
    auto leave = [&]() {
        if constexpr (COOP) {
            if (lane < 24) coop_area[wave * 64 + lane] = 0.f;
            __syncthreads();
            if (wave == 0) coop_inverse6_x8(coop_area, lane);
            __syncthreads();
        }
    };
    // TAB: [NTA*64] float pairs (x_local, y_local), then [NTA*64] byte offsets; the waves' own arrays follow
    f2* __restrict__ tab_xy = reinterpret_cast<f2*>(lds);
    unsigned* __restrict__ tab_off = reinterpret_cast<unsigned*>(lds + 2 * NTA * kWave);
    float* const wave_lds = lds + (TAB ? 3 * NTA * kWave : 0);
    if constexpr (TAB) {
        const int Wt = 2 * P.rx + 1;
        const unsigned w4t = (unsigned)P.width * 4u;
        for (int s = threadIdx.x; s < NTA * kWave; s += kWave * WPB) {
            const int r = s / Wt, c = s - r * Wt;
            tab_xy[s] = mk2((float)(c - P.rx), (float)(r - P.ry));
            tab_off[s] = (unsigned)r * w4t + ((unsigned)c << 2);
        }
        __syncthreads();  // waves that leave early below are past this barrier (and pass COOP's two in leave())
    }
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD a
    // contiguous range of the queue so POIs that share LUT lines meet in the same L2.
    unsigned long long grp = blockIdx.x;
    if (L.xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * L.xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long slot = grp * WPB + wave;
    if (slot >= L.count) {
        leave();
        return;
    }
    // the k-th wave of the launch solves POI perm[k] (a locality schedule) or simply POI k
    const unsigned long long idx = P.perm ? (unsigned long long)__builtin_amdgcn_readfirstlane((int)P.perm[slot]) : slot;
    // (no __restrict__: in MODE 4 the two names denote one array -- the reference values pass through it before the
    // first sweep overwrites them)
    float* l_rs = wave_lds + (size_t)wave * ARRAYS * NTA * kWave + lane;
    float* l_ts = KEEP_RS ? l_rs + NTA * kWave : l_rs;
    float* __restrict__ l_gx = l_ts + NTA * kWave;  // MODE 0 only
    float* __restrict__ l_gy = l_gx + NTA * kWave;

    float* poi = pois + idx * (unsigned long long)L.stride_f;
    const float rec = lane < poi2d::FLOATS ? poi[lane] : 0.f;
    const float px = wave_bcast(rec, poi2d::X), py = wave_bcast(rec, poi2d::Y);
    const float u_in = wave_bcast(rec, poi2d::U), ux_in = wave_bcast(rec, poi2d::UX), uy_in = wave_bcast(rec, poi2d::UY);
    const float v_in = wave_bcast(rec, poi2d::V), vx_in = wave_bcast(rec, poi2d::VX), vy_in = wave_bcast(rec, poi2d::VY);
    const float zncc_in = wave_bcast(rec, poi2d::ZNCC);
    const int height = P.height, width = P.width;
    int rx = P.rx, ry = P.ry;
    if (P.self_adaptive) {
        // poi->subset_radius is a Point2D of floats; the reference passes it to int parameters
        rx = (int)wave_bcast(rec, poi2d::SRX);
        ry = (int)wave_bcast(rec, poi2d::SRY);
        // a radius the batch was not sized for (negative, or changed behind the engine's back): the
        // reference would reallocate; here the POI is rejected like any other unusable POI
        if (rx < 0 || ry < 0 || (2 * rx + 1) * (2 * ry + 1) > NTA * kWave) {
            if (lane == 0) poi[poi2d::ZNCC] = zncc_in >= 0 ? -3.f : zncc_in;
            leave();
            return;
        }
    }
    float offx = 0.f, offy = 0.f;
    if constexpr (OFFS) {
        offx = uni(P.offsets[2 * idx]);
        offy = uni(P.offsets[2 * idx + 1]);
    }

    // guard, src/oc_icgn.cpp:160-167 (2D2: 705-712)
    if (py - ry < 0 || px - rx < 0 || py + ry > height - 1 || px + rx > width - 1 || fabsf(u_in) >= width ||
        fabsf(v_in) >= height || zncc_in < 0 || isnan(u_in) || isnan(v_in)) {
        if (lane == 0) poi[poi2d::ZNCC] = zncc_in >= 0 ? -3.f : zncc_in;
        leave();
        return;
    }
    const int W = 2 * rx + 1, N = W * (2 * ry + 1);
    const float fN = (float)N;
    const int NT = (N + kWave - 1) / kWave;
    const int NF = N / kWave;  // passes in which every lane owns a sample; pass NF (if any) is partial
    const int q64 = kWave / W, r64 = kWave - q64 * W;
    const int r0 = lane / W;
    const int c0 = lane - r0 * W;
    // subset origin as a wave-uniform byte offset (images are <= 2^28 bytes)
    const unsigned goff = (unsigned)__builtin_amdgcn_readfirstlane((((int)py - ry) * width + ((int)px - rx)) * 4);
    const __amdgpu_buffer_rsrc_t r_gx = make_rsrc(P.gx), r_gy = make_rsrc(P.gy), r_ref = make_rsrc(P.ref);
    const LutPlanes4 r_lut(P.lut, height, width);
    const unsigned w4 = (unsigned)width * 4u;
    // byte offset of sample (r, c) from the subset origin
    auto soff = [&](const SampleWalk& w) { return __umul24((unsigned)w.r, w4) + ((unsigned)w.c << 2); };

    // per-pass sample coordinates: from the workgroup's table, or walked
    auto tab_at = [&](int t) { return tab_xy[t * kWave + lane]; };
    auto off_at = [&](int t) { return tab_off[t * kWave + lane]; };

    // ---- reference subset, zero-mean + norm (src/oc_icgn.cpp:174-176, src/oc_subset.cpp:39-53)
    float ref_norm, ref_mean;
    const int x0r = (int)(px - rx), y0r = (int)(py - ry);
    const unsigned roff = (unsigned)__builtin_amdgcn_readfirstlane((y0r * width + x0r) * 4);
    float* const setup_rec = (kSetupOnly || kIterOnly) ? P.setup + idx * (unsigned long long)kSetupFloats : nullptr;
    if constexpr (kIterOnly) {
        ref_mean = uni(setup_rec[0]);
        ref_norm = uni(setup_rec[1]);
    } else if constexpr (kSetupOnly) {
        // no per-wave LDS array in the set-up kernel: the raw values are read a second time (coalesced, L2) for the norm
        float acc = 0.f;
        passes_batched<kSetupBatch>(
            NF, NT, (NF * kWave + lane) < N,
            [&](int t, bool valid) { return valid ? buf_f32(r_ref, off_at(t), roff) : 0.f; },
            [&](int, bool valid, float v) { acc = valid ? acc + v : acc; });
        const float mean = wave_allreduce_sum(acc) / fN;
        ref_mean = uni(mean);
        acc = 0.f;
        passes_batched<kSetupBatch>(
            NF, NT, (NF * kWave + lane) < N,
            [&](int t, bool valid) { return valid ? buf_f32(r_ref, off_at(t), roff) : 0.f; },
            [&](int, bool valid, float v) {
                const float d = v - mean;
                acc = valid ? mad(d, d, acc) : acc;
            });
        ref_norm = uni(sqrtf(wave_allreduce_sum(acc)));
    } else {
        float acc = 0.f;
        SampleWalk w(lane, r0, c0, W, q64, r64);
        // (MODE 4 parks the raw values in the target array, which is idle until the first sweep)
        passes_batched<kSetupBatch>(
            NF, NT, (NF * kWave + lane) < N,
            [&](int t, bool valid) {
                const unsigned off = TAB ? off_at(t) : soff(w);
                w.next();
                return valid ? buf_f32(r_ref, off, roff) : 0.f;
            },
            [&](int t, bool valid, float v) {
                acc = valid ? acc + v : acc;
                l_rs[t * kWave] = v;
            });
        const float mean = wave_allreduce_sum(acc) / fN;
        ref_mean = uni(mean);
        acc = 0.f;
#pragma unroll 3
        for (int t = 0; t < NF; t++) {
            const float d = l_rs[t * kWave] - mean;
            if constexpr (KEEP_RS) l_rs[t * kWave] = d;
            acc = mad(d, d, acc);
        }
        if (NF < NT) {
            const float d = l_rs[NF * kWave] - mean;
            if constexpr (KEEP_RS) l_rs[NF * kWave] = d;
            acc = (NF * kWave + lane) < N ? mad(d, d, acc) : acc;
        }
        ref_norm = uni(sqrtf(wave_allreduce_sum(acc)));
    }

    // ---- steepest-descent image + Hessian (src/oc_icgn.cpp:179-207; 2D2: 716-756), inverse (:210 / :759)
    float hinv_col[DOF];  // lane j < DOF: column j of H^-1
    float hcol[LM ? DOF : 1];  // LM: column j of H itself
    if constexpr (!kIterOnly) {
        float h[NH];
#pragma unroll
        for (int i = 0; i < NH; i++) h[i] = 0.f;
        SampleWalk w(lane, r0, c0, W, q64, r64);
        if constexpr (DOF == 6) {
            // The 21 running sums H(i,j) += sd[i]*sd[j] as packed-fp32 pairs (v_pk_mul_f32 / v_pk_add_f32: two
            // IEEE operations per issue slot, each rounded on its own -- every H(i,j) still receives the same
            // products in the same order).  With A = (sd1, sd2) = g_x*(x, y), B = (sd4, sd5) = g_y*(x, y):
            f2 hAA = mk2(0.f, 0.f), hBB = hAA, hAB = hAA, hAs = hAA, hxA = hAA, hyA = hAA, hxB = hAA, hyB = hAA;
            float h00 = 0.f, h33 = 0.f, h30 = 0.f, h21 = 0.f, h54 = 0.f;
            auto fetch = [&](int t, bool valid) {
                GradSample v;
                const unsigned off = TAB ? off_at(t) : soff(w);
                v.gx = valid ? buf_f32(r_gx, off, goff) : 0.f;
                v.gy = valid ? buf_f32(r_gy, off, goff) : 0.f;
                if constexpr (!TAB) v.xy = mk2((float)(w.c - rx) - offx, (float)(w.r - ry) - offy);
                w.next();
                return v;
            };
            auto sample = [&](int t, bool valid, const GradSample& v) {
                const float g_x = v.gx, g_y = v.gy;
                if constexpr (MODE == 0) {
                    l_gx[t * kWave] = g_x;
                    l_gy[t * kWave] = g_y;
                }
                f2 xy;
                if constexpr (TAB) xy = tab_at(t) - mk2(offx, offy);
                else xy = v.xy;
                const f2 A = g_x * xy, B = g_y * xy;
                // (mad: product and add rounded separately, or -- OC_FMA -- one fused multiply-add per running sum)
                const f2 nAA = mad(A, A, hAA), nBB = mad(B, B, hBB), nAB = mad(A, B, hAB), nAs = mad(A, B.yx, hAs);
                const f2 nxA = mad(g_x, A, hxA), nyA = mad(g_y, A, hyA), nxB = mad(g_x, B, hxB), nyB = mad(g_y, B, hyB);
                const float n00 = mad(g_x, g_x, h00), n33 = mad(g_y, g_y, h33), n30 = mad(g_y, g_x, h30);
                const float n21 = mad(A.y, A.x, h21), n54 = mad(B.y, B.x, h54);
                if (valid) {
                    hAA = nAA; hBB = nBB; hAB = nAB; hAs = nAs; hxA = nxA; hyA = nyA; hxB = nxB; hyB = nyB;
                    h00 = n00; h33 = n33; h30 = n30; h21 = n21; h54 = n54;
                }
            };
            passes_batched<kHessBatch>(NF, NT, (NF * kWave + lane) < N, fetch, sample);
            // back to the row-major lower triangle h[i*(i+1)/2 + j]
            h[0] = h00;                                                           // (0,0)
            h[1] = hxA.x; h[2] = hAA.x;                                           // (1,0) (1,1)
            h[3] = hxA.y; h[4] = h21; h[5] = hAA.y;                               // (2,0) (2,1) (2,2)
            h[6] = h30; h[7] = hyA.x; h[8] = hyA.y; h[9] = h33;                   // (3,0) (3,1) (3,2) (3,3)
            h[10] = hxB.x; h[11] = hAB.x; h[12] = hAs.y; h[13] = hyB.x; h[14] = hBB.x;                // (4,0..4)
            h[15] = hxB.y; h[16] = hAs.x; h[17] = hAB.y; h[18] = hyB.y; h[19] = h54; h[20] = hBB.y;   // (5,0..5)
        } else {
            // 12 DoF: the 78 running sums as packed pairs over adjacent columns.  sd = g * (1, x | y, xx | xy, yy)
            // for g = g_x, g_y (sd_row<12>): six pairs; row r takes the column pairs below the diagonal and, when r
            // is even, the single diagonal term.  Same products, same order per sum.
            f2 hp[12][6];
            float hd[12];
#pragma unroll
            for (int r = 0; r < 12; r++) {
                hd[r] = 0.f;
#pragma unroll
                for (int q = 0; q < 6; q++) hp[r][q] = mk2(0.f, 0.f);
            }
            auto fetch = [&](int t, bool valid) {
                GradSample v;
                const unsigned off = TAB ? off_at(t) : soff(w);
                v.gx = valid ? buf_f32(r_gx, off, goff) : 0.f;
                v.gy = valid ? buf_f32(r_gy, off, goff) : 0.f;
                if constexpr (!TAB) v.xy = mk2((float)(w.c - rx) - offx, (float)(w.r - ry) - offy);
                w.next();
                return v;
            };
            auto sample = [&](int t, bool valid, const GradSample& v) {
                const float g_x = v.gx, g_y = v.gy;
                if constexpr (MODE == 0) {
                    l_gx[t * kWave] = g_x;
                    l_gy[t * kWave] = g_y;
                }
                f2 lxy;
                if constexpr (TAB) lxy = tab_at(t) - mk2(offx, offy);
                else lxy = v.xy;
                const float fxl = lxy.x, fyl = lxy.y;
                const float xx = (fxl * fxl) * 0.5f, xy = fxl * fyl, yy = (fyl * fyl) * 0.5f;
                const f2 m01 = mk2(1.f, fxl), m23 = mk2(fyl, xx), m45 = mk2(xy, yy);  // g * 1.f is exact
                const f2 sdp[6] = {g_x * m01, g_x * m23, g_x * m45, g_y * m01, g_y * m23, g_y * m45};
#pragma unroll
                for (int r = 0; r < 12; r++) {
                    const float sr = (r & 1) ? sdp[r / 2].y : sdp[r / 2].x;
#pragma unroll
                    for (int q = 0; q < (r + 1) / 2; q++) {
                        const f2 nv = mad(sr, sdp[q], hp[r][q]);
                        hp[r][q] = valid ? nv : hp[r][q];
                    }
                    if ((r & 1) == 0) hd[r] = valid ? mad(sr, sr, hd[r]) : hd[r];
                }
            };
            passes_prefetched(NF, NT, (NF * kWave + lane) < N, fetch, sample);
            int k = 0;
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int c = 0; c <= r; c++, k++)
                    h[k % NH] = (c == r && (r & 1) == 0) ? hd[r] : ((c & 1) ? hp[r][c / 2].y : hp[r][c / 2].x);
        }
        lap(0);
        if constexpr (COOP) {
            // the totals go to LDS, wave 0 inverts the workgroup's eight Hessians at once
            wave_reduce_sum_multi_to_lds<NH>(h, lane, coop_area + wave * 64);
            __syncthreads();
            if (wave == 0) coop_inverse6_x8(coop_area, lane);
            __syncthreads();
        } else {
            // lane j < DOF assembles column j of the symmetric Hessian
            float col[DOF];
#pragma unroll
            for (int i = 0; i < DOF; i++) col[i] = 0.f;
            wave_allreduce_sum_multi<NH>(h, lane);  // all NH sums in one transposing butterfly (oc_device.h)
            int k = 0;
#pragma unroll
            for (int i = 0; i < DOF; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    const float v = h[k++];
                    if (lane == j) col[i] = v;  // H(i,j)
                    if (lane == i) col[j] = v;  // H(j,i)
                }
            if constexpr (LM) {
#pragma unroll
                for (int i = 0; i < DOF; i++) hcol[i] = col[i];
            } else {
                if constexpr ((OC_ABLATE2D & 32) != 0) {
#pragma unroll
                    for (int i = 0; i < DOF; i++) hinv_col[i] = col[i] * 1e-9f;  // ablation: no inverse at all
                } else {
                    lu_inverse_lanes<DOF>(col, hinv_col, lane);
                }
            }
        }
    }
    // IC-GN: H^-1 stays fixed, so it is transposed once -- lane i (< DOF) gets ROW i of H^-1 -- and every iteration's
    // dp[i] = sum_j H^-1(i,j) * num[j] is formed inside lane i (ascending j, as the reference's loop) and handed round
    // with DOF broadcasts, instead of DOF x DOF v_readlane per iteration (6.2 cycles each on gfx950).
    float hinv_row[LM ? 1 : DOF];
    if constexpr (kSetupOnly) {
        // file the POI's set-up record: mean, norm, H^-1 row-major -- the values PHASE 2 loads into hinv_row
        if (lane == 0) {
            setup_rec[0] = ref_mean;
            setup_rec[1] = ref_norm;
        }
        if constexpr (COOP) {
            if (lane < DOF * DOF) setup_rec[2 + lane] = coop_area[wave * 64 + 24 + lane];
        } else {
            if (lane < DOF) {
#pragma unroll
                for (int i = 0; i < DOF; i++) setup_rec[2 + i * DOF + lane] = hinv_col[i];  // H^-1(i, lane)
            }
        }
        return;
    } else if constexpr (kIterOnly) {
        const float* __restrict__ row = setup_rec + 2 + min(lane, DOF - 1) * DOF;
#pragma unroll
        for (int j = 0; j < DOF; j++) hinv_row[j] = lane < DOF ? row[j] : 0.f;
    } else if constexpr (COOP) {
        const float* __restrict__ row = coop_area + wave * 64 + 24 + min(lane, DOF - 1) * DOF;
#pragma unroll
        for (int j = 0; j < DOF; j++) hinv_row[j] = lane < DOF ? row[j] : 0.f;
    } else if constexpr (!LM) {
#pragma unroll
        for (int j = 0; j < DOF; j++) hinv_row[j] = 0.f;
#pragma unroll
        for (int i = 0; i < DOF; i++)
#pragma unroll
            for (int j = 0; j < DOF; j++) {
                const float v = wave_bcast(hinv_col[i], j);  // H^-1(i, j)
                hinv_row[j] = lane == i ? v : hinv_row[j];
            }
    }

    lap(1);
    // ---- IC-GN loop (src/oc_icgn.cpp:216-307; 2D2: 762-858)
    // 2D1: 3x3 warp matrix, wave-uniform in SGPRs.  2D2: 6x6 warp matrix, column j in lane j;
    // rows 3 and 4 (the ones Deformation2D2::warp needs) are broadcast once per iteration.
    float Wm[9];      // 2D1
    float Wcol[6];    // 2D2
    float row3[6], row4[6];
    if constexpr (DOF == 6) {
        set_warp_2d1(Wm, u_in, ux_in, uy_in, v_in, vx_in, vy_in);
    } else {
        // first-order initial guess promoted to second order (src/oc_icgn.cpp:765-770,
        // src/oc_deformation.cpp:249-266)
        const float q[12] = {u_in, ux_in, uy_in, 0.f, 0.f, 0.f, v_in, vx_in, vy_in, 0.f, 0.f, 0.f};
        float w36[36];
        set_warp_2d2(w36, q);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float c = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) c = lane == j ? w36[i * 6 + j] : c;
            Wcol[i] = c;
        }
    }
    const float tcx = px + offx, tcy = py + offy;  // centre of the target subset
    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
    float cur[12];
#pragma unroll
    for (int i = 0; i < 12; i++) cur[i] = 0.f;
    float znssd0 = 4.f, lambda = 0.f;  // IC-LM state (src/oc_iclm.cpp:226)
    if constexpr (LM) {
        // p_current.setDeformation(p_initial) copies the parameters themselves (src/oc_iclm.cpp:223 / :598)
        cur[0] = u_in; cur[1] = ux_in; cur[2] = uy_in;
        cur[6] = v_in; cur[7] = vx_in; cur[8] = vy_in;
    }
#if OC_ABLATE2D & 4
    LutFetch f_stale[G] = {};  // coefficients survive from the first iteration
#endif
#pragma nounroll
    do {
        iter++;
        if constexpr (DOF == 12) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                row3[k] = wave_bcast(Wcol[3], k);
                row4[k] = wave_bcast(Wcol[4], k);
            }
        }
        lap(3);
        // warped target subset (src/oc_icgn.cpp:230-242; 2D2: 784-796)
        bool negative = false;
        float acc = 0.f;
        // Round 6: the range rule of BicubicBspline::compute (src/oc_cubic_bspline.cpp:137-142) hoisted out of the sweep for the
        // AFFINE warp.  A sample's target coordinate is tc + ((W0 x + W1 y) + W2) with every float operation monotone in x and
        // in y, so over the subset's index rectangle it takes its extremes at the four corner samples: all four inside
        // [1, size - 2) <=> every sample inside.  A corner outside is a sample outside, i.e. a -1.f in the target subset, and the
        // reference abandons the POI (:251-255, only zncc = -3 is written) -- which is decided here, before the sweep, from four
        // lanes' worth of arithmetic instead of two subtractions, two compares and an or per sample (5 of the sweep's ~67 VALU
        // instructions).  The quadratic warp of ICGN2D2 is not monotone and IC-LM keeps out-of-range samples as values: both keep
        // the per-sample test.
        constexpr bool kCornerTest = DOF == 6 && LM == 0 && !(OC_ABLATE2D & 6);
        if constexpr (kCornerTest) {
            const float cxl = (float)((lane & 1) ? rx : -rx) - offx, cyl = (float)((lane & 2) ? ry : -ry) - offy;
            const float cax = tcx + (mad(Wm[1], cyl, Wm[0] * cxl) + Wm[2]), cay = tcy + (mad(Wm[4], cyl, Wm[3] * cxl) + Wm[5]);
            const int cxi = floor_to_int(cax), cyi = floor_to_int(cay);
            const bool cout = (unsigned)(cxi - 1) > (unsigned)(width - 4) || (unsigned)(cyi - 1) > (unsigned)(height - 4);
            if (!(OC_ABLATE2D & 1) && wave_any(cout)) {
                if (lane == 0) poi[poi2d::ZNCC] = -3.f;
                return;
            }
        }
        {
            SampleWalk w(lane, r0, c0, W, q64, r64);
            constexpr bool kWarpV = (OC_UNIFORM_IN_VGPR & 1) != 0 && DOF == 6, kSizeV = (OC_UNIFORM_IN_VGPR & 2) != 0;
            float Wv[6];
#pragma unroll
            for (int i = 0; i < 6; i++) Wv[i] = kWarpV ? in_vgpr(Wm[i]) : Wm[i];
            const float tcxv = kWarpV ? in_vgpr(tcx) : tcx, tcyv = kWarpV ? in_vgpr(tcy) : tcy;
            const int heightv = kSizeV ? in_vgpr(height) : height, widthv = kSizeV ? in_vgpr(width) : width;
            // warp the next G samples of this lane and issue their LUT gathers.  CHECKED = std::false_type: every
            // lane owns a sample in all G passes (the common case: no validity selects at all).
            // !LM: a sample outside the interpolatable range makes the reference abandon the POI (:251-255: -1.f is
            // the only way out of range shows, and any negative value aborts), so "outside" is folded into `negative`
            // right here and the sample's value is never looked at; LM keeps the -1.f sentinel as a value (MARK).
            auto issue = [&](LutFetch(&f)[G], bool(&valid)[G], int t0, auto checked) {
                constexpr bool CHECKED = decltype(checked)::value;
#pragma unroll
                for (int g = 0; g < G; g++, w.next()) {
                    valid[g] = CHECKED ? w.s < N : true;
                    // local_coor = (c - rx) - center_offset (src/oc_icgn.cpp:447-450); "- 0.f" is exact
                    float xl, yl;
                    if constexpr (TAB) {
                        // a pass past the table's end (only in the CHECKED tail) reads the last pass: its value is unused
                        const f2 lxy = tab_at(CHECKED ? min(t0 + g, NT - 1) : t0 + g);
                        xl = lxy.x - offx;
                        yl = lxy.y - offy;
                    } else {
                        xl = (float)(w.c - rx) - offx;
                        yl = (float)(w.r - ry) - offy;
                    }
                    float wx, wy;
                    if constexpr (DOF == 6) {
                        // Deformation2D1::warp, src/oc_deformation.cpp:94-105: (W0 x + W1 y) + W2 * 1 (the product
                        // with 1.f is exact and dropped); OC_FMA: the second product joins the first sum
                        wx = mad(Wv[1], yl, Wv[0] * xl) + Wv[2];
                        wy = mad(Wv[4], yl, Wv[3] * xl) + Wv[5];
                    } else {
                        // Deformation2D2::warp, src/oc_deformation.cpp:268-282: rows 3, 4 of W * [x^2 xy y^2 x y 1]
                        const float pv[6] = {xl * xl, xl * yl, yl * yl, xl, yl, 1.f};
                        wx = row3[0] * pv[0];
                        wy = row4[0] * pv[0];
#pragma unroll
                        for (int k = 1; k < 6; k++) {
                            wx = mad(row3[k], pv[k], wx);
                            wy = mad(row4[k], pv[k], wy);
                        }
                    }
                    // tar_subset->center = POI + center_offset, then + warped_coor (src/oc_icgn.cpp:425-426,452);
                    // a lane past the end of the subset fetches a harmless in-range point
                    float ax = tcxv + wx, ay = tcyv + wy;
                    if constexpr (CHECKED) {
                        ax = valid[g] ? ax : 1.f;
                        ay = valid[g] ? ay : 1.f;
                    }
                    bool out = false;
                    if constexpr (kCornerTest) {
                        // every sample is inside the interpolatable range (the corner test above): no range test, no sentinel
                        const int xi = floor_to_int(ax), yi = floor_to_int(ay);
                        f[g].dx = __builtin_amdgcn_fractf(ax);
                        f[g].dy = __builtin_amdgcn_fractf(ay);
                        r_lut.load(f[g], (__umul24((unsigned)yi, (unsigned)widthv) + (unsigned)xi) << 4);
                    } else if constexpr ((OC_ABLATE2D & 6) != 0) {
                        unsigned off = lut_locate<LM != 0>(f[g], heightv, widthv, ax, ay, out);
                        if (OC_ABLATE2D & 2) off = iter > 1 ? (off & 0x3ff0u) : off;
                        if (!(OC_ABLATE2D & 4) || iter == 1) r_lut.load(f[g], off);
                    } else {
                        lut_fetch<LM != 0>(f[g], r_lut, heightv, widthv, ax, ay, out);
                    }
                    if constexpr (!LM && !kCornerTest) negative = negative || out;
                }
            };
            auto consume = [&](const LutFetch(&f)[G], const bool(&valid)[G], int t0, auto checked) {
                constexpr bool CHECKED = decltype(checked)::value;
#pragma unroll
                for (int g = 0; g < G; g++) {
#if OC_ABLATE2D & 8
                    const float v = iter > 1 ? f[g].c0.x + f[g].c1.y * f[g].dx + f[g].c2.z * f[g].dy + f[g].c3.w : lut_value(f[g]);
#else
                    const float v = LM ? lut_eval(f[g]) : lut_value(f[g]);
#endif
                    if constexpr (CHECKED) {
                        negative = negative || (valid[g] && v < 0.f);
                        acc = valid[g] ? acc + v : acc;
                        if (t0 + g < NT) l_ts[(t0 + g) * kWave] = v;
                    } else {
                        negative = negative || v < 0.f;
                        acc = acc + v;
                        l_ts[(t0 + g) * kWave] = v;
                    }
                }
            };
            // groups of G passes in which every lane owns a sample need no validity logic
            const int full_groups = NF / G;
            int t0 = 0;
#pragma nounroll
            for (int q = 0; q < full_groups; q++, t0 += G) {
                if constexpr (SWEEP_SYNC > 0) {
                    if (q % SWEEP_SYNC == 0) __builtin_amdgcn_s_barrier();
                }
#if OC_ABLATE2D & 4
                LutFetch(&f)[G] = f_stale;
#else
                LutFetch f[G];
#endif
                bool valid[G];
#if OC_SWEEP_PRIO == 1
                __builtin_amdgcn_s_setprio(2);
#elif OC_SWEEP_PRIO == 2
                __builtin_amdgcn_s_setprio(0);
#endif
                issue(f, valid, t0, std::false_type{});
#if OC_SWEEP_PRIO == 1
                __builtin_amdgcn_s_setprio(0);
#elif OC_SWEEP_PRIO == 2
                __builtin_amdgcn_s_setprio(2);
#endif
                consume(f, valid, t0, std::false_type{});
            }
#if OC_SWEEP_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#pragma nounroll
            for (; t0 < NT; t0 += G) {
#if OC_ABLATE2D & 4
                LutFetch(&f)[G] = f_stale;
#else
                LutFetch f[G];
#endif
                bool valid[G];
                issue(f, valid, t0, std::true_type{});
                consume(f, valid, t0, std::true_type{});
            }
        }
        lap(2);
        // src/oc_icgn.cpp:251-255 (the IC-LM classes have no such check)
        if constexpr (!LM) {
            if (!(OC_ABLATE2D & 1) && wave_any(negative)) {
                if (lane == 0) poi[poi2d::ZNCC] = -3.f;
                return;
            }
        }
        // zeroMeanNorm of the target subset (src/oc_icgn.cpp:257)
        float tmean = wave_allreduce_sum(acc) / fN;
        if constexpr ((OC_UNIFORM_IN_VGPR & 4) != 0) tmean = in_vgpr(tmean);
        acc = 0.f;
#pragma unroll 6
        for (int t = 0; t < NF; t++) {
            const float d = l_ts[t * kWave] - tmean;
            acc = mad(d, d, acc);
        }
        if (NF < NT) {
            const float d = l_ts[NF * kWave] - tmean;
            acc = (NF * kWave + lane) < N ? mad(d, d, acc) : acc;
        }
        const float tar_norm = uni(sqrtf(wave_allreduce_sum(acc)));
        // error image, ZNSSD, numerator (src/oc_icgn.cpp:260-276)
        float factor = ref_norm / tar_norm;
        float ref_mean_v = ref_mean;
        if constexpr ((OC_UNIFORM_IN_VGPR & 4) != 0) {
            factor = in_vgpr(factor);
            ref_mean_v = in_vgpr(ref_mean_v);
        }
        float num[DOF];
#pragma unroll
        for (int i = 0; i < DOF; i++) num[i] = 0.f;
        float ssd = 0.f;
        {
            SampleWalk w(lane, r0, c0, W, q64, r64);
            FloatWalk fw(r0, c0, rx, ry, W, q64, r64, w4);  // DOF 6 without centre offsets
            constexpr bool kFloatWalk = DOF == 6 && !OFFS && !TAB;
            f2 nA = mk2(0.f, 0.f), nB = nA;  // DOF 6: (num1, num2) and (num4, num5) as packed pairs
            f2 np12[6];                       // DOF 12: (num0, num1) .. (num10, num11)
#pragma unroll
            for (int q = 0; q < 6; q++) np12[q] = mk2(0.f, 0.f);
            auto fetch = [&](int t, bool valid) {
                GradSample v;
                if constexpr (MODE == 0) {
                    v.gx = l_gx[t * kWave];
                    v.gy = l_gy[t * kWave];
                } else {
                    const unsigned off = TAB ? off_at(t) : (kFloatWalk ? fw.off : soff(w));
                    v.gx = valid ? buf_f32(r_gx, off, goff) : 0.f;
                    v.gy = valid ? buf_f32(r_gy, off, goff) : 0.f;
                    if constexpr (!KEEP_RS) v.ref = valid ? buf_f32(r_ref, off, roff) : 0.f;
                }
                if constexpr (!TAB) v.xy = kFloatWalk ? fw.xy : mk2((float)(w.c - rx) - offx, (float)(w.r - ry) - offy);
                if constexpr (kFloatWalk) fw.next();
                else w.next();
                return v;
            };
            auto sample = [&](int t, bool valid, const GradSample& v) {
                const float g_x = v.gx, g_y = v.gy;
                [[maybe_unused]] const float ref_v = v.ref;
                const float tz = l_ts[t * kWave] - tmean;  // same bits as in the norm pass
                // the zero-mean reference value: parked in LDS, or re-formed from the image (same subtraction, same bits)
                const float rsv = KEEP_RS ? l_rs[t * kWave] : ref_v - ref_mean_v;
                const float e = mad(tz, factor, -rsv);
                ssd = valid ? mad(e, e, ssd) : ssd;
                if constexpr (DOF == 6) {
                    f2 xy;
                    if constexpr (TAB) xy = tab_at(t) - mk2(offx, offy);
                    else xy = v.xy;
                    const f2 A = g_x * xy, B = g_y * xy;  // (sd1, sd2), (sd4, sd5)
                    const f2 mA = mad(A, e, nA), mB = mad(B, e, nB);
                    const float m0 = mad(g_x, e, num[0]), m3 = mad(g_y, e, num[3]);
                    if (valid) {
                        nA = mA; nB = mB; num[0] = m0; num[3] = m3;
                    }
                } else {
                    f2 lxy;
                    if constexpr (TAB) lxy = tab_at(t) - mk2(offx, offy);
                    else lxy = v.xy;
                    const float fxl = lxy.x, fyl = lxy.y;
                    const float xx = (fxl * fxl) * 0.5f, xy = fxl * fyl, yy = (fyl * fyl) * 0.5f;
                    const f2 m01 = mk2(1.f, fxl), m23 = mk2(fyl, xx), m45 = mk2(xy, yy);
                    const f2 sdp[6] = {g_x * m01, g_x * m23, g_x * m45, g_y * m01, g_y * m23, g_y * m45};
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        const f2 nv = mad(sdp[q], e, np12[q]);
                        np12[q] = valid ? nv : np12[q];
                    }
                }
            };
            passes_batched<kNumBatch>(NF, NT, (NF * kWave + lane) < N, fetch, sample);
            if constexpr (DOF == 6) {
                num[1] = nA.x; num[2] = nA.y; num[4] = nB.x; num[5] = nB.y;
            } else {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    num[(2 * q) % DOF] = np12[q].x;
                    num[(2 * q + 1) % DOF] = np12[q].y;
                }
            }
        }
        // the DOF numerator sums and the sum of squared errors in one transposing butterfly
        float red[DOF + 1];
#pragma unroll
        for (int j = 0; j < DOF; j++) red[j] = num[j];
        red[DOF] = ssd;
        wave_allreduce_sum_multi<DOF + 1>(red, lane);
        znssd = uni(red[DOF]) / (ref_norm * ref_norm);
        if constexpr (LM) {
            // src/oc_iclm.cpp:250-257: lambda from the first ZNSSD; (H + lambda * I)^-1 every iteration
            if (iter == 1) lambda = uni(pow_lambda(P.lm_log_lambda, znssd / znssd0) - 1.f);
            float dcol[DOF];
#pragma unroll
            for (int i = 0; i < DOF; i++) dcol[i] = hcol[i] + lambda * (lane == i ? 1.f : 0.f);
            lu_inverse_lanes<DOF>(dcol, hinv_col, lane);
        }
        // dp = H^-1 * numerator (src/oc_icgn.cpp:279-286): lane j forms H^-1(i,j) * num[j], the
        // products of row i are then added in ascending j exactly like the reference loop
        float dp[DOF];
        if constexpr (LM) {
            float numj = 0.f;
#pragma unroll
            for (int j = 0; j < DOF; j++) numj = lane == j ? red[j] : numj;
#pragma unroll
            for (int i = 0; i < DOF; i++) {
                const float prod = hinv_col[i] * numj;
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < DOF; j++) v += wave_bcast(prod, j);
                dp[i] = v;
            }
        } else {
            float mine = 0.f;  // lane i < DOF: dp[i]
#pragma unroll
            for (int j = 0; j < DOF; j++) mine += hinv_row[j] * red[j];
#pragma unroll
            for (int i = 0; i < DOF; i++) dp[i] = wave_bcast(mine, i);
        }
        // IC-LM applies the step only when the ZNSSD went down (src/oc_iclm.cpp:283-300); wave-uniform
        const bool accept = !LM || znssd < znssd0;
        if constexpr (LM) {
            lambda = lambda * (accept ? P.lm_alpha : P.lm_beta);
            znssd0 = accept ? znssd : znssd0;
        }
        // W <- W * (dW)^-1 ; p <- W (src/oc_icgn.cpp:287-293 / 828-834)
        const int rx2 = rx * rx, ry2 = ry * ry;
        if constexpr (DOF == 6) {
            float dW[9], dWi[9], Wn[9];
            set_warp_2d1(dW, dp[0], dp[1], dp[2], dp[3], dp[4], dp[5]);
            inverse3(dW, dWi);
            mat_mul<3>(Wm, dWi, Wn);
            if (accept) {
#pragma unroll
                for (int i = 0; i < 9; i++) Wm[i] = uni(Wn[i]);
                // src/oc_deformation.cpp:107-115
                cur[0] = Wm[2]; cur[1] = Wm[0] - 1.f; cur[2] = Wm[1];
                cur[6] = Wm[5]; cur[7] = Wm[3]; cur[8] = Wm[4] - 1.f;
            }
            // convergence norm (src/oc_icgn.cpp:296-306)
            const float d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3] * dp[3] +
                            dp[4] * dp[4] * rx2 + dp[5] * dp[5] * ry2;
            dp_norm = uni(sqrtf(d));
        } else {
            float dW[36];
            set_warp_2d2(dW, dp);
            // (dW)^-1 by the lane-distributed LU (Eigen PartialPivLU for 6x6, src/oc_icgn.cpp:831)
            float dcol[6], dinv[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                float c = 0.f;
#pragma unroll
                for (int j = 0; j < 6; j++) c = lane == j ? dW[i * 6 + j] : c;
                dcol[i] = c;
            }
            lu_inverse_lanes<6>(dcol, dinv, lane);
            // lane j: column j of W * dW^-1, inner index ascending
            float ncol[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                float v = wave_bcast(Wcol[i], 0) * dinv[0];
#pragma unroll
                for (int k = 1; k < 6; k++) v = v + wave_bcast(Wcol[i], k) * dinv[k];
                ncol[i] = v;
            }
            if (accept) {
#pragma unroll
            for (int i = 0; i < 6; i++) Wcol[i] = ncol[i];
            // Deformation2D2::setDeformation(), src/oc_deformation.cpp:284-299
            const float r30 = wave_bcast(Wcol[3], 0), r31 = wave_bcast(Wcol[3], 1), r32 = wave_bcast(Wcol[3], 2);
            const float r33 = wave_bcast(Wcol[3], 3), r34 = wave_bcast(Wcol[3], 4), r35 = wave_bcast(Wcol[3], 5);
            const float r40 = wave_bcast(Wcol[4], 0), r41 = wave_bcast(Wcol[4], 1), r42 = wave_bcast(Wcol[4], 2);
            const float r43 = wave_bcast(Wcol[4], 3), r44 = wave_bcast(Wcol[4], 4), r45 = wave_bcast(Wcol[4], 5);
            cur[0] = r35; cur[1] = r33 - 1.f; cur[2] = r34; cur[3] = r30 * 2.f; cur[4] = r31; cur[5] = r32 * 2.f;
            cur[6] = r45; cur[7] = r43; cur[8] = r44 - 1.f; cur[9] = r40 * 2.f; cur[10] = r41; cur[11] = r42 * 2.f;
            }
            const int rxy2 = rx2 * ry2;
            constexpr int D = DOF;  // keeps the dp[] indices in range when this branch is discarded
            float d;
            if constexpr (!LM) {
                // src/oc_icgn.cpp:837-857 (integer-truncated weights are reference behaviour)
                const int rx4 = (int)(rx2 * rx2 * 0.25f), ry4 = (int)(ry2 * ry2 * 0.25f);
                d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3 % D] * dp[3 % D] * rx4 +
                    dp[5 % D] * dp[5 % D] * ry4 + dp[4 % D] * dp[4 % D] * rxy2 + dp[6 % D] * dp[6 % D] +
                    dp[7 % D] * dp[7 % D] * rx2 + dp[8 % D] * dp[8 % D] * ry2 + dp[9 % D] * dp[9 % D] * rx4 +
                    dp[11 % D] * dp[11 % D] * ry4 + dp[10 % D] * dp[10 % D] * rxy2;
            } else {
                // src/oc_iclm.cpp:675-687: p * p * rx2 * rx2 * 0.25f, left to right in float
                d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3 % D] * dp[3 % D] * rx2 * rx2 * 0.25f +
                    dp[5 % D] * dp[5 % D] * ry2 * ry2 * 0.25f + dp[4 % D] * dp[4 % D] * rxy2 + dp[6 % D] * dp[6 % D] +
                    dp[7 % D] * dp[7 % D] * rx2 + dp[8 % D] * dp[8 % D] * ry2 + dp[9 % D] * dp[9 % D] * rx2 * rx2 * 0.25f +
                    dp[11 % D] * dp[11 % D] * ry2 * ry2 * 0.25f + dp[10 % D] * dp[10 % D] * rxy2;
            }
            dp_norm = uni(sqrtf(d));
        }
    } while ((OC_ABLATE2D & 1) ? iter < 3 : (iter < P.stop && dp_norm >= P.conv));

    lap(3);
    // ---- outputs (src/oc_icgn.cpp:310-340; 2D2: 860-897)
    if (lane == 0) {
        if constexpr ((OC_ABLATE2D & 16) != 0) {
            poi[20] = (float)tl[0] * 1e-3f;  // strain exx, eyy, exy and result.feature (src/oc_poi.h:102-136)
            poi[21] = (float)tl[1] * 1e-3f;
            poi[22] = (float)tl[2] * 1e-3f;
            poi[19] = (float)tl[3] * 1e-3f;
        }
        float zncc = 0.5f * (2 - znssd);
        const float fiter = (float)iter;
        if (dp_norm >= P.conv && fiter >= P.stop) zncc = -4.f;
        float out_u = cur[0], out_v = cur[6];
        if (isnan(zncc) || isnan(out_u) || isnan(out_v)) {
            out_u = u_in;
            out_v = v_in;
            zncc = -5.f;
        }
        poi[poi2d::U] = out_u;
        poi[poi2d::UX] = cur[1];
        poi[poi2d::UY] = cur[2];
        poi[poi2d::V] = out_v;
        poi[poi2d::VX] = cur[7];
        poi[poi2d::VY] = cur[8];
        if constexpr (DOF == 12) {
            poi[poi2d::UXX] = cur[3];
            poi[poi2d::UXY] = cur[4];
            poi[poi2d::UYY] = cur[5];
            poi[poi2d::VXX] = cur[9];
            poi[poi2d::VXY] = cur[10];
            poi[poi2d::VYY] = cur[11];
        }
        poi[poi2d::U0] = u_in;
        poi[poi2d::V0] = v_in;
        poi[poi2d::ZNCC] = zncc;
        poi[poi2d::ITER] = fiter;
        poi[poi2d::CONV] = dp_norm;
        poi[poi2d::SRX] = (float)rx;
        poi[poi2d::SRY] = (float)ry;
    }
}

// ---------------------------------------------------------------------------
// launch: a table of kernel variants (all bit-identical), selected per engine through
// oc_hip_set_tuning("icgn2d_variant", i) / ("icgn2d_xcd", 0|1); defaults chosen from the
// MI355X sweep in DESIGN.md section 4.
// ---------------------------------------------------------------------------
constexpr int kLdsBudget = 160 * 1024 - 2048;  // dynamic LDS; 2 KB stay free for the static area of the cooperative inverse

struct VariantInfo {
    int g, mode, pipe, wpb, occ;
};

template <int DOF, int G, int MODE, int PIPE, int WPB, int OCC, int OFFS, int LM = 0, int PHASE = 0>
static hipError_t launch_t(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int nt, bool xcd,
                           hipStream_t stream) {
    constexpr int arrays = PHASE == 1 ? 0 : (MODE == 0 ? 4 : ((MODE == 4 || MODE == 2) ? 1 : 2));  // (the set-up kernel: table only)
    size_t lds = ((size_t)arrays * WPB + (MODE >= 3 ? 3 : 0)) * nt * kWave * sizeof(float);
    if (lds > (size_t)kLdsBudget) return hipErrorInvalidValue;
#if OC_BUILD_AB
    // experiments only, A/B build only (tools/icgn2d_occupancy_probe.py): OC_ICGN2D_LDS_PAD=<bytes> raises the workgroup's LDS
    // request, i.e. lowers the number of workgroups a CU holds, with nothing else changed
    static const size_t lds_pad = std::getenv("OC_ICGN2D_LDS_PAD") ? (size_t)std::atol(std::getenv("OC_ICGN2D_LDS_PAD")) : 0;
    if (lds_pad && lds_pad <= (size_t)kLdsBudget && lds_pad > lds) lds = lds_pad;
#endif
    if (PHASE != 0 && !p.setup) return hipErrorInvalidValue;
    auto kern = icgn2d_kernel<DOF, G, MODE, PIPE, WPB, OCC, OFFS, LM, PHASE>;
    // the dynamic-LDS limit is a per-device property of the loaded function: raise it once on every
    // device this process launches on (one engine per device is a supported host layout)
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    hipError_t derr = hipGetDevice(&dev);
    if (derr != hipSuccess) return derr;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (err != hipSuccess) return err;
        attr_devices.fetch_or(bit, std::memory_order_release);
    }
    const size_t groups = (count + WPB - 1) / WPB;
    Icgn2dLaunch L;
    L.stride_f = stride_f;
    L.nt = nt;
    L.count = count;
    L.xcd_chunk = xcd ? (int)((groups + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)L.xcd_chunk * 8 : groups;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WPB), lds, stream, p, pois, L);
    return hipGetLastError();
}

// 0: everything per sample parked in LDS, one wave per workgroup (the first design; kept as the reference point of the
//    sweeps), 1: LDS-light single wave (the fallback for large subsets and the IC-LM launch shape), 2: the ICGN2D1
//    default, 3: the ICGN2D2 default (deeper gathers, fewer registers per wave budget), 4: per-workgroup coordinate
//    table, target array only, 5: the same with G = 2 inside 80 VGPRs, i.e. three 8-wave workgroups (6 waves per SIMD)
//    per CU -- the ICGN2D1 default.  Interleaved A/B timing on config B (tools/variant_ab.py,
//    profiles/r02w_icgn2d1_variant_ab.json): 2: 3.71 ms, 4: 3.65, 5: 3.64; G = 3 at 6 waves per SIMD spills (4.29 ms),
//    G = 4 / G = 1 / 5 waves per SIMD lose 1 - 4 %.  The kernel is VALU-issue bound: occupancy barely matters.
//    6 (round 4): variant 5's table and lockstep sweeps in 4-wave workgroups, four per CU (more independent workgroups per CU,
//    fewer waves per SIMD): 3.47 against 3.30 ms (profiles/r4i_icgn2d1_variant_ab_4wave_lockstep.json) -- not selected automatically.
//    7 (round 5): variant 2 with the target array only (MODE 2) -- the self-adaptive default: half the LDS, so that a batch
//    whose largest subset is 41 x 41 still runs four 4-wave workgroups per CU instead of two.
// Round 1's G = 2 and software-pipelined variants never won a sweep and are gone.
// Variants 0, 6 and 8 are measured losers kept as A/B partners: they are compiled only into the A/B build of the library
// (-DOC_BUILD_AB=1, opencorr_amd/build.py build_ab(); tests that compare them load that build).
//        id  G mode pipe wpb occ
#define OC_ICGN2D_VARIANTS_PRODUCT(X) \
    X(1, 3, 1, 0, 1, 4)       \
    X(2, 3, 1, 0, 4, 4)       \
    X(3, 4, 1, 0, 1, 3)       \
    X(4, 3, 4, 0, 8, 4)       \
    X(5, OC_V5_G, 4, 0, 8, OC_V5_OCC) \
    X(7, 3, 2, 0, 4, 4)
#define OC_ICGN2D_VARIANTS_AB(X) \
    X(0, 3, 0, 0, 1, 1)       \
    X(6, 2, 4, 0, 4, 4)
#if OC_BUILD_AB
#define OC_ICGN2D_VARIANTS(X) OC_ICGN2D_VARIANTS_PRODUCT(X) OC_ICGN2D_VARIANTS_AB(X)
#else
#define OC_ICGN2D_VARIANTS(X) OC_ICGN2D_VARIANTS_PRODUCT(X)
#endif
#define OC_ICGN2D_VARIANTS_ALL(X) OC_ICGN2D_VARIANTS_PRODUCT(X) OC_ICGN2D_VARIANTS_AB(X)

constexpr int kSplitVariant = 8;

template <int DOF>
static hipError_t launch_dof(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int variant, bool xcd,
                             hipStream_t stream, int phase = 0) {
    if (count == 0) return hipSuccess;
    // p.rx, p.ry: the engine's radius, or in self-adaptive mode the largest radii of the batch
    const int N = (2 * p.rx + 1) * (2 * p.ry + 1);
    const int nt = (N + 63) / 64;
    switch (variant) {
#define X(ID, GG, MM, PP, WW, OO)                                                                           \
    case ID:                                                                                                \
        return p.offsets ? launch_t<DOF, GG, MM, PP, WW, OO, 1>(p, pois, stride_f, count, nt, xcd, stream)  \
                         : launch_t<DOF, GG, MM, PP, WW, OO, 0>(p, pois, stride_f, count, nt, xcd, stream);
        OC_ICGN2D_VARIANTS(X)
#undef X
#if OC_BUILD_AB
        case kSplitVariant: {
            // the split launch shape: the set-up kernel (table only in LDS, 8 / 4 waves per SIMD by registers) files
            // mean, norm and H^-1 per POI in p.setup, the iteration kernel -- the big-queue default's shape, variant 5 for
            // 6 DoF and variant 4 for 12 -- reads them.  Back to back on one stream here; capi.hip may instead feed the two
            // kernels chunk-wise on two streams so that set-up workgroups of chunk k + 1 are co-resident with iteration
            // workgroups of chunk k (`phase` = 1 / 2 launches one of them alone).
            // MEASURED in round 5 and SLOWER in every form (profiles/r5d_icgn2d_split_launch_shape_ab.json: config B 3.43 vs
            // 3.39 ms back to back, 3.62 in four chunks on two streams; config C 3.87 vs 3.61): the fused kernel's phases of
            // different workgroups already overlap on a CU, and three iteration workgroups leave no registers for a
            // set-up workgroup to move in beside them.  A/B build only.
            constexpr int SOCC = DOF == 6 ? 8 : 4, IG = DOF == 6 ? OC_V5_G : 3, IOCC = DOF == 6 ? OC_V5_OCC : 4;
            hipError_t err = hipSuccess;
            if (phase == 0 || phase == 1)
                err = p.offsets ? launch_t<DOF, 1, 4, 0, 8, SOCC, 1, 0, 1>(p, pois, stride_f, count, nt, xcd, stream)
                                : launch_t<DOF, 1, 4, 0, 8, SOCC, 0, 0, 1>(p, pois, stride_f, count, nt, xcd, stream);
            if (err != hipSuccess || phase == 1) return err;
            return p.offsets ? launch_t<DOF, IG, 4, 0, 8, IOCC, 1, 0, 2>(p, pois, stride_f, count, nt, xcd, stream)
                             : launch_t<DOF, IG, 4, 0, 8, IOCC, 0, 0, 2>(p, pois, stride_f, count, nt, xcd, stream);
        }
#endif
        default: return hipErrorNotSupported;  // an A/B partner that this build does not contain, or an unknown id
    }
}

hipError_t launch_icgn2d1(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int variant, bool xcd,
                          hipStream_t stream, int phase) {
    return launch_dof<6>(p, pois, stride_f, count, variant, xcd, stream, phase);
}

hipError_t launch_icgn2d2(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int variant, bool xcd,
                          hipStream_t stream, int phase) {
    return launch_dof<12>(p, pois, stride_f, count, variant, xcd, stream, phase);
}

// IC-LM: one launch shape each (G = 3, gradients re-read, one wave per workgroup so the LDS limit is the
// largest; the per-iteration LU keeps more registers live, hence occupancy 3)
hipError_t launch_iclm2d1(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int nt = ((2 * p.rx + 1) * (2 * p.ry + 1) + 63) / 64;
    return launch_t<6, 3, 1, 0, 1, 4, 0, 1>(p, pois, stride_f, count, nt, xcd, stream);
}

hipError_t launch_iclm2d2(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int nt = ((2 * p.rx + 1) * (2 * p.ry + 1) + 63) / 64;
    return launch_t<12, 4, 1, 0, 1, 3, 0, 1>(p, pois, stride_f, count, nt, xcd, stream);
}

}  // namespace OC_ARITH

#if !OC_FMA
// ---- the arithmetic-independent part and the dispatch between the two builds (this translation unit only) ----
using OC_ARITH::kLdsBudget;
constexpr int kIcgn2dVariants = 10;  // 0 ... 7: one kernel each; 8: the split launch shape (set-up kernel + iteration kernel);
                                     // 9: the band kernel of icgn2d_band.hip (capi.hip launches it; listed here for the tuning key)

int icgn2d_variant_count() { return kIcgn2dVariants; }

// the launch shape of a variant id, whether this build contains it or not
int icgn2d_variant_info(int variant, int* g, int* mode, int* pipe, int* wpb, int* occ) {
    switch (variant) {
#define X(ID, GG, MM, PP, WW, OO) \
    case ID: *g = GG; *mode = MM; *pipe = PP; *wpb = WW; *occ = OO; return 0;
        OC_ICGN2D_VARIANTS_ALL(X)
#undef X
        case 8: *g = OC_V5_G; *mode = 4; *pipe = 0; *wpb = 8; *occ = OC_V5_OCC; return 0;  // (the iteration kernel's shape at 6 DoF)
        case 9: *g = 1; *mode = 5; *pipe = 0; *wpb = 8; *occ = 6; return 0;  // icgn2d_band.hip: table + LDS band, subset in registers
        default: return -1;
    }
}

bool icgn2d_variant_built(int variant) {
    switch (variant) {
#define X(ID, GG, MM, PP, WW, OO) \
    case ID: return true;
        OC_ICGN2D_VARIANTS(X)
#undef X
        case 8: return OC_BUILD_AB != 0;
        case 9: return OC_BUILD_AB != 0;   // icgn2d_band.hip: measured 1.7 x slower, A/B build only
        default: return false;
    }
}

int icgn2d_setup_record_floats(int dof) { return sep::icgn2d_setup_floats(dof); }

// largest sample count a variant can hold in LDS
int icgn2d_max_samples(int variant) {
    int g, mode, pipe, wpb, occ;
    if (icgn2d_variant_info(variant, &g, &mode, &pipe, &wpb, &occ)) return 0;
    if (variant == 9) return 0;  // (capi.hip asks icgn2d_band_supported instead)
    const int arrays = mode == 0 ? 4 : ((mode == 4 || mode == 2) ? 1 : 2);
    return kLdsBudget / ((arrays * wpb + (mode >= 3 ? 3 : 0)) * (int)sizeof(float) * kWave) * kWave;
}

int iclm2d_max_samples() { return kLdsBudget / (2 * (int)sizeof(float) * kWave) * kWave; }

// p.arith_fma selects the build whose per-sample multiply-adds are fused (oc_device.h; icgn2d_fma.o)
hipError_t launch_icgn2d1(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int variant, bool xcd,
                          hipStream_t stream, int phase) {
    return p.arith_fma ? fma::launch_icgn2d1(p, pois, stride_f, count, variant, xcd, stream, phase)
                       : sep::launch_icgn2d1(p, pois, stride_f, count, variant, xcd, stream, phase);
}
hipError_t launch_icgn2d2(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int variant, bool xcd,
                          hipStream_t stream, int phase) {
    return p.arith_fma ? fma::launch_icgn2d2(p, pois, stride_f, count, variant, xcd, stream, phase)
                       : sep::launch_icgn2d2(p, pois, stride_f, count, variant, xcd, stream, phase);
}
hipError_t launch_iclm2d1(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    return p.arith_fma ? fma::launch_iclm2d1(p, pois, stride_f, count, xcd, stream) : sep::launch_iclm2d1(p, pois, stride_f, count, xcd, stream);
}
hipError_t launch_iclm2d2(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    return p.arith_fma ? fma::launch_iclm2d2(p, pois, stride_f, count, xcd, stream) : sep::launch_iclm2d2(p, pois, stride_f, count, xcd, stream);
}

// largest subset radii of a POI queue (self-adaptive mode sizes the LDS arrays from them)
__global__ __launch_bounds__(256) void poi2d_max_radius_kernel(const float* __restrict__ pois, int stride_f,
                                                               unsigned long long count, int* __restrict__ out) {
    int mx = 0, my = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        mx = max(mx, (int)pois[i * stride_f + poi2d::SRX]);
        my = max(my, (int)pois[i * stride_f + poi2d::SRY]);
    }
    atomicMax(out, mx);
    atomicMax(out + 1, my);
}

hipError_t launch_poi2d_max_radius(const float* pois, int stride_f, size_t count, int* out2, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(out2, 0, 2 * sizeof(int), stream);
    if (err != hipSuccess || count == 0) return err;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    (void)hipGetLastError();
    hipLaunchKernelGGL(poi2d_max_radius_kernel, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned long long)count,
                       out2);
    return hipGetLastError();
}

#endif  // !OC_FMA

}  // namespace ochip
