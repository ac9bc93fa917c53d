// valu_ubench.hip -- how many cycles does one wave64 VALU instruction occupy a SIMD on gfx950?
//
// Settles the 2-vs-4-cycle question behind "the ICGN kernels are VALU-issue-bound" (VERDICT round 1, item 2):
// every wave runs REPS x 32 instructions of one kind on 32 independent registers (no dependency closer than 32
// instructions), with W waves resident per SIMD; the figure of merit is
//     cycles per instruction per SIMD = kernel cycles x SIMDs / wave-instructions issued,
// cycles from s_memtime stamps inside the waves (shader clock).  Every wave also records WHERE it ran (HW_REG_HW_ID:
// SE / SH / CU / SIMD, and the XCC id), so the host groups the waves by SIMD and divides that SIMD's busy span
// (last end - first start) by the instructions its waves issued -- no assumption about how the dispatcher spreads the
// workgroups.  The hipEvent time is reported beside it (launch overhead included).  Instruction kinds: plain fp32 (v_mul_f32, v_add_f32, v_fma_f32), packed fp32
// (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32: two IEEE results per lane), the movers the compiler inserts to form
// register pairs (v_mov_b32), integer address arithmetic (v_mul_u32_u24, v_add_u32), v_cndmask_b32, DPP adds and
// v_readlane_b32 (the reductions), v_floor_f32 / v_cvt_i32_f32 (the range rule).
// Build: hipcc --offload-arch=gfx950 -O3 valu_ubench.hip -o valu_ubench ; prints one JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 16384;

// 32 instructions, destination/source register i of a 32-register file (packed kinds use 16 register pairs, twice)
#define R4(M, a) M(a) M(a + 1) M(a + 2) M(a + 3)
#define R16(M, a) R4(M, a) R4(M, a + 4) R4(M, a + 8) R4(M, a + 12)
#define R32(M) R16(M, 0) R16(M, 16)

enum Kind { MUL, ADD, FMA, PK_MUL, PK_ADD, PK_FMA, MOV, MUL_U24, ADD_U32, CNDMASK, DPP_ADD, READLANE, FLOOR, CVT_I32, NKIND };
static const char* kNames[NKIND] = {"v_mul_f32", "v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_mov_b32",
                                    "v_mul_u32_u24", "v_add_u32", "v_cndmask_b32", "v_add_f32_dpp", "v_readlane_b32", "v_floor_f32",
                                    "v_cvt_i32_f32"};

struct WaveRec {
    unsigned long long t0, t1;
    unsigned hw_id, xcc_id;
};

template <int KIND>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, WaveRec* __restrict__ cycles, float seed) {
    float r[32];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 32; i++) r[i] = seed + (float)i * 1.0e-3f + (float)threadIdx.x * 1.0e-6f;
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = f2{r[2 * i], r[2 * i + 1]};
    const float c = 1.0000001f;
    const f2 c2 = {c, c};
    int sink = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma nounroll
    for (int it = 0; it < REPS; it++) {
        if constexpr (KIND == MUL) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (KIND == ADD) {
#define M(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (KIND == FMA) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (KIND == PK_MUL) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) & 15]) : "v"(c2));
            R32(M)
#undef M
        } else if constexpr (KIND == PK_ADD) {
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i) & 15]) : "v"(c2));
            R32(M)
#undef M
        } else if constexpr (KIND == PK_FMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(i) & 15]) : "v"(c2));
            R32(M)
#undef M
        } else if constexpr (KIND == MOV) {
#define M(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[((i) + 1) & 31]));
            R32(M)
#undef M
        } else if constexpr (KIND == MUL_U24) {
#define M(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (KIND == ADD_U32) {
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (KIND == CNDMASK) {
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
            R32(M)
#undef M
        } else if constexpr (KIND == DPP_ADD) {
#define M(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
            R32(M)
#undef M
        } else if constexpr (KIND == READLANE) {
#define M(i) { int s_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s_) : "v"(r[i])); sink += s_; }
            R32(M)
#undef M
        } else if constexpr (KIND == FLOOR) {
#define M(i) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));
            R32(M)
#undef M
        } else if constexpr (KIND == CVT_I32) {
#define M(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
            R32(M)
#undef M
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = (float)sink;
#pragma unroll
    for (int i = 0; i < 32; i++) acc += r[i];
#pragma unroll
    for (int i = 0; i < 16; i++) acc += p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        WaveRec r;
        r.t0 = t0;
        r.t1 = t1;
        r.hw_id = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);    // HW_REG_HW_ID
        r.xcc_id = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20);  // HW_REG_XCC_ID
        cycles[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
}

struct Result {
    double ms, wave_cycles_mean, cyc_per_inst_simd, waves_per_simd_seen;
};

template <int KIND>
Result run(int blocks, float* out, WaveRec* cyc) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.f);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<WaveRec> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(WaveRec), hipMemcpyDeviceToHost));
    double sum = 0;
    // group by SIMD: key = (xcc, se, sh, cu, simd) out of HW_ID[15:4] (simd 5:4, cu 11:8, sh 12, se 15:13) and XCC_ID[3:0]
    std::map<unsigned, std::pair<std::pair<unsigned long long, unsigned long long>, int>> simd;
    for (const WaveRec& r : h) {
        sum += (double)(r.t1 - r.t0);
        const unsigned key = ((r.xcc_id & 0xf) << 16) | (r.hw_id & 0xff30);
        auto it = simd.find(key);
        if (it == simd.end()) simd[key] = {{r.t0, r.t1}, 1};
        else {
            it->second.first.first = std::min(it->second.first.first, r.t0);
            it->second.first.second = std::max(it->second.first.second, r.t1);
            it->second.second++;
        }
    }
    double cpi = 0, waves = 0;
    for (auto& kv : simd) {
        cpi += (double)(kv.second.first.second - kv.second.first.first) / ((double)kv.second.second * REPS * 32);
        waves += kv.second.second;
    }
    return {ms, sum / h.size(), cpi / simd.size(), waves / simd.size()};
}

typedef Result (*RunFn)(int, float*, WaveRec*);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out;
    WaveRec* cyc;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(WaveRec)));
    RunFn fns[NKIND] = {run<MUL>, run<ADD>, run<FMA>, run<PK_MUL>, run<PK_ADD>, run<PK_FMA>, run<MOV>, run<MUL_U24>, run<ADD_U32>,
                        run<CNDMASK>, run<DPP_ADD>, run<READLANE>, run<FLOOR>, run<CVT_I32>};
    printf("{\"device\": \"%s\", \"cus\": %d, \"reps\": %d, \"instructions_per_wave\": %d, \"results\": [\n", prop.gcnArchName, cus, REPS, REPS * 32);
    bool first = true;
    // a 256-thread block puts one wave on each SIMD of a CU: blocks per CU = waves per SIMD
    for (int wps : {1, 2, 4, 8}) {
        for (int kind = 0; kind < NKIND; kind++) {
            const Result r = fns[kind](cus * wps, out, cyc);
            const double inst = (double)REPS * 32;  // per wave
            printf("%s  {\"inst\": \"%s\", \"blocks_per_cu\": %d, \"waves_per_simd_seen\": %.2f, \"ms\": %.4f, \"wave_cycles_per_inst\": %.3f, "
                   "\"cycles_per_inst_per_simd\": %.3f, \"ginst_per_s_chip\": %.1f}",
                   first ? "" : ",\n", kNames[kind], wps, r.waves_per_simd_seen, r.ms, r.wave_cycles_mean / inst, r.cyc_per_inst_simd,
                   inst * cus * wps * 4 / (r.ms * 1e6));
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
