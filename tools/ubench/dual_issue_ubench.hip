// dual_issue_ubench.hip -- which wave64 VALU instructions does gfx950 issue two at a time, and does the pairing survive in
// mixed instruction streams?  (valu_ubench.hip found v_mul/add_f32, v_mov_b32, v_add_u32 at 2.4 cycles per instruction
// and SIMD -- two per 4-cycle slot, from two waves -- and everything else at 4.1 - 4.4.  The ICGN kernels retire only
// ~0.9 instructions per slot although three quarters of their instructions are of the fast kind.)
//
// Every wave runs REPS x 32 instructions of one PATTERN on independent registers; W waves per SIMD (W blocks of 256
// threads per CU).  Figure of merit: s_memtime units per instruction and SIMD = mean wave span x 1 / (REPS * 32 * W) ...
// reported relative to nothing: compare rows.  Patterns: single kinds (to classify more opcodes) and 1:1 / 3:1 mixes of
// the fast kind with slow kinds, with LDS reads, with SALU work, and a fully dependent chain.
// Build: hipcc --offload-arch=gfx950 -O3 dual_issue_ubench.hip -o dual_issue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 8192;
typedef float f2 __attribute__((ext_vector_type(2)));

#define R4(M, a) M(a) M(a + 1) M(a + 2) M(a + 3)
#define R16(M, a) R4(M, a) R4(M, a + 4) R4(M, a + 8) R4(M, a + 12)
#define R32(M) R16(M, 0) R16(M, 16)

enum Pat {
    P_MUL, P_SUB, P_MAX, P_AND, P_LSHL, P_FMAC, P_CMP, P_CNDMASK_S, P_MUL_SGPR, P_MUL_LIT,
    P_CHAIN, P_MUL_PK, P_MUL_CVT, P_MUL_FMA, P_MUL_DPP, P_MUL3_CVT1, P_MUL_LDS, P_MUL_SALU, P_ADD_MUL, NPAT
};
static const char* kNames[NPAT] = {
    "v_mul_f32 x32", "v_sub_f32 x32", "v_max_f32 x32", "v_and_b32 x32", "v_lshlrev_b32 x32", "v_fmac_f32 x32", "v_cmp_gt_f32 x32",
    "v_cndmask_b32 (sgpr mask) x32", "v_mul_f32 with an SGPR source x32", "v_mul_f32 with a literal x32",
    "v_mul_f32 dependent chain x32", "v_mul_f32 / v_pk_mul_f32 1:1", "v_mul_f32 / v_cvt_i32_f32 1:1", "v_mul_f32 / v_fma_f32 1:1",
    "v_mul_f32 / v_add_f32_dpp 1:1", "v_mul_f32 / v_cvt_i32_f32 3:1", "v_mul_f32 x28 + ds_read_b32 x4", "v_mul_f32 / s_add_u32 1:1",
    "v_add_f32 / v_mul_f32 1:1"};

template <int PAT>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, unsigned long long* __restrict__ span, float seed, float sc) {
    __shared__ float lds[256];
    lds[threadIdx.x] = seed;
    __syncthreads();
    float r[32];
    f2 p[16];
    int q[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        r[i] = seed + (float)i * 1.0e-3f + (float)threadIdx.x * 1.0e-6f;
        q[i] = i + threadIdx.x;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = f2{r[2 * i], r[2 * i + 1]};
    const float c = 1.0000001f;
    const f2 c2 = {c, c};
    const unsigned long long mask = 0x5555555555555555ull;
    const unsigned addr = threadIdx.x * 4;
    unsigned sacc = 0;
    float ld0 = 0.f, ld1 = 0.f, ld2 = 0.f, ld3 = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma nounroll
    for (int it = 0; it < REPS; it++) {
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) & 15]) : "v"(c2));
#define CVT(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(r[i]));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
#define DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
        if constexpr (PAT == P_MUL) {
            R32(MUL)
        } else if constexpr (PAT == P_SUB) {
#define M(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (PAT == P_MAX) {
#define M(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (PAT == P_AND) {
#define M(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 31]));
            R32(M)
#undef M
        } else if constexpr (PAT == P_LSHL) {
#define M(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(q[i]));
            R32(M)
#undef M
        } else if constexpr (PAT == P_FMAC) {
#define M(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[i]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (PAT == P_CMP) {
#define M(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(c) : "vcc");
            R32(M)
#undef M
        } else if constexpr (PAT == P_CNDMASK_S) {
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "s"(mask));
            R32(M)
#undef M
        } else if constexpr (PAT == P_MUL_SGPR) {
#define M(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "s"(sc));
            R32(M)
#undef M
        } else if constexpr (PAT == P_MUL_LIT) {
#define M(i) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(r[i]));
            R32(M)
#undef M
        } else if constexpr (PAT == P_CHAIN) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[0]) : "v"(c));
            R32(M)
#undef M
        } else if constexpr (PAT == P_MUL_PK) {
#define M(i) MUL(i) PKM(i)
            R16(M, 0)
#undef M
        } else if constexpr (PAT == P_MUL_CVT) {
#define M(i) MUL(i) CVT(i + 16)
            R16(M, 0)
#undef M
        } else if constexpr (PAT == P_MUL_FMA) {
#define M(i) MUL(i) FMA(i + 16)
            R16(M, 0)
#undef M
        } else if constexpr (PAT == P_MUL_DPP) {
#define M(i) MUL(i) DPP(i + 16)
            R16(M, 0)
#undef M
        } else if constexpr (PAT == P_MUL3_CVT1) {
#define M(i) MUL(i) MUL(i + 8) MUL(i + 16) CVT(i + 24)
            R4(M, 0) R4(M, 4)
#undef M
        } else if constexpr (PAT == P_MUL_LDS) {
#define M(i) MUL(i)
            R4(M, 0) R4(M, 4) R4(M, 8) R4(M, 12) R4(M, 16) R4(M, 20) R4(M, 24)
#undef M
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(ld0), "=v"(ld1), "=v"(ld2), "=v"(ld3) : "v"(addr & 0x3f0));
        } else if constexpr (PAT == P_MUL_SALU) {
#define M(i) MUL(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
            R16(M, 0)
#undef M
        } else {
#define M(i) ADD(i) MUL(i + 16)
            R16(M, 0)
#undef M
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = ld0 + ld1 + ld2 + ld3 + (float)sacc;
#pragma unroll
    for (int i = 0; i < 32; i++) acc += r[i] + (float)q[i];
#pragma unroll
    for (int i = 0; i < 16; i++) acc += p[i].x + p[i].y;
    if ((threadIdx.x & 63) == 0) span[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int PAT>
static double run(int blocks, float* out, unsigned long long* span, std::vector<unsigned long long>& hs) {
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, out, span, 1.0f, 1.0000001f);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(hs.data(), span, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < blocks * 4; i++) s += (double)hs[i];
    return s / (blocks * 4);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int maxw = 6;
    float* out;
    unsigned long long* span;
    CHECK(hipMalloc(&out, (size_t)cus * maxw * 256 * sizeof(float)));
    CHECK(hipMalloc(&span, (size_t)cus * maxw * 4 * sizeof(unsigned long long)));
    std::vector<unsigned long long> hs((size_t)cus * maxw * 4);
    typedef double (*Fn)(int, float*, unsigned long long*, std::vector<unsigned long long>&);
    const Fn fns[NPAT] = {run<P_MUL>, run<P_SUB>, run<P_MAX>, run<P_AND>, run<P_LSHL>, run<P_FMAC>, run<P_CMP>, run<P_CNDMASK_S>,
                          run<P_MUL_SGPR>, run<P_MUL_LIT>, run<P_CHAIN>, run<P_MUL_PK>, run<P_MUL_CVT>, run<P_MUL_FMA>, run<P_MUL_DPP>,
                          run<P_MUL3_CVT1>, run<P_MUL_LDS>, run<P_MUL_SALU>, run<P_ADD_MUL>};
    printf("{\"device\": \"%s\", \"cus\": %d, \"reps\": %d, \"unit\": \"s_memtime units per VALU instruction and SIMD (all waves of a SIMD resident together)\", \"results\": [\n",
           prop.gcnArchName, cus, REPS);
    bool first = true;
    for (int w : {1, 2, 4, 6})
        for (int pat = 0; pat < NPAT; pat++) {
            const double wave_span = fns[pat](cus * w, out, span, hs);
            const int valu_per_it = pat == P_MUL_LDS ? 28 : (pat == P_MUL_SALU ? 16 : 32);
            printf("%s{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"units_per_valu_inst_per_simd\": %.3f}", first ? "" : ",\n", kNames[pat], w,
                   wave_span / ((double)REPS * valu_per_it * w));
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
