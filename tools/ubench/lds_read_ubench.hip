// lds_read_ubench.hip -- what does a wave64 LDS read of 2 / 4 consecutive dwords cost on gfx950 when the address is only
// 4-byte aligned (the x-taps of the tricubic evaluation in icgn3d.hip start at an arbitrary float)?
//
// One 256-thread workgroup per CU (so the LDS pipe of a CU is loaded by 4 waves, one per SIMD) runs REPS x 16 reads of
// one kind; lane l reads at  base + 4 * (l * STRIDE + MIS)  bytes, i.e. consecutive lanes STRIDE dwords apart, the whole
// pattern shifted by MIS dwords off a 16-byte boundary.  Reported: LDS cycles per wave-instruction per CU
// (s_memtime span x CUs / instructions issued) and whether the values read are the ones a dword-wise read returns.
// Kinds: ds_read2_b32 (offset1 = offset0 + 1), ds_read_b64, ds_read2_b32 x2 vs ds_read_b128 for four dwords.
// Build: hipcc --offload-arch=gfx950 -O3 lds_read_ubench.hip -o lds_read_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 4096;
enum Kind { READ2_B32, READ_B64, READ2X2_B32, READ_B128, NKIND };
static const char* kNames[NKIND] = {"ds_read2_b32", "ds_read_b64", "2 x ds_read2_b32", "ds_read_b128"};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, unsigned long long* __restrict__ span, int stride, int mis) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // waves read disjoint 4096-float quarters
    const unsigned addr = (unsigned)((wave * 4096 + lane * stride + mis) * 4);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma nounroll
    for (int it = 0; it < REPS; it++) {
        // 16 (or 32) reads in flight, then ONE wait that names every destination (so no use is scheduled before it)
        f2 v[16], w[16];
        f4 q[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if constexpr (KIND == READ2_B32) {
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[u]) : "v"(addr), "n"(u * 8), "n"(u * 8 + 1));
            } else if constexpr (KIND == READ_B64) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[u]) : "v"(addr), "n"(u * 32));
            } else if constexpr (KIND == READ2X2_B32) {
                asm volatile("ds_read2_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2_b32 %1, %2 offset0:%5 offset1:%6"
                             : "=v"(v[u]), "=v"(w[u]) : "v"(addr), "n"(u * 8), "n"(u * 8 + 1), "n"(u * 8 + 2), "n"(u * 8 + 3));
            } else {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[u]) : "v"(addr), "n"(u * 32));
            }
        }
        if constexpr (KIND == READ_B128) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]),
                           "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]));
#pragma unroll
            for (int u = 0; u < 16; u++) acc += q[u];
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                           "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
            if constexpr (KIND == READ2X2_B32) {
                asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]),
                             "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                acc.x += v[u].x; acc.y += v[u].y;
                if constexpr (KIND == READ2X2_B32) { acc.z += w[u].x; acc.w += w[u].y; }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) span[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// the same sum formed on the host from what a dword-wise read returns (lds[i] = i)
static double expect(int kind, int stride, int mis, int lane, int wave) {
    const int n = (kind == READ2_B32 || kind == READ_B64) ? 2 : 4;
    double s = 0;
    for (int u = 0; u < 16; u++)
        for (int d = 0; d < n; d++) s += (double)(wave * 4096 + lane * stride + mis + u * 8 + d);
    return s * REPS;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out;
    unsigned long long* span;
    CHECK(hipMalloc(&out, (size_t)cus * 256 * sizeof(float)));
    CHECK(hipMalloc(&span, (size_t)cus * 4 * sizeof(unsigned long long)));
    std::vector<float> h((size_t)cus * 256);
    std::vector<unsigned long long> hs((size_t)cus * 4);
    printf("{\"device\": \"%s\", \"cus\": %d, \"reps\": %d, \"note\": \"cycles = s_memtime units per wave-instruction per CU, 4 waves per CU\", \"results\": [\n",
           prop.gcnArchName, cus, REPS);
    bool first = true;
    for (int kind = 0; kind < NKIND; kind++)
        for (int stride : {1, 2, 4})
            for (int mis = 0; mis < 4; mis++) {
                for (int rep = 0; rep < 2; rep++) {
                    switch (kind) {
                        case READ2_B32: hipLaunchKernelGGL(k<READ2_B32>, dim3(cus), dim3(256), 0, 0, out, span, stride, mis); break;
                        case READ_B64: hipLaunchKernelGGL(k<READ_B64>, dim3(cus), dim3(256), 0, 0, out, span, stride, mis); break;
                        case READ2X2_B32: hipLaunchKernelGGL(k<READ2X2_B32>, dim3(cus), dim3(256), 0, 0, out, span, stride, mis); break;
                        default: hipLaunchKernelGGL(k<READ_B128>, dim3(cus), dim3(256), 0, 0, out, span, stride, mis); break;
                    }
                    CHECK(hipDeviceSynchronize());
                }
                CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(hs.data(), span, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double cyc = 0;
                for (auto v : hs) cyc += (double)v;
                cyc /= hs.size();  // mean span of a wave
                const int per_it = (kind == READ2X2_B32) ? 32 : 16;
                // 4 waves share the CU's LDS: instructions issued per CU in that span = 4 * REPS * per_it
                const double per_inst_cu = cyc / ((double)REPS * per_it * 4);
                bool ok = true;
                for (int t = 0; t < 256 && ok; t++) {
                    const double e = expect(kind, stride, mis, t & 63, t >> 6);
                    ok = std::abs((double)h[t] - e) <= 1e-3 * e;
                }
                printf("%s{\"inst\": \"%s\", \"lane_stride_dwords\": %d, \"misaligned_dwords\": %d, \"wave_span_units\": %.0f, \"units_per_inst_per_cu\": %.3f, \"values_ok\": %s}",
                       first ? "" : ",\n", kNames[kind], stride, mis, cyc, per_inst_cu, ok ? "true" : "false");
                first = false;
            }
    printf("\n]}\n");
    return 0;
}
