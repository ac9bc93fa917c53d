// l2_req_calib.hip -- how many bytes is one TCC_REQ?  A read-once stream of known size: every lane loads 16 B, a wave 1 KiB of
// consecutive bytes, nothing is read twice, so (bytes read) / TCC_REQ_sum is the size of an L2 request as the counter
// tallies it for wide vector loads (the ICGN2D table gather uses the same 16-byte-per-lane loads).  Run under
//   rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex stream_read -- ./l2_req_calib
// (tools/gpu_traffic.sh); prints the byte count of one launch.
// Build: hipcc --offload-arch=gfx950 -O3 l2_req_calib.hip -o l2_req_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ in, float* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (; i < n4; i += step) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;  // keeps the loads alive without a write stream
}

int main() {
    const size_t bytes = 1ull << 30, n4 = bytes / 16;
    float4* in = nullptr;
    float* out = nullptr;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(in, 0, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, in, out, n4);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("launch %d: %zu bytes read once, %.3f ms, %.2f TB/s\n", rep, bytes, ms, bytes / ms / 1e9);
    }
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
