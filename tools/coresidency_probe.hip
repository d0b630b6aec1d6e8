// Co-residency probe (round 6; no product code): an HBM-streaming kernel held to 64 VGPRs and no LDS, to be run on a second HIP stream
// while gemm_w4 (192 VGPR + 256 AGPR = 448 of a SIMD lane's 512 registers, 128 KB of LDS, one wave per SIMD) loops on the first.
// Built by tools/coresidency_probe.py: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// (256, 8): eight waves per SIMD must fit -> at most 64 VGPRs per lane
extern "C" __global__ __launch_bounds__(256, 8) void stream_probe_kernel(const u32x4* __restrict__ src, size_t n16, int passes, unsigned int* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int p = 0; p < passes; ++p) {
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {          // four 16-byte non-temporal loads in flight per lane
            const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
            const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
            acc ^= a ^ b ^ c ^ d;
        }
        for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) out[0] = 1u;      // keeps the loads alive; practically never taken
}

extern "C" int stream_probe_launch(const void* src, size_t bytes, int passes, int grid, void* out, void* stream) {
    hipLaunchKernelGGL(stream_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, bytes / 16, passes, (unsigned int*)out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
