// Shader clock under load: every CU runs an MFMA+VALU loop; clock64() (s_memtime, shader cycles) against wall_clock64() (100 MHz).
// hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe ; ./tools/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ __launch_bounds__(256) void spin(unsigned long long* out, int iters, int mode) {
    f32x16_t acc0 = {0}, acc1 = {0};
    bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, (short)threadIdx.x};
    float x = threadIdx.x * 1e-3f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (mode & 1) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0); }
        if (mode & 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) x = fmaf(x, 1.0001f, 0.5f);
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    if (acc0[0] + acc1[3] + x == 12345.678f) out[0] = 0;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 2048 * 16);
    unsigned long long h[4];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, d, 200000, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, d + 2 * 512, 32, hipMemcpyDeviceToHost);
        printf("mode %d (1 = MFMA, 2 = VALU, 3 = both): shader cycles %llu, 100 MHz ticks %llu -> %.3f GHz\n", mode, h[0], h[1], (double)h[0] / h[1] * 0.1);
    }
    return 0;
}
