#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) __bf16 b4;
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x;
    const int i = lane & 15, g = lane >> 4;
    // 16-lane group g reads the 4x16 block at rows g*4.. of a [rows][16] image (32-byte rows)
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + (g * 4 + (i >> 2)) * 16 + (i & 3) * 4));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
}
