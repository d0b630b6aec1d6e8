// On-device frame preprocessing: what process_video does on the host per frame (trace/mm_utils.py:456-462 ->
// expand2square :259-270 -> HF CLIPImageProcessor.preprocess: Pillow BICUBIC resize, centre crop, x/255, (x-mean)/std),
// for uint8 RGB frames already in HBM.  The resize is Pillow's 8-bit resampler integer for integer (src/libImaging/
// Resample.c): a horizontal pass over the source rows the vertical pass needs, uint8 in between, a vertical pass, each
// a 22-bit fixed-point dot product with round-half-up and clamp.  The tap tables come from the host (engine.hip, double
// arithmetic in Pillow's operation order); the rescale + normalise of an 8-bit value is a 3 x 256 table built on the
// host with the reference's float64 -> float32 -> float32 arithmetic, so the output is bit-identical by construction.
// HBM-bound and tiny next to the ViT (128 frames of 720p: 354 MB in, 87 MB out).
#include "common.h"
#include "kernels.h"

namespace {

// virtual (padded) source pixel: rows / columns outside the pasted frame read the background colour
__device__ __forceinline__ int src_px(const uint8_t* __restrict__ f, int H, int W, int y, int x, int c, int y0, int x0, int bg) {
    const int yy = y - y0, xx = x - x0;
    return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (int)f[((size_t)yy * W + xx) * 3 + c] : bg;
}

// tmp[t][r][j][c] = horizontal resample of padded-source row (row_first + r) at output column (crop-left + j)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ frames, int H, int W, int y0, int x0, uint32_t bg,
                                                       const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                       int row_first, int nrows, int S, uint8_t* __restrict__ tmp) {
    const int t = blockIdx.y, r = blockIdx.x;
    const uint8_t* f = frames + (size_t)t * H * W * 3;
    const int y = row_first + r;
    for (int idx = threadIdx.x; idx < S * 3; idx += 256) {
        const int j = idx / 3, c = idx - j * 3;
        const int xmin = bounds[2 * j], n = bounds[2 * j + 1];
        const int32_t* k = kk + (size_t)j * ksize;
        const int bgc = (bg >> (8 * c)) & 255;
        int acc = 1 << 21;
        for (int i = 0; i < n; ++i) acc += src_px(f, H, W, y, xmin + i, c, y0, x0, bgc) * k[i];
        acc >>= 22;
        tmp[(((size_t)t * nrows + r) * S + j) * 3 + c] = (uint8_t)min(max(acc, 0), 255);
    }
}

// out[t][c][i][j] = lut[c][ vertical resample of tmp at output row (crop-top + i) ]
template <bool F32>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int nrows, int S,
                                                            const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                            int ksize, const float* __restrict__ lut, void* __restrict__ out) {
    const int t = blockIdx.y, i = blockIdx.x;
    const int ymin = bounds[2 * i], n = bounds[2 * i + 1];          // relative to the first staged row
    const int32_t* k = kk + (size_t)i * ksize;
    const uint8_t* base = tmp + ((size_t)t * nrows + ymin) * S * 3;
    for (int idx = threadIdx.x; idx < S * 3; idx += 256) {
        const int j = idx / 3, c = idx - j * 3;
        int acc = 1 << 21;
        for (int q = 0; q < n; ++q) acc += (int)base[((size_t)q * S + j) * 3 + c] * k[q];
        acc = min(max(acc >> 22, 0), 255);
        const float v = lut[c * 256 + acc];
        const size_t o = (((size_t)t * 3 + c) * S + i) * S + j;
        if (F32) reinterpret_cast<float*>(out)[o] = v;
        else reinterpret_cast<bf16_t*>(out)[o] = f2bf(v);
    }
}
}  // namespace

int launch_resize_h(const uint8_t* frames, int T, int H, int W, int y0, int x0, uint32_t bg, const int32_t* bounds,
                    const int32_t* kk, int ksize, int row_first, int nrows, int S, uint8_t* tmp, hipStream_t s) {
    if (T < 1 || nrows < 1 || S < 1) return TRACE_ERR_ARG;
    hipLaunchKernelGGL(resize_h_kernel, dim3(nrows, T), dim3(256), 0, s, frames, H, W, y0, x0, bg, bounds, kk, ksize, row_first,
                       nrows, S, tmp);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}

int launch_resize_v_norm(const uint8_t* tmp, int T, int nrows, int S, const int32_t* bounds, const int32_t* kk, int ksize,
                         const float* lut, void* out, int out_f32, hipStream_t s) {
    if (T < 1 || nrows < 1 || S < 1) return TRACE_ERR_ARG;
    if (out_f32) hipLaunchKernelGGL(resize_v_norm_kernel<true>, dim3(S, T), dim3(256), 0, s, tmp, nrows, S, bounds, kk, ksize, lut, out);
    else hipLaunchKernelGGL(resize_v_norm_kernel<false>, dim3(S, T), dim3(256), 0, s, tmp, nrows, S, bounds, kk, ksize, lut, out);
    return hipGetLastError() == hipSuccess ? TRACE_OK : TRACE_ERR_HIP;
}
