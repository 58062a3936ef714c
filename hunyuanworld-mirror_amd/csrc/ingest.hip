// Image ingest after decode (reference: src/utils/inference_utils.py:67-108): uint8 RGB ->
// Pillow-exact bicubic resize -> /255 -> centre crop / white pad -> planar float32 [3][H'][W'].
//
// The resize is Pillow's 8-bit resample (libImaging/Resample.c, the call at inference_utils.py:86): two separable
// passes, horizontal first with a uint8 intermediate, per-output-pixel windows of normalised Keys-bicubic weights
// quantised to 22 fractional bits.  The integer weight tables are built on the host in double precision exactly as
// Pillow builds them (wm_model.cpp: wm_resample_coeffs); the kernels only do integer multiply-adds, so the result is
// bit-identical to Pillow's.  HBM-bound byte work: one thread per output pixel, 3 channels each.
#include "wm_common.h"
#include "wm_kernels.h"

namespace {

constexpr int PREC = 22;

// in [H][Wi][3] u8 -> out [H][Wo][3] u8
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H,
                                                         int Wi, int Wo, const int* __restrict__ bounds, const int* __restrict__ kk,
                                                         int ksize) {
  const size_t total = (size_t)H * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int y = (int)(i / Wo), xo = (int)(i - (size_t)y * Wo);
    const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
    const int* k = kk + (size_t)xo * ksize;
    const unsigned char* src = in + ((size_t)y * Wi + x0) * 3;
    int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      s0 += src[3 * x] * w; s1 += src[3 * x + 1] * w; s2 += src[3 * x + 2] * w;
    }
    unsigned char* o = out + i * 3;
    s0 >>= PREC; s1 >>= PREC; s2 >>= PREC;
    o[0] = (unsigned char)(s0 < 0 ? 0 : s0 > 255 ? 255 : s0);
    o[1] = (unsigned char)(s1 < 0 ? 0 : s1 > 255 ? 255 : s1);
    o[2] = (unsigned char)(s2 < 0 ? 0 : s2 > 255 ? 255 : s2);
  }
}

// tmp [Hi][W][3] u8 -> out planar float [3][Hf][Wf]; output pixel (yf, xf) is resized pixel (yf + ry0, xf + rx0) when that
// lies inside [0, Ho) x [0, W), else the white pad value 1.0; vertical weights `kk` (all ones shortcut when Hi == Ho).
__global__ __launch_bounds__(256) void resample_v_tensor_kernel(const unsigned char* __restrict__ tmp, float* __restrict__ out, int Hi,
                                                                int W, int Ho, int Hf, int Wf, int ry0, int rx0,
                                                                const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const size_t total = (size_t)Hf * Wf, plane = total;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int yf = (int)(i / Wf), xf = (int)(i - (size_t)yf * Wf);
    const int yr = yf + ry0, xr = xf + rx0;
    float v0 = 1.0f, v1 = 1.0f, v2 = 1.0f;
    if (yr >= 0 && yr < Ho && xr >= 0 && xr < W) {
      int s0, s1, s2;
      if (bounds) {
        const int y0 = bounds[2 * yr], n = bounds[2 * yr + 1];
        const int* k = kk + (size_t)yr * ksize;
        s0 = s1 = s2 = 1 << (PREC - 1);
        for (int y = 0; y < n; ++y) {
          const unsigned char* src = tmp + ((size_t)(y0 + y) * W + xr) * 3;
          const int w = k[y];
          s0 += src[0] * w; s1 += src[1] * w; s2 += src[2] * w;
        }
        s0 >>= PREC; s1 >>= PREC; s2 >>= PREC;
        s0 = s0 < 0 ? 0 : s0 > 255 ? 255 : s0; s1 = s1 < 0 ? 0 : s1 > 255 ? 255 : s1; s2 = s2 < 0 ? 0 : s2 > 255 ? 255 : s2;
      } else {  // no vertical resize
        const unsigned char* src = tmp + ((size_t)yr * W + xr) * 3;
        s0 = src[0]; s1 = src[1]; s2 = src[2];
      }
      v0 = __fdiv_rn((float)s0, 255.0f); v1 = __fdiv_rn((float)s1, 255.0f); v2 = __fdiv_rn((float)s2, 255.0f);  // ToTensor
    }
    out[i] = v0; out[plane + i] = v1; out[2 * plane + i] = v2;
  }
}

inline unsigned grid_for_px(size_t n) {
  size_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 65535 * 16 ? 65535 * 16 : b);
}

}  // namespace

hipError_t wm_launch_resample_h(const unsigned char* in, unsigned char* out, int H, int Wi, int Wo, const int* bounds, const int* kk,
                                int ksize, hipStream_t s) {
  if ((size_t)H * Wo == 0) return hipSuccess;
  hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for_px((size_t)H * Wo)), dim3(256), 0, s, in, out, H, Wi, Wo, bounds, kk, ksize);
  return hipGetLastError();
}

hipError_t wm_launch_resample_v_tensor(const unsigned char* tmp, float* out, int Hi, int W, int Ho, int Hf, int Wf, int ry0, int rx0,
                                       const int* bounds, const int* kk, int ksize, hipStream_t s) {
  if ((size_t)Hf * Wf == 0) return hipSuccess;
  hipLaunchKernelGGL(resample_v_tensor_kernel, dim3(grid_for_px((size_t)Hf * Wf)), dim3(256), 0, s, tmp, out, Hi, W, Ho, Hf, Wf, ry0, rx0,
                     bounds, kk, ksize);
  return hipGetLastError();
}
