// 3x3 / stride 1 / pad 1 convolution with 32 output channels on a 16-BIT NHWC input — the last 3x3 conv of a DPT head
// (output_conv2[0], 128 -> 32 at full resolution, reference: src/models/heads/dense_head.py:97-105,217-251), fed by
// bilinear16_kernel (elementwise.hip), which writes interpolate(feat, (H, W), align_corners) + 0.1 pos already rounded to the
// conv's operand type.  Rounding happens at the same point as in the fused-resize kernel (conv3x3.hip: interpolate in fp32,
// round to 16 bits, multiply), so the results agree with it to the rounding of the interpolation.
//
// Why a separate pass: with the resize fused into the halo staging every halo item is a 4-corner gather (10 x 16-B loads into
// registers, ~7 GB of L2 -> CU traffic per launch for 0.36 GB of source at 8 views) and the kernel waits on those gathers (PMC:
// 46 % of the wave cycles parked, matrix pipes 12 % busy, 571 us).  Here the halo is one 16-B piece per (pixel, 8 channels),
// so halo AND weights arrive by LDS-DMA with no staging registers, a whole 64-channel chunk ahead, while the previous
// chunk's 36 MFMA steps run; blocks are persistent, so the next tile's first chunk is requested during this tile's last.
//
// Block = 8 waves = 16 x 16 output pixels x 32 channels (one 32 x 32 MFMA tile per wave).  LDS: the weights of ALL chunks stay
// resident (Cin <= 128: 2 x 36 KiB, loaded once per block) + 2 x 41 KiB of halo = 154 KiB; only the halo streams.  Ordering without any assumption about loads vs stores: a tile's output stores are issued one step
// late, BEFORE the next DMA batch, so the only wait is a full drain at the top of a step, by which time everything in flight
// has had a whole MFMA phase to complete.
#include "wm_common.h"
#include "wm_kernels.h"

namespace {

__device__ __forceinline__ void dma16(const void* g, uint32_t lds_dst) {  // see attention.hip: DMA hidden from hipcc's waitcnt pass
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

constexpr int TP = 16, HWD = TP + 2, HROWS = HWD * HWD;  // 18 x 18 halo
constexpr int HP = (HROWS * 8 + 63) / 64;                // 41 halo pieces of 1 KiB (64 x 16 B)
constexpr int HB = HP * 1024, WB = 36 * 1024;
constexpr int MAXCH = 2;                                 // resident weight chunks (Cin <= 128)
constexpr int PPW = (HP + 7) / 8;                        // halo DMA pieces per wave per chunk (5 or 6)

template <int T>
__global__ __launch_bounds__(512) void conv3x3_n32_in16_kernel(const WmConvN32Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [nchunks][WB] weights | [2][HB] halo
  typedef __attribute__((address_space(3))) void* lds_vp;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_vp)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W, Cin = p.Cin;
  const int tiles_x = (W + TP - 1) / TP, tiles_y = (H + TP - 1) / TP;
  const int ntiles = p.N * tiles_y * tiles_x, nchunks = Cin / 64, K = 9 * Cin;
  const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nsteps = my_tiles * nchunks;
  if (nsteps == 0) return;

  auto tile_origin = [&](int j, int& n, int& y0, int& x0) {
    int t = (int)blockIdx.x + j * (int)gridDim.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    n = t / tiles_y; y0 = ty * TP; x0 = tx * TP;
  };
  const uint32_t halo0 = lds0 + nchunks * WB;
  // the 9 weight tiles [tap][32 cout][8 x 16 B] of every chunk: loaded once (row-pair XOR swizzle built on the SOURCE side:
  // LDS-DMA writes lane-linear)
  for (int wp = wave; wp < nchunks * 36; wp += 8) {
    const int cc = wp / 36, q = wp - cc * 36, tap = q >> 2, r = (q & 3) * 8 + (lane >> 3), c = (lane & 7) ^ swz(r);
    dma16(p.w + (size_t)r * K + tap * Cin + cc * 64 + c * 8, lds0 + wp * 1024);
  }
  // one chunk (64 input channels) of one tile: halo image [hr][8 x 16 B], swizzled the same way
  auto stage = [&](int s, int buf) {
    const int j = s / nchunks, cc = s - j * nchunks;
    int n, y0, x0;
    tile_origin(j, n, y0, x0);
    const uint32_t base = halo0 + buf * HB;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 8 * i;  // wave-uniform
      if (pc >= HP) break;
      const u16* src = p.zero;
      const int q = pc * 64 + lane, hr = q >> 3, ch = (q & 7) ^ swz(hr);
      const int hy = hr / HWD, hx = hr - hy * HWD;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (hr < HROWS && iy >= 0 && iy < H && ix >= 0 && ix < W) src = p.x + (((size_t)n * H + iy) * W + ix) * Cin + cc * 64 + ch * 8;
      dma16(src, base + pc * 1024);
    }
  };

  const int r = wave * 32 + (lane & 31);                 // this lane's pixel of the tile
  const int hbase = (r >> 4) * HWD + (r & 15);
  const int h4 = (lane >> 5) * 4;
  float4 bs[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bs[g] = p.bias ? *(const float4*)(p.bias + 8 * g + h4) : make_float4(0, 0, 0, 0);

  f32x16 acc, outv;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; outv[e] = 0.f; }
  int out_tile = -1;  // block-uniform: tile whose finished accumulators wait in outv
  // optional fused tail (dense_head.py:97-105,297-344): ReLU -> 1x1 conv 32 -> C -> split (attr C - 1, conf) -> activations; a lane
  // holds 16 of its pixel's 32 channels (the other 16 sit in lane ^ 32), so the 1x1 conv is a 16-term dot product + one cross-half sum
  const bool tail = p.tail_w != nullptr;
  float tw[4][16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) tw[c][4 * g + e] = (tail && c < p.tail_C) ? p.tail_w[c * 32 + 8 * g + h4 + e] : 0.f;
  float tres[4] = {0.f, 0.f, 0.f, 0.f};  // fused tail: up to 3 attributes + confidence of this lane's pixel
  auto flush = [&]() {
    int n, y0, x0;
    tile_origin(out_tile, n, y0, x0);
    const int y = y0 + (r >> 4), x = x0 + (r & 15);
    if (tail) {  // block-uniform: the values were finished when the tile completed; only the stores are left
      if (y < H && x < W && (lane >> 5) == 0) {
        const size_t i = ((size_t)n * H + y) * W + x;
        const int A = p.tail_C - 1;
        for (int c = 0; c < A; ++c) p.tail_attr[i * A + c] = tres[c];
        p.tail_conf[i] = tres[3];
      }
      out_tile = -1;
      return;
    }
    if (y < H && x < W) {
      float* o = p.y + (((size_t)n * H + y) * W + x) * 32 + h4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(outv[4 * g] + bs[g].x, outv[4 * g + 1] + bs[g].y, outv[4 * g + 2] + bs[g].z, outv[4 * g + 3] + bs[g].w);
        if (p.relu_out) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *(float4*)(o + 8 * g) = v;
      }
    }
    out_tile = -1;
  };

  stage(0, 0);
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk s landed (and any older stores completed)
    __syncthreads();  // every wave's pieces of chunk s landed; every wave finished reading the other buffer (chunk s - 1)
    if (out_tile >= 0) flush();                  // stores first, so that they are OLDER than the DMA batch issued next
    if (s + 1 < nsteps) stage(s + 1, buf ^ 1);
    const char* hb = smem + nchunks * WB + buf * HB;
    const char* wb = smem + (s % nchunks) * WB;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hr = hbase + (tap / 3) * HWD + (tap % 3);
      const int row = tap * 32 + (lane & 31);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 2 * ks + (lane >> 5);
        const s16x8 a = *(const s16x8*)(hb + hr * 128 + ((ch ^ swz(hr)) << 4));
        const s16x8 b = *(const s16x8*)(wb + row * 128 + ((ch ^ swz(row & 31)) << 4));
        acc = mfma32<T>(b, a, acc);  // D[cout][pixel]: lane = pixel, regs 4g..4g+3 <-> channels 8g + 4h + {0..3}
      }
    }
    const int j = s / nchunks;
    if (s - j * nchunks == nchunks - 1) {  // last chunk of the tile: hand the results to the deferred store
      if (tail) {  // finish the head here, under the DMA flight of the next chunk: ReLU, 1x1 conv, activations
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bias_e = e == 0 ? bs[g].x : e == 1 ? bs[g].y : e == 2 ? bs[g].z : bs[g].w;
            const float v = fmaxf(acc[4 * g + e] + bias_e, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] += v * tw[c][4 * g + e];
          }
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = xhalf_sum(o[c]) + (c < p.tail_C ? p.tail_b[c] : 0.f);  // all lanes take part
        const int A = p.tail_C - 1;
        if (p.tail_act == WM_ACT_NORM) {
          float nn = 0.f;
          for (int c = 0; c < A; ++c) nn += o[c] * o[c];
          nn = sqrtf(nn);
          for (int c = 0; c < A; ++c) tres[c] = o[c] / nn;
        } else if (p.tail_act == WM_ACT_EXP) {
          for (int c = 0; c < A; ++c) tres[c] = expf(o[c]);
        } else {
          for (int c = 0; c < A; ++c) {
            const float e = expm1f(fabsf(o[c]));
            tres[c] = o[c] > 0.f ? e : (o[c] < 0.f ? -e : 0.f);
          }
        }
        tres[3] = 1.0f + expf(o[A]);
      } else {
        outv = acc;
      }
      out_tile = j;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    }
  }
  if (out_tile >= 0) flush();
}

}  // namespace

hipError_t wm_launch_conv3x3_n32_in16(const WmConvN32Args& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  if (a.Cin % 64 || a.Cin <= 0 || a.Cin > 64 * MAXCH || !a.x || !a.w || !a.zero) return hipErrorInvalidValue;
  if (a.tail_w ? (!a.tail_b || !a.tail_attr || !a.tail_conf || a.tail_C < 2 || a.tail_C > 4) : !a.y) return hipErrorInvalidValue;
  static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
  const int ntiles = a.N * ((a.H + TP - 1) / TP) * ((a.W + TP - 1) / TP);
  const int grid = ntiles < ncu ? ntiles : ncu;
  const size_t shm = (size_t)(a.Cin / 64) * WB + 2 * HB;
  static bool attr[2] = {false, false};
  if (a.dtype == WM_T_BF16) {
    if (!attr[0]) { (void)hipFuncSetAttribute((const void*)conv3x3_n32_in16_kernel<WM_T_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr[0] = true; }
    hipLaunchKernelGGL((conv3x3_n32_in16_kernel<WM_T_BF16>), dim3(grid), dim3(512), shm, s, a);
  } else {
    if (!attr[1]) { (void)hipFuncSetAttribute((const void*)conv3x3_n32_in16_kernel<WM_T_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr[1] = true; }
    hipLaunchKernelGGL((conv3x3_n32_in16_kernel<WM_T_F16>), dim3(grid), dim3(512), shm, s, a);
  }
  return hipGetLastError();
}
