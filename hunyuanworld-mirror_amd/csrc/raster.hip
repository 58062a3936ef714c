// 3D-Gaussian-splat rasteriser forward — the call the reference makes through
// Rasterizer.rasterize_splats (src/models/models/rasterization.py:29-66): gsplat.rasterization(packed, "classic",
// pinhole, render_mode "RGB+ED", sh_degree 0 | None).  Stages (gsplat/rendering.py:853-992):
//   project    quat/scale -> covariance, world -> camera, perspective EWA projection, + 0.3 blur, conic, 3.33-sigma
//              radii, near/far + screen culling, tile rectangle        (_torch_impl.py:45-61,78-133,250-375)
//   scan       exclusive sum of tiles-per-Gaussian                     (hipCUB)
//   emit       one (key, value) pair per touched tile: key = ((camera << tile_bits | tile) << 32) | depth bits
//   sort       64-bit radix sort of the pairs                          (hipCUB; _torch_impl.py:378-474)
//   offsets    first pair of every (camera, tile)                      (_torch_impl.py:477-503)
//   composite  front-to-back alpha blending per 16 x 16 tile, colour + depth channel, expected-depth normalisation
//              (csrc/RasterizeToPixels3DGSFwd.cu:118-184, rendering.py:984-992)
// All of it is HBM / VALU-bound integer and fp32 work (no MFMA).  Compositing is written for 64-wide waves: a wave owns an
// 8 x 8-pixel quadrant of a tile and walks the tile's depth-sorted list by itself — the Gaussian of a step is wave-uniform,
// so its 48-byte record (written once per (camera, Gaussian) by the projection) arrives through SCALAR loads and its fields
// are SGPR operands of the per-pixel arithmetic: no LDS staging, no workgroup barrier, no shared early-out.  A step is
// branch-free per lane (the reference's skip / stop / blend decisions are three lane masks); the two branches are wave-uniform:
// skip the blend when no pixel of the quadrant is hit, leave the list when all 64 are saturated.
#include "wm_common.h"
#include "wm_kernels.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int TILE = 16;
constexpr float SH_C0 = 0.28209479177387814f;
constexpr float ALPHA_THRESHOLD = 1.0f / 255.0f;

struct __attribute__((aligned(16))) G2D {  // per (camera, Gaussian): 48 B = three 16-byte scalar loads of the compositing pass
  float mx, my;         // pixel-space mean
  float ca, cb;         // conic
  float cc, opacity;
  float depth;
  float r, g, b;        // colour (view-independent: degree-0 SH or given colours)
  int rect;             // x0 | y0 << 8 | x1 << 16 | y1 << 24 in tiles (tile grids up to 255 x 255); the compositing pass reads words 0-9 only
  int pad;
};
static_assert(sizeof(G2D) == 48, "G2D is read as three dwordx4");

__global__ __launch_bounds__(256) void raster_project_kernel(const float* __restrict__ means, const float* __restrict__ quats,
                                                             const float* __restrict__ scales, const float* __restrict__ viewmats,
                                                             const float* __restrict__ Ks, int N, int C, int width, int height,
                                                             float near_plane, float far_plane, const float* __restrict__ opac,
                                                             const float4* __restrict__ rgb, G2D* __restrict__ g2d,
                                                             unsigned long long* __restrict__ counts, int* __restrict__ radii_out) {
  const int g = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (g >= N) return;
  // covariance = (R S)(R S)^T, quats wxyz normalised (_torch_impl.py:11-29,45-61)
  float qw = quats[4 * g], qx = quats[4 * g + 1], qy = quats[4 * g + 2], qz = quats[4 * g + 3];
  const float qn = fmaxf(sqrtf(qw * qw + qx * qx + qy * qy + qz * qz), 1e-12f);
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  const float Rm[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                       2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                       2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
  const float s[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
  float M[9], cov[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[3 * i + j] = Rm[3 * i + j] * s[j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) cov[3 * i + k] = M[3 * i] * M[3 * k] + M[3 * i + 1] * M[3 * k + 1] + M[3 * i + 2] * M[3 * k + 2];
  // world -> camera (_torch_impl.py:250-283)
  const float* V = viewmats + 16 * c;
  const float Rv[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
  const float m[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
  const float tx = Rv[0] * m[0] + Rv[1] * m[1] + Rv[2] * m[2] + V[3];
  const float ty = Rv[3] * m[0] + Rv[4] * m[1] + Rv[5] * m[2] + V[7];
  const float tz = Rv[6] * m[0] + Rv[7] * m[1] + Rv[8] * m[2] + V[11];
  float RC[9], cc[9];  // R cov, then (R cov) R^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) RC[3 * i + k] = Rv[3 * i] * cov[k] + Rv[3 * i + 1] * cov[3 + k] + Rv[3 * i + 2] * cov[6 + k];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int l = 0; l < 3; ++l) cc[3 * i + l] = RC[3 * i] * Rv[3 * l] + RC[3 * i + 1] * Rv[3 * l + 1] + RC[3 * i + 2] * Rv[3 * l + 2];
  // perspective projection (_torch_impl.py:78-133)
  const float* K = Ks + 9 * c;
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const float tfx = 0.5f * width / fx, tfy = 0.5f * height / fy;
  const float lxp = (width - cx) / fx + 0.3f * tfx, lxn = cx / fx + 0.3f * tfx;
  const float lyp = (height - cy) / fy + 0.3f * tfy, lyn = cy / fy + 0.3f * tfy;
  const float txc = tz * fminf(fmaxf(tx / tz, -lxn), lxp), tyc = tz * fminf(fmaxf(ty / tz, -lyn), lyp);
  const float tz2 = tz * tz;
  const float J0 = fx / tz, J2 = -fx * txc / tz2, J4 = fy / tz, J5 = -fy * tyc / tz2;  // J = [[J0 0 J2] [0 J4 J5]]
  // cov2d = J cc J^T
  const float a0 = J0 * cc[0] + J2 * cc[6], a1 = J0 * cc[1] + J2 * cc[7], a2 = J0 * cc[2] + J2 * cc[8];
  const float b0 = J4 * cc[3] + J5 * cc[6], b1 = J4 * cc[4] + J5 * cc[7], b2 = J4 * cc[5] + J5 * cc[8];
  float c00 = a0 * J0 + a2 * J2, c01 = a1 * J4 + a2 * J5, c10 = b0 * J0 + b2 * J2, c11 = b1 * J4 + b2 * J5;
  const float mx = (K[0] * tx + K[1] * ty + K[2] * tz) / tz, my = (K[3] * tx + K[4] * ty + K[5] * tz) / tz;
  c00 += 0.3f; c11 += 0.3f;  // eps2d
  float det = c00 * c11 - c01 * c10;
  det = fmaxf(det, 1e-10f);
  G2D o;
  o.mx = mx; o.my = my;
  o.ca = c11 / det; o.cb = -(c01 + c10) / 2.0f / det; o.cc = c00 / det;
  o.depth = tz;
  float rx = ceilf(3.33f * sqrtf(c00)), ry = ceilf(3.33f * sqrtf(c11));
  const bool valid = det > 0.f && tz > near_plane && tz < far_plane;
  if (!valid) { rx = 0.f; ry = 0.f; }
  const bool inside = mx + rx > 0.f && mx - rx < (float)width && my + ry > 0.f && my - ry < (float)height;
  if (!inside) { rx = 0.f; ry = 0.f; }
  if (!(rx == rx) || !(ry == ry) || isinf(rx) || isinf(ry)) { rx = 0.f; ry = 0.f; }
  const int irx = (int)rx, iry = (int)ry;
  // tile rectangle (_torch_impl.py:404-415)
  const int tw = (width + TILE - 1) / TILE, th = (height + TILE - 1) / TILE;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  unsigned long long cnt = 0;
  if (irx > 0 && iry > 0) {
    const float tmx = mx / TILE, tmy = my / TILE, trx = (float)irx / TILE, try_ = (float)iry / TILE;
    x0 = min(max((int)floorf(tmx - trx), 0), tw); y0 = min(max((int)floorf(tmy - try_), 0), th);
    x1 = min(max((int)ceilf(tmx + trx), 0), tw); y1 = min(max((int)ceilf(tmy + try_), 0), th);
    cnt = (unsigned long long)((x1 - x0) * (y1 - y0));
  }
  o.rect = x0 | (y0 << 8) | (x1 << 16) | (y1 << 24);
  const float4 col = rgb[g];
  o.opacity = opac[g]; o.r = col.x; o.g = col.y; o.b = col.z;
  o.pad = 0;
  const size_t idx = (size_t)c * N + g;
  g2d[idx] = o;
  counts[idx] = cnt;
  if (radii_out) { radii_out[2 * idx] = irx; radii_out[2 * idx + 1] = iry; }
}

// colours: degree-0 SH -> clamp_min(C0 sh + 0.5, 0) (rendering.py:919-923), or the given colours as they are (sh_degree None)
__global__ __launch_bounds__(256) void raster_color_kernel(const float* __restrict__ colors_in, int N, int is_sh, float4* __restrict__ rgb) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  float r = colors_in[3 * g], gg = colors_in[3 * g + 1], b = colors_in[3 * g + 2];
  if (is_sh) { r = fmaxf(SH_C0 * r + 0.5f, 0.f); gg = fmaxf(SH_C0 * gg + 0.5f, 0.f); b = fmaxf(SH_C0 * b + 0.5f, 0.f); }
  rgb[g] = make_float4(r, gg, b, 0.f);
}

__global__ __launch_bounds__(256) void raster_emit_kernel(const G2D* __restrict__ g2d, const unsigned long long* __restrict__ offsets,
                                                          size_t CN, int N, int tw, int tile_bits, unsigned long long* __restrict__ keys,
                                                          unsigned int* __restrict__ vals) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= CN) return;
  const G2D o = g2d[idx];
  const int x0 = o.rect & 255, y0 = (o.rect >> 8) & 255, x1 = (o.rect >> 16) & 255, y1 = (o.rect >> 24) & 255;
  if (x1 <= x0 || y1 <= y0) return;
  const unsigned long long cam = idx / N;
  const unsigned long long dbits = (unsigned long long)__float_as_uint(o.depth);
  unsigned long long w = offsets[idx];
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      keys[w] = (((cam << tile_bits) | (unsigned long long)(y * tw + x)) << 32) | dbits;
      vals[w] = (unsigned int)idx;
      ++w;
    }
}

// offsets[(camera, tile)] = first sorted pair whose (camera << tile_bits | tile) is >= this one's; offsets[last + 1] = n
__global__ __launch_bounds__(256) void raster_offsets_kernel(const unsigned long long* __restrict__ keys, unsigned int n, int C, int tiles,
                                                             int tile_bits, unsigned int* __restrict__ offs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t > C * tiles) return;
  if (t == C * tiles) { offs[t] = n; return; }
  const unsigned long long want = ((unsigned long long)(t / tiles) << tile_bits) | (unsigned long long)(t % tiles);
  unsigned int lo = 0, hi = n;
  while (lo < hi) {
    const unsigned int mid = (lo + hi) >> 1;
    if ((keys[mid] >> 32) < want) lo = mid + 1; else hi = mid;
  }
  offs[t] = lo;
}

// One wave per tile region: a lane owns PX x PY neighbouring pixels, the wave 8 PX x 8 PY of them — PX = PY = 2: the wave is the
// whole 16 x 16 tile and every record is fetched once per tile; PX = PY = 1: four independent waves per tile, one per 8 x 8
// quadrant (finer skip / stop granularity, four times the scalar traffic).  Blending rules of the reference's forward
// (csrc/RasterizeToPixels3DGSFwd.cu:118-184): a Gaussian is skipped for a pixel when its exponent is negative or its alpha below
// 1/255; alpha is capped at 0.999; a pixel whose transmittance would fall to 1e-4 stops BEFORE blending that Gaussian; expected
// depth = sum(depth * weight) / alpha (rendering.py:984-992).
template <int PX, int PY>
__global__ __launch_bounds__(256 / (PX * PY)) void raster_composite_kernel(const G2D* __restrict__ g2d, const unsigned int* __restrict__ vals,
                                                                         const unsigned int* __restrict__ offs, int tw, int th, int width,
                                                                         int height, float* __restrict__ out_rgb, float* __restrict__ out_depth,
                                                                         float* __restrict__ out_alpha) {
  constexpr int NP = PX * PY;               // pixels per lane
  const int tile = blockIdx.x, cam = blockIdx.y;
  const int ty = tile / tw, tx = tile - ty * tw;
  const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // part: which 8 PX x 8 PY region of the tile
  constexpr int PARTS_X = TILE / (8 * PX);
  const int i0 = ty * TILE + (part / PARTS_X) * 8 * PY + (lane >> 3) * PY, j0 = tx * TILE + (part % PARTS_X) * 8 * PX + (lane & 7) * PX;
  float px[NP], py[NP], T[NP], r[NP], g[NP], b[NP], d[NP];
  bool open[NP];                            // the pixel still takes contributions
  bool any_open = false;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int i = i0 + q / PX, j = j0 + q % PX;
    px[q] = (float)j + 0.5f; py[q] = (float)i + 0.5f;
    open[q] = i < height && j < width;
    any_open |= open[q];
    T[q] = 1.0f; r[q] = g[q] = b[q] = d[q] = 0.f;
  }
  unsigned int k = offs[cam * tw * th + tile];
  const unsigned int end = offs[cam * tw * th + tile + 1];
  if (__builtin_amdgcn_ballot_w64(any_open) == 0ull) return;   // a region outside the image (wave-uniform)
  if (k < end) {
    // Two-deep scalar pipeline over two register sets: while step k is blended, the record of step k + 1 is in flight together with
    // the list entry its set will need next (two steps ahead).  The requests are asm statements because the compiler sinks a plain
    // load below the saturation exit, next to its use; a set is settled (s_waitcnt, tied to its registers) one step after its
    // request and before any exit, so no register of the loop is ever read, copied or left behind while a load still owns it.
    const unsigned int last = end - 1;
    typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    struct Rec { u32x8 lo; u32x2 hi; unsigned int nxt; };   // words 0-7 and 8-9 of a G2D record; nxt: the list entry this set fetches next
    auto request = [&](Rec& o, unsigned int v, unsigned int at) __attribute__((always_inline)) {
      const G2D* rp = g2d + v;
      const unsigned int* ep = vals + (at < last ? at : last);
      asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x20\n\ts_load_dword %2, %4, 0x0"
                   : "=&s"(o.lo), "=&s"(o.hi), "=&s"(o.nxt) : "s"(rp), "s"(ep));
    };
    auto settle = [&](Rec& o) __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(o.lo), "+s"(o.hi), "+s"(o.nxt)); };
    // one step; false when every pixel of the wave is saturated
    auto step = [&](const Rec& w) __attribute__((always_inline)) {
      const float mx = __uint_as_float(w.lo[0]), my = __uint_as_float(w.lo[1]), ca = __uint_as_float(w.lo[2]), cb = __uint_as_float(w.lo[3]);
      const float cc = __uint_as_float(w.lo[4]), op = __uint_as_float(w.lo[5]), depth = __uint_as_float(w.lo[6]);
      const float cr = __uint_as_float(w.lo[7]), cg = __uint_as_float(w.hi[0]), cbl = __uint_as_float(w.hi[1]);
      float alpha[NP];
      bool hit[NP], any_hit = false;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const float dx = mx - px[q], dy = my - py[q];
        const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        alpha[q] = fminf(0.999f, op * __expf(-sigma));
        hit[q] = open[q] && !(sigma < 0.f) && !(alpha[q] < ALPHA_THRESHOLD);
        any_hit |= hit[q];
      }
      if (__builtin_amdgcn_ballot_w64(any_hit) == 0ull) return true;   // nobody in the region sees this Gaussian
      bool still = false;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const float nT = T[q] * (1.0f - alpha[q]);
        const bool stop = hit[q] && nT <= 1e-4f;
        const bool blend = hit[q] && !stop;
        const float vis = blend ? alpha[q] * T[q] : 0.f;
        r[q] += cr * vis; g[q] += cg * vis; b[q] += cbl * vis; d[q] += depth * vis;
        T[q] = blend ? nT : T[q];
        open[q] = open[q] && !stop;
        still |= open[q];
      }
      return __builtin_amdgcn_ballot_w64(still) != 0ull;
    };
    Rec A, B;
    request(A, __builtin_amdgcn_readfirstlane(vals[k]), k + 2); settle(A);
    request(B, __builtin_amdgcn_readfirstlane(vals[k + 1 < last ? k + 1 : last]), k + 3);
    for (;;) {
      bool alive = step(A);
      settle(B);
      if (!alive || ++k >= end) break;
      request(A, A.nxt, k + 3);
      alive = step(B);
      settle(A);
      if (!alive || ++k >= end) break;
      request(B, B.nxt, k + 3);
    }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int i = i0 + q / PX, j = j0 + q % PX;
    if (i < height && j < width) {
      const size_t pix = ((size_t)cam * height + i) * width + j;
      const float al = 1.0f - T[q];
      out_rgb[3 * pix] = r[q]; out_rgb[3 * pix + 1] = g[q]; out_rgb[3 * pix + 2] = b[q];
      out_depth[pix] = d[q] / fmaxf(al, 1e-10f);
      out_alpha[pix] = al;
    }
  }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RasterWs {
  G2D* g2d; unsigned long long* counts; unsigned long long* offsets; float4* rgb;
  unsigned long long* keys[2]; unsigned int* vals[2]; unsigned int* tile_offs; void* cub; size_t cub_bytes; size_t total;
};

RasterWs carve(char* base, size_t N, size_t C, int tiles, size_t max_isects) {
  RasterWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += align256(bytes); return p; };
  const size_t CN = N * C;
  w.g2d = (G2D*)take(CN * sizeof(G2D));
  w.counts = (unsigned long long*)take((CN + 1) * 8);
  w.offsets = (unsigned long long*)take((CN + 1) * 8);
  w.rgb = (float4*)take(N * sizeof(float4));
  w.keys[0] = (unsigned long long*)take(max_isects * 8); w.keys[1] = (unsigned long long*)take(max_isects * 8);
  w.vals[0] = (unsigned int*)take(max_isects * 4); w.vals[1] = (unsigned int*)take(max_isects * 4);
  w.tile_offs = (unsigned int*)take((C * (size_t)tiles + 1) * 4);
  size_t scan_b = 0, sort_b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_b, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)(CN + 1));
  hipcub::DoubleBuffer<unsigned long long> dk((unsigned long long*)nullptr, (unsigned long long*)nullptr);
  hipcub::DoubleBuffer<unsigned int> dv((unsigned int*)nullptr, (unsigned int*)nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, dk, dv, (int)(max_isects > 0 ? max_isects : 1), 0, 64);
  w.cub_bytes = scan_b > sort_b ? scan_b : sort_b;
  w.cub = take(w.cub_bytes);
  w.total = o;
  return w;
}

}  // namespace

size_t wm_raster_workspace_bytes(int N, int C, int width, int height, size_t max_isects) {
  const int tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
  return carve(nullptr, (size_t)N, (size_t)C, tiles, max_isects).total;
}

// returns hipSuccess and *n_isects_out; if the intersection count exceeds max_isects nothing is rendered and
// *n_isects_out holds the required count (the caller re-sizes the workspace)
hipError_t wm_launch_rasterize(const WmRasterArgs& a, hipStream_t s, unsigned long long* n_isects_out) {
  const int tw = (a.width + TILE - 1) / TILE, th = (a.height + TILE - 1) / TILE, tiles = tw * th;
  if (a.N <= 0 || a.C <= 0 || tw > 255 || th > 255) return hipErrorInvalidValue;
  const size_t N = a.N, C = a.C, CN = N * C;
  if (CN >= (1ull << 31)) return hipErrorInvalidValue;
  RasterWs w = carve((char*)a.workspace, N, C, tiles, a.max_isects);
  if (w.total > a.workspace_bytes) return hipErrorInvalidValue;
  int tile_bits = 0;
  while ((1 << tile_bits) <= tiles) ++tile_bits;  // = bit_length(tiles), as the reference
  int cam_bits = 0;
  while ((1ull << cam_bits) < C) ++cam_bits;
  hipLaunchKernelGGL(raster_color_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, a.colors, a.N, a.is_sh, w.rgb);
  hipLaunchKernelGGL(raster_project_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)C), dim3(256), 0, s, a.means, a.quats, a.scales, a.viewmats,
                     a.Ks, a.N, a.C, a.width, a.height, 0.01f, 1e10f, a.opacities, w.rgb, w.g2d, w.counts, a.radii_out);
  hipError_t e = hipMemsetAsync(w.counts + CN, 0, 8, s);
  if (e != hipSuccess) return e;
  size_t tb = w.cub_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(w.cub, tb, w.counts, w.offsets, (int)(CN + 1), s);
  if (e != hipSuccess) return e;
  unsigned long long n_isects = 0;
  e = hipMemcpyAsync(&n_isects, w.offsets + CN, 8, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);  // the list length sizes the sort (the reference's isect_tiles does .item() here too)
  if (e != hipSuccess) return e;
  if (n_isects_out) *n_isects_out = n_isects;
  if (n_isects > a.max_isects || n_isects >= (1ull << 31)) return hipSuccess;  // caller checks n_isects_out against max_isects
  const unsigned long long* sorted_keys = w.keys[0];
  const unsigned int* sorted_vals = w.vals[0];
  if (n_isects > 0) {
    hipLaunchKernelGGL(raster_emit_kernel, dim3((unsigned)((CN + 255) / 256)), dim3(256), 0, s, w.g2d, w.offsets, CN, a.N, tw, tile_bits, w.keys[0], w.vals[0]);
    hipcub::DoubleBuffer<unsigned long long> dk(w.keys[0], w.keys[1]);
    hipcub::DoubleBuffer<unsigned int> dv(w.vals[0], w.vals[1]);
    tb = w.cub_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(w.cub, tb, dk, dv, (int)n_isects, 0, 32 + tile_bits + cam_bits, s);
    if (e != hipSuccess) return e;
    sorted_keys = dk.Current(); sorted_vals = dv.Current();
  }
  hipLaunchKernelGGL(raster_offsets_kernel, dim3((unsigned)((C * tiles + 1 + 255) / 256)), dim3(256), 0, s, sorted_keys, (unsigned int)n_isects, a.C, tiles,
                     tile_bits, w.tile_offs);
  // pixels per lane of the compositing pass: a whole tile per wave when that still gives every SIMD several waves, half a tile
  // otherwise (8 views at 518^2, 8 712 tiles: 4 / 2 / 1 pixels per lane 3.14 / 3.31 / 4.37 ms; 2 views, 2 178 tiles: end to end
  // 0.87 / 0.74 / 0.79 ms; the LDS-staged workgroup-per-tile form it replaces: 3.70 ms, 0.83 ms — profiles/r03_raster_ab.md)
  static const int ppl_env = [] { const char* e = wm_env("WM_RASTER_PPL"); return e ? atoi(e) : 0; }();   // A/B: 1, 2, 4
  const int ppl = ppl_env ? ppl_env : ((long)tiles * C >= 8192 ? 4 : 2);
  if (ppl == 1)
    hipLaunchKernelGGL((raster_composite_kernel<1, 1>), dim3((unsigned)tiles, (unsigned)C), dim3(256), 0, s, w.g2d, sorted_vals, w.tile_offs, tw, th, a.width,
                       a.height, a.out_rgb, a.out_depth, a.out_alpha);
  else if (ppl == 2)
    hipLaunchKernelGGL((raster_composite_kernel<2, 1>), dim3((unsigned)tiles, (unsigned)C), dim3(128), 0, s, w.g2d, sorted_vals, w.tile_offs, tw, th, a.width,
                       a.height, a.out_rgb, a.out_depth, a.out_alpha);
  else
    hipLaunchKernelGGL((raster_composite_kernel<2, 2>), dim3((unsigned)tiles, (unsigned)C), dim3(64), 0, s, w.g2d, sorted_vals, w.tile_offs, tw, th, a.width,
                       a.height, a.out_rgb, a.out_depth, a.out_alpha);
  return hipGetLastError();
}
