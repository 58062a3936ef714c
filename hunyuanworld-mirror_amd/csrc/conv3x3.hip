// 3x3 / stride 1 / pad 1 NHWC convolution with halo reuse — the dominant DPT-head operator
// (reference: src/models/heads/dense_head.py ResidualConvUnit :426-455, scratch.layer*_rn :394-406,
//  output_conv1 / output_conv2[0] :97-105; 95 % of the heads' 298.6 GFLOP/view).
//
// A block owns a 16x16 output patch (256 pixels) x BN output channels of one image.  Per 64-channel
// chunk the 18x18 input halo is staged ONCE into LDS (fp32 -> 16-bit, optional ReLU, zero padding)
// and then feeds all 9 taps: every tap's MFMA A-rows are the same LDS image read at a shifted row.
// The generic implicit-GEMM kernel (conv.hip) re-fetches and re-converts every input pixel 9 times.
// Weights [Cout][ky][kx][Cin] stream per (chunk, tap) by LDS-DMA, double buffered.  MFMA operands are
// swapped (D = W_frag * X_frag) so a lane owns a pixel and 4 consecutive channels: float4 epilogue
// with bias + relu?(resid) + resid2 (RCU skip and fusion add).
#include "wm_common.h"
#include "wm_kernels.h"

#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

constexpr int TP = 16;                      // patch edge
constexpr int HW_ = TP + 2;                 // halo edge (18)
constexpr int HROWS = HW_ * HW_;            // 324 halo pixels
constexpr int HALO_BYTES = HROWS * 128;     // 64 channels x 2 B per pixel
constexpr int HCH = HROWS * 8;              // 16-B chunks per halo
constexpr int HPT = (HCH + 511) / 512;      // chunks per thread (6)

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int T, int WM, int WN, int TM, int TN, int NARROW = 0>
__global__ __launch_bounds__(512) void conv3x3_kernel(const WmConvArgs p) {
  constexpr int BN = WN * TN * 32;
  constexpr int B_BYTES = BN * 128;
  constexpr int PB = BN / 8;  // 1-KiB LDS-DMA pieces per weight tile
  static_assert(WM * WN == 8 && WM * TM * 32 == 256, "8 waves x 256 pixels");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 halo][2 B]
  char* hbuf = smem;
  char* bbuf = smem + (NARROW ? 1 : 2) * HALO_BYTES;  // NARROW: one halo buffer, then all 9 taps' weights

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int H = p.Hi, W = p.Wi, Cin = p.Cin, Cout = p.Cout;
  const int tiles_x = (W + TP - 1) / TP, tiles_y = (H + TP - 1) / TP, ctiles = (Cout + BN - 1) / BN;
  const int nblk = p.N * tiles_y * tiles_x * ctiles;
  int lid = xcd_remap(blockIdx.x, nblk);
  const int ct = lid % ctiles; lid /= ctiles;
  const int tx = lid % tiles_x; lid /= tiles_x;
  const int ty = lid % tiles_y;
  const int n = lid / tiles_y;
  const int y0 = ty * TP, x0 = tx * TP, n0 = ct * BN;
  const float* xin = p.x + (size_t)n * H * W * Cin;  // re-pointed below in up mode
  const u16* Wt = (const u16*)p.w;
  const int K = 9 * Cin;

  // ---- halo staging (global fp32 -> regs -> 16-bit LDS), one item (= one 8-channel chunk of one halo pixel) per
  // thread per call: chunk cc+1's halo is brought in over taps 0..HPT of chunk cc, an item's loads issued before one
  // tap's MFMAs and converted / written to the other halo buffer at the start of the next tap.  With up_hs > 0 the item is interpolated on
  // the fly from the low-resolution source (4 corner loads, the weights of bilinear_kernel) — the upsampled tensor is
  // never written to HBM.
  const bool up = p.up_hs > 0;
  const float usy = up && H > 1 ? (float)(p.up_hs - 1) / (float)(H - 1) : 0.f;
  const float usx = up && W > 1 ? (float)(p.up_ws - 1) / (float)(W - 1) : 0.f;
  if (up) xin = p.x + (size_t)n * p.up_hs * p.up_ws * Cin;
  struct HaloItem {
    f32x4 hv[4][2];      // plain: hv[0]; up: the 4 corners (native vectors: they are tied to the wait below)
    f32x4 ha[2];         // position-table add (up mode with tables)
    float hw[4];         // corner weights
    bool in;             // the halo pixel lies inside the image
  };
  HaloItem it0;
  it0.in = false;
  auto halo_load_it = [&](HaloItem& it, int cc, int i) {
    f32x4 (&hv)[4][2] = it.hv; f32x4 (&ha)[2] = it.ha; float (&hw)[4] = it.hw; bool& h_in = it.in;
    const int id = tid + i * 512;
    const int hr = id >> 3, ch = id & 7;
    const int hy = hr / HW_, hx = hr - hy * HW_;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    h_in = id < HCH && iy >= 0 && iy < H && ix >= 0 && ix < W;
    if (!h_in) return;
    const int c0 = cc * 64 + ch * 8;
    if (!up) {
      const float* s = xin + ((size_t)iy * W + ix) * Cin + c0;
      hv[0][0] = *(const f32x4*)s;
      hv[0][1] = *(const f32x4*)(s + 4);
    } else {
      const float fy = usy * iy, fx = usx * ix;
      int ya = (int)fy, xa = (int)fx;
      ya = ya < p.up_hs - 1 ? ya : p.up_hs - 1;
      xa = xa < p.up_ws - 1 ? xa : p.up_ws - 1;
      const int yb = ya < p.up_hs - 1 ? ya + 1 : ya, xb = xa < p.up_ws - 1 ? xa + 1 : xa;
      const float wy = fy - ya, wx = fx - xa;
      hw[0] = (1.f - wy) * (1.f - wx); hw[1] = (1.f - wy) * wx; hw[2] = wy * (1.f - wx); hw[3] = wy * wx;
      const float* s00 = xin + ((size_t)ya * p.up_ws + xa) * Cin + c0;
      const float* s01 = xin + ((size_t)ya * p.up_ws + xb) * Cin + c0;
      const float* s10 = xin + ((size_t)yb * p.up_ws + xa) * Cin + c0;
      const float* s11 = xin + ((size_t)yb * p.up_ws + xb) * Cin + c0;
      hv[0][0] = *(const f32x4*)s00; hv[0][1] = *(const f32x4*)(s00 + 4);
      hv[1][0] = *(const f32x4*)s01; hv[1][1] = *(const f32x4*)(s01 + 4);
      hv[2][0] = *(const f32x4*)s10; hv[2][1] = *(const f32x4*)(s10 + 4);
      hv[3][0] = *(const f32x4*)s11; hv[3][1] = *(const f32x4*)(s11 + 4);
      if (p.up_addx) {
        const int half = Cin >> 1;
        const float* t = c0 < half ? p.up_addx + (size_t)ix * half + c0 : p.up_addy + (size_t)iy * half + (c0 - half);
        ha[0] = *(const f32x4*)t; ha[1] = *(const f32x4*)(t + 4);
      }
    }
  };
  auto halo_store_it = [&](HaloItem& it, char* dst, int i) {
    // The item's registers are released by an explicit full drain that they are tied to: hipcc's own waitcnt
    // insertion consumed loop-carried loads too early in these kernels (LDS-DMA in flight), and untied register-only
    // math may be hoisted above a bare asm wait.  Every call site sits right behind a vmcnt(0) anyway.
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(it.hv[0][0]), "+v"(it.hv[0][1]), "+v"(it.hv[1][0]), "+v"(it.hv[1][1]), "+v"(it.hv[2][0]),
                 "+v"(it.hv[2][1]), "+v"(it.hv[3][0]), "+v"(it.hv[3][1]), "+v"(it.ha[0]), "+v"(it.ha[1]) : : "memory");
    f32x4 (&hv)[4][2] = it.hv; f32x4 (&ha)[2] = it.ha; float (&hw)[4] = it.hw; const bool h_in = it.in;
    const int id = tid + i * 512;
    if (id >= HCH) return;
    const int hr = id >> 3, ch = id & 7;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (h_in) {
      if (!up) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = hv[0][q][e];
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = hw[0] * hv[0][q][e] + hw[1] * hv[1][q][e] + hw[2] * hv[2][q][e] + hw[3] * hv[3][q][e];
        }
        if (p.up_addx) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] += ha[q][e];
        }
      }
      if (p.relu_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
    }
    uint4 u;
    u.x = (uint32_t)f2t<T>(v[0]) | ((uint32_t)f2t<T>(v[1]) << 16);
    u.y = (uint32_t)f2t<T>(v[2]) | ((uint32_t)f2t<T>(v[3]) << 16);
    u.z = (uint32_t)f2t<T>(v[4]) | ((uint32_t)f2t<T>(v[5]) << 16);
    u.w = (uint32_t)f2t<T>(v[6]) | ((uint32_t)f2t<T>(v[7]) << 16);
    *(uint4*)(dst + hr * 128 + ((ch ^ swz(hr)) << 4)) = u;
  };
  auto halo_load = [&](int cc, int i) { halo_load_it(it0, cc, i); };
  auto halo_store = [&](char* dst, int i) { halo_store_it(it0, dst, i); };
  // ---- weight tile (chunk cc, tap) by LDS-DMA, swizzle on the source address
  auto stage_w = [&](int cc, int tap, char* dst) {
    for (int pc = wave; pc < PB; pc += 8) {
      const int r = pc * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz(r);
      int co = n0 + r;
      co = co < Cout ? co : Cout - 1;
      __builtin_amdgcn_global_load_lds((glb_vp)(Wt + (size_t)co * K + tap * Cin + cc * 64 + c * 8), (lds_vp)(dst + pc * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane halo row of the patch pixel this lane feeds into MFMA tile i (tap offset added later)
  int hbase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wm * TM + i) * 32 + (lane & 31);
    hbase[i] = (r >> 4) * HW_ + (r & 15);
  }

  const int nchunks = Cin / 64;
  if constexpr (NARROW) {
    // Narrow variant (Cout <= 32: output_conv2[0], 128 -> 32 at full resolution with the resize fused): the MFMA work
    // per tap is 4 instructions per wave, so the per-tap DMA wait + barrier of the loops below IS the run time
    // (20 us per block, measured).  Here a chunk's whole halo and all 9 weight tiles (36 KiB) are staged up front and
    // the 36 MFMA steps run without a barrier; the halo comes in two batches of three items per thread (12 KiB of
    // loads in flight per wave), i.e. two memory latencies per chunk instead of nine: 740 -> 650 us for the fused
    // 296 -> 518 resize + conv at 8 views.  What remains is L2 bandwidth: the 4-corner gather reads every source value
    // ~5 times (5.6 GB per launch).  Staging the ~12 x 12 source patch in LDS by DMA and blending from LDS was tried
    // and was slower (spills at 256 VGPRs); left for a dedicated kernel.
    static_assert(!NARROW || (TM == 1 && TN == 1 && WM == 8), "narrow variant: 32 px x 32 ch per wave");
    HaloItem it1, it2;
    it1.in = it2.in = false;
    for (int cc = 0; cc < nchunks; ++cc) {
      if (cc) __syncthreads();  // everybody finished reading the previous chunk's halo and weights
      for (int pc = wave; pc < 36; pc += 8) {  // weights [tap][cout 0..31][64 cin]: 4 pieces per tap
        const int tap = pc >> 2, r = (pc & 3) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(r);
        int co = n0 + r;
        co = co < Cout ? co : Cout - 1;
        __builtin_amdgcn_global_load_lds((glb_vp)(Wt + (size_t)co * K + tap * Cin + cc * 64 + c * 8), (lds_vp)(bbuf + pc * 1024), 16, 0, 0);
      }
#pragma unroll 1
      for (int i = 0; i < HPT; i += 3) {
        halo_load_it(it0, cc, i);
        if (i + 1 < HPT) halo_load_it(it1, cc, i + 1);
        if (i + 2 < HPT) halo_load_it(it2, cc, i + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // register loads are consumed only after a full drain (DMA pending)
        halo_store_it(it0, hbuf, i);
        if (i + 1 < HPT) halo_store_it(it1, hbuf, i + 1);
        if (i + 2 < HPT) halo_store_it(it2, hbuf, i + 2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * HW_ + (tap % 3);
        const int hr = hbase[0] + toff;
        const int row = tap * 32 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int ch = 2 * ks + (lane >> 5);
          const s16x8 a = *(const s16x8*)(hbuf + hr * 128 + ((ch ^ swz(hr)) << 4));
          const s16x8 b = *(const s16x8*)(bbuf + row * 128 + ((ch ^ swz(row & 31)) << 4));
          acc[0][0] = mfma32<T>(b, a, acc[0][0]);
        }
      }
    }
  } else {
  static_assert(HPT <= 7, "one halo item per tap, stored one tap later");
#pragma unroll 1
  for (int i = 0; i < HPT; ++i) {
    halo_load(0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    halo_store(hbuf, i);
  }
  stage_w(0, 0, bbuf);
  int kt = 0;
  for (int cc = 0; cc < nchunks; ++cc) {
    const char* hcur = hbuf + (cc & 1) * HALO_BYTES;
    for (int tap = 0; tap < 9; ++tap, ++kt) {
      // weights(kt) must have landed: hipcc does NOT reliably drain LDS-DMA before __syncthreads() here (it
      // emitted lgkmcnt(0) only), so the vmcnt(0) is explicit.  Also publishes the halo writes.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // halo item tap-1 of the next chunk: its global loads were issued one tap ago and are covered by the vmcnt(0)
      // above.  (Never consume a register load on a compiler-counted vmcnt(N > 0) while LDS-DMA is in flight: DMA and
      // register loads return out of order with respect to each other — measured: 16-lane groups with stale data.)
      if (tap >= 1 && tap <= HPT && cc + 1 < nchunks) halo_store(hbuf + ((cc + 1) & 1) * HALO_BYTES, tap - 1);
      const bool last = cc + 1 == nchunks && tap == 8;
      if (!last) stage_w(tap == 8 ? cc + 1 : cc, tap == 8 ? 0 : tap + 1, bbuf + ((kt + 1) & 1) * B_BYTES);
      const bool stage_h = tap < HPT && cc + 1 < nchunks;  // wave-uniform
      if (stage_h) halo_load(cc + 1, tap);
      const char* tB = bbuf + (kt & 1) * B_BYTES;
      const int toff = (tap / 3) * HW_ + (tap % 3);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 2 * ks + (lane >> 5);
        s16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int hr = hbase[i] + toff;
          a[i] = *(const s16x8*)(hcur + hr * 128 + ((ch ^ swz(hr)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = (wn * TN + j) * 32 + (lane & 31);
          b[j] = *(const s16x8*)(tB + row * 128 + ((ch ^ swz(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<T>(b[j], a[i], acc[i][j]);  // D[cout][pixel]
      }
    }
  }

  }  // not narrow
  // ---- epilogue: lane = pixel (lane&31) of tile i, regs 4g..4g+3 <-> channels 8g + 4h + {0..3}
  const int h4 = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wm * TM + i) * 32 + (lane & 31);
    const int y = y0 + (r >> 4), x = x0 + (r & 15);
    if (y >= H || x >= W) continue;
    const size_t obase = (((size_t)n * H + y) * W + x) * Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cb = n0 + (wn * TN + j) * 32 + h4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = cb + 8 * g;
        if (col >= Cout) continue;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        if (p.bias) {
          const float4 bs = *(const float4*)(p.bias + col);
          v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
        }
        if (p.resid) {
          float4 rr = *(const float4*)(p.resid + obase + col);
          if (p.resid_relu) rr = make_float4(fmaxf(rr.x, 0.f), fmaxf(rr.y, 0.f), fmaxf(rr.z, 0.f), fmaxf(rr.w, 0.f));
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.resid2) {
          const float4 rr = *(const float4*)(p.resid2 + obase + col);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.relu_out) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *(float4*)(p.y + obase + col) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Register-staged variant (plain input, 256- and 128-channel tiles).  In the kernel above a tap's MFMA work
// (0.45-0.9 us) is shorter than one memory round trip and every tap drains vmcnt(0) — it has to, LDS-DMA and register
// loads must not be mixed under counted waits — so the loop runs at the memory latency, not at the matrix pipe.
// Here NOTHING uses LDS-DMA: weights and halo items are ordinary global loads into registers (returned in order, so
// the compiler's own counted waits are safe), the weights of tap kt+2 are requested at the top of tap kt and written
// to LDS at the end of tap kt+1 (two taps of flight), a halo item of the next chunk is requested at the top of a tap
// and written at the end of the next one, and the tap barrier is a raw s_barrier behind an lgkmcnt(0) (a
// __syncthreads() would drain the loads in flight).
// TPX x (256 / TPX) output pixels per block: 16 x 16, or 32 wide x 8 high where that needs fewer rounds over the CUs
// (148^2 at 8 views: 800 tiles = 3.1 rounds with 16 x 16, 760 = 2.97 with 32 x 8).
template <int T, int WM, int WN, int TM, int TN, int UP = 0, int TPX = 16>
__global__ __launch_bounds__(512) void conv3x3_rs_kernel(const WmConvArgs p) {
  constexpr int TPY = 256 / TPX, LGX = TPX == 16 ? 4 : 5;
  static_assert(TPX == 16 || TPX == 32, "pixel tile 16 x 16 or 32 x 8");
  constexpr int HWX = TPX + 2, HWY = TPY + 2, HROWS = HWX * HWY, HALO_BYTES = HROWS * 128, HCH = HROWS * 8, HPT = (HCH + 511) / 512;
  constexpr int BN = WN * TN * 32;
  constexpr int B_BYTES = BN * 128;
  constexpr int WPT = BN * 8 / 512;  // 16-B weight chunks per thread per tap: 4 (256 ch) or 2 (128 ch)
  static_assert(WM * WN == 8 && WM * TM * 32 == 256 && WPT >= 1, "8 waves x 256 pixels");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 halo][2 B]
  char* hbuf = smem;
  char* bbuf = smem + 2 * HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int H = p.Hi, W = p.Wi, Cin = p.Cin, Cout = p.Cout;
  const int tiles_x = (W + TPX - 1) / TPX, tiles_y = (H + TPY - 1) / TPY, ctiles = (Cout + BN - 1) / BN;
  const int nblk = p.N * tiles_y * tiles_x * ctiles;
  int lid = xcd_remap(blockIdx.x, nblk);
  const int ct = lid % ctiles; lid /= ctiles;
  const int tx = lid % tiles_x; lid /= tiles_x;
  const int ty = lid % tiles_y;
  const int n = lid / tiles_y;
  const int y0 = ty * TPY, x0 = tx * TPX, n0 = ct * BN;
  const float* xin = UP ? p.x + (size_t)n * p.up_hs * p.up_ws * Cin : p.x + (size_t)n * H * W * Cin;
  const float usy = UP && H > 1 ? (float)(p.up_hs - 1) / (float)(H - 1) : 0.f;
  const float usx = UP && W > 1 ? (float)(p.up_ws - 1) / (float)(W - 1) : 0.f;
  const u16* Wt = (const u16*)p.w;
  const int K = 9 * Cin;
  const int nchunks = Cin / 64, NT = 9 * nchunks;

  // ---- weights: thread -> WPT (cout row, 16-B chunk) pairs of the tile
  const u16* wsrc[WPT];
  int wdst[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int id = tid + i * 512, r = id >> 3, c = id & 7;
    int co = n0 + r;
    co = co < Cout ? co : Cout - 1;
    wsrc[i] = Wt + (size_t)co * K + c * 8;
    wdst[i] = r * 128 + ((c ^ swz(r)) << 4);
  }
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // native vectors: HIP's uint4 / float4 structs kept these arrays in scratch
  struct WSet { u32x4 v[WPT]; };
  WSet wA, wB;  // two named register sets (an array of sets indexed by tap parity ends up in scratch memory)
  auto w_load = [&](int kt, WSet& ws) {  // kt = 9 * chunk + tap
    u32x4 (&w)[WPT] = ws.v;
    const int cc = kt / 9, tap = kt - cc * 9;
#pragma unroll
    for (int i = 0; i < WPT; ++i) w[i] = *(const u32x4*)(wsrc[i] + tap * Cin + cc * 64);
  };
  auto w_store = [&](const WSet& ws, char* dst) {
    const u32x4 (&w)[WPT] = ws.v;
#pragma unroll
    for (int i = 0; i < WPT; ++i) *(u32x4*)(dst + wdst[i]) = w[i];
  };
  // ---- halo items (plain input): two in flight
  struct HItem { f32x4 v[UP ? 4 : 1][2]; f32x4 t[2]; float w[4]; bool in; };  // UP: 4 corners, position-table add, weights
  HItem hA, hB;
  hA.in = hB.in = false;
  auto h_load = [&](int cc, int i, HItem& it) {
    const int id = tid + i * 512;
    const int hr = id >> 3, ch = id & 7;
    const int hy = hr / HWX, hx = hr - hy * HWX;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    it.in = id < HCH && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int c0 = cc * 64 + ch * 8;
    if (it.in) {
      if constexpr (!UP) {
        if (p.in16) {   // block-uniform: the input is already a 16-bit NHWC tensor of the operand type (RCU conv1 -> conv2): 16 B, no conversion
          const u16* sp = (const u16*)p.x + ((size_t)n * H * W + (size_t)iy * W + ix) * Cin + c0;
          it.v[0][0] = __builtin_bit_cast(f32x4, *(const u32x4*)sp);
        } else {
          const float* sp = xin + ((size_t)iy * W + ix) * Cin + c0;
          it.v[0][0] = *(const f32x4*)sp;
          it.v[0][1] = *(const f32x4*)(sp + 4);
        }
      } else {
        const float fy = usy * iy, fx = usx * ix;
        int ya = (int)fy, xa = (int)fx;
        ya = ya < p.up_hs - 1 ? ya : p.up_hs - 1;
        xa = xa < p.up_ws - 1 ? xa : p.up_ws - 1;
        const int yb = ya < p.up_hs - 1 ? ya + 1 : ya, xb = xa < p.up_ws - 1 ? xa + 1 : xa;
        const float wy = fy - ya, wx = fx - xa;
        it.w[0] = (1.f - wy) * (1.f - wx); it.w[1] = (1.f - wy) * wx; it.w[2] = wy * (1.f - wx); it.w[3] = wy * wx;
        const float* s00 = xin + ((size_t)ya * p.up_ws + xa) * Cin + c0;
        const float* s01 = xin + ((size_t)ya * p.up_ws + xb) * Cin + c0;
        const float* s10 = xin + ((size_t)yb * p.up_ws + xa) * Cin + c0;
        const float* s11 = xin + ((size_t)yb * p.up_ws + xb) * Cin + c0;
        it.v[0][0] = *(const f32x4*)s00; it.v[0][1] = *(const f32x4*)(s00 + 4);
        it.v[1][0] = *(const f32x4*)s01; it.v[1][1] = *(const f32x4*)(s01 + 4);
        it.v[2][0] = *(const f32x4*)s10; it.v[2][1] = *(const f32x4*)(s10 + 4);
        it.v[3][0] = *(const f32x4*)s11; it.v[3][1] = *(const f32x4*)(s11 + 4);
        if (p.up_addx) {
          const int half = Cin >> 1;
          const float* t = c0 < half ? p.up_addx + (size_t)ix * half + c0 : p.up_addy + (size_t)iy * half + (c0 - half);
          it.t[0] = *(const f32x4*)t; it.t[1] = *(const f32x4*)(t + 4);
        }
      }
    }
  };
  auto h_store = [&](char* dst, int i, const HItem& it) {
    const int id = tid + i * 512;
    if (id >= HCH) return;
    const int hr = id >> 3, ch = id & 7;
    if constexpr (!UP) {
      if (p.in16) {   // stored as it was loaded (zero outside the image)
        const u32x4 z = {0, 0, 0, 0};
        *(u32x4*)(dst + hr * 128 + ((ch ^ swz(hr)) << 4)) = it.in ? __builtin_bit_cast(u32x4, it.v[0][0]) : z;
        return;
      }
    }
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (it.in) {
      if constexpr (!UP) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = it.v[0][q][e];
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[4 * q + e] = it.w[0] * it.v[0][q][e] + it.w[1] * it.v[1][q][e] + it.w[2] * it.v[2][q][e] + it.w[3] * it.v[3][q][e];
        if (p.up_addx) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] += it.t[q][e];
        }
      }
      if (p.relu_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
    }
    uint4 u;
    u.x = (uint32_t)f2t<T>(v[0]) | ((uint32_t)f2t<T>(v[1]) << 16);
    u.y = (uint32_t)f2t<T>(v[2]) | ((uint32_t)f2t<T>(v[3]) << 16);
    u.z = (uint32_t)f2t<T>(v[4]) | ((uint32_t)f2t<T>(v[5]) << 16);
    u.w = (uint32_t)f2t<T>(v[6]) | ((uint32_t)f2t<T>(v[7]) << 16);
    *(uint4*)(dst + hr * 128 + ((ch ^ swz(hr)) << 4)) = u;
  };

  // Explicit waits, tied to the registers they release.  hipcc's own waitcnt insertion is NOT relied upon for these
  // loop-carried loads: with the loads issued a tap (or two) before their use, under conditions, it consumed stale
  // registers (sparse wrong halo pixels, run-to-run different), while the hardware itself completes loads in issue
  // order (tools/micro/vmorder.hip).
  auto wait_w = [&](int, WSet& ws) {  // full drain (hipcc's own merged waits drained here too; a switch over exact counts cost more than it saved)
    if constexpr (WPT == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ws.v[0]), "+v"(ws.v[1]), "+v"(ws.v[2]), "+v"(ws.v[3]) : : "memory");
    else if constexpr (WPT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ws.v[0]), "+v"(ws.v[1]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ws.v[0]) : : "memory");
  };
  auto wait_h = [&](int, HItem& it) {
    if constexpr (UP)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(it.v[0][0]), "+v"(it.v[0][1]), "+v"(it.v[1][0]), "+v"(it.v[1][1]), "+v"(it.v[2][0]), "+v"(it.v[2][1]),
                   "+v"(it.v[UP ? 3 : 0][0]), "+v"(it.v[UP ? 3 : 0][1]), "+v"(it.t[0]), "+v"(it.t[1]) : : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(it.v[0][0]), "+v"(it.v[0][1]) : : "memory");
  };
  const int LI = UP ? (p.up_addx ? 10 : 8) : 2;  // loads per halo item

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int hbase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wm * TM + i) * 32 + (lane & 31);
    hbase[i] = (r >> LGX) * HWX + (r & (TPX - 1));
  }

  // ---- prologue: weights(0) and (1) requested, chunk 0's halo staged, weights(0) written
  w_load(0, wA);
  if (NT > 1) w_load(1, wB);
  static_assert(HPT <= 7, "halo items: one per tap, written one tap later");
#pragma unroll 1
  for (int i = 0; i < HPT; i += 2) {
    h_load(0, i, hA);
    if (i + 1 < HPT) h_load(0, i + 1, hB);
    wait_h(0, hA);
    if (i + 1 < HPT) wait_h(0, hB);
    h_store(hbuf, i, hA);
    if (i + 1 < HPT) h_store(hbuf, i + 1, hB);
  }
  wait_w(0, wA);
  wait_w(0, wB);
  w_store(wA, bbuf);

  // one tap; `mine` held weights(kt) (already in LDS) and receives weights(kt+2); `other` holds weights(kt+1).
  // Two taps per loop iteration with the register sets swapped (a set array indexed by tap parity, or HIP's uint4 /
  // float4 structs instead of native vectors, end up in scratch memory).  hipcc's waitcnt bookkeeping merges
  // conservatively across the loop back-edge, so the ds_writes at the end of a tap still wait for the newest loads
  // (fully unrolling 18 taps makes the waits exact but spills); even so this loop measures +8-22 % over the LDS-DMA
  // one (tools/bench_conv.py (rounds 1-2; git history)): raw barrier, no vmcnt(0) drain at the top of the tap, no DMA issue.  Needs an even
  // number of 64-channel chunks (the two-tap step walks chunk pairs).
  auto tap_body = [&](int cc, int tap, int PAR, WSet& w_mine, WSet& w_other, HItem& h_mine, HItem& h_other) {
    const int kt = cc * 9 + tap;
    const char* hcur = hbuf + (cc & 1) * HALO_BYTES;
    char* hnext = hbuf + ((cc + 1) & 1) * HALO_BYTES;
    const bool more_chunks = cc + 1 < nchunks;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS writes of the previous tap are done
    __builtin_amdgcn_s_barrier();                        // everybody finished tap kt-1; its LDS writes are visible
#ifdef WM_CONV_TIMING_EXPERIMENT
    const bool do_w = !(p.dbg & 4), do_h = !(p.dbg & 1);
#else
    constexpr bool do_w = true, do_h = true;
#endif
    if (do_w && kt + 2 < NT) w_load(kt + 2, w_mine);
    if (do_h && more_chunks && tap < HPT) h_load(cc + 1, tap, h_mine);
    const char* tB = bbuf + PAR * B_BYTES;
    const int toff = (tap / 3) * HWX + (tap % 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = 2 * ks + (lane >> 5);
      s16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int hr = hbase[i] + toff;
        a[i] = *(const s16x8*)(hcur + hr * 128 + ((ch ^ swz(hr)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + (lane & 31);
        b[j] = *(const s16x8*)(tB + row * 128 + ((ch ^ swz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32<T>(b[j], a[i], acc[i][j]);
    }
    // end of tap: weights(kt+1) (requested at the top of tap kt-1) -> the buffer read during tap kt-1; the halo item
    // requested at the top of tap kt-1 -> the next chunk's halo buffer (nobody reads it during this chunk)
    // in flight, youngest last: W(kt+1) | item(tap-1) | W(kt+2) | item(tap)  (each only if it was requested)
    const int young = (kt + 2 < NT ? WPT : 0) + (more_chunks && tap < HPT ? LI : 0);  // requested at the top of THIS tap
    const bool st_h = more_chunks && tap >= 1 && tap <= HPT;
    if (do_w && kt + 1 < NT) {
      wait_w(young + (st_h ? LI : 0), w_other);
      w_store(w_other, bbuf + (PAR ^ 1) * B_BYTES);
    }
    if (do_h && st_h) {
      wait_h(young, h_other);
      h_store(hnext, tap - 1, h_other);
    }
  };
  for (int cc0 = 0; cc0 < nchunks; cc0 += 2) {
#pragma unroll 1
    for (int u = 0; u < 18; u += 2) {
      tap_body(cc0 + u / 9, u % 9, 0, wA, wB, hA, hB);
      tap_body(cc0 + (u + 1) / 9, (u + 1) % 9, 1, wB, wA, hB, hA);
    }
  }

#ifdef WM_CONV_TIMING_EXPERIMENT
  if ((p.dbg & 2) && acc[0][0][0] != 1.2345e-30f) return;
#endif
  // ---- epilogue, through LDS.  The MFMA leaves lane (pixel = lane & 31, h = lane >> 5) with channels 8g + 4h + {0..3} of its
  // pixel: stored from there, one instruction touches 32 pixels x 32 B = 32 cache lines (and the residual loads likewise); measured
  // with the epilogue switched off (tools/conv_timing_experiment.py): 130 of the 330 us of an RCU conv2 at 148^2, i.e. 537 MB at
  // 4.1 TB/s.  Here a wave parks one 32-pixel x (TN * 32)-channel group of accumulators in its own LDS slab (the main loop's
  // buffers are dead) and reads it back row-major: 8 TN lanes cover one pixel's contiguous channels, an instruction covers
  // 8 / TN whole pixels = 8 lines; bias, relu(resid), resid2 are added in that layout from equally coalesced loads, requested one
  // group ahead.  Same values, same order of additions: bit-identical to the direct form.
  constexpr int LP = TN * 8, PPI = 64 / LP, NI = 32 / PPI;  // lanes per pixel, pixels per instruction, instructions per group
  constexpr int SROW = TN * 32 + 4;                         // slab row stride in floats (+16 B: rows start 4 banks apart)
  static_assert(8 * 32 * SROW * 4 <= 2 * HALO_BYTES + 2 * B_BYTES, "epilogue slabs fit the main loop's LDS");
  __syncthreads();                                          // every wave is done reading the halo / weight buffers
  float* slab = (float*)smem + wave * 32 * SROW;
  const int h4 = (lane >> 5) * 4;
  const int lp = lane % LP, pq = lane / LP;                 // this lane's 4-channel chunk and pixel-in-instruction (read-back layout)
  const int cb = n0 + wn * TN * 32 + lp * 4;
  const bool cok = cb < Cout;
  const float4 bs = (p.bias && cok) ? *(const float4*)(p.bias + cb) : make_float4(0, 0, 0, 0);
  constexpr int NS = NI >= 8 ? 2 : 1, NH = NI / NS;         // a group is read back in NS sub-steps of NH instructions (register budget)
  float4 r1[2][NH], r2[2][NH];
  auto pix = [&](int i, int k, size_t& obase) {  // pixel of tile row i handled by this lane in instruction k -> in range
    const int r = (wm * TM + i) * 32 + k * PPI + pq;
    const int y = y0 + (r >> LGX), x = x0 + (r & (TPX - 1));
    obase = (((size_t)n * H + y) * W + x) * Cout;
    return y < H && x < W;
  };
  auto res_load = [&](int q, float4 (&a)[NH], float4 (&b)[NH]) {  // sub-step q = NS * i + half
    const int i = q / NS, k0 = (q % NS) * NH;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      size_t obase;
      const bool ok = pix(i, k0 + k, obase) && cok;
      a[k] = (p.resid && ok) ? *(const float4*)(p.resid + obase + cb) : make_float4(0, 0, 0, 0);
      b[k] = (p.resid2 && ok) ? *(const float4*)(p.resid2 + obase + cb) : make_float4(0, 0, 0, 0);
    }
  };
  res_load(0, r1[0], r2[0]);
#pragma unroll
  for (int q = 0; q < TM * NS; ++q) {
    const int i = q / NS, k0 = (q % NS) * NH;
    if (q + 1 < TM * NS) res_load(q + 1, r1[(q + 1) & 1], r2[(q + 1) & 1]);
    if (q % NS == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous group's read-back is complete (LDS is in order per wave; this pins the compiler)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(slab + (lane & 31) * SROW + j * 32 + 8 * g + h4) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const float4 a4 = *(const float4*)(slab + ((k0 + k) * PPI + pq) * SROW + lp * 4);
      float4 v = make_float4(a4.x + bs.x, a4.y + bs.y, a4.z + bs.z, a4.w + bs.w);
      float4 rr = r1[q & 1][k];
      if (p.resid_relu) rr = make_float4(fmaxf(rr.x, 0.f), fmaxf(rr.y, 0.f), fmaxf(rr.z, 0.f), fmaxf(rr.w, 0.f));
      const float4 r2v = r2[q & 1][k];
      v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      v.x += r2v.x; v.y += r2v.y; v.z += r2v.z; v.w += r2v.w;
      if (p.relu_out) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      size_t obase;
      if (pix(i, k0 + k, obase) && cok) {
        if (p.out16) {  // block-uniform: 16-bit output only (8 B per lane, 128 B per pixel and instruction)
          uint2 u;
          u.x = (uint32_t)f2t<T>(v.x) | ((uint32_t)f2t<T>(v.y) << 16);
          u.y = (uint32_t)f2t<T>(v.z) | ((uint32_t)f2t<T>(v.w) << 16);
          *(uint2*)((u16*)p.y + obase + cb) = u;
        } else {
          *(float4*)(p.y + obase + cb) = v;
        }
      }
    }
  }
}

template <int T, int WM, int WN, int TM, int TN, int UP = 0, int TPX = 16>
hipError_t launch_rs(const WmConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * TN * 32, TPY = 256 / TPX;
  const size_t shm = 2 * (size_t)(TPX + 2) * (TPY + 2) * 128 + 2 * BN * 128;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_rs_kernel<T, WM, WN, TM, TN, UP, TPX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  const int nblk = a.N * ((a.Hi + TPY - 1) / TPY) * ((a.Wi + TPX - 1) / TPX) * ((a.Cout + BN - 1) / BN);
  hipLaunchKernelGGL((conv3x3_rs_kernel<T, WM, WN, TM, TN, UP, TPX>), dim3(nblk), dim3(512), shm, s, a);
  return hipGetLastError();
}

template <int T, int WM, int WN, int TM, int TN, int NARROW = 0>
hipError_t launch_cfg(const WmConvArgs& a, hipStream_t s) {
  constexpr int BN = WN * TN * 32;
  const size_t shm = NARROW ? HALO_BYTES + 9 * 32 * 128 : 2 * HALO_BYTES + 2 * BN * 128;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3x3_kernel<T, WM, WN, TM, TN, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr = true;
  }
  const int nblk = a.N * ((a.Hi + TP - 1) / TP) * ((a.Wi + TP - 1) / TP) * ((a.Cout + BN - 1) / BN);
  hipLaunchKernelGGL((conv3x3_kernel<T, WM, WN, TM, TN, NARROW>), dim3(nblk), dim3(512), shm, s, a);
  return hipGetLastError();
}

int cu_count() {
  static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
  return ncu;
}
int pick_bn(const WmConvArgs& a) {  // output-channel tile (see launch_T)
  const int ncu = cu_count();
  const long ptiles = (long)a.N * ((a.Hi + TP - 1) / TP) * ((a.Wi + TP - 1) / TP);
  int bn = a.Cout > 128 ? 256 : a.Cout > 64 ? 128 : a.Cout > 32 ? 64 : 32;
  while (bn > 64 && 2 * ptiles * ((a.Cout + bn - 1) / bn) < ncu) bn >>= 1;
  if (wm_tuning[WM_TUNE_CONV_BN] > 0) bn = wm_tuning[WM_TUNE_CONV_BN];
  return bn;
}

template <int T>
hipError_t launch_T(const WmConvArgs& a, hipStream_t s) {
  // output-channel tile: as wide as Cout allows, but narrower while the launch would cover less than half the chip
  // (the 19^2 and 37^2 DPT levels: 32 / 72 pixel tiles; measured with tools/bench_conv.py (rounds 1-2; git history): 37^2 161 -> 249 TF/s at
  // 128 channels, 19^2 46 -> 112 at 64; 74^2 with 200 tiles stays fastest at 256) — the halo is then re-staged per
  // channel tile, from L2.  conv_bn (tuning) forces a width.
  static const int ncu = cu_count();
  const long ptiles = (long)a.N * ((a.Hi + TP - 1) / TP) * ((a.Wi + TP - 1) / TP);
  const int bn = pick_bn(a);
  const bool rs_ok = (a.Cin / 64) % 2 == 0 && wm_tuning[WM_TUNE_CONV_RS] != 0;
  const bool rs = rs_ok && a.up_hs == 0;  // register-staged main loop (plain input, even chunk count)
  if (rs_ok && a.up_hs > 0 && bn == 128) {  // fused resize, 128-channel tile
    if (wm_tuning[WM_TUNE_CONV_TPX] == 32) return launch_rs<T, 4, 2, 2, 2, 1, 32>(a, s);  // A/B: 32 x 8 pixel tile (no LDS bank conflicts, 2.5 % more tiles at 296^2)
    return launch_rs<T, 4, 2, 2, 2, 1>(a, s);
  }
  if (rs && bn >= 128 && wm_tuning[WM_TUNE_CONV_TPX] != 16) {
    // pixel-tile shape: rounds over the CUs (one block per CU) with 16 x 16 vs 32 x 8 tiles
    const long ct = (a.Cout + bn - 1) / bn;
    const long t16 = ptiles * ct, t32 = (long)a.N * ((a.Hi + 7) / 8) * ((a.Wi + 31) / 32) * ct;
    const long r16 = (t16 + ncu - 1) / ncu, r32 = (t32 + ncu - 1) / ncu;
    if (r32 < r16 || (r32 == r16 && t32 < t16) || wm_tuning[WM_TUNE_CONV_TPX] == 32)  // fewer rounds, else fewer (less ragged) tiles
      return bn >= 256 ? launch_rs<T, 2, 4, 4, 2, 0, 32>(a, s) : launch_rs<T, 4, 2, 2, 2, 0, 32>(a, s);
  }
  if (bn >= 256) return rs ? launch_rs<T, 2, 4, 4, 2>(a, s) : launch_cfg<T, 2, 4, 4, 2>(a, s);   // 256 px x 256 ch (a two-group ping-pong main loop was tried here: bit-identical, no faster)
  if (bn >= 128) return rs ? launch_rs<T, 4, 2, 2, 2>(a, s) : launch_cfg<T, 4, 2, 2, 2>(a, s);   // 256 px x 128 ch
  if (bn >= 64) return launch_cfg<T, 4, 2, 2, 1>(a, s);    // 256 px x 64 ch (register staging measured equal here)
  if (wm_tuning[WM_TUNE_CONV_NARROW] == 0) return launch_cfg<T, 8, 1, 1, 1>(a, s);  // per-tap loop (A/B)
  return launch_cfg<T, 8, 1, 1, 1, 1>(a, s);               // 256 px x 32 ch, narrow variant
}

}  // namespace

bool wm_conv_force_generic() {  // WM_CONV_GENERIC (A/B switch): every conv through the generic kernel of conv.hip
  static const bool g = wm_env("WM_CONV_GENERIC") != nullptr;
  return g;
}
bool wm_conv3x3_out16_ok(const WmConvArgs& a) {  // exactly the launches launch_T sends to conv3x3_rs_kernel with a plain input
  return !wm_conv_force_generic() && wm_conv3x3_applicable(a) && a.up_hs == 0 && (a.Cin / 64) % 2 == 0 && wm_tuning[WM_TUNE_CONV_RS] != 0 && pick_bn(a) >= 128;
}

bool wm_conv3x3_applicable(const WmConvArgs& a) {
  return a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 64 == 0 && a.Cout % 4 == 0 && a.Hi * a.Wi >= 256;
}

hipError_t wm_launch_conv3x3(const WmConvArgs& a_in, hipStream_t s) {
  WmConvArgs a = a_in;
#ifdef WM_CONV_TIMING_EXPERIMENT
  { const char* e = wm_env("WM_CONV_DBG"); a.dbg = e ? atoi(e) : 0; }
#else
  a.dbg = 0;
#endif
  if (a.out16 && !wm_conv3x3_out16_ok(a)) return hipErrorInvalidValue;
  if (a.in16 && (!wm_conv3x3_out16_ok(a) || a.relu_in)) return hipErrorInvalidValue;   // the register-staged kernel with a plain input only
  return a.dtype == WM_T_BF16 ? launch_T<WM_T_BF16>(a, s) : launch_T<WM_T_F16>(a, s);
}
