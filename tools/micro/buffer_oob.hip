// Which offsets does the raw-buffer range check of gfx950 see?  (attn_v4's staged tiles rely on it: rows beyond a key segment's
// end must read as zeros.)  num_records = 20000 bytes over a 64 KiB array of 0xAB; 16-byte loads at voffset = lane * 128 with
// (a) soffset = 16384, (b) the same offset folded into voffset, (c) soffset = 0.  Prints the lanes that returned data.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/micro/buffer_oob tools/micro/buffer_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void probe(const unsigned char* base, unsigned* out, int soff) {
  const int lane = threadIdx.x;
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 20000, 0x00020000);
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 128, soff, 0);              // soffset carries the tile offset
  u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 128 + 16384, 0, 0);         // voffset carries it
  u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 128, 0, 0);
  u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 128 + 19990 - 63 * 128, 0, 0);   // lane 63 straddles the end (19990 + 16 > 20000)
  out[lane * 4 + 0] = a[0] | a[3]; out[lane * 4 + 1] = b[0] | b[3]; out[lane * 4 + 2] = c[0] | c[3]; out[lane * 4 + 3] = d[3];
}
int main() {
  unsigned char* buf; unsigned* out;
  hipMalloc(&buf, 65536); hipMemset(buf, 0xAB, 65536); hipMalloc(&out, 64 * 4 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out, 16384);
  std::vector<unsigned> h(256); hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"soffset=16384 voffset=lane*128 (in range iff lane < 28 when soffset counts)", "voffset=16384+lane*128 (lane < 28)", "voffset=lane*128 (all)", "straddle: lane 63's last dword"};
  for (int k = 0; k < 4; ++k) {
    int n = 0, last = -1;
    for (int l = 0; l < 64; ++l) if (h[l * 4 + k]) { ++n; last = l; }
    printf("%s: %d lanes returned data, highest lane %d\n", names[k], n, last);
  }
  return 0;
}
