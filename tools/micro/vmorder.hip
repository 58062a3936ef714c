// Micro-benchmark: do ordinary global loads of one wave complete in issue order, i.e. is a counted s_waitcnt vmcnt(N)
// enough to consume the OLDEST load while N younger ones are still in flight?  Each lane issues one load from a cold
// address (its own never-touched cache line: HBM miss) followed by 4 loads from a hot line (L1/L2 hits), waits with
// vmcnt(4) and immediately copies the cold load's register.  A stale copy (-1) means the hot loads overtook the miss.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(const int* __restrict__ cold, const int* __restrict__ hot, int* __restrict__ out,
                                         size_t stride) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int* pc = cold + t * stride;
  const int* ph = hot + (threadIdx.x & 63);
  int a, b0, b1, b2, b3, r;
  asm volatile(
      "v_mov_b32 %0, -1\n\t"
      "s_nop 4\n\t"
      "global_load_dword %0, %6, off\n\t"
      "global_load_dword %1, %7, off\n\t"
      "global_load_dword %2, %7, off offset:256\n\t"
      "global_load_dword %3, %7, off offset:512\n\t"
      "global_load_dword %4, %7, off offset:768\n\t"
      "s_waitcnt vmcnt(4)\n\t"
      "v_mov_b32 %5, %0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      : "=&v"(a), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(r)
      : "v"(pc), "v"(ph)
      : "memory");
  out[t] = r + 0 * (a + b0 + b1 + b2 + b3);
}

// Variant 2: the four younger operations are LDS-DMA loads (global_load_lds_dword, hot line), the consumed one is the
// cold register load.  Variant 3: the oldest is a cold LDS-DMA and the consumed younger one... (not needed).
__global__ __launch_bounds__(256) void k2(const int* __restrict__ cold, const int* __restrict__ hot, int* __restrict__ out,
                                          size_t stride) {
  __shared__ int lds[4 * 256];
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int* pc = cold + t * stride;
  const int* ph = hot + (threadIdx.x & 63);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x >> 6) * 1024);
  int a, r;
  asm volatile(
      "v_mov_b32 %0, -1\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 4\n\t"
      "global_load_dword %0, %2, off\n\t"
      "global_load_lds_dword %3, off\n\t"
      "global_load_lds_dword %3, off\n\t"
      "global_load_lds_dword %3, off\n\t"
      "global_load_lds_dword %3, off\n\t"
      "s_waitcnt vmcnt(4)\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      : "=&v"(a), "=&v"(r)
      : "v"(pc), "v"(ph), "s"(base)
      : "memory", "m0");
  out[t] = r + 0 * (a + lds[threadIdx.x]);
}

int main() {
  const size_t n = 256 * 4096, stride = 64;  // 256 B apart: every lane its own line, 268 MB of cold data
  int *cold, *hot, *out;
  hipMalloc(&cold, n * stride * 4); hipMalloc(&hot, 4096); hipMalloc(&out, n * 4);
  std::vector<int> h(n * stride);
  for (size_t i = 0; i < n; ++i) h[i * stride] = (int)(i & 0x7fffff) + 1;
  hipMemcpy(cold, h.data(), n * stride * 4, hipMemcpyHostToDevice);
  hipMemset(hot, 0, 4096);
  std::vector<int> o(n);
  for (int rep = 0; rep < 10; ++rep) {
    hipMemset(out, 0, n * 4);
    // evict: touch another big buffer? the cold buffer (268 MB) exceeds L2 (32 MB) and MALL (256 MB) across reps
    if (rep < 5) k<<<4096, 256>>>(cold, hot, out, stride);
    else k2<<<4096, 256>>>(cold, hot, out, stride);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
    size_t stale = 0, wrong = 0;
    for (size_t i = 0; i < n; ++i) {
      if (o[i] == -1) ++stale;
      else if (o[i] != (int)(i & 0x7fffff) + 1) ++wrong;
    }
    printf("%s rep %d: stale (cold load consumed before it landed) %zu of %zu, other mismatches %zu\n", rep < 5 ? "[4 younger register loads]" : "[4 younger LDS-DMA loads]", rep, stale, n, wrong);
  }
  return 0;
}
