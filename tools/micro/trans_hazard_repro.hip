// Minimal reproducer attempt for the multi-queue finding: a kernel whose ARITHMETIC goes wrong — inputs read
// correctly — when a different, transcendental-heavy kernel shares its SIMDs from another queue.
//
// Observed in the product (tools/dbg_c5b.py, 32 views, Gaussian head, 5 queues): gs_splat_kernel reads the right camera vector
// (dumped by the same threads), yet tc[1] = -(R[1] t0 + R[4] t1 + R[7] t2) comes out as -(R[1] t0 + R[7] t2): ONE fma term is
// lost, in groups of 16 consecutive lanes, a few hundred lanes per launch, one launch in three; never with a single queue.
// Here: kernel A = that arithmetic (v_rcp division, tanf, packed fp32 fma chain) on known inputs, every lane checked against a
// double-precision host value; kernel B = back-to-back v_exp_f32 / v_rcp_f32 on another stream.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/micro/trans_hazard_repro.hip -o tools/micro/libtrans_hazard.so
//        (or without -shared -fPIC, -DWITH_MAIN, for a standalone binary)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void kernel_a(const float* __restrict__ cam, const float* __restrict__ depth, float* __restrict__ means,
                                                float* __restrict__ tcout, int N, int H, int W) {
  const size_t npix = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((size_t)W * H));
    const float* v = cam + n * 9;
    const float qi = v[3], qj = v[4], qk = v[5], qr = v[6];
    const float s2 = 2.0f / (qi * qi + qj * qj + qk * qk + qr * qr);
    const float R[9] = {1 - s2 * (qj * qj + qk * qk), s2 * (qi * qj - qk * qr), s2 * (qi * qk + qj * qr),
                        s2 * (qi * qj + qk * qr), 1 - s2 * (qi * qi + qk * qk), s2 * (qj * qk - qi * qr),
                        s2 * (qi * qk - qj * qr), s2 * (qj * qk + qi * qr), 1 - s2 * (qi * qi + qj * qj)};
    const float fy = H * 0.5f / tanf(v[7] * 0.5f), fx = W * 0.5f / tanf(v[8] * 0.5f);
    const float d = depth[i];
    const float xc = ((float)x - W * 0.5f) * d / fx, yc = ((float)y - H * 0.5f) * d / fy, zc = d;
    float tc[3];
    for (int a = 0; a < 3; ++a) tc[a] = -(R[0 * 3 + a] * v[0] + R[1 * 3 + a] * v[1] + R[2 * 3 + a] * v[2]);
    for (int a = 0; a < 3; ++a) means[i * 3 + a] = R[0 * 3 + a] * xc + R[1 * 3 + a] * yc + R[2 * 3 + a] * zc + tc[a];
    for (int a = 0; a < 3; ++a) tcout[i * 3 + a] = tc[a];
  }
}

__global__ __launch_bounds__(256) void kernel_b(float* __restrict__ out, int iters) {
  float a = 0.001f * threadIdx.x, b = 1.0f + 0.002f * threadIdx.x, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) {
    a = __builtin_amdgcn_exp2f(a) * 0.5f;
    b = __builtin_amdgcn_rcpf(b) + 1.0f;
    c = __builtin_amdgcn_exp2f(c) * 0.25f;
    d = __builtin_amdgcn_rcpf(d + 1.0f);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

extern "C" int trans_hazard_run(int rounds, int streams_b) {
  const int N = 8, H = 518, W = 518;
  const size_t npix = (size_t)N * H * W;
  std::vector<float> cam(N * 9), dep(npix), tc_ref(N * 3);
  srand(1);
  for (int n = 0; n < N; ++n) {
    float* v = &cam[n * 9];
    for (int k = 0; k < 7; ++k) v[k] = (rand() / (float)RAND_MAX - 0.5f) * 1.2f;
    v[7] = 0.8f + 0.4f * rand() / (float)RAND_MAX; v[8] = 1.2f + 0.3f * rand() / (float)RAND_MAX;
    const double qi = v[3], qj = v[4], qk = v[5], qr = v[6], s2 = 2.0 / (qi * qi + qj * qj + qk * qk + qr * qr);
    const double R[9] = {1 - s2 * (qj * qj + qk * qk), s2 * (qi * qj - qk * qr), s2 * (qi * qk + qj * qr),
                         s2 * (qi * qj + qk * qr), 1 - s2 * (qi * qi + qk * qk), s2 * (qj * qk - qi * qr),
                         s2 * (qi * qk - qj * qr), s2 * (qj * qk + qi * qr), 1 - s2 * (qi * qi + qj * qj)};
    for (int a = 0; a < 3; ++a) tc_ref[n * 3 + a] = (float)-(R[a] * v[0] + R[3 + a] * v[1] + R[6 + a] * v[2]);
  }
  for (auto& x : dep) x = 0.5f + rand() / (float)RAND_MAX;
  float *d_cam, *d_dep, *d_means, *d_tc, *d_b;
  hipMalloc(&d_cam, cam.size() * 4); hipMalloc(&d_dep, npix * 4); hipMalloc(&d_means, npix * 12); hipMalloc(&d_tc, npix * 12);
  hipMalloc(&d_b, (size_t)8192 * 256 * 4);
  hipMemcpy(d_cam, cam.data(), cam.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_dep, dep.data(), npix * 4, hipMemcpyHostToDevice);
  hipStream_t sa; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  std::vector<hipStream_t> sb(streams_b);
  for (auto& s : sb) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  std::vector<float> tc(npix * 3);
  long bad_total = 0;
  for (int r = 0; r < rounds; ++r) {
    for (auto& s : sb) hipLaunchKernelGGL(kernel_b, dim3(2048), dim3(256), 0, s, d_b, 20000);
    for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(kernel_a, dim3(2048), dim3(256), 0, sa, d_cam, d_dep, d_means, d_tc, N, H, W);
    hipDeviceSynchronize();
    hipMemcpy(tc.data(), d_tc, npix * 12, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < npix; ++i)
      for (int a = 0; a < 3; ++a)
        if (std::fabs(tc[i * 3 + a] - tc_ref[(i / ((size_t)H * W)) * 3 + a]) > 1e-4f) { if (!bad) printf("round %d: first wrong lane %zu comp %d got %.6f ref %.6f\n", r, i, a, tc[i * 3 + a], tc_ref[(i / ((size_t)H * W)) * 3 + a]); ++bad; }
    bad_total += bad;
  }
  int rt = 0; hipRuntimeGetVersion(&rt);
  printf("trans_hazard: hip runtime %d, %d rounds, %d trans-heavy queues beside the arithmetic kernel: %ld wrong values\n", rt, rounds, streams_b, bad_total);
  return (int)(bad_total > 2000000000 ? 2000000000 : bad_total);
}
#ifdef WITH_MAIN
int main(int argc, char** argv) { trans_hazard_run(argc > 1 ? atoi(argv[1]) : 20, argc > 2 ? atoi(argv[2]) : 2); return 0; }
#endif
