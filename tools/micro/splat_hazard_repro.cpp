// In-situ reproducer of the multi-queue finding with the library's own kernels, NO torch, NO Python:
// queue 0 runs gs_splat_kernel (WM_DBG_SPLAT=1: it also dumps the camera vector each thread READ and the tc it COMPUTED),
// queues 1..NQ-1 run conv3x3 -> conv3x3 -> bilinear chains.  Checked per thread: camera vector read == the input (always, so far)
// and tc == -(R^T t) recomputed on the host.  Build + run: tools/micro/run_splat_hazard.sh [rounds] [queues]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/wm_hip.h"
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 30, NQ = argc > 2 ? atoi(argv[2]) : 5;
  setenv("WM_DBG_SPLAT", "1", 1);
  const int N = 8, H = 518, W = 518, Cc = 128, Hh = 296, Nc = 4;
  const size_t npix = (size_t)N * H * W;
  std::mt19937 g(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> cam(N * 9), tmp(npix * 12);
  for (int n = 0; n < N; ++n) { for (int k = 0; k < 7; ++k) cam[n * 9 + k] = 0.4f * nd(g); cam[n * 9 + 7] = 0.9f; cam[n * 9 + 8] = 1.3f; }
  std::vector<double> tc_ref(N * 3);
  for (int n = 0; n < N; ++n) {
    const float* v = &cam[n * 9];
    const double qi = v[3], qj = v[4], qk = v[5], qr = v[6], s2 = 2.0 / (qi * qi + qj * qj + qk * qk + qr * qr);
    const double R[9] = {1 - s2 * (qj * qj + qk * qk), s2 * (qi * qj - qk * qr), s2 * (qi * qk + qj * qr), s2 * (qi * qj + qk * qr), 1 - s2 * (qi * qi + qk * qk),
                         s2 * (qj * qk - qi * qr), s2 * (qi * qk - qj * qr), s2 * (qj * qk + qi * qr), 1 - s2 * (qi * qi + qj * qj)};
    for (int a = 0; a < 3; ++a) tc_ref[n * 3 + a] = -(R[a] * v[0] + R[3 + a] * v[1] + R[6 + a] * v[2]);
  }
  float *d_gp, *d_img, *d_dep, *d_cam, *d_means, *d_quats, *d_scales, *d_opac, *d_sh, *d_wts;
  CK(hipMalloc(&d_gp, npix * 48)); CK(hipMalloc(&d_img, npix * 12)); CK(hipMalloc(&d_dep, npix * 4)); CK(hipMalloc(&d_cam, N * 36));
  CK(hipMalloc(&d_means, npix * 12)); CK(hipMalloc(&d_quats, npix * 16)); CK(hipMalloc(&d_scales, npix * 12)); CK(hipMalloc(&d_opac, npix * 4));
  CK(hipMalloc(&d_sh, npix * 12)); CK(hipMalloc(&d_wts, npix * 4));
  for (auto& x : tmp) x = 0.3f * nd(g);
  CK(hipMemcpy(d_gp, tmp.data(), npix * 48, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_img, tmp.data(), npix * 12, hipMemcpyHostToDevice));
  for (size_t i = 0; i < npix; ++i) tmp[i] = 0.5f + std::fabs(tmp[i]);
  CK(hipMemcpy(d_dep, tmp.data(), npix * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cam, cam.data(), N * 36, hipMemcpyHostToDevice));
  // conv chains on the other queues
  const size_t nin = (size_t)Nc * Hh * Hh * Cc, nup = (size_t)Nc * 518 * 518 * Cc, nw = (size_t)Cc * 9 * Cc;
  struct Job { float *x, *b, *b0, *b1, *up; void* w; };
  std::vector<Job> jobs(NQ - 1);
  std::vector<float> h(nin), hw(nw);
  std::vector<uint16_t> hw16(nw);
  for (auto& j : jobs) {
    for (auto& v : h) v = nd(g);
    CK(hipMalloc(&j.x, nin * 4)); CK(hipMemcpy(j.x, h.data(), nin * 4, hipMemcpyHostToDevice));
    for (auto& v : hw) v = nd(g) / std::sqrt((float)(Cc * 9));
    wm_host_to_16(hw.data(), hw16.data(), nw, WM_DT_F16);
    CK(hipMalloc(&j.w, nw * 2)); CK(hipMemcpy(j.w, hw16.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&j.b, Cc * 4)); CK(hipMemset(j.b, 0, Cc * 4));
    CK(hipMalloc(&j.b0, nin * 4)); CK(hipMalloc(&j.b1, nin * 4)); CK(hipMalloc(&j.up, nup * 4));
  }
  std::vector<hipStream_t> st(NQ);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<float> opac(npix), wts(npix), sh(npix * 3), scales(npix * 3), quats(npix * 4);
  long bad_read = 0, bad_tc = 0;
  int rt = 0; CK(hipRuntimeGetVersion(&rt));
  for (int r = 0; r < rounds; ++r) {
    for (int k = 0; k < NQ - 1; ++k) {
      Job& j = jobs[k];
      if (wm_op_conv(WM_DT_F16, j.x, j.w, j.b, nullptr, nullptr, j.b0, Nc, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, st[k + 1]) != WM_OK) return 3;
      if (wm_op_conv(WM_DT_F16, j.b0, j.w, j.b, nullptr, nullptr, j.b1, Nc, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, st[k + 1]) != WM_OK) return 3;
      if (wm_op_bilinear(j.b1, j.up, Nc, Hh, Hh, 518, 518, Cc, st[k + 1]) != WM_OK) return 3;
    }
    for (int rep = 0; rep < 8; ++rep)
      if (wm_op_gs_splat(d_gp, d_img, d_dep, d_cam, d_means, d_quats, d_scales, d_opac, d_sh, d_wts, N, H, W, st[0]) != WM_OK) return 3;
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(opac.data(), d_opac, npix * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(wts.data(), d_wts, npix * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sh.data(), d_sh, npix * 12, hipMemcpyDeviceToHost));
    long br = 0, bt = 0;
    for (size_t i = 0; i < npix; ++i) {
      const int n = (int)(i / ((size_t)H * W));
      if (opac[i] != cam[n * 9] || wts[i] != cam[n * 9 + 1]) ++br;
      for (int a = 0; a < 3; ++a)
        if (std::fabs(sh[i * 3 + a] - tc_ref[n * 3 + a]) > 1e-4) { if (!bt) printf("round %d: thread %zu (view %d) tc[%d] = %.6f, expected %.6f; camera t_y it read %.6f (input %.6f)\n", r, i, n, a, sh[i * 3 + a], tc_ref[n * 3 + a], wts[i], cam[n * 9 + 1]); ++bt; }
    }
    bad_read += br; bad_tc += bt;
  }
  printf("splat_hazard: hip runtime %d, %d rounds x 8 launches, %d queues: threads that read a wrong camera vector %ld, threads with a wrong tc %ld\n", rt, rounds, NQ, bad_read, bad_tc);
  return 0;
}
