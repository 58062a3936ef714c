// Standalone reproducer for the multi-queue finding (DESIGN.md §4 "Multi-stream finding"): NO torch, NO Python.
// Three independent chains  conv3x3 -> conv3x3 -> conv3x3 -> bilinear 296 -> 518  (libwm_hip.so's own kernels through the C ABI,
// the shapes of tools/dbg_chain2.py) run (a) one after the other on one stream = reference, (b) concurrently on three streams,
// ROUNDS times; every output word is compared with the reference.  Linked against /opt/rocm (the ROCm 7.2 runtime this
// tree's code objects are built for); the same binary logic under torch (tools/dbg_chain2.py) runs on the ROCm 7.0 runtime that
// torch 2.10.0+rocm7.0 bundles.  Build + run: see tools/micro/run_multiqueue_repro.sh.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/wm_hip.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(2); } } while (0)

static const int Cc = 128, Hh = 296, N = 4, Ho = 518;

struct Job {
  float *x, *b, *buf[2], *up;
  void* w[3];
  std::vector<float> ref_up, ref_b0, ref_b1;
};

static void chain(Job& j, hipStream_t s) {
  if (wm_op_conv(WM_DT_F16, j.x, j.w[0], j.b, nullptr, nullptr, j.buf[0], N, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, s) != WM_OK) exit(3);
  if (wm_op_conv(WM_DT_F16, j.buf[0], j.w[1], j.b, nullptr, nullptr, j.buf[1], N, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, s) != WM_OK) exit(3);
  if (wm_op_conv(WM_DT_F16, j.buf[1], j.w[2], j.b, nullptr, nullptr, j.buf[0], N, Hh, Hh, Cc, Cc, 3, 1, 1, 1, 0, s) != WM_OK) exit(3);
  if (wm_op_bilinear(j.buf[0], j.up, N, Hh, Hh, Ho, Ho, Cc, s) != WM_OK) exit(3);
}

static size_t diff(const float* dev, const std::vector<float>& ref, std::vector<float>& tmp, const char* what, int job, int round) {
  tmp.resize(ref.size());
  CK(hipMemcpy(tmp.data(), dev, ref.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, first = 0;
  for (size_t i = 0; i < ref.size(); ++i)
    if (memcmp(&tmp[i], &ref[i], 4) != 0) { if (!bad) first = i; ++bad; }
  if (bad) {
    const size_t pix = first / Cc;
    printf("round %d job %d %s: %zu wrong words, first at pixel %zu channel %zu: got %.6g ref %.6g\n", round, job, what, bad, pix, first % Cc,
           tmp[first], ref[first]);
  }
  return bad;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 12;
  const int NQ = argc > 2 ? atoi(argv[2]) : 3;   // queues = independent chains
  int rt = 0, drv = 0;
  CK(hipRuntimeGetVersion(&rt)); CK(hipDriverGetVersion(&drv));
  printf("hip runtime %d driver %d\n", rt, drv);
  const size_t nin = (size_t)N * Hh * Hh * Cc, nup = (size_t)N * Ho * Ho * Cc, nw = (size_t)Cc * 9 * Cc;
  std::vector<Job> jobs(NQ);
  std::vector<float> h(nin), hw(nw), hb(Cc);
  std::vector<uint16_t> hw16(nw);
  for (int k = 0; k < NQ; ++k) {
    std::mt19937 g(k + 1);
    std::normal_distribution<float> nd(0.f, 1.f);
    Job& j = jobs[k];
    for (auto& v : h) v = nd(g);
    CK(hipMalloc(&j.x, nin * 4)); CK(hipMemcpy(j.x, h.data(), nin * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) {
      for (auto& v : hw) v = nd(g) / std::sqrt((float)(Cc * 9));
      wm_host_to_16(hw.data(), hw16.data(), nw, WM_DT_F16);
      CK(hipMalloc(&j.w[i], nw * 2)); CK(hipMemcpy(j.w[i], hw16.data(), nw * 2, hipMemcpyHostToDevice));
    }
    for (auto& v : hb) v = nd(g);
    CK(hipMalloc(&j.b, Cc * 4)); CK(hipMemcpy(j.b, hb.data(), Cc * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&j.buf[0], nin * 4)); CK(hipMalloc(&j.buf[1], nin * 4)); CK(hipMalloc(&j.up, nup * 4));
  }
  hipStream_t s0;
  std::vector<hipStream_t> st(NQ);
  CK(hipStreamCreate(&s0));
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (auto& j : jobs) {  // single-queue reference
    chain(j, s0);
    CK(hipDeviceSynchronize());
    j.ref_up.resize(nup); j.ref_b0.resize(nin); j.ref_b1.resize(nin);
    CK(hipMemcpy(j.ref_up.data(), j.up, nup * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(j.ref_b0.data(), j.buf[0], nin * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(j.ref_b1.data(), j.buf[1], nin * 4, hipMemcpyDeviceToHost));
  }
  std::vector<float> tmp;
  size_t bad_single = 0, bad_multi = 0;
  for (int r = 0; r < rounds; ++r) {  // single queue again: must be bit-exact
    for (int k = 0; k < NQ; ++k) chain(jobs[k], s0);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < NQ; ++k) bad_single += diff(jobs[k].up, jobs[k].ref_up, tmp, "up (single queue)", k, r);
  }
  for (int r = 0; r < rounds; ++r) {  // three queues
    for (int k = 0; k < NQ; ++k) chain(jobs[k], st[k]);
    CK(hipDeviceSynchronize());
    for (int k = 0; k < NQ; ++k) {
      bad_multi += diff(jobs[k].buf[1], jobs[k].ref_b1, tmp, "conv2 out", k, r);
      bad_multi += diff(jobs[k].buf[0], jobs[k].ref_b0, tmp, "conv3 out", k, r);
      bad_multi += diff(jobs[k].up, jobs[k].ref_up, tmp, "bilinear out", k, r);
    }
  }
  printf("RESULT single-queue wrong words %zu, %d-queue wrong words %zu over %d rounds\n", bad_single, NQ, bad_multi, rounds);
  return 0;
}
