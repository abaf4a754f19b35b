// The odometer's dependent frame loop through the C++ adaptor's device-side entry points (fast_vgicp_hip.hpp): setInputTargetReframed
// (RGC_odometer.cpp:1248-1256 + :1007), alignEndReframe, holdSourceUntilTargetOf, setLazyTarget, rgc::DependentSequence.  Three routes over
// the same frames must give the same motions bit for bit:
//   A  one registration, the reference's call sequence per frame: setInputTargetReframed, setInputSourceDevice, align
//   B  rgc::DependentSequence on two registrations taking turns (alignEndReframe enqueues the next frame's target)
//   C  B with setLazyTarget(2) on both
// Clouds come from the raw float files the Python test writes (int32 n, then n x 3 floats); poses0.bin: 16 doubles, the world pose
// before the first frame.
//   test_dependent map.bin pose0.bin n_scans scan0.bin scan1.bin ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../rgc-slam_amd/cpp/fast_vgicp_hip.hpp"

struct Dev { float* p = nullptr; int n = 0; };

static Dev load(rgc_ctx* c, const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  int n = 0;
  if (fread(&n, 4, 1, f) != 1) exit(2);
  std::vector<float> xyz((size_t)n * 3), xyzw((size_t)n * 4, 0.f);
  if (fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) exit(2);
  fclose(f);
  for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) xyzw[(size_t)i * 4 + a] = xyz[(size_t)i * 3 + a];
  Dev d; d.n = n;
  void* p = nullptr;
  if (rgc_device_alloc(c, xyzw.size() * 4, &p) != RGC_OK || rgc_upload(c, p, xyzw.data(), xyzw.size() * 4) != RGC_OK || rgc_synchronize(c) != RGC_OK) exit(3);
  d.p = (float*)p;
  return d;
}

static void configure(rgc::FastVGICPHip& v) {   // RGC_odometer.cpp:998-1006
  v.setResolution(1.0); v.setMaximumIterations(25); v.setMaxCorrespondenceDistance(2); v.setTransformationEpsilon(1e-6);
  v.setEuclideanFitnessEpsilon(1e-6); v.setRANSACIterations(0); v.setNumThreads(14);
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int ns = atoi(argv[3]);
  if (argc < 4 + ns) return 2;
  double Tw0[16];
  {
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(Tw0, 8, 16, f) != 16) return 2;
    fclose(f);
  }
  try {
    rgc::FastVGICPHip a, b;
    configure(a); configure(b);
    const Dev map = load(a.context(), argv[1]);
    std::vector<Dev> scans;
    for (int s = 0; s < ns; s++) scans.push_back(load(a.context(), argv[4 + s]));
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    auto set_source = [&](int i, rgc::FastVGICPHip& v) { v.setInputSourceDevice(scans[(size_t)i].p, scans[(size_t)i].n, 16); };

    // ---- A: the reference's call sequence, one frame at a time ----
    std::vector<float> A((size_t)ns * 16), B((size_t)ns * 16), C((size_t)ns * 16);
    std::vector<double> fA((size_t)ns), fB((size_t)ns), fC((size_t)ns);
    void* scratch = nullptr;
    if (rgc_device_alloc(a.context(), (size_t)map.n * 16, &scratch) != RGC_OK) return 3;
    double Tw[16];
    std::memcpy(Tw, Tw0, sizeof(Tw));
    float g[16];
    std::memcpy(g, I, sizeof(g));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < ns; i++) {
      double q[4], t[3];
      rgc::DependentSequence::worldToBody(Tw, q, t);
      a.setInputTargetReframed(map.p, map.n, 16, q, t, (float*)scratch);
      set_source(i, a);
      a.alignBegin(g, true);
      a.alignEnd();
      const float* T = a.getFinalTransformation();
      double W[16];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
          double v = 0.0;
          for (int k = 0; k < 4; k++) v += Tw[r * 4 + k] * (double)T[k * 4 + c];
          W[r * 4 + c] = v;
        }
      std::memcpy(Tw, W, sizeof(W));
      std::memcpy(g, T, sizeof(g));
      std::memcpy(&A[(size_t)i * 16], T, sizeof(g));
      fA[(size_t)i] = a.getFitnessScore();
    }
    const double ms_a = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / ns;
    double TwA[16];
    std::memcpy(TwA, Tw, sizeof(Tw));
    rgc_device_free(a.context(), scratch);

    // ---- B: two registrations taking turns; C: the same with the lazy target ----
    double ms_b = 0, ms_c = 0, TwB[16], TwC[16];
    {
      rgc::DependentSequence seq(a, &b, map.p, map.n, 16);
      std::memcpy(TwB, Tw0, sizeof(TwB));
      t0 = std::chrono::steady_clock::now();
      seq.run(ns, TwB, I, true, set_source, [&](int i, rgc::FastVGICPHip& v) {
        std::memcpy(&B[(size_t)i * 16], v.getFinalTransformation(), 16 * sizeof(float));
        fB[(size_t)i] = v.getFitnessScore();
      });
      ms_b = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / ns;
      a.setLazyTarget(2); b.setLazyTarget(2);
      std::memcpy(TwC, Tw0, sizeof(TwC));
      t0 = std::chrono::steady_clock::now();
      seq.run(ns, TwC, I, true, set_source, [&](int i, rgc::FastVGICPHip& v) {
        std::memcpy(&C[(size_t)i * 16], v.getFinalTransformation(), 16 * sizeof(float));
        fC[(size_t)i] = v.getFitnessScore();
      });
      ms_c = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / ns;
      a.setLazyTarget(0); b.setLazyTarget(0);
    }
    int same_b = 1, same_c = 1;
    for (size_t k = 0; k < A.size(); k++) { same_b &= A[k] == B[k]; same_c &= A[k] == C[k]; }
    for (int i = 0; i < ns; i++) { same_b &= fA[(size_t)i] == fB[(size_t)i]; same_c &= fA[(size_t)i] == fC[(size_t)i]; }
    for (int k = 0; k < 16; k++) { same_b &= TwA[k] == TwB[k]; same_c &= TwA[k] == TwC[k]; }
    for (int i = 0; i < ns; i++) {
      printf("T%d", i);
      for (int k = 0; k < 16; k++) printf(" %.9g", A[(size_t)i * 16 + k]);
      printf("\n");
    }
    printf("world");
    for (int k = 0; k < 16; k++) printf(" %.17g", TwA[k]);
    printf("\nsame_two_contexts %d\nsame_lazy %d\nlazy_misses %d\nms_per_frame_reference_calls %.4f\nms_per_frame_two_contexts %.4f\nms_per_frame_lazy %.4f\n", same_b,
           same_c, a.stats().lazy_misses + b.stats().lazy_misses, ms_a, ms_b, ms_c);
  } catch (const std::exception& e) {
    printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
