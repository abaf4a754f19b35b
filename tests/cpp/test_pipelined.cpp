// rgc::PipelinedVGICP (fast_vgicp_hip.hpp) against rgc::FastVGICPHip one frame at a time, on device-resident clouds read from the raw
// float files the Python test writes: prints both trajectories and the time per frame of each.
//   test_pipelined tgt.bin n_scans scan0.bin scan1.bin ... [repeat]
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
  if (argc < 4) return 2;
  const int ns = atoi(argv[2]);
  if (argc < 3 + ns) return 2;
  const int repeat = argc > 3 + ns ? atoi(argv[3 + ns]) : 1;
  try {
    rgc::PipelinedVGICP pipe(0, 2);
    for (int k = 0; k < pipe.depth(); k++) configure(pipe.context(k));
    rgc_ctx* c0 = pipe.context(0).context();
    const Dev tgt = load(c0, argv[1]);
    std::vector<Dev> scans;
    for (int s = 0; s < ns; s++) scans.push_back(load(c0, argv[3 + s]));
    const int N = ns * repeat;
    float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    auto set_clouds = [&](int i, rgc::FastVGICPHip& v) {
      v.setInputTargetDevice(tgt.p, tgt.n, 16);
      v.setInputSourceDevice(scans[(size_t)(i % ns)].p, scans[(size_t)(i % ns)].n, 16);
    };
    std::vector<float> seq((size_t)N * 16), par((size_t)N * 16);
    std::vector<double> fit_seq((size_t)N), fit_par((size_t)N);
    // warm-up of both contexts
    for (int k = 0; k < pipe.depth(); k++) { set_clouds(0, pipe.context(k)); pipe.context(k).align(I); }
    rgc::FastVGICPHip& v = pipe.context(0);
    float g[16];
    std::memcpy(g, I, sizeof(g));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      set_clouds(i, v);
      v.alignBegin(g, true);
      v.alignEnd();
      std::memcpy(g, v.getFinalTransformation(), sizeof(g));
      std::memcpy(&seq[(size_t)i * 16], g, sizeof(g));
      fit_seq[(size_t)i] = v.getFitnessScore();
    }
    const double ms_seq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / N;
    t0 = std::chrono::steady_clock::now();
    pipe.run(N, set_clouds, I, true, [&](int i, rgc::FastVGICPHip& w) {
      std::memcpy(&par[(size_t)i * 16], w.getFinalTransformation(), 16 * sizeof(float));
      fit_par[(size_t)i] = w.getFitnessScore();
    });
    const double ms_par = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / N;
    int same = 1;
    for (size_t k = 0; k < seq.size(); k++) same &= seq[k] == par[k];
    for (int i = 0; i < N; i++) same &= fit_seq[(size_t)i] == fit_par[(size_t)i];
    for (int i = 0; i < ns && i < N; i++) {
      printf("T%d", i);
      for (int k = 0; k < 16; k++) printf(" %.9g", par[(size_t)i * 16 + k]);
      printf("\n");
    }
    printf("same %d\nms_per_frame_one_at_a_time %.4f\nms_per_frame_pipelined %.4f\nframes %d\n", same, ms_seq, ms_par, N);
  } catch (const std::exception& e) {
    printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
