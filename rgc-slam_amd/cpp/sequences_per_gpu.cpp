// sequences_per_gpu -- what ONE MI355X can carry: S independent dependent sequences (the odometer's frame loop, RGC_odometer.cpp:976-1256,
// through rgc::DependentSequence) on one device, each on its own pair of registrations and its own HOST THREAD -- no interpreter lock, no
// Python between the frames.  BASELINE.json's config 4 (8 sequences on 8 GPUs, one each, no collective) is S = 1 per device; this is the
// same path with more bags than GPUs, and the only hardware evidence of how sequences share a device that a one-GPU box can give.
//
//   sequences_per_gpu <device> <frames> <reps> <reuse 0|1|2> <S list, e.g. 1,2,4,8> <n datasets> <dir_0> [<dir_1> ...]
//
// A dataset directory holds map.bin, pose0.bin and s0.bin .. s<frames-1>.bin (int32 n, then n x 3 floats; pose0: 16 doubles, the world
// pose before the first frame) -- bench.py writes them.  Sequence s runs dataset s % n_datasets on its OWN device copies.
// Every sequence is first run alone (its reference motions); then, for each S, S threads leave a barrier together and run their frames
// `reps` times from the same start; the job is done when the slowest is.  One JSON line on stdout.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fast_vgicp_hip.hpp"

namespace {

struct Dev { float* p = nullptr; int n = 0; };

Dev load(rgc_ctx* c, const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { perror(path.c_str()); exit(2); }
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

void configure(rgc::FastVGICPHip& v, int reuse) {   // RGC_odometer.cpp:998-1006
  v.setResolution(1.0); v.setMaximumIterations(25); v.setMaxCorrespondenceDistance(2); v.setTransformationEpsilon(1e-6);
  v.setEuclideanFitnessEpsilon(1e-6); v.setRANSACIterations(0); v.setNumThreads(14);
  v.setNeighbourReuse(reuse);
}

struct Sequence {
  std::unique_ptr<rgc::FastVGICPHip> a, b;
  std::unique_ptr<rgc::DependentSequence> seq;
  Dev map;
  std::vector<Dev> scans;
  double Tw0[16];
  std::vector<float> ref;   // frames x 16: the motions of the run alone
  bool same = true;

  void run(int frames, std::vector<float>* out) {
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double Tw[16];
    std::memcpy(Tw, Tw0, sizeof(Tw));
    seq->run(frames, Tw, I, true,
             [&](int i, rgc::FastVGICPHip& v) { v.setInputSourceDevice(scans[(size_t)i].p, scans[(size_t)i].n, 16); },
             [&](int i, rgc::FastVGICPHip& v) {
               if (out) std::memcpy(&(*out)[(size_t)i * 16], v.getFinalTransformation(), 16 * sizeof(float));
               else if (std::memcmp(&ref[(size_t)i * 16], v.getFinalTransformation(), 16 * sizeof(float)) != 0) same = false;
             });
  }
};

struct Gate {   // a reusable barrier (C++14: no std::barrier)
  std::mutex m; std::condition_variable cv; int waiting = 0, generation = 0, n;
  explicit Gate(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int g = generation;
    if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); }
    else cv.wait(lk, [&] { return g != generation; });
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s device frames reps reuse S,S,... n_datasets dir...\n", argv[0]); return 2; }
  const int device = atoi(argv[1]), frames = atoi(argv[2]), reps = atoi(argv[3]), reuse = atoi(argv[4]);
  std::vector<int> S_list;
  for (char* tok = strtok(argv[5], ","); tok; tok = strtok(nullptr, ",")) S_list.push_back(atoi(tok));
  const int nd = atoi(argv[6]);
  if (argc < 7 + nd || frames < 1 || reps < 1 || S_list.empty()) return 2;
  int S_max = 0;
  for (int s : S_list) S_max = s > S_max ? s : S_max;
  try {
    std::vector<std::unique_ptr<Sequence>> seqs;
    for (int s = 0; s < S_max; s++) {
      const std::string dir = argv[7 + s % nd];
      std::unique_ptr<Sequence> q(new Sequence());
      q->a.reset(new rgc::FastVGICPHip(device));
      q->b.reset(new rgc::FastVGICPHip(device));
      configure(*q->a, reuse); configure(*q->b, reuse);
      q->map = load(q->a->context(), dir + "/map.bin");
      for (int i = 0; i < frames; i++) q->scans.push_back(load(q->a->context(), dir + "/s" + std::to_string(i) + ".bin"));
      FILE* f = fopen((dir + "/pose0.bin").c_str(), "rb");
      if (!f || fread(q->Tw0, 8, 16, f) != 16) return 2;
      fclose(f);
      q->seq.reset(new rgc::DependentSequence(*q->a, q->b.get(), q->map.p, q->map.n, 16));
      q->ref.assign((size_t)frames * 16, 0.f);
      q->run(frames, &q->ref);      // alone: start-up (allocations, the first bounding box) and the reference motions
      q->run(frames, nullptr);      // ... and once more: must repeat itself
      seqs.push_back(std::move(q));
    }
    printf("{\"frames_per_pass\": %d, \"passes\": %d, \"knn_reuse\": %d, \"n_target\": %d, \"n_source\": %d, \"host_threads_available\": %u, \"runs\": [", frames, reps, reuse,
           seqs[0]->map.n, seqs[0]->scans[0].n, std::thread::hardware_concurrency());
    bool first = true;
    for (int S : S_list) {
      Gate gate(S + 1);
      std::vector<std::thread> th;
      std::vector<double> per_seq_ms((size_t)S, 0.0);
      for (int s = 0; s < S; s++)
        th.emplace_back([&, s] {
          gate.wait();
          const auto t0 = std::chrono::steady_clock::now();
          for (int r = 0; r < reps; r++) seqs[(size_t)s]->run(frames, nullptr);
          rgc_synchronize(seqs[(size_t)s]->a->context());
          rgc_synchronize(seqs[(size_t)s]->b->context());
          per_seq_ms[(size_t)s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (reps * frames);
        });
      gate.wait();
      const auto t0 = std::chrono::steady_clock::now();
      for (auto& t : th) t.join();
      const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      bool same = true;
      double worst = 0, best = 1e30;
      for (int s = 0; s < S; s++) {
        same = same && seqs[(size_t)s]->same;
        worst = per_seq_ms[(size_t)s] > worst ? per_seq_ms[(size_t)s] : worst;
        best = per_seq_ms[(size_t)s] < best ? per_seq_ms[(size_t)s] : best;
      }
      printf("%s{\"S\": %d, \"aggregate_scans_per_s\": %.3f, \"ms_per_frame_per_sequence_min\": %.4f, \"ms_per_frame_per_sequence_max\": %.4f, "
             "\"poses_equal_each_sequence_alone\": %s}", first ? "" : ", ", S, (double)S * reps * frames / wall, best, worst, same ? "true" : "false");
      first = false;
    }
    printf("]}\n");
  } catch (const std::exception& e) {
    printf("{\"error\": \"%s\"}\n", e.what());
    return 1;
  }
  return 0;
}
