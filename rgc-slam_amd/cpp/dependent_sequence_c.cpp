// librgc_seq.so -- the odometer's dependent frame loop (rgc::DependentSequence, fast_vgicp_hip.hpp; RGC_odometer.cpp:976-1023, 1201-1203,
// 1248-1256) behind one extern "C" call, for callers that hold rgc_ctx handles of their own (bench.py's timed region, via ctypes): the K
// frames of a sequence run in C++, the reference's host language, with nothing of the caller's interpreter between a frame's result and the
// next frame's first launch.  Same calls in the same order as rgc::DependentSequence::run -- which it is checked against, bit for bit
// (tests/test_gpu_sequence.py).  Plain C-ABI over the plain C-ABI: pointers and sizes only.
#include <chrono>
#include <cstring>

#include "fast_vgicp_hip.hpp"

extern "C" {

// a, b: two contexts taking turns (b == NULL: one frame at a time on a).  d_map: the world-frame map (n_map points, stride_bytes), scratch_a /
// scratch_b: n_map * 16 bytes each on the device.  d_scans[i], n_scans[i]: frame i's scan on the device (scan_stride_bytes).
// world_T: in the world pose before the first frame, out the pose after the last (row-major 4x4 fp64).  guess0: the first frame's guess;
// every later frame starts from the previous frame's motion.  Outputs (any may be NULL): motions n_frames x 16 floats, worlds n_frames x 16
// doubles, fitness / iterations per frame, stamps = seconds since the call started at which each frame's result was in the host's hands.
// Returns RGC_OK or the failing call's status (rgc_last_error of the context that failed).
RGC_API int rgc_seq_run_dependent(rgc_ctx* a, rgc_ctx* b, const float* d_map, int n_map, int stride_bytes, float* scratch_a, float* scratch_b,
                                  const float* const* d_scans, const int* n_scans, int scan_stride_bytes, int n_frames, double world_T[16],
                                  const float guess0[16], int want_fitness, float* motions, double* worlds, double* fitness, int* iterations,
                                  double* stamps) {
  if (!a || !d_map || !scratch_a || (b && !scratch_b) || !d_scans || !n_scans || !world_T || !guess0 || n_frames < 0) return RGC_ERR_INVALID;
  const auto t0 = std::chrono::steady_clock::now();
  rgc_ctx* regs[2] = {a, b ? b : a};
  float* scratch[2] = {scratch_a, b ? scratch_b : scratch_a};
  const int D = b ? 2 : 1;
  float g[16];
  std::memcpy(g, guess0, sizeof(g));
  int rc;
  if (D == 2 && n_frames > 0 && (rc = rgc_set_source_device(regs[0], d_scans[0], n_scans[0], scan_stride_bytes))) return rc;
  double q[4], t[3];
  rgc::DependentSequence::worldToBody(world_T, q, t);
  if (n_frames > 0 && (rc = rgc_set_target_reframed(regs[0], d_map, n_map, stride_bytes, q, t, scratch[0]))) return rc;
  for (int i = 0; i < n_frames; i++) {
    rgc_ctx* cur = regs[i % D];
    rgc_ctx* nxt = regs[(i + 1) % D];
    if (D == 1 && (rc = rgc_set_source_device(cur, d_scans[i], n_scans[i], scan_stride_bytes))) return rc;
    if ((rc = rgc_align_begin(cur, g, want_fitness))) return rc;
    if (D == 2 && i + 1 < n_frames) {
      if ((rc = rgc_hold_source_until_target_of(nxt, cur))) return rc;
      if ((rc = rgc_set_source_device(nxt, d_scans[i + 1], n_scans[i + 1], scan_stride_bytes))) return rc;
    }
    float T[16];
    double fit = 0.0;
    int it = 0, conv = 0, fail = 0;
    if (i + 1 < n_frames) {
      if ((rc = rgc_align_end_reframe(cur, nxt, world_T, d_map, n_map, stride_bytes, scratch[(i + 1) % D], T, nullptr, want_fitness ? &fit : nullptr, &it, &conv, &fail)))
        return rc;
    } else {
      if ((rc = rgc_align_end(cur, T, nullptr, want_fitness ? &fit : nullptr, &it, &conv, &fail))) return rc;
      double W[16];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
          double v = 0.0;
          for (int k = 0; k < 4; k++) v += world_T[r * 4 + k] * (double)T[k * 4 + c];
          W[r * 4 + c] = v;
        }
      std::memcpy(world_T, W, sizeof(W));
    }
    if (stamps) stamps[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (motions) std::memcpy(motions + (size_t)i * 16, T, sizeof(T));
    if (worlds) std::memcpy(worlds + (size_t)i * 16, world_T, 16 * sizeof(double));
    if (fitness) fitness[i] = fit;
    if (iterations) iterations[i] = it;
    std::memcpy(g, T, sizeof(g));
  }
  return RGC_OK;
}

}  // extern "C"
