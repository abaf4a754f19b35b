// Drives the C++ adaptor exactly like RGC_odometer.cpp:998-1011 drives fast_gicp::FastVGICP, on clouds read from a
// raw float file written by the Python test; prints the final transformation, fitness and flags.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../rgc-slam_amd/cpp/fast_vgicp_hip.hpp"

struct PointXYZI { float x, y, z, pad0, intensity, pad1, pad2, pad3; };  // 32 bytes like pcl::PointXYZI
struct Cloud { std::vector<PointXYZI> points; size_t size() const { return points.size(); } };
struct Mat4 { float m[16]; float& operator()(int r, int c) { return m[r * 4 + c]; } float operator()(int r, int c) const { return m[r * 4 + c]; } };

static Cloud* load(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  int n = 0;
  if (fread(&n, 4, 1, f) != 1) exit(2);
  std::vector<float> xyz((size_t)n * 3);
  if (fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) exit(2);
  fclose(f);
  Cloud* c = new Cloud;
  c->points.resize(n);
  for (int i = 0; i < n; i++) { c->points[i] = PointXYZI{xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], 1.f, 0.f, 0, 0, 0}; }
  return c;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  Cloud* target = load(argv[1]);
  Cloud* source = load(argv[2]);
  Mat4 T2{};
  for (int i = 0; i < 4; i++) T2(i, i) = 1.f;
  try {
    rgc::FastVGICPHip vgicp;
    Cloud aligned;
    vgicp.setResolution(1.0);
    vgicp.setMaximumIterations(25);
    vgicp.setMaxCorrespondenceDistance(2);
    vgicp.setTransformationEpsilon(1e-6);
    vgicp.setEuclideanFitnessEpsilon(1e-6);
    vgicp.setRANSACIterations(0);
    vgicp.setNumThreads(14);
    vgicp.setInputTarget(target);
    vgicp.setInputSource(source);
    vgicp.align(aligned, T2);
    double score = vgicp.getFitnessScore();
    Mat4 T = vgicp.getFinalTransformation<Mat4>();
    printf("T");
    for (int i = 0; i < 16; i++) printf(" %.9g", T.m[i]);
    printf("\nfitness %.17g\nconverged %d iterations %d aligned %zu first %.9g %.9g %.9g\n", score, (int)vgicp.hasConverged(),
           vgicp.iterations(), aligned.points.size(), aligned.points[0].x, aligned.points[0].y, aligned.points[0].z);
    // fast_gicp.hpp:55-61 -- the rest of the coarse interface: covariances handed back in, roles swapped twice, clouds dropped
    const float* Tf = vgicp.getFinalTransformation();
    float T1[16];
    for (int i = 0; i < 16; i++) T1[i] = Tf[i];
    vgicp.setRegularizationMethod(rgc::RegularizationMethod::PLANE);
    vgicp.setVoxelAccumulationMode(rgc::VoxelAccumulationMode::ADDITIVE);
    vgicp.setSourceCovariances(vgicp.getSourceCovariances());
    vgicp.setTargetCovariances(vgicp.getTargetCovariances());
    vgicp.align(aligned, T2);
    int same = 1;
    for (int i = 0; i < 16; i++) same &= vgicp.getFinalTransformation()[i] == T1[i];
    vgicp.swapSourceAndTarget();
    vgicp.swapSourceAndTarget();
    vgicp.align(aligned, T2);
    for (int i = 0; i < 16; i++) same &= vgicp.getFinalTransformation()[i] == T1[i];
    int refused = 0;
    try { std::vector<double> bad((size_t)source->points.size() * 9, 0.25); vgicp.setSourceCovariances(bad); } catch (const std::exception&) { refused++; }
    // the reference's setters cannot fail: none throws here either.  Every RegularizationMethod / VoxelAccumulationMode is implemented (the
    // odometer's on the tuned kernels, the others on the general route); a change of route drops the clouds -- the reference would compute
    // covariances under the new method at align() -- so they are set again.  An invalid parameter is remembered and refused by the next call
    // that would compute something.
    int setters = 0;
    vgicp.setRegularizationMethod(rgc::RegularizationMethod::FROBENIUS);
    setters += vgicp.lastSetterStatus() == RGC_OK;
    try { vgicp.align(aligned, T2); } catch (const std::exception&) { setters++; }        // the clouds were dropped with the change of method
    vgicp.setVoxelAccumulationMode(rgc::VoxelAccumulationMode::MULTIPLICATIVE);
    setters += vgicp.lastSetterStatus() == RGC_OK;
    vgicp.setInputTarget(target);
    vgicp.setInputSource(source);
    vgicp.align(aligned, T2);                                                              // FROBENIUS + MULTIPLICATIVE: the general route
    float dmax = 0.f;
    for (int i = 0; i < 16; i++) { const float d = vgicp.getFinalTransformation()[i] - T1[i]; dmax = d > dmax ? d : (-d > dmax ? -d : dmax); }
    setters += dmax < 0.05f;                                                               // the same scan, the same map: about the same motion
    vgicp.setRegularizationMethod(rgc::RegularizationMethod::PLANE);
    vgicp.setVoxelAccumulationMode(rgc::VoxelAccumulationMode::ADDITIVE_WEIGHTED);         // = ADDITIVE in the vendored FastVGICP (fast_vgicp_voxel.hpp:137-141)
    vgicp.setResolution(-1.0);
    setters += vgicp.lastSetterStatus() == RGC_ERR_INVALID;
    try { vgicp.setInputTarget(target); } catch (const std::exception&) { setters++; }
    vgicp.setResolution(1.0);
    setters += vgicp.lastSetterStatus() == RGC_OK;
    vgicp.setInputTarget(target);
    vgicp.setInputSource(source);
    vgicp.align(aligned, T2);
    for (int i = 0; i < 16; i++) same &= vgicp.getFinalTransformation()[i] == T1[i];
    refused += setters == 7;
    vgicp.clearSource();
    try { vgicp.align(aligned, T2); } catch (const std::exception&) { refused++; }
    vgicp.setInputSource(source);
    vgicp.clearTarget();
    try { vgicp.align(aligned, T2); } catch (const std::exception&) { refused++; }
    printf("leftovers same %d refused %d\n", same, refused);
  } catch (const std::exception& e) {
    printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
