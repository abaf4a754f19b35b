// valu_issue.hip -- micro-benchmark behind the "VALU issue roof" of bench.py / DESIGN.md (MI355X, gfx950).
// Streams of INDEPENDENT vector instructions of one kind (8 accumulator registers, no dependency closer than 8 instructions apart),
// W waves per SIMD (one 256-thread workgroup = one wave per SIMD; W workgroups per CU, enforced through the LDS allocation),
// every CU busy.  Cycles come from s_memtime inside the kernel (shader clock), so the result is cycles per wave-instruction per
// SIMD, independent of DVFS:   cycles_per_instr_per_simd = elapsed_cycles / (W * instructions per wave).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; ./valu_issue > profiles/r02_valu_issue.jsonl
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 8;     // 8 x 8 = 64 instructions per loop trip
constexpr int kTrips = 4096;

#define OP8(ASM)                                                                                               \
  asm volatile(ASM(0) "\n\t" ASM(1) "\n\t" ASM(2) "\n\t" ASM(3) "\n\t" ASM(4) "\n\t" ASM(5) "\n\t" ASM(6) "\n\t" ASM(7) \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
               : "v"(x), "v"(y) : "vcc", "s10", "s11");

#define FMA_F32(i) "v_fma_f32 %" #i ", %8, %9, %" #i
#define MIN_I32(i) "v_min_i32 %" #i ", %" #i ", %8"
#define MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %8"
#define MED3_I32(i) "v_med3_i32 %" #i ", %8, %" #i ", %9"
#define ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8"
#define AND_OR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9"
#define LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8"
#define MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %8"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define MIN_F32(i) "v_min_f32 %" #i ", %" #i ", %8"
#define MAX_F32(i) "v_max_f32 %" #i ", %" #i ", %8"
#define MED3_F32(i) "v_med3_f32 %" #i ", %8, %" #i ", %9"
#define MIN3_F32(i) "v_min3_f32 %" #i ", %8, %" #i ", %9"
#define SUB_F32(i) "v_sub_f32 %" #i ", %" #i ", %8"
#define OR_B32(i) "v_or_b32 %" #i ", %" #i ", %8"
#define LSHLREV(i) "v_lshlrev_b32 %" #i ", 1, %" #i
#define BFI_B32(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9"
#define ADD3_U32(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define MIN_U32(i) "v_min_u32 %" #i ", %" #i ", %8"
#define CNDMASK_S(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]"
#define CMP_ADDC(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n\tv_addc_co_u32 %" #i ", vcc, %" #i ", %9, vcc"
#define MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
#define MOV_B32(i) "v_mov_b32 %" #i ", %8"
#define PK_MIN_I16(i) "v_pk_min_i16 %" #i ", %" #i ", %8"
#define CMP_LT(i) "v_cmp_lt_i32 s[10:11], %" #i ", %8"
#define AND_B32(i) "v_and_b32 %" #i ", %" #i ", %8"
#define SUB_U32(i) "v_sub_u32 %" #i ", %" #i ", %8"
#define FMAC_F32(i) "v_fmac_f32 %" #i ", %8, %9"
#define MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8"
#define CMP_CND(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc"
#define XAD_U32(i) "v_xad_u32 %" #i ", %" #i ", %8, %9"
#define MAX3_I32(i) "v_max3_i32 %" #i ", %" #i ", %8, %9"

template <int OP>
__global__ void __launch_bounds__(256) k_stream(unsigned long long* cycles, int* sink) {
  extern __shared__ int lds_pad[];
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const int x = 0x3f800001 + (int)blockIdx.x, y = 0x3f000000 + (int)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < kTrips; t++) {
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      if (OP == 0) { OP8(FMA_F32) }
      if (OP == 1) { OP8(MIN_I32) }
      if (OP == 2) { OP8(MAX_I32) }
      if (OP == 3) { OP8(MED3_I32) }
      if (OP == 4) { OP8(ADD_U32) }
      if (OP == 5) { OP8(AND_OR) }
      if (OP == 6) { OP8(LSHL_ADD) }
      if (OP == 7) { OP8(MUL_F32) }
      if (OP == 8) { OP8(CNDMASK) }
      if (OP == 9) { OP8(MIN_F32) }
      if (OP == 10) { OP8(MAX_F32) }
      if (OP == 11) { OP8(MED3_F32) }
      if (OP == 12) { OP8(MIN3_F32) }
      if (OP == 13) { OP8(SUB_F32) }
      if (OP == 14) { OP8(OR_B32) }
      if (OP == 15) { OP8(LSHLREV) }
      if (OP == 16) { OP8(BFI_B32) }
      if (OP == 17) { OP8(ADD3_U32) }
      if (OP == 18) { OP8(MIN_U32) }
      if (OP == 19) { OP8(CNDMASK_S) }
      if (OP == 20) { OP8(CMP_ADDC) }
      if (OP == 21) { OP8(MAD_U24) }
      if (OP == 22) { OP8(MOV_B32) }
      if (OP == 23) { OP8(PK_MIN_I16) }
      if (OP == 24) { OP8(CMP_LT) }
      if (OP == 25) { OP8(AND_B32) }
      if (OP == 26) { OP8(SUB_U32) }
      if (OP == 27) { OP8(FMAC_F32) }
      if (OP == 28) { OP8(MUL_LO) }
      if (OP == 29) { OP8(CMP_CND) }
      if (OP == 30) { OP8(XAD_U32) }
      if (OP == 31) { OP8(MAX3_I32) }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678) sink[0] = lds_pad[0];
}

// fp64 and packed-fp32 streams need register PAIRS: written with builtins on doubles / float2 (the compiler keeps them independent)
template <int OP>
__global__ void __launch_bounds__(256) k_stream64(unsigned long long* cycles, double* sink) {
  extern __shared__ int lds_pad[];
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-9 * (double)(threadIdx.x + i);
  const double x = 1.0 + 1e-12 * (double)blockIdx.x, y = 1e-13 * (double)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < kTrips; t++) {
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(y));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
        if (OP == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i];
  if (s == 0.123456789) sink[0] = s + lds_pad[0];
}

struct Case { const char* name; void (*fn)(unsigned long long*, int*); void (*fn64)(unsigned long long*, double*); };

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const Case cases[] = {
      {"v_fma_f32", k_stream<0>, nullptr},     {"v_min_i32", k_stream<1>, nullptr},      {"v_max_i32", k_stream<2>, nullptr},
      {"v_med3_i32", k_stream<3>, nullptr},    {"v_add_u32", k_stream<4>, nullptr},      {"v_and_or_b32", k_stream<5>, nullptr},
      {"v_lshl_add_u32", k_stream<6>, nullptr}, {"v_mul_f32", k_stream<7>, nullptr},     {"v_cndmask_b32", k_stream<8>, nullptr},
      {"v_min_f32", k_stream<9>, nullptr},     {"v_max_f32", k_stream<10>, nullptr},     {"v_med3_f32", k_stream<11>, nullptr},
      {"v_min3_f32", k_stream<12>, nullptr},   {"v_sub_f32", k_stream<13>, nullptr},     {"v_or_b32", k_stream<14>, nullptr},
      {"v_lshlrev_b32", k_stream<15>, nullptr}, {"v_bfi_b32", k_stream<16>, nullptr},    {"v_add3_u32", k_stream<17>, nullptr},
      {"v_min_u32", k_stream<18>, nullptr},    {"v_cndmask_b32 (sgpr mask)", k_stream<19>, nullptr},
      {"v_cmp_lt_i32 + v_addc_co_u32 (2 instr)", k_stream<20>, nullptr}, {"v_mad_u32_u24", k_stream<21>, nullptr},
      {"v_mov_b32", k_stream<22>, nullptr},    {"v_pk_min_i16", k_stream<23>, nullptr},
      {"v_cmp_lt_i32 (to sgpr pair)", k_stream<24>, nullptr}, {"v_and_b32", k_stream<25>, nullptr}, {"v_sub_u32", k_stream<26>, nullptr},
      {"v_fmac_f32", k_stream<27>, nullptr},   {"v_mul_lo_u32", k_stream<28>, nullptr},
      {"v_cmp_lt_i32 + v_cndmask_b32 vcc (2 instr)", k_stream<29>, nullptr}, {"v_xad_u32", k_stream<30>, nullptr}, {"v_max3_i32", k_stream<31>, nullptr},
      {"v_fma_f64", nullptr, k_stream64<0>},   {"v_add_f64", nullptr, k_stream64<1>},    {"v_pk_fma_f32", nullptr, k_stream64<2>},
      {"v_pk_add_f32", nullptr, k_stream64<3>}, {"v_mul_f64", nullptr, k_stream64<4>},
  };
  unsigned long long* d_cycles;
  int* d_sink;
  const int max_blocks = cus * 8;
  CHK(hipMalloc(&d_cycles, sizeof(unsigned long long) * max_blocks * 4));
  CHK(hipMalloc(&d_sink, 64));
  std::vector<unsigned long long> h(max_blocks * 4);
  hipEvent_t ev0, ev1;
  CHK(hipEventCreate(&ev0));
  CHK(hipEventCreate(&ev1));
  const double instr_per_wave = (double)kTrips * kUnroll * 8;
  for (const Case& c : cases) {
    for (int W : {1, 2, 4, 8}) {
      // W workgroups (one wave per SIMD each) per CU: LDS per workgroup = floor(160 KiB / W) keeps a (W+1)-th one out
      const size_t lds = (size_t)(160 * 1024 / W) - (W == 1 ? 0 : 512);
      const void* f = c.fn ? (const void*)c.fn : (const void*)c.fn64;
      CHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int blocks = cus * W;
      float ms = 0.f;
      for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(ev0, 0));
        if (c.fn) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), lds, 0, d_cycles, d_sink);
        else hipLaunchKernelGGL(c.fn64, dim3(blocks), dim3(256), lds, 0, d_cycles, (double*)d_sink);
        CHK(hipEventRecord(ev1, 0));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, ev0, ev1));
      }
      CHK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> v(h.begin(), h.begin() + blocks * 4);
      std::sort(v.begin(), v.end());
      const double med = (double)v[v.size() / 2];
      // all W waves of a SIMD run the whole time (same program, same start): the SIMD issues W * instr_per_wave in `med` cycles
      printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_one_wave\": %.3f, \"cycles_per_instr_per_simd\": %.3f, "
             "\"kernel_ms\": %.4f, \"memtime_ticks_per_us\": %.1f, \"wave_instr_per_s_chip\": %.4g}\n",
             c.name, W, med / instr_per_wave, med / (instr_per_wave * W), ms, med / (ms * 1e3),
             (double)cus * 4 * W * instr_per_wave / (ms * 1e-3));
      fflush(stdout);
    }
  }
  return 0;
}
