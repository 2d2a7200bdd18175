// Micro-benchmark (measurement tooling, not product code): issue cost of the VALU instruction classes the
// composite kernels are made of, on gfx950.  Each test runs REPS x 64 independent instructions per wave,
// with 1..8 waves per SIMD, and reports SIMD cycles per wave-instruction (s_memtime) and wall-clock rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int REPS = 2000;

// 8 independent chains x 8 = 64 instructions per loop body
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY64(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP)

#define KERNEL_F(NAME, OP)                                                                              \
  __global__ void NAME(float *out, long long *cyc, float s) {                                           \
    float a[8], b = s, c = s * 0.5f;                                                                    \
    const unsigned long long m64 = s > 0.f ? 0x5555555555555555ull : 0xffffull; (void)m64;              \
    for (int i = 0; i < 8; i++) a[i] = s + threadIdx.x * 1e-3f + i;                                     \
    long long t0 = __builtin_readcyclecounter();                                                        \
    for (int r = 0; r < REPS; r++) { BODY64(OP) }                                                       \
    long long t1 = __builtin_readcyclecounter();                                                        \
    float acc = 0; for (int i = 0; i < 8; i++) acc += a[i];                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                    \
  }

#define KERNEL_F2(NAME, OP)                                                                             \
  __global__ void NAME(float *out, long long *cyc, float s) {                                           \
    f2 a[8], b = {s, s * 1.1f}, c = {s * 0.5f, s * 0.4f};                                               \
    for (int i = 0; i < 8; i++) a[i] = f2{s + threadIdx.x * 1e-3f + i, s - i};                          \
    long long t0 = __builtin_readcyclecounter();                                                        \
    for (int r = 0; r < REPS; r++) { BODY64(OP) }                                                       \
    long long t1 = __builtin_readcyclecounter();                                                        \
    float acc = 0; for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                    \
  }

#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define OP_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define OP_CMP(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define OP_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m64));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define OP_ADDABS(i) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
#define OP_FMAABS(i) asm volatile("v_fma_f32 %0, |%0|, |%1|, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define OP_SWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
// mixes: one transcendental per 8 slots among FMAs (does the transcendental unit overlap with the FMA pipe?)
#define OP_MIX_EXP(i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(b), "v"(c));

KERNEL_F(k_fma, OP_FMA)
KERNEL_F(k_mul, OP_MUL)
KERNEL_F(k_add, OP_ADD)
KERNEL_F(k_exp, OP_EXP)
KERNEL_F(k_rcp, OP_RCP)
KERNEL_F(k_min, OP_MIN)
KERNEL_F(k_cnd, OP_CND)
KERNEL_F(k_cmp, OP_CMP)
KERNEL_F(k_cnd64, OP_CND64)
KERNEL_F(k_mov, OP_MOV)
KERNEL_F(k_addabs, OP_ADDABS)
KERNEL_F(k_fmaabs, OP_FMAABS)
KERNEL_F(k_and, OP_AND)
KERNEL_F(k_dpp, OP_DPP)
KERNEL_F(k_swap, OP_SWAP)
KERNEL_F(k_mixexp, OP_MIX_EXP)
KERNEL_F2(k_pkfma, OP_PKFMA)
KERNEL_F2(k_pkmul, OP_PKMUL)
KERNEL_F2(k_pkadd, OP_PKADD)

// LDS broadcast reads (every lane reads the same 16 bytes), as in the composite inner loops
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_ldsb128(float *out, long long *cyc, float s) {
  __shared__ float4 sm[64];
  sm[threadIdx.x & 63] = make_float4(s, s, s, s);
  __syncthreads();
  f4 acc = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int t = 0; t < 64; t += 4) {
      f4 v0, v1, v2, v3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"((unsigned)(t * 16)));
      acc += v0 + v1 + v2 + v3;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// float atomics: 12 lanes of a wave add to one 64-byte record (AoS) vs five separate arrays (SoA), random records
__global__ void k_atomic_aos(float *rec, const int *ids, int n, int per_wave) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int i = 0; i < per_wave; i++) {
    const int g = ids[(w * per_wave + i) % n];
    if (lane < 12) atomicAdd(rec + (long)g * 16 + lane, 1.0f);
  }
}
__global__ void k_atomic_soa(float *col, float *con, float *m2, float *ab, float *op, const int *ids, int n, int per_wave) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int i = 0; i < per_wave; i++) {
    const long g = ids[(w * per_wave + i) % n];
    float *p = nullptr;
    if (lane < 4) p = col + g * 4 + lane;
    else if (lane < 7) p = con + g * 3 + (lane - 4);
    else if (lane < 9) p = m2 + g * 2 + (lane - 7);
    else if (lane < 11) p = ab + g * 2 + (lane - 9);
    else if (lane == 11) p = op + g;
    if (p) atomicAdd(p, 1.0f);
  }
}

template <typename K>
static void run(const char *name, K kern, int insts_per_rep, double flop_per_inst_lane) {
  float *out; long long *cyc;
  const int cus = 256;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 4 * 8 * 64 * 2));
  CHECK(hipMalloc(&cyc, sizeof(long long) * cus * 8));
  for (int wps : {1, 2, 4, 8}) {               // waves per SIMD (4 SIMDs per CU): block = 256 threads = 1 wave per SIMD
    const int blocks = cus * wps;              // wps blocks of 256 threads per CU
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double n_inst = (double)REPS * insts_per_rep;                 // per wave
    const double wave_insts_total = n_inst * blocks * 4;                // 4 waves per block
    const double per_simd_inst_per_us = wave_insts_total / (cus * 4) / (ms * 1e3);
    printf("%-10s waves/SIMD=%d  memtime-ticks/inst/wave=%7.3f  wall: %8.1f wave-inst/us/SIMD  (%.2f ns per wave-inst per SIMD)  %.1f TFLOP/s-equiv\n",
           name, wps, mean / n_inst, per_simd_inst_per_us, 1e3 / per_simd_inst_per_us,
           wave_insts_total * 64 * flop_per_inst_lane / (ms * 1e-3) / 1e12);
  }
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

__global__ void k_rcp_bits(unsigned *out, float x) {
  float r;
  asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x));
  float e;
  asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x - 1.0f));
  out[0] = __float_as_uint(r); out[1] = __float_as_uint(e);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  {
    unsigned *d, h[2];
    CHECK(hipMalloc(&d, 8));
    hipLaunchKernelGGL(k_rcp_bits, dim3(1), dim3(64), 0, 0, d, 1.0f);
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    printf("v_rcp_f32(1.0) = 0x%08x (exact: 0x3f800000)   v_exp_f32(0.0) = 0x%08x\n", h[0], h[1]);
  }
  run("v_fma", k_fma, 64, 2);
  run("v_pk_fma", k_pkfma, 64, 4);
  run("v_mul", k_mul, 64, 1);
  run("v_pk_mul", k_pkmul, 64, 2);
  run("v_add", k_add, 64, 1);
  run("v_pk_add", k_pkadd, 64, 2);
  run("v_exp", k_exp, 64, 1);
  run("v_rcp", k_rcp, 64, 1);
  run("v_min", k_min, 64, 1);
  run("v_cndmask", k_cnd, 64, 1);
  run("v_cmp", k_cmp, 64, 1);
  run("cndmask_s", k_cnd64, 64, 1);
  run("v_mov", k_mov, 64, 1);
  run("add_abs_e64", k_addabs, 64, 1);
  run("fma_abs", k_fmaabs, 64, 2);
  run("v_and", k_and, 64, 1);
  run("add_dpp", k_dpp, 64, 1);
  run("perm32swap", k_swap, 64, 1);
  run("exp+3fma", k_mixexp, 64 * 4, 1);
  run("lds_b128bc", k_ldsb128, 64, 0);
  // atomics
  {
    const int n = 300000, per_wave = 256, waves = 8160 * 2;
    std::vector<int> ids(n);
    unsigned s = 12345; for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; ids[i] = (int)(s % n); }
    int *d_ids; float *rec, *col, *con, *m2, *ab, *op;
    CHECK(hipMalloc(&d_ids, 4 * n)); CHECK(hipMemcpy(d_ids, ids.data(), 4 * n, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&rec, 4L * n * 16)); CHECK(hipMalloc(&col, 4L * n * 4)); CHECK(hipMalloc(&con, 4L * n * 3));
    CHECK(hipMalloc(&m2, 4L * n * 2)); CHECK(hipMalloc(&ab, 4L * n * 2)); CHECK(hipMalloc(&op, 4L * n));
    CHECK(hipMemset(rec, 0, 4L * n * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_atomic_aos, dim3(waves), dim3(64), 0, 0, rec, d_ids, n, per_wave);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("atomic AoS (12 lanes -> one 64-B record): %.1f us for %.2f M records  (%.1f ns/record/chip)\n", ms * 1e3, waves * (double)per_wave / 1e6, ms * 1e6 / (waves * (double)per_wave));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_atomic_soa, dim3(waves), dim3(64), 0, 0, col, con, m2, ab, op, d_ids, n, per_wave);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("atomic SoA (12 lanes -> five arrays)    : %.1f us for %.2f M records  (%.1f ns/record/chip)\n", ms * 1e3, waves * (double)per_wave / 1e6, ms * 1e6 / (waves * (double)per_wave));
    }
  }
  return 0;
}
