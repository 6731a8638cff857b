// issue_probe.hip -- what one SIMD of gfx950 issues per cycle for the integer / cross-lane / LDS
// instructions the traversal kernels are made of, at 1..8 waves per SIMD (diagnostic, GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/issue_probe.hip -o gpurun_out/issue_probe && gpurun_out/issue_probe
// Every wave runs REPS x 64 instructions of one kind on 8 independent chains and stamps s_memtime
// around the loop; the table prints shader cycles per wave-instruction per SIMD (wall cycles of the
// slowest wave x 1 / (instructions per wave x waves per SIMD)).  DESIGN.md section 6 quotes it as the
// issue ceiling of the VALU-bound k_pip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int REPS = 2000;

enum Op {
  OP_ADD, OP_CMP_CNDMASK, OP_OR3, OP_READLANE, OP_READFIRSTLANE, OP_BPERMUTE, OP_DPP, OP_MUL_LO, OP_MUL_HI,
  OP_MAD_U64, OP_ADD64, OP_LSHL64, OP_SALU, OP_DS_READ_B32, OP_DS_READ_B128, OP_DS_WRITE_B32, OP_BALLOT, OP_MBCNT, OP_COUNT
};
// wave-instructions (or pairs, where the name says so) per asm block
static const int kPerBlock[OP_COUNT] = {8, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 4, 4};
static const char* kNames[OP_COUNT] = {
    "v_add_u32", "v_cmp+v_cndmask (pair)", "v_or3_b32", "v_readlane_b32", "v_readfirstlane_b32", "ds_bpermute_b32",
    "v_mov_b32_dpp", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshl_add_u64", "v_lshlrev_b64", "s_add_u32",
    "ds_read_b32", "ds_read_b128", "ds_write_b32", "v_cmp + s_and (ballot use)", "v_mbcnt_lo+hi (pair)"};

template <int OP>
__global__ __launch_bounds__(256) void k_probe(unsigned long long* cycles, unsigned int* sink) {
  __shared__ unsigned int lds[256 * 4 + 64];
  const int t = threadIdx.x;
  for (int i = t; i < 256 * 4 + 64; i += 256) lds[i] = i;
  __syncthreads();
  unsigned int a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
  unsigned int b = t * 3 + 1;
  unsigned long long w0 = t, w1 = t + 9, w2 = t + 17, w3 = t + 5;
  unsigned int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  unsigned long long bm = ~0ull;
  uint4 q0 = make_uint4(t, t, t, t), q1 = q0;
  const unsigned int laddr = (unsigned int) (size_t) (&lds[0]) + (unsigned int) (t * 4);  // LDS byte address (low 32 bits of the generic pointer's offset)
  const unsigned int lbase = (unsigned int) (t * 16);
  (void) laddr;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == OP_ADD) {
        asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                     "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (OP == OP_CMP_CNDMASK) {
        asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_i32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                     "v_cmp_gt_i32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_gt_i32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
      } else if (OP == OP_OR3) {
        asm volatile("v_or3_b32 %0, %0, %8, %1\n v_or3_b32 %1, %1, %8, %2\n v_or3_b32 %2, %2, %8, %3\n v_or3_b32 %3, %3, %8, %4\n"
                     "v_or3_b32 %4, %4, %8, %5\n v_or3_b32 %5, %5, %8, %6\n v_or3_b32 %6, %6, %8, %7\n v_or3_b32 %7, %7, %8, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (OP == OP_READLANE) {
        asm volatile("v_readlane_b32 %0, %4, 7\n v_readlane_b32 %1, %5, 15\n v_readlane_b32 %2, %6, 23\n v_readlane_b32 %3, %7, 31\n"
                     "v_readlane_b32 %0, %4, 39\n v_readlane_b32 %1, %5, 47\n v_readlane_b32 %2, %6, 55\n v_readlane_b32 %3, %7, 63"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      } else if (OP == OP_READFIRSTLANE) {
        asm volatile("v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7\n"
                     "v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      } else if (OP == OP_BPERMUTE) {
        asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                     "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(lbase >> 2));
      } else if (OP == OP_DPP) {
        asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %4, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_ror:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (OP == OP_MUL_LO) {
        asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                     "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (OP == OP_MUL_HI) {
        asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                     "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (OP == OP_MAD_U64) {
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                     "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                     : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(b) : "vcc");
      } else if (OP == OP_ADD64) {
        asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                     "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
                     : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w0));
      } else if (OP == OP_LSHL64) {
        asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                     "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3"
                     : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
      } else if (OP == OP_SALU) {
        asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                     "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
      } else if (OP == OP_DS_READ_B32) {
        asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                     "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"((unsigned int) (t * 4)));
      } else if (OP == OP_DS_READ_B128) {
        asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n"
                     "ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(q0), "=v"(q1) : "v"(lbase));
      } else if (OP == OP_DS_WRITE_B32) {
        asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:4\n ds_write_b32 %8, %2 offset:8\n ds_write_b32 %8, %3 offset:12\n"
                     "ds_write_b32 %8, %4 offset:16\n ds_write_b32 %8, %5 offset:20\n ds_write_b32 %8, %6 offset:24\n ds_write_b32 %8, %7 offset:28"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"((unsigned int) (t * 4)));
      } else if (OP == OP_BALLOT) {
        asm volatile("v_cmp_gt_i32 vcc, %0, %5\n s_and_b64 %4, %4, vcc\n v_cmp_gt_i32 vcc, %1, %5\n s_and_b64 %4, %4, vcc\n"
                     "v_cmp_gt_i32 vcc, %2, %5\n s_and_b64 %4, %4, vcc\n v_cmp_gt_i32 vcc, %3, %5\n s_and_b64 %4, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(bm) : "v"(b) : "vcc");
      } else if (OP == OP_MBCNT) {
        asm volatile("v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %4, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n"
                     "v_mbcnt_lo_u32_b32 %2, %4, 0\n v_mbcnt_hi_u32_b32 %2, %4, %2\n v_mbcnt_lo_u32_b32 %3, %4, 0\n v_mbcnt_hi_u32_b32 %3, %4, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((t & 63) == 0) cycles[(blockIdx.x * 256 + t) >> 6] = t1 - t0;
  sink[blockIdx.x * 256 + t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned int) (w0 + w1 + w2 + w3) + s0 + s1 + s2 + s3 + (unsigned int) bm + q0.x + q1.y;
}

template <int OP>
static void run(int cus, unsigned long long* d_cycles, unsigned int* d_sink) {
  printf("%-28s", kNames[OP]);
  for (int wps = 1; wps <= 8; wps++) {
    const int blocks = cus * wps;  // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink);  // warm
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc((size_t) blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cycles, cyc.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    double sum = 0;
    for (auto c : cyc) {
      mx = c > mx ? c : mx;
      sum += (double) c;
    }
    const double per_wave = (double) REPS * 8 * kPerBlock[OP];  // instructions (or pairs) per wave
    // cycles per instruction per SIMD = mean wave lifetime / (instructions of the wps waves sharing the SIMD)
    printf(" %6.2f", sum / cyc.size() / (per_wave * wps));
    if (wps == 8) printf("   | slowest wave %.0f ticks in %.3f ms of wall = %.0f ticks/us", (double) mx, ms, (double) mx / (ms * 1e3));
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_cycles;
  unsigned int* d_sink;
  CHECK(hipMalloc(&d_cycles, (size_t) cus * 8 * 4 * 8));
  CHECK(hipMalloc(&d_sink, (size_t) cus * 8 * 256 * 4));
  printf("%s, %d CUs: shader cycles per wave-instruction per SIMD (s_memtime), by waves per SIMD\n", prop.gcnArchName, cus);
  printf("%-28s %6d %6d %6d %6d %6d %6d %6d %6d\n", "instruction", 1, 2, 3, 4, 5, 6, 7, 8);
  run<OP_ADD>(cus, d_cycles, d_sink);
  run<OP_CMP_CNDMASK>(cus, d_cycles, d_sink);
  run<OP_OR3>(cus, d_cycles, d_sink);
  run<OP_READLANE>(cus, d_cycles, d_sink);
  run<OP_READFIRSTLANE>(cus, d_cycles, d_sink);
  run<OP_BPERMUTE>(cus, d_cycles, d_sink);
  run<OP_DPP>(cus, d_cycles, d_sink);
  run<OP_MUL_LO>(cus, d_cycles, d_sink);
  run<OP_MUL_HI>(cus, d_cycles, d_sink);
  run<OP_MAD_U64>(cus, d_cycles, d_sink);
  run<OP_ADD64>(cus, d_cycles, d_sink);
  run<OP_LSHL64>(cus, d_cycles, d_sink);
  run<OP_SALU>(cus, d_cycles, d_sink);
  run<OP_DS_READ_B32>(cus, d_cycles, d_sink);
  run<OP_DS_READ_B128>(cus, d_cycles, d_sink);
  run<OP_DS_WRITE_B32>(cus, d_cycles, d_sink);
  run<OP_BALLOT>(cus, d_cycles, d_sink);
  run<OP_MBCNT>(cus, d_cycles, d_sink);
  return 0;
}
