// How fast does one gfx950 SIMD issue wave64 VALU instructions, and how many resident waves does it take to get there?
// (Round-3 review, "What's weak" #6: DESIGN.md assumed 4 cycles per wave64 instruction, MI355X_MICROARCH.md says 2.)
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/valu_issue_bench tools/valu_issue_bench.hip && tools/_build/valu_issue_bench
//
// Every wave runs ITERS x 64 instructions of one kind over 16 independent accumulator registers (a register is re-used every 16
// instructions, far beyond the ~4-8 cycle dependent latency), bracketed by s_memtime.  Two placements:
//   (A) ONE workgroup of 4 W waves on one CU (a workgroup's waves are dealt round the CU's four SIMDs): W = 1, 2, 4 waves per SIMD;
//   (B) the whole chip: 1024 W single-wave workgroups (the shape of the step kernels), W = 1 .. 8, wall time by hipEvents.
// Reported: shader cycles per instruction per WAVE (what a wave sees) and per SIMD (= per wave / W: the issue rate of the SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define OPS16 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])

#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define FMAC_DPP(i) "v_fmac_f32_dpp %" #i ", %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define ADD_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define MED3(i) "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define MUL(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n"
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define FMA_CHAIN(i) "v_fma_f32 %0, %0, %16, %17\n"
#define RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define MOV_DPP(i) "v_mov_b32_dpp %" #i ", %16 row_ror:4 row_mask:0xf bank_mask:0xf\n"

enum { K_FMA, K_FMAC_DPP, K_ADD_DPP, K_MED3, K_MUL, K_CNDMASK, K_PKMUL, K_PKFMA, K_FMA_CHAIN, K_RCP, K_MOV_DPP, K_MIX_SOLVER, K_COUNT };
static const char* KNAME[K_COUNT] = {"v_fma_f32", "v_fmac_f32_dpp row_newbcast", "v_add_f32_dpp quad_perm", "v_med3_f32", "v_mul_f32", "v_cndmask_b32 (vcc)",
                                    "v_pk_mul_f32", "v_pk_fma_f32", "v_fma_f32 dependent chain", "v_rcp_f32 (transcendental)", "v_mov_b32_dpp row_ror",
                                    "solver turn: med3, cndmask, s_nop, fmac_dpp"};
static const int FLOP_PER_LANE[K_COUNT] = {2, 2, 1, 0, 1, 0, 2, 4, 2, 1, 0, 0};

struct WaveRec { uint64_t cycles; uint32_t hw_id, xcc_id; };

template <int KIND>
__global__ __launch_bounds__(1024) void bench(WaveRec* out, int iters, float b, float c) {
  uint64_t t0, t1;
  if constexpr (KIND == K_PKMUL || KIND == K_PKFMA) {
    float2v a[16], bb = {b, b}, cc = {c, c};
    for (int i = 0; i < 16; i++) a[i] = float2v{(float)threadIdx.x + i, (float)i};
    asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; it++) {
      if constexpr (KIND == K_PKMUL) asm volatile(R16(PKMUL) R16(PKMUL) R16(PKMUL) R16(PKMUL) : OPS16 : "v"(bb), "v"(cc));
      else asm volatile(R16(PKFMA) R16(PKFMA) R16(PKFMA) R16(PKFMA) : OPS16 : "v"(bb), "v"(cc));
    }
    asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[0].cycles = 0;
  } else {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = (float)threadIdx.x + i;
    asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; it++) {
      if constexpr (KIND == K_FMA) asm volatile(R16(FMA) R16(FMA) R16(FMA) R16(FMA) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_FMAC_DPP) asm volatile(R16(FMAC_DPP) R16(FMAC_DPP) R16(FMAC_DPP) R16(FMAC_DPP) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_ADD_DPP) asm volatile(R16(ADD_DPP) R16(ADD_DPP) R16(ADD_DPP) R16(ADD_DPP) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_MED3) asm volatile(R16(MED3) R16(MED3) R16(MED3) R16(MED3) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_MUL) asm volatile(R16(MUL) R16(MUL) R16(MUL) R16(MUL) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_CNDMASK) asm volatile(R16(CNDMASK) R16(CNDMASK) R16(CNDMASK) R16(CNDMASK) : OPS16 : "v"(b), "v"(c) : "vcc");
      if constexpr (KIND == K_FMA_CHAIN) asm volatile(R16(FMA_CHAIN) R16(FMA_CHAIN) R16(FMA_CHAIN) R16(FMA_CHAIN) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_RCP) asm volatile(R16(RCP) R16(RCP) R16(RCP) R16(RCP) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_MOV_DPP) asm volatile(R16(MOV_DPP) R16(MOV_DPP) R16(MOV_DPP) R16(MOV_DPP) : OPS16 : "v"(b), "v"(c));
      if constexpr (KIND == K_MIX_SOLVER) {
        // the Gauss-Seidel turn of csrc/pmc_step.hpp: clamp, select, one wait state, broadcast multiply-add -- x16 = 64 issue slots
#define TURN(i) "v_med3_f32 %1, %0, %16, %17\nv_cndmask_b32 %2, %2, %1, vcc\ns_nop 0\nv_fmac_f32_dpp %0, %1, %3 row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n"
        asm volatile(TURN(0) TURN(1) TURN(2) TURN(3) TURN(4) TURN(5) TURN(6) TURN(7) TURN(8) TURN(9) TURN(10) TURN(11) TURN(12) TURN(13) TURN(14) TURN(15)
                     : OPS16 : "v"(b), "v"(c) : "vcc");
      }
    }
    asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    if (s == 12345.678f) out[0].cycles = 0;
  }
  if ((threadIdx.x & 63) == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    WaveRec r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
    out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
  }
}

typedef void (*KernelFn)(WaveRec*, int, float, float);
static KernelFn FN[K_COUNT] = {bench<K_FMA>, bench<K_FMAC_DPP>, bench<K_ADD_DPP>, bench<K_MED3>, bench<K_MUL>, bench<K_CNDMASK>, bench<K_PKMUL>, bench<K_PKFMA>,
                               bench<K_FMA_CHAIN>, bench<K_RCP>, bench<K_MOV_DPP>, bench<K_MIX_SOLVER>};

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("# %s, %d CUs, clockRate %d kHz; %d iterations x 64 instructions per wave\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, iters);
  WaveRec* d;
  const int MAXW = 8 * 1024 * 2;
  CHECK(hipMalloc(&d, sizeof(WaveRec) * MAXW));
  std::vector<WaveRec> h(MAXW);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double ninst = (double)iters * 64.0;
  printf("\n## (A) one workgroup of 4 W waves on one CU: shader cycles per instruction\n");
  printf("%-44s %18s %18s %18s\n", "instruction", "W=1 wave | SIMD", "W=2 wave | SIMD", "W=4 wave | SIMD");
  for (int k = 0; k < K_COUNT; k++) {
    printf("%-44s", KNAME[k]);
    for (int W : {1, 2, 4}) {
      const int waves = 4 * W;
      for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(FN[k], dim3(1), dim3(64 * waves), 0, 0, d, iters, 1.0001f, 1e-6f); }
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), d, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost));
      double mx = 0;
      for (int w = 0; w < waves; w++) mx = std::max(mx, (double)h[w].cycles);
      printf("   %6.2f | %6.2f ", mx / ninst, mx / ninst / W);
    }
    printf("\n");
  }
  printf("\n## (B) the whole chip, 1024 W single-wave workgroups: cycles per instruction per wave (median / max over waves), per SIMD, wall time, and the\n"
         "##     chip-level rate; 'waves per SIMD seen' = how the dispatcher spread them (from HW_ID: max waves that shared one SIMD)\n");
  for (int k : {K_FMA, K_PKFMA, K_FMAC_DPP, K_MIX_SOLVER}) {
    for (int W : {1, 2, 3, 4, 6, 8}) {
      const int blocks = 1024 * W;
      hipLaunchKernelGGL(FN[k], dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 1e-6f);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(FN[k], dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 1e-6f);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h.data(), d, sizeof(WaveRec) * blocks, hipMemcpyDeviceToHost));
      std::vector<double> c(blocks);
      std::vector<uint64_t> key(blocks);
      for (int w = 0; w < blocks; w++) {
        c[w] = (double)h[w].cycles;
        const uint32_t hw = h[w].hw_id;       // gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        key[w] = ((uint64_t)(h[w].xcc_id & 0xf) << 32) | (hw & 0xfff0 & ~0xc0u);
      }
      std::sort(c.begin(), c.end());
      std::sort(key.begin(), key.end());
      int maxshare = 0, run = 0, distinct = 0;
      for (int w = 0; w < blocks; w++) { run = (w && key[w] == key[w - 1]) ? run + 1 : 1; if (run == 1) distinct++; maxshare = std::max(maxshare, run); }
      const double med = c[blocks / 2] / ninst, mx = c[blocks - 1] / ninst;
      const double tot = ninst * blocks;
      printf("%-30s W=%d: wave %6.2f / %6.2f  SIMD %5.2f   wall %8.3f ms  -> %7.1f G wave-instr/s = %5.2f per SIMD-ns; %6.1f TFLOP/s; SIMDs used %d, max sharing %d\n", KNAME[k], W, med, mx,
             med / W, ms, tot / ms * 1e-6, tot / ms * 1e-6 / 1024.0, tot * 64.0 * FLOP_PER_LANE[k] / ms * 1e-9, distinct, maxshare);
    }
  }
  return 0;
}
