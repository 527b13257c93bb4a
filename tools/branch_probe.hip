// What a TAKEN branch costs a wave that is alone on its SIMD (tools/issue_probe.hip prices straight-line code only; the ledger charged a branch 4.4 cycles).
// One workgroup of 4 waves on one CU, every pattern against the same number of plain v_fma_f32; time from wall_clock64 (100 MHz), reported per pattern element
// in units of one v_fma_f32 of the baseline (5.4 - 5.7 cycles: profiles/r06_issue_probe.txt).   hipcc --offload-arch=gfx950 -O3 -o branch_probe tools/branch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
#define R64(X) R16(X) R16(X) R16(X) R16(X)
#define FMA "v_fma_f32 %0, %0, %1, %2\n"
#define FMA4 FMA FMA FMA FMA
enum { K_FMA, K_FMA_NOT_TAKEN, K_FMA_TAKEN_NEXT, K_FMA_TAKEN_SKIP4, K_FMA_TAKEN_SKIP16, K_FMA_EXECZ_NOT_TAKEN, K_LOOP8, K_LOOP32, K_LOOP128, K_COUNT };
static const char* NAME[K_COUNT] = {
  "v_fma_f32 alone (baseline)", "v_fma + s_cbranch_scc0, NOT taken", "v_fma + s_cbranch_scc1 TAKEN to the next instruction", "v_fma + s_cbranch_scc1 TAKEN over 4 v_fma (32 B)",
  "v_fma + s_cbranch_scc1 TAKEN over 16 v_fma (128 B)", "v_fma + s_cbranch_execz, NOT taken", "loop: 8 v_fma + s_add + s_cmp + s_cbranch_scc1 back", "loop: 32 v_fma + ... back", "loop: 128 v_fma + ... back"};
static const int ELEMS[K_COUNT] = {64, 64, 64, 64, 64, 64, 64, 64, 64};     // elements per outer iteration
static const int FMAS[K_COUNT] = {1, 1, 1, 1, 1, 1, 8, 32, 128};            // v_fma executed per element

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, float b, float c) {
  float a = (float)threadIdx.x;
  unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    if constexpr (KIND == K_FMA) asm volatile(R64(FMA) : "+v"(a) : "v"(b), "v"(c));
    if constexpr (KIND == K_FMA_NOT_TAKEN) asm volatile("s_cmp_eq_u32 0, 0\n" R64(FMA "s_cbranch_scc0 9f\n") "9:\n" : "+v"(a) : "v"(b), "v"(c) : "scc");
    if constexpr (KIND == K_FMA_TAKEN_NEXT) asm volatile("s_cmp_eq_u32 0, 0\n" R64(FMA "s_cbranch_scc1 1f\n1:\n") : "+v"(a) : "v"(b), "v"(c) : "scc");
    if constexpr (KIND == K_FMA_TAKEN_SKIP4) asm volatile("s_cmp_eq_u32 0, 0\n" R64(FMA "s_cbranch_scc1 1f\n" FMA4 "1:\n") : "+v"(a) : "v"(b), "v"(c) : "scc");
    if constexpr (KIND == K_FMA_TAKEN_SKIP16) asm volatile("s_cmp_eq_u32 0, 0\n" R64(FMA "s_cbranch_scc1 1f\n" FMA4 FMA4 FMA4 FMA4 "1:\n") : "+v"(a) : "v"(b), "v"(c) : "scc");
    if constexpr (KIND == K_FMA_EXECZ_NOT_TAKEN) asm volatile(R64(FMA "s_cbranch_execz 9f\n") "9:\n" : "+v"(a) : "v"(b), "v"(c));
    if constexpr (KIND == K_LOOP8) asm volatile("s_mov_b32 s40, 0\n1:\n" FMA4 FMA4 "s_add_u32 s40, s40, 1\ns_cmp_lt_u32 s40, 64\ns_cbranch_scc1 1b\n" : "+v"(a) : "v"(b), "v"(c) : "scc", "s40");
    if constexpr (KIND == K_LOOP32) asm volatile("s_mov_b32 s40, 0\n1:\n" R4(FMA4 FMA4) "s_add_u32 s40, s40, 1\ns_cmp_lt_u32 s40, 64\ns_cbranch_scc1 1b\n" : "+v"(a) : "v"(b), "v"(c) : "scc", "s40");
    if constexpr (KIND == K_LOOP128) asm volatile("s_mov_b32 s40, 0\n1:\n" R16(FMA4 FMA4) "s_add_u32 s40, s40, 1\ns_cmp_lt_u32 s40, 64\ns_cbranch_scc1 1b\n" : "+v"(a) : "v"(b), "v"(c) : "scc", "s40");
  }
  unsigned long long t1 = wall_clock64();
  if (a == 12345.678f) out[1] = 1;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> double run(unsigned long long* d, int iters) {
  unsigned long long h = 0;
  probe<KIND><<<1, 256>>>(d, 10, 0.999f, 0.001f);
  probe<KIND><<<1, 256>>>(d, iters, 0.999f, 0.001f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
  return (double)h / iters / ELEMS[KIND];          // 100 MHz ticks per element
}
int main() {
  unsigned long long* d;
  CHECK(hipMalloc(&d, 16));
  const int iters = 4000;
  double t[K_COUNT];
  t[K_FMA] = run<K_FMA>(d, iters); t[K_FMA_NOT_TAKEN] = run<K_FMA_NOT_TAKEN>(d, iters); t[K_FMA_TAKEN_NEXT] = run<K_FMA_TAKEN_NEXT>(d, iters);
  t[K_FMA_TAKEN_SKIP4] = run<K_FMA_TAKEN_SKIP4>(d, iters); t[K_FMA_TAKEN_SKIP16] = run<K_FMA_TAKEN_SKIP16>(d, iters); t[K_FMA_EXECZ_NOT_TAKEN] = run<K_FMA_EXECZ_NOT_TAKEN>(d, iters);
  t[K_LOOP8] = run<K_LOOP8>(d, iters / 8); t[K_LOOP32] = run<K_LOOP32>(d, iters / 32); t[K_LOOP128] = run<K_LOOP128>(d, iters / 128);
  printf("# one workgroup of 4 waves on one CU (one wave per SIMD); unit = one v_fma_f32 of the baseline pattern\n");
  printf("%-62s %10s %14s\n", "pattern (per element)", "in v_fma", "branch extra");
  for (int k = 0; k < K_COUNT; k++)
    printf("%-62s %10.2f %14.2f\n", NAME[k], t[k] / t[K_FMA], t[k] / t[K_FMA] - FMAS[k]);
  return 0;
}
