// mfma_gram_probe.hip -- is the MFMA form of the solver's Gram block (lanes.hpp gram16) the same numbers as the DPP form (gram4)?
//   g_lane[L] = sum_i y_lane[i] * x_{lane L of my 16-lane row}[i],  i < K, for the four rows (envs) of a wavefront at once:
//   v_mfma_f32_16x16x1_4b_f32 (four independent 16x16 outer products = the four env rows), A = x (lane i of a row holds x_i[k]), B = y (lane j holds y_j[k]):
//   D_b[i][j] = x_i[k] y_j[k] lands in lane 16 (i / 4) + j, register 4 b + i % 4 -- lane j of row-group g holds g_{b,j}[4 g + r]: the owner's own entries, in
//   the wrong row group; a 4 x 4 block transpose between register block and row group (8 v_permlane32_swap + 8 v_permlane16_swap) brings them home.
// hipcc --offload-arch=gfx950 -O3 -o _build/mfma_gram_probe mfma_gram_probe.hip && _build/mfma_gram_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f16v __attribute__((ext_vector_type(16)));
#define K 6
__global__ void probe(const float* x, const float* y, float* out_ref, float* out_mfma) {
  const int lane = threadIdx.x, row = lane >> 4;
  float xs[K], ys[K];
  for (int i = 0; i < K; i++) { xs[i] = x[lane * K + i]; ys[i] = y[lane * K + i]; }
  // reference by shuffles
  for (int L = 0; L < 16; L++) {
    float g = 0.0f;
    for (int i = 0; i < K; i++) g += ys[i] * __shfl(xs[i], 16 * row + L);
    out_ref[lane * 16 + L] = g;
  }
  f16v acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  for (int i = 0; i < K; i++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(xs[i], ys[i], acc, 0, 0, 0);
  unsigned R[16];
  for (int i = 0; i < 16; i++) R[i] = __float_as_uint(acc[i]);
  for (int r = 0; r < 4; r++) {
    auto a = __builtin_amdgcn_permlane32_swap(R[r], R[8 + r], false, false); R[r] = a[0]; R[8 + r] = a[1];
    auto b = __builtin_amdgcn_permlane32_swap(R[4 + r], R[12 + r], false, false); R[4 + r] = b[0]; R[12 + r] = b[1];
  }
  for (int r = 0; r < 4; r++) {
    auto a = __builtin_amdgcn_permlane16_swap(R[r], R[4 + r], false, false); R[r] = a[0]; R[4 + r] = a[1];
    auto b = __builtin_amdgcn_permlane16_swap(R[8 + r], R[12 + r], false, false); R[8 + r] = b[0]; R[12 + r] = b[1];
  }
  for (int i = 0; i < 16; i++) out_mfma[lane * 16 + i] = __uint_as_float(R[i]);
}
int main() {
  float hx[64 * K], hy[64 * K], r0[1024], r1[1024];
  srand(3);
  for (int i = 0; i < 64 * K; i++) { hx[i] = rand() / (float)RAND_MAX - 0.5f; hy[i] = rand() / (float)RAND_MAX - 0.5f; }
  float *dx, *dy, *d0, *d1;
  hipMalloc(&dx, sizeof hx); hipMalloc(&dy, sizeof hy); hipMalloc(&d0, sizeof r0); hipMalloc(&d1, sizeof r1);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, dy, d0, d1);
  hipMemcpy(r0, d0, sizeof r0, hipMemcpyDeviceToHost); hipMemcpy(r1, d1, sizeof r1, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  for (int i = 0; i < 1024; i++) { double d = fabs(r0[i] - r1[i]); if (d > worst) worst = d; if (d > 1e-5) bad++; }
  printf("mfma gram vs shuffle reference: worst |diff| %.3e, entries off by > 1e-5: %d of 1024\n", worst, bad);
  if (bad) {   // which permutation did we get?  print where lane 17's entries went
    for (int L = 0; L < 16; L++) printf("lane 17 col %2d ref % .5f mfma % .5f\n", L, r0[17 * 16 + L], r1[17 * 16 + L]);
  }
  return bad != 0;
}
