// What does ONE wavefront alone on a gfx950 SIMD pay per instruction, by instruction FORM -- and what runs beside the VALU for free?
// (Round-5 review, "Next round" #1: the occupancy-1 step kernel is paced by the single-wave issue interval; the ledger in
//  profiles/r06_issue_ledger.md prices its instruction classes with the numbers this prints.)
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/issue_probe tools/issue_probe.hip && tools/_build/issue_probe
//
// One workgroup of four waves on one CU = one wave per SIMD (the step kernel's shape).  Every pattern is a hand-written asm body run ITERS
// times between two s_memtime; printed: shader cycles per body and per instruction (max over the four waves).  Independent forms work on
// 16 accumulators so that no form waits for its own result.  Mixed patterns answer "is B free beside A": cost(A + B) against cost(A), cost(B).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

#define R4(OP, a, b, c, d) OP(a) OP(b) OP(c) OP(d)
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define R64(OP) R16(OP) R16(OP) R16(OP) R16(OP)
#define OPS16 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
// operands: %0..%15 accumulators, %16 = b (VGPR), %17 = c (VGPR), %18 = SGPR pair mask, %19 = SGPR scalar

#define ADD32(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n"
#define MUL32(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\n"
#define FMAC32(i) "v_fmac_f32_e32 %" #i ", %16, %17\n"
#define FMA64(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define MUL64(i) "v_mul_f32_e64 %" #i ", %" #i ", -%16\n"
#define FMAK(i) "v_fma_f32 %" #i ", %" #i ", 2.0, 1.0\n"
#define FMAS(i) "v_fma_f32 %" #i ", %" #i ", %19, %17\n"
#define ADDLIT(i) "v_add_f32_e32 %" #i ", 0x3f8ccccd, %" #i "\n"
#define MOV32(i) "v_mov_b32_e32 %" #i ", %16\n"
#define CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, %18\n"
#define MED3(i) "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define FMACDPP(i) "v_fmac_f32_dpp %" #i ", %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define MOVDPP(i) "v_mov_b32_dpp %" #i ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define LSHL(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
#define RSQ(i) "v_rsq_f32_e32 %" #i ", %" #i "\n"
#define ACCW(i) "v_accvgpr_write_b32 a" #i ", %" #i "\n"
#define ACCR(i) "v_accvgpr_read_b32 %" #i ", a" #i "\n"
#define SNOP(i) "s_nop 0\n"
#define SNOP1(i) "s_nop 1\n"
#define SMOV(i) "s_mov_b32 s40, s41\n"
#define SADD(i) "s_add_u32 s40, s40, s41\n"
#define FMA_SNOP(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_nop 0\n"
#define FMA_SMOV(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_mov_b32 s40, s41\n"
#define MUL_SMOV(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\ns_mov_b32 s40, s41\n"
#define MUL_SNOP(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\ns_nop 0\n"
#define FMA_ACCR(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\nv_accvgpr_read_b32 %" #i ", a" #i "\n"
#define READLANE(i) "v_readlane_b32 s40, %" #i ", 3\n"
#define DSR(i) "ds_read_b32 %" #i ", %20 offset:" #i "*4\n"
#define DSR_FMA(i) "ds_read_b32 %" #i ", %20 offset:" #i "*4\nv_fma_f32 %16, %16, %17, %17\n"
#define PL16(i) "v_permlane16_swap_b32 %" #i ", %16\n"
#define PL32(i) "v_permlane32_swap_b32 %" #i ", %16\n"
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n"

enum {
  K_ADD32, K_MUL32, K_FMAC32, K_FMA64, K_MUL64, K_FMAK, K_FMAS, K_ADDLIT, K_MOV32, K_CND64, K_MED3, K_FMACDPP, K_MOVDPP, K_LSHL, K_RSQ, K_ACCW, K_ACCR,
  K_SNOP, K_SNOP1, K_SMOV, K_SADD, K_FMA_SNOP, K_FMA_SMOV, K_MUL_SMOV, K_MUL_SNOP, K_FMA_ACCR, K_READLANE, K_DSR, K_DSR_FMA, K_PL16, K_PL32, K_PKFMA,
  K_MFMA16_DEP, K_MFMA16_IND, K_MFMA4_DEP, K_MFMA4_IND, K_MFMA16x6_ALONE, K_MFMA16x6_I4, K_MFMA16x6_I6, K_MFMA16x6_I8, K_V36_ALONE, K_MFMA16x6_V16, K_MFMA16x6_V32, K_MFMA16x6_V48, K_V48_ALONE, K_MFMA16x6_READ, K_MFMA4x3_READ,
  K_TURN, K_CONE, K_COUNT
};
struct Info { const char* name; int n; };
static const Info INFO[K_COUNT] = {
  {"v_add_f32_e32 (VOP2, 4 B)", 64}, {"v_mul_f32_e32 (VOP2, 4 B)", 64}, {"v_fmac_f32_e32 (VOP2, 4 B)", 64}, {"v_fma_f32 (VOP3, 8 B)", 64}, {"v_mul_f32_e64 (VOP3, 8 B)", 64},
  {"v_fma_f32 v, v, 2.0, 1.0 (inline constants)", 64}, {"v_fma_f32 v, v, s, v (one SGPR source)", 64}, {"v_add_f32_e32 v, literal, v (8 B)", 64}, {"v_mov_b32_e32", 64},
  {"v_cndmask_b32_e64 (SGPR-pair mask)", 64}, {"v_med3_f32", 64}, {"v_fmac_f32_dpp row_newbcast", 64}, {"v_mov_b32_dpp quad_perm", 64}, {"v_lshlrev_b32_e32", 64}, {"v_rsq_f32_e32", 64},
  {"v_accvgpr_write_b32", 64}, {"v_accvgpr_read_b32", 64}, {"s_nop 0", 64}, {"s_nop 1", 64}, {"s_mov_b32", 64}, {"s_add_u32 (dependent)", 64},
  {"v_fma_f32 + s_nop 0 (pairs)", 128}, {"v_fma_f32 + s_mov_b32 (pairs)", 128}, {"v_mul_f32_e32 + s_mov_b32 (pairs)", 128}, {"v_mul_f32_e32 + s_nop 0 (pairs)", 128}, {"v_fma_f32 + v_accvgpr_read (pairs)", 128},
  {"v_readlane_b32", 64}, {"ds_read_b32 (16 in flight, then wait)", 64}, {"ds_read_b32 + v_fma_f32 (pairs)", 128}, {"v_permlane16_swap_b32", 64}, {"v_permlane32_swap_b32", 64}, {"v_pk_fma_f32", 64},
  {"v_mfma_f32_16x16x1_4b_f32, one accumulator (dependent)", 16}, {"v_mfma_f32_16x16x1_4b_f32, 4 accumulators (independent)", 16},
  {"v_mfma_f32_4x4x1_16b_f32, one accumulator (dependent)", 16}, {"v_mfma_f32_4x4x1_16b_f32, 4 accumulators (independent)", 16},
  {"chain of 6 MFMA 16x16x1_4b, nothing else", 6},
  {"6 x (MFMA 16x16x1_4b + 4 independent v_fma) interleaved", 30}, {"6 x (MFMA 16x16x1_4b + 6 independent v_fma) interleaved", 42}, {"6 x (MFMA 16x16x1_4b + 8 independent v_fma) interleaved", 54},
  {"36 independent v_fma alone", 36}, {"chain of 6 MFMA 16x16x1_4b + 16 independent v_fma", 22}, {"chain of 6 MFMA 16x16x1_4b + 32 independent v_fma", 38},
  {"chain of 6 MFMA 16x16x1_4b + 48 independent v_fma", 54}, {"48 independent v_fma alone", 48}, {"chain of 6 MFMA 16x16x1_4b, then a VALU read of the result", 7},
  {"chain of 3 MFMA 4x4x1_16b, then a VALU read of the result", 4},
  {"solver turn x16: v_med3, v_cndmask_e64, s_nop 0, v_fmac_dpp", 64}, {"cone turn x4 (12 slots each, pipelined form)", 49}};

struct WaveRec { uint64_t cycles; };

template <int KIND>
__global__ __launch_bounds__(256) void probe(WaveRec* out, int iters, float b, float c) {
  __shared__ float lds[256];
  lds[threadIdx.x] = b;
  __syncthreads();
  uint64_t t0, t1;
  float a[16];
  for (int i = 0; i < 16; i++) a[i] = (float)threadIdx.x + i;
  f16 acc0, acc1, acc2, acc3;
  f4 q0, q1, q2, q3;
  for (int i = 0; i < 16; i++) { acc0[i] = (float)i; acc1[i] = b; acc2[i] = c; acc3[i] = 1.0f; }
  for (int i = 0; i < 4; i++) { q0[i] = (float)i; q1[i] = b; q2[i] = c; q3[i] = 1.0f; }
  unsigned long long mask = 0x0001000100010001ull << (iters & 3);
  float sc = c;
  unsigned ldsa = (threadIdx.x & 63) * 4;
  asm volatile("" : "+s"(mask), "+s"(sc));
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; it++) {
#define BODY(S) asm volatile(S : OPS16 : "v"(b), "v"(c), "s"(mask), "s"(sc), "v"(ldsa) : "scc", "s40", "s41", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")
    if constexpr (KIND == K_ADD32) BODY(R64(ADD32));
    if constexpr (KIND == K_MUL32) BODY(R64(MUL32));
    if constexpr (KIND == K_FMAC32) BODY(R64(FMAC32));
    if constexpr (KIND == K_FMA64) BODY(R64(FMA64));
    if constexpr (KIND == K_MUL64) BODY(R64(MUL64));
    if constexpr (KIND == K_FMAK) BODY(R64(FMAK));
    if constexpr (KIND == K_FMAS) BODY(R64(FMAS));
    if constexpr (KIND == K_ADDLIT) BODY(R64(ADDLIT));
    if constexpr (KIND == K_MOV32) BODY(R64(MOV32));
    if constexpr (KIND == K_CND64) BODY(R64(CND64));
    if constexpr (KIND == K_MED3) BODY(R64(MED3));
    if constexpr (KIND == K_FMACDPP) BODY(R64(FMACDPP));
    if constexpr (KIND == K_MOVDPP) BODY(R64(MOVDPP));
    if constexpr (KIND == K_LSHL) BODY(R64(LSHL));
    if constexpr (KIND == K_RSQ) BODY(R64(RSQ));
    if constexpr (KIND == K_ACCW) BODY(R64(ACCW));
    if constexpr (KIND == K_ACCR) BODY(R64(ACCR));
    if constexpr (KIND == K_SNOP) BODY(R64(SNOP));
    if constexpr (KIND == K_SNOP1) BODY(R64(SNOP1));
    if constexpr (KIND == K_SMOV) BODY(R64(SMOV));
    if constexpr (KIND == K_SADD) BODY(R64(SADD));
    if constexpr (KIND == K_FMA_SNOP) BODY(R64(FMA_SNOP));
    if constexpr (KIND == K_FMA_SMOV) BODY(R64(FMA_SMOV));
    if constexpr (KIND == K_MUL_SMOV) BODY(R64(MUL_SMOV));
    if constexpr (KIND == K_MUL_SNOP) BODY(R64(MUL_SNOP));
    if constexpr (KIND == K_FMA_ACCR) BODY(R64(FMA_ACCR));
    if constexpr (KIND == K_READLANE) BODY(R64(READLANE));
    if constexpr (KIND == K_DSR) BODY(R16(DSR) "s_waitcnt lgkmcnt(0)\n" R16(DSR) "s_waitcnt lgkmcnt(0)\n" R16(DSR) "s_waitcnt lgkmcnt(0)\n" R16(DSR) "s_waitcnt lgkmcnt(0)\n");
    if constexpr (KIND == K_DSR_FMA) BODY(R16(DSR_FMA) "s_waitcnt lgkmcnt(0)\n" R16(DSR_FMA) "s_waitcnt lgkmcnt(0)\n" R16(DSR_FMA) "s_waitcnt lgkmcnt(0)\n" R16(DSR_FMA) "s_waitcnt lgkmcnt(0)\n");
    if constexpr (KIND == K_PL16) BODY(R64(PL16));
    if constexpr (KIND == K_PL32) BODY(R64(PL32));
    if constexpr (KIND == K_PKFMA) {
      f2 p[8], bb = {b, b}, cc = {c, c};
      for (int i = 0; i < 8; i++) p[i] = f2{a[2 * i], a[2 * i + 1]};
#define PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define PK8 PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
      asm volatile(PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(bb), "v"(cc));
      for (int i = 0; i < 8; i++) { a[2 * i] = p[i].x; a[2 * i + 1] = p[i].y; }
    }
#define M16(A) "v_mfma_f32_16x16x1_4b_f32 %" #A ", %4, %5, %" #A "\n"
#define M4(A) "v_mfma_f32_4x4x1_16b_f32 %" #A ", %4, %5, %" #A "\n"
#define MOPS "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(b), "v"(c)
#define QOPS "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b), "v"(c)
    if constexpr (KIND == K_MFMA16_DEP) asm volatile(M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) : MOPS);
    if constexpr (KIND == K_MFMA16_IND) asm volatile(M16(0) M16(1) M16(2) M16(3) M16(0) M16(1) M16(2) M16(3) M16(0) M16(1) M16(2) M16(3) M16(0) M16(1) M16(2) M16(3) : MOPS);
    if constexpr (KIND == K_MFMA4_DEP) asm volatile(M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) M4(0) : QOPS);
    if constexpr (KIND == K_MFMA4_IND) asm volatile(M4(0) M4(1) M4(2) M4(3) M4(0) M4(1) M4(2) M4(3) M4(0) M4(1) M4(2) M4(3) M4(0) M4(1) M4(2) M4(3) : QOPS);
    // a chain of six MFMAs (one Gram block of the solver) with k independent VALU instructions behind it: how many are free?
#define VF(i) "v_fma_f32 %" #i ", %" #i ", %17, %18\n"
#define V16 VF(1) VF(2) VF(3) VF(4) VF(5) VF(6) VF(7) VF(8) VF(9) VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) VF(16)
#define CH6 "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n" "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n" "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n" \
            "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n" "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n" "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n"
#define COPS "+v"(acc0), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c)
    if constexpr (KIND == K_MFMA16x6_ALONE) asm volatile(CH6 : COPS);
#define M1 "v_mfma_f32_16x16x1_4b_f32 %0, %17, %18, %0\n"
#define V4(a_, b_, c_, d_) VF(a_) VF(b_) VF(c_) VF(d_)
    if constexpr (KIND == K_MFMA16x6_I4) asm volatile(M1 V4(1, 2, 3, 4) M1 V4(5, 6, 7, 8) M1 V4(9, 10, 11, 12) M1 V4(13, 14, 15, 16) M1 V4(1, 2, 3, 4) M1 V4(5, 6, 7, 8) : COPS);
    if constexpr (KIND == K_MFMA16x6_I6) asm volatile(M1 V4(1, 2, 3, 4) VF(5) VF(6) M1 V4(7, 8, 9, 10) VF(11) VF(12) M1 V4(13, 14, 15, 16) VF(1) VF(2) M1 V4(3, 4, 5, 6) VF(7) VF(8) M1 V4(9, 10, 11, 12) VF(13) VF(14) M1 V4(15, 16, 1, 2) VF(3) VF(4) : COPS);
    if constexpr (KIND == K_MFMA16x6_I8) asm volatile(M1 V4(1, 2, 3, 4) V4(5, 6, 7, 8) M1 V4(9, 10, 11, 12) V4(13, 14, 15, 16) M1 V4(1, 2, 3, 4) V4(5, 6, 7, 8) M1 V4(9, 10, 11, 12) V4(13, 14, 15, 16) M1 V4(1, 2, 3, 4) V4(5, 6, 7, 8) M1 V4(9, 10, 11, 12) V4(13, 14, 15, 16) : COPS);
    if constexpr (KIND == K_V36_ALONE) asm volatile(V16 V16 V4(1, 2, 3, 4) : COPS);
    if constexpr (KIND == K_MFMA16x6_V16) asm volatile(CH6 V16 : COPS);
    if constexpr (KIND == K_MFMA16x6_V32) asm volatile(CH6 V16 V16 : COPS);
    if constexpr (KIND == K_MFMA16x6_V48) asm volatile(CH6 V16 V16 V16 : COPS);
    if constexpr (KIND == K_V48_ALONE) asm volatile(V16 V16 V16 : COPS);
    // the compiler places the wait states between the chain and the read of its result: written as two statements
    if constexpr (KIND == K_MFMA16x6_READ) {
      for (int i = 0; i < 6; i++) acc0 = __builtin_amdgcn_mfma_f32_16x16x1f32(b, c, acc0, 0, 0, 0);
      asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[0]) : "v"(acc0[0]));
      asm volatile("" : "+v"(acc0));
    }
    if constexpr (KIND == K_MFMA4x3_READ) {
      for (int i = 0; i < 3; i++) q0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, c, q0, 0, 0, 0);
      asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[0]) : "v"(q0[0]));
      asm volatile("" : "+v"(q0));
    }
    if constexpr (KIND == K_TURN) {
#define TURN(i) "v_med3_f32 %1, %0, %16, %17\nv_cndmask_b32_e64 %2, %2, %1, %18\ns_nop 0\nv_fmac_f32_dpp %0, %1, %3 row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n"
      BODY(TURN(0) TURN(1) TURN(2) TURN(3) TURN(4) TURN(5) TURN(6) TURN(7) TURN(8) TURN(9) TURN(10) TURN(11) TURN(12) TURN(13) TURN(14) TURN(15));
    }
    if constexpr (KIND == K_CONE) {
      // lanes.hpp cone_turns4, LL_CONE_PIPE form: S1 %0, S2 %1, d1 %2, d2 %3, t %4, e1 %5, e2 %6, lam %7 %8, lim %9, k %10..%13
#define CT(L) "v_mul_f32_e32 %4, %1, %1\nv_fmac_f32_e32 %4, %0, %0\nv_rsq_f32_e32 %4, %4\nv_cndmask_b32_e64 %3, %3, %6, %18\nv_mul_legacy_f32_e64 %4, %9, %4 clamp\n" \
              "v_fma_f32 %5, %0, %4, -%7\nv_fma_f32 %6, %1, %4, -%8\nv_cndmask_b32_e64 %2, %2, %5, %18\n"                                                     \
              "v_fmac_f32_dpp %0, %5, %10 row_newbcast:" #L " row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp %1, %5, %11 row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n" \
              "v_fmac_f32_dpp %0, %6, %12 row_newbcast:" #L " row_mask:0xf bank_mask:0xf\nv_fmac_f32_dpp %1, %6, %13 row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n"
      BODY(CT(0) CT(4) CT(8) CT(12) "v_cndmask_b32_e64 %3, %3, %6, %18\n");
    }
  }
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i] + acc0[i] + acc1[i] + acc2[i] + acc3[i];
  for (int i = 0; i < 4; i++) s += q0[i] + q1[i] + q2[i] + q3[i];
  if (s == 12345.678f) out[63].cycles = (uint64_t)lds[5];
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6].cycles = t1 - t0;
}

typedef void (*Fn)(WaveRec*, int, float, float);
template <int K> struct Tab { static void fill(Fn* f) { f[K] = probe<K>; Tab<K - 1>::fill(f); } };
template <> struct Tab<-1> { static void fill(Fn*) {} };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("# %s, %d CUs, clockRate %d kHz; one workgroup of 4 waves on one CU (one wave per SIMD), %d iterations per pattern\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, iters);
  printf("# cycles = s_memtime ticks (100 MHz constant clock on gfx9 counts shader-clock-independent: see the v_mul row for the scale -- round 4 measured 4.44 there)\n");
  Fn fn[K_COUNT];
  Tab<K_COUNT - 1>::fill(fn);
  WaveRec* d;
  CHECK(hipMalloc(&d, sizeof(WaveRec) * 64));
  WaveRec h[4];
  printf("%-62s %6s %12s %12s\n", "pattern", "instr", "cycles/body", "cycles/instr");
  for (int k = 0; k < K_COUNT; k++) {
    printf("%-62s ", INFO[k].name);
    fflush(stdout);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(fn[k], dim3(1), dim3(256), 0, 0, d, iters, 1.0001f, 1e-6f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, sizeof(WaveRec) * 4, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int w = 0; w < 4; w++) mx = std::max(mx, (double)h[w].cycles);
    printf("%6d %12.1f %12.2f\n", INFO[k].n, mx / iters, mx / iters / INFO[k].n);
  }
  return 0;
}
