// llenv.hip -- the product: HIP kernels for gfx950 + the C ABI of include/llenv.h.
//
//   pmc_step_kernel     one 50 Hz control step of every env, fully fused (pmc_step.hpp); 1 env = one 16-lane DPP row,
//                       64-thread workgroups (one wavefront, 4 envs), constraint rows in registers, LDS only for constants
//   pmc_reset_kernel    PLE:150-171 for a list of envs
//   pmc_table_kernel    folds finished-episode statistics into the prioritized sampling table (PLE:235-240)
//   pmc_actions_kernel  draws the synthetic random-policy actions (Philox + Box-Muller)
//
// There is no CPU path: without a HIP device ll_create fails with LL_ENODEV.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <new>
#include <type_traits>

#include "lanes.hpp"
#include "epmc_engine.hpp"
#include "sepmc_engine.hpp"
#include "pmc_engine.hpp"
#include "pmc_step.hpp"

#define HIPCHK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess) throw PmcError(LL_EHIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

typedef Pmc<GpuLanes> K;
typedef GpuLanesPinned<LC_COUNT> GpuLanes1;                   // the per-leg table in registers (the PMC kernel's variant below pins more)
// EPMC / SEPMC at one wave per SIMD read their tables from LDS: with terrain, rays and (SEPMC) pair rows on top of the substep the
// registers are needed for the state -- pinned, the SEPMC kernel spilled 224 B per lane to scratch (45 MB of HBM traffic per launch
// against 22.5 MB algorithmic); A/B on one box: SEPMC 0.326 -> 0.320 ms, EPMC 0.298 -> 0.296 ms
#ifndef LL_PIN_EPMC
#define LL_PIN_EPMC 0
#endif
#ifndef LL_PIN_SEPMC
#define LL_PIN_SEPMC 0
#endif
#ifndef LL_PIN_PMC
#define LL_PIN_PMC 1
#endif
// re-reading of the argument block (lanes.hpp WithParamsReload), per kernel by A/B; the one-wave-per-SIMD SEPMC kernels keep the plain lane policy
#ifndef LL_CONE_LDS
#define LL_CONE_LDS 1   // the larger-batch PMC / EPMC builds keep the cone round's cross scalars in the row's LDS scratch (lanes.hpp WithConeInLds):
#endif                  // 65536 envs 2.305 -> 2.255 ms (PMC), 3.862 -> 3.724 ms (EPMC hurdles); SEPMC 5.035 -> 5.079 ms at 32768 arenas: not there (profiles/r04_cone_lds_ab.txt)
#ifndef LL_CONE_LDS_SEPMC
#define LL_CONE_LDS_SEPMC 1   // ... and since round 5 the larger-batch SEPMC build too: 32768 arenas 6.57 -> 5.13 ms per step on one box (profiles/r05_sepmc_large_batch_ab.txt).  Round 4 had measured
#endif                        // it useless there (5.035 -> 5.079); the round-5 source landed on a worse register allocation (5.04 -> 6.96 ms with 400 more scratch instructions) and this buys it back
#ifndef LL_CONE_LDS_SEPMC1
#define LL_CONE_LDS_SEPMC1 1   // the one-wave-per-SIMD SEPMC builds too: they are the ones that do not fit 512 registers (scratch 364 -> 184 B multi-step, 184 -> 0 single; 0.324 -> 0.318 ms)
#endif
#ifndef LL_PARK
#define LL_PARK 1      // the larger-batch EPMC / SEPMC builds park their per-row scalars in LDS across the substep loop
#endif
#ifndef LL_RELOAD_PMC
#define LL_RELOAD_PMC 1
#endif
#ifndef LL_RELOAD_EPMC1
#define LL_RELOAD_EPMC1 2
#endif
#ifndef LL_RELOAD_EPMC2
#define LL_RELOAD_EPMC2 1
#endif
#ifndef LL_SEPMC_RAY_CHUNK
#define LL_SEPMC_RAY_CHUNK 3      // rays per chunk in the one-wave-per-SIMD SEPMC kernels (lanes.hpp WithRayChunk; 7 fails on the GPU: tools/diag_sepmc_rays.py)
#endif
#ifndef LL_RELOAD_SEPMC2
#define LL_RELOAD_SEPMC2 1      // (with the episode scalars parked in LDS the re-read pays here too: 32768 arenas 14.8 -> 16.0 M robot-steps/s)
#endif
#ifndef LL_SEPMC_ONE_WAVE_DEFAULT
#define LL_SEPMC_ONE_WAVE_DEFAULT 1
#endif
#ifndef LL_GRAM_PIPE_PMC1
#define LL_GRAM_PIPE_PMC1 0     // 1: the one-wave-per-SIMD PMC cone kernels form the Gram blocks of their contact rows on the matrix cores, one MFMA at a time between pieces of the next
                                // row's arithmetic (lanes.hpp WithGramPipe; round 6, the round-5 review's "pipelined producer").  Built, held to the oracle on the host build, and measured
                                // (profiles/r06_gram_pipe_ab.txt, one box): single-step launches 0.2073 -> 0.2037 ms (- 1.7 %), the contract's multi-step launches 0.1896 -> 0.1910 (+ 0.7 %: 176 B of
                                // scratch).  Why so little: a LONE wave does not overlap an MFMA with its own VALU work at all -- 6 x (v_mfma_f32_16x16x1_4b + k independent v_fma) takes 6 x (32 + 5.7 k)
                                // cycles however they are interleaved (profiles/r06_issue_probe.txt) -- so a block costs 6 x 32 cycles of MFMA + 16 v_accvgpr_read + 16 v_permlane swaps at 8.6 cycles
                                // + 16 v_fmac against 96 v_fmac_dpp + 16 selects: 500 against 600 cycles, before register pressure.  And MFMA ignores EXEC: in a wave whose last rows hold no env it
                                // overwrites registers of the rows that sit the step out (a GPU fault in test_partial_wave_and_odd_batch_sizes).  OFF; the code, the host statement and its test stay
#endif
typedef GpuLanesPinned<LC_COUNT, LL_PIN_PMC ? 7 : 0, LL_PIN_PMC ? BC_COUNT : 0, LL_PIN_PMC ? LK_BASE : 0> GpuLanesPmc1Plain;  // PMC at one wave per SIMD: candidate fields and base constants too
typedef std::conditional<LL_GRAM_PIPE_PMC1 != 0, WithGramPipe<GpuLanesPmc1Plain>, GpuLanesPmc1Plain>::type GpuLanesPmc1;

// PLE:235-240 for the batch, by one wavefront: fold the statistics published by finished episodes into the per-clip table
// (lane = clip, 64 per pass), then rebuild p ~ (1 - avg_reward_sum)^factor and its inclusive CDF.  Called by the last
// workgroup of a step kernel to finish, so the table a step leaves behind already contains the episodes it ended.
__device__ __forceinline__ double wave_sum(double x) {
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  return x;
}
// (Everything the fold reads and writes goes through device-scope accesses: inside a multi-step launch consecutive folds run on different
// wavefronts, possibly on different XCDs whose L2s are not coherent with each other, and the versions are read by re-seeding waves anywhere.)
__device__ __forceinline__ double ld_agent(const double* p) {
  return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the kernel's first by-value argument where it lies in the kernarg segment (StepParams::kFirstKernelArgument; lanes.hpp WithParamsReload::params)
__device__ __forceinline__ const StepParams& kernarg_params() {
  const __attribute__((address_space(4))) void* k = (const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return *(const StepParams*)(const __attribute__((address_space(4))) StepParams*)k;
}
// (table_fold_wave_out_of_line) Out of line on purpose in the multi-step kernels: inlined into the step loop its double-precision pow and scans cost the step kernel registers it does not have (the
// one-wave-per-SIMD multi-step build went from 234 to 256 AGPRs plus 76 B of scratch).
// sl: the control step of the launch whose finished episodes are folded; the resulting CDF goes to `cdf` itself after the launch's last step and to
// version slot sl + 1 otherwise (what a re-seed at step sl + 1 samples from: Pmc::table_for_reseed)
__device__ __forceinline__ void table_fold_wave(const StepParams& P, int sl, bool last_step) {
  const int lane = threadIdx.x & 63, n = P.n_clips;
  unsigned long long* pend_r = P.pending_reward + (long)sl * n;
  unsigned long long* pend_l = P.pending_len + (long)sl * n;
  const double* prev = sl == 0 ? P.cdf : P.cdf_ver + (long)sl * n;
  double* dst = last_step ? P.cdf : P.cdf_ver + (long)(sl + 1) * n;
  bool mine = false;
  for (int c = lane; c < n; c += PMC_WAVE) {
    const unsigned long long pr = __hip_atomic_load(pend_r + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long pl = __hip_atomic_load(pend_l + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pr) {
      st_agent(P.avg_reward + c, (double)__uint_as_float((uint32_t)pr));
      st_agent(P.avg_len + c, (double)__uint_as_float((uint32_t)pl));
      __hip_atomic_store(pend_r + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pend_l + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mine = true;
    }
  }
  if (!__any(mine)) {                                 // nothing ended in this step: the next version is the previous one
    if (dst != prev) for (int c = lane; c < n; c += PMC_WAVE) st_agent(dst + c, ld_agent(prev + c));
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (a lane re-reads what it has just stored)
  double part = 0.0;
  for (int c = lane; c < n; c += PMC_WAVE) {
    const double p = pow(1.0 - ld_agent(P.avg_reward + c), P.sample_factor);
    st_agent(P.prob + c, p);
    part += p;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const double inv = 1.0 / wave_sum(part);
  double carry = 0.0;
  for (int c0 = 0; c0 < n; c0 += PMC_WAVE) {       // inclusive scan, 64 clips per pass
    const int c = c0 + lane;
    double x = 0.0;
    if (c < n) { x = ld_agent(P.prob + c) * inv; st_agent(P.prob + c, x); }
    for (int d = 1; d < PMC_WAVE; d <<= 1) {
      const double y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    x += carry;
    if (c < n) st_agent(dst + c, (c == n - 1) ? 1.0 : x);
    carry = __shfl(x, PMC_WAVE - 1);
  }
}
__device__ __attribute__((noinline)) void table_fold_wave_out_of_line(const StepParams& P, int sl, bool last_step) { table_fold_wave(P, sl, last_step); }
// The last workgroup to finish control step `sl` of the launch folds that step's finished episodes into the sampling table.  The statistics travel
// by device-scope atomics only (publish_max), so no cache write-back is needed -- a __threadfence() here would flush this XCD's L2 once per
// workgroup.  Waiting for the wave's own atomics to be acknowledged before it takes its ticket orders them ahead of the ticket.  Folds run in step
// order: the wave that folds step sl first waits for the mark of step sl - 1 (every wave has finished step sl - 1 by then, so its folding wave is
// running or done).  `multi`: the launch runs several steps and re-seeds read the versions, so the fold is published with a mark.
__device__ __forceinline__ void step_done_fold(const StepParams& P, int sl, bool last_step, bool multi) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned int ticket = 0;
  if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(P.block_ticket + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket != gridDim.x - 1) return;
  if (multi && sl > 0) {
    while (__hip_atomic_load(P.ver_ready + (sl - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.launch_serial) __builtin_amdgcn_s_sleep(2);
    LL_VER_FENCE(__ATOMIC_ACQUIRE);                    // the previous fold's version is read after its mark -- said in the memory model too (pmc_params.hpp LL_VER_FENCE)
  }
  asm volatile("" ::: "memory");
  if (multi) table_fold_wave_out_of_line(P, sl, last_step);      // (P: a reference into the kernarg segment -- nothing is copied to the stack for the call)
  else table_fold_wave(P, sl, last_step);                        // the single-step kernels fold inline at their very end, as they always have
  if (threadIdx.x == 0) __hip_atomic_store(P.block_ticket + sl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (multi) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the version's (write-through, device-scope) stores are acknowledged before its mark goes out
    LL_VER_FENCE(__ATOMIC_RELEASE);
    if (threadIdx.x == 0) {
      __hip_atomic_store(P.ver_ready + sl, P.launch_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (last_step) __hip_atomic_store(P.resident, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The synthetic random policy a ~ N(0, sigma^2): Philox4x32-10 keyed on (seed; group of four actions, step) + Box-Muller.
__device__ __forceinline__ float4 random_action_group(const StepParams& P, uint32_t gid, float sigma, int sl = 0) {
  uint32_t r[4];
  const uint64_t step = P.step_count + (uint64_t)sl;
  philox4x32(gid, (uint32_t)step, (uint32_t)(step >> 32), 0xAC710u, (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r);
  const float k = 2.3283064365386963e-10f;   // 2^-32
  float u1 = fminf(((float)r[0] + 1.0f) * k, 1.0f), u2 = (float)r[1] * k, u3 = fminf(((float)r[2] + 1.0f) * k, 1.0f), u4 = (float)r[3] * k;
  float m1 = sqrtf(-2.0f * logf(u1)), m2 = sqrtf(-2.0f * logf(u3));
  float4 o;
  o.x = sigma * m1 * cosf(6.283185307179586f * u2); o.y = sigma * m1 * sinf(6.283185307179586f * u2);
  o.z = sigma * m2 * cosf(6.283185307179586f * u4); o.w = sigma * m2 * sinf(6.283185307179586f * u4);
  return o;
}

// The actions of one control step for the lane's leg: drawn here (action_sigma > 0: lanes 0..2 of the row draw the env's three groups of
// four actions, record them, and hand them to the leg lanes through 48 B of LDS) or read from P.actions.
template <class Lanes>
__device__ __forceinline__ void step_actions(const StepParams& P, const Lanes& ln, float* lds, int env, int sl, float* act) {
  if (P.action_sigma > 0.0f) {
    float* stash = lds + LC_COUNT * 4 + CAND_TABLE_WORDS * PMC_ROW + (threadIdx.x >> 4) * 12;
    const int l16 = threadIdx.x & 15;
    const uint32_t g = (uint32_t)(l16 < 3 ? l16 : 0);
    const float4 o = random_action_group(P, (uint32_t)env * 3u + g, P.action_sigma, sl);
    if (l16 < 3) {
      reinterpret_cast<float4*>(P.actions_out)[(long)env * 3 + l16] = o;
      reinterpret_cast<float4*>(stash)[l16] = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the row's LDS writes have landed (one wave)
    for (int j = 0; j < 3; j++) act[j] = stash[ln.leg() * 3 + j];
  } else {
    for (int j = 0; j < 3; j++) act[j] = ln.ldl(P.actions, (long)env * 12 + j, 3);
  }
}

// OCC = wavefronts per SIMD the register budget allows: 1 (512 registers, nothing spills to scratch) while the grid fits the
// chip one wavefront per SIMD, 2 (256 registers) for larger batches, where a second resident wavefront hides issue stalls.
// OBST = the set_obstacle build (Pmc::step_env<true>): the jump obstacle takes part in the substeps as a static box.
// MULTI = the launch may run several control steps (ll_step_random_n); single-step launches run the loop-free build.
// CONE = the cone-coupled friction solve (LLM_SPEC_FRICTION_MODE = 2, Pmc::gs_cone_round): its own instantiations of every step kernel.
// XROWS = the build that can carry the extended contact rows (Pmc::substep_impl; launched when LLM_SPEC_SELF_FRICTION / _PAIR_FRICTION / _MAX_PAIR are off their defaults: cone builds only).
template <int OCC, bool OBST = false, bool MULTI = false, bool CONE = false, bool XROWS = false>
__global__ __launch_bounds__(PMC_WAVE, OCC) void pmc_step_kernel(StepParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env0 = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);      // one env = one 16-lane DPP row
  typedef WithParamsReload<typename std::conditional<OCC == 1, GpuLanesPmc1, GpuLanes>::type, LL_RELOAD_PMC> Lanes0;
  typedef typename std::conditional<(OCC == 2 && CONE && LL_CONE_LDS), WithConeInLds<Lanes0>, Lanes0>::type Lanes;     // (launched with the row scratch allocated: launch_step)
  Lanes ln(lds);
  if constexpr (OCC == 1) ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS, P.basec);   // all 64 lanes copy, also those without an env
  else ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if constexpr (!MULTI) {
    if (env0 < P.n_envs) {
      float act[3];
      step_actions(P, ln, lds, env0, 0, act);
      Pmc<Lanes>::template step_env<OBST, CONE, XROWS>(ln, P, env0, act, 0);
    }
    step_done_fold(P, 0, true, false);
  } else {
    // ll_step_random_n: n_steps control steps back to back.  A wave walks its four envs through them on its own -- no other wave is waited
    // for, so a slow step of one wave (leg-leg rows, a re-seed) is not a slow step of the whole chip -- and between two steps it only has
    // to see its own stores (state, obs row, bookkeeping: workgroup-scope fence = wait for the wave's outstanding memory operations).
    // (Dealing the steps out dynamically -- a per-XCD ready queue of (step, group) items -- was built and measured in round 4: 0.8 % slower,
    // there is no imbalance left to win inside a multi-step launch; profiles/r04_dyn_steps.txt.)
    // The sampling table keeps PLE:235-240 inside the launch: every step is folded by the last wave to finish it into a version of its own, and an
    // episode that re-seeds at step s draws from the version steps 0 .. s - 1 left (step_done_fold, Pmc::table_for_reseed): k steps in one launch are
    // k launches, bit for bit, table included.
    if (threadIdx.x == 0) __hip_atomic_fetch_add(P.resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // "this wave has started"
    for (int sl = 0; sl < P.n_steps; sl++) {
      if (sl) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      ln.new_step();
      int env = env0;
      asm volatile("" : "+v"(env));      // ... and no address of the step's ~150 loads and stores either (they all derive from env)
      if (env < P.n_envs) {
        float act[3];
        PMC_PHASE("step.actions");
        step_actions(P, ln, lds, env, sl, act);
        Pmc<Lanes>::template step_env<OBST, CONE, XROWS>(ln, P, env, act, sl);
      }
      PMC_PHASE("step.table_fold");
      const StepParams& Pr = kernarg_params();      // (re-read from the kernarg segment like the step itself: nothing of it is parked in spilled SGPRs across the step)
      step_done_fold(Pr, sl, sl == Pr.n_steps - 1, true);
    }
  }
}

// EPMC (epmc_step.hpp): one control step of PlayGroundEnv for every env; same execution model and register budgets as
// pmc_step_kernel.  The 778 rays of an env are dealt out over the 16 lanes of its row.
// MULTI: the launch may run several control steps (ll_epmc_step_random_n).  The step loop costs the larger-batch build registers it does
// not have (788 instead of 552 B of scratch per lane; 65536 envs: 18.6 -> 16.8 M env-steps/s), and a grid of sixteen wavefronts per SIMD has
// neither a launch gap nor a slowest wave worth hiding, so multi-step launches exist for the one-wave-per-SIMD build only; larger batches
// run their steps as single launches (same results: the step's draws are keyed on env, episode and draw index).
template <int OCC, bool MULTI = false, bool CONE = false, bool XROWS = false>   // CONE, XROWS: see pmc_step_kernel
__global__ __launch_bounds__(PMC_WAVE, OCC) void epmc_step_kernel(StepParams P, EpmcParams E) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env0 = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);
  typedef WithRayChunk<WithShapePrefetch<WithParamsReload<typename std::conditional<OCC == 1 && LL_PIN_EPMC, GpuLanes1, GpuLanes>::type, (OCC == 1 ? LL_RELOAD_EPMC1 : LL_RELOAD_EPMC2)>, OCC == 2>, (OCC == 1 ? (MULTI ? 7 : 3) : 1)> Lanes0;   // (single launches at 7: 68 B of scratch)
  typedef typename std::conditional<(OCC == 2 && CONE && LL_CONE_LDS), WithConeInLds<Lanes0>, Lanes0>::type Lanes;
  Lanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if (env0 >= P.n_envs) return;
  if constexpr (!MULTI) {
    float act[3];
    step_actions(P, ln, lds, env0, 0, act);
    Epmc<Lanes>::template step_env<((OCC == 2 && LL_PARK) || LL_PARK > 1), CONE, XROWS>(ln, P, E, env0, act);
  } else {
    for (int sl = 0; sl < P.n_steps; sl++) {               // ll_epmc_step_random_n: see pmc_step_kernel
      if (sl) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      ln.new_step();
      int env = env0;
      asm volatile("" : "+v"(env));
      float act[3];
      step_actions(P, ln, lds, env, sl, act);
      Epmc<Lanes>::template step_env<(LL_PARK > 1), CONE, XROWS>(ln, P, E, env, act);
    }
  }
}
__global__ __launch_bounds__(PMC_WAVE) void epmc_reset_kernel(StepParams P, EpmcParams E, const int32_t* ids, int n, const float* draws, const float* prev_orn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int i = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);
  GpuLanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if (i >= n) return;
  const int env = ids ? ids[i] : i;
  Epmc<GpuLanes>::reset_env(ln, P, E, env, draws ? draws + (long)i * EPMC_MAX_DRAWS : nullptr, prev_orn ? prev_orn + (long)i * 4 : nullptr);
}

// The 778 rays of a row's observation as a kernel of their own (round 6; epmc_step.hpp percept_rays): one workgroup of two waves per row -- eight waves per SIMD at 4096 rows, where the
// step kernel has one -- stages the row's box records in LDS, sorts them into the three ray families' lists (any order: a ray's answer is a minimum / maximum over boxes), and thread t casts
// its share of every ray family.  Launched behind a step kernel that ran with EpmcParams::split_rays (it left the row's pose in ray_pose); serves EPMC rows and SEPMC robot rows alike.
#define PERCEPT_THREADS 128      // 325 + 128 + 325 rays in 3 + 1 + 3 passes of two waves: 87 % of the lanes busy (256 threads: 61 %)
__global__ __launch_bounds__(PERCEPT_THREADS) void epmc_percept_kernel(StepParams P, EpmcParams E) {
  __shared__ __attribute__((aligned(16))) float lists[3][EPMC_MAX_BOXES * EPMC_BOX_WORDS];
  __shared__ int cnt[3];
  const int row = blockIdx.x;
  const float* rec = E.ray_pose + (long)row * EPMC_RAY_POSE;
  if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
  __syncthreads();
  typedef Epmc<GpuLanes> EP;
  const int nb = (int)rec[14];
  if ((int)threadIdx.x < nb) {
    M3<float> R;
    for (int i = 0; i < 9; i++) R.m[i] = rec[3 + i];
    const EP::RayBounds bd = EP::ray_bounds(rec, R);
    const BoxRec r = EP::ray_box(rec, E.boxes + (long)row * EPMC_MAX_BOXES * EPMC_BOX_WORDS, threadIdx.x);
    for (int fam = 0; fam < 3; fam++)
      if (EP::box_in_family(fam, r, rec, bd)) store_box(lists[fam] + atomicAdd(&cnt[fam], 1) * EPMC_BOX_WORDS, r);
  }
  __syncthreads();
  const float* lp[3] = {lists[0], lists[1], lists[2]};
  const int n[3] = {cnt[0], cnt[1], cnt[2]};
  EP::percept_rays(E, row, rec, lp, n, P.obs + (long)row * P.obs_dim + 3L * P.prop_dim + 36, threadIdx.x, PERCEPT_THREADS);
}

// SEPMC (sepmc_step.hpp): one control step of ChaseTagGameEnv; row = 2 * arena + robot, the two robots of an arena are
// neighbouring rows of one wave and exchange state with v_permlane16_swap.
template <int OCC, bool MULTI = false, bool CONE = false, bool XROWS = false>          // MULTI: see epmc_step_kernel; CONE, XROWS: see pmc_step_kernel
__global__ __launch_bounds__(PMC_WAVE, OCC) void sepmc_step_kernel(StepParams P, SepmcParams S) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row0 = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);
  typedef typename std::conditional<OCC == 1 && LL_PIN_SEPMC, GpuLanes1, GpuLanes>::type PlainLanes;
  typedef WithRayChunk<WithShapePrefetch<typename std::conditional<(OCC == 2 && LL_RELOAD_SEPMC2), WithParamsReload<PlainLanes, LL_RELOAD_SEPMC2>, PlainLanes>::type, OCC == 1>, (OCC == 1 ? LL_SEPMC_RAY_CHUNK : 1)> Lanes0;   // (chunk 7 fails the arena invariants on the GPU in this kernel -- at 256 + 255 registers; 3 is what was validated: profiles/r04_ray_ab.txt)
  typedef typename std::conditional<(CONE && ((OCC == 2 && LL_CONE_LDS_SEPMC) || (OCC == 1 && LL_CONE_LDS_SEPMC1))), WithConeInLds<Lanes0>, Lanes0>::type Lanes;
  Lanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if (row0 >= P.n_envs) return;
  if constexpr (!MULTI) {
    float act[3];
    step_actions(P, ln, lds, row0, 0, act);
    Sepmc<Lanes>::template step_env<((OCC == 2 && LL_PARK) || LL_PARK > 1), CONE, XROWS>(ln, P, S, row0, act);
  } else {
    for (int sl = 0; sl < P.n_steps; sl++) {               // ll_sepmc_step_random_n: see pmc_step_kernel
      if (sl) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      ln.new_step();
      int row = row0;
      asm volatile("" : "+v"(row));
      float act[3];
      step_actions(P, ln, lds, row, sl, act);
      Sepmc<Lanes>::template step_env<(LL_PARK > 1), CONE, XROWS>(ln, P, S, row, act);
    }
  }
}
__global__ __launch_bounds__(PMC_WAVE) void sepmc_reset_kernel(StepParams P, SepmcParams S, const int32_t* ids, int n, const float* draws, const float* prev_orn) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int i = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);     // ids lists rows in (robot 0, robot 1) pairs: pairs stay neighbours
  GpuLanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if (i >= n) return;
  const int row = ids ? ids[i] : i;
  Sepmc<GpuLanes>::reset_env(ln, P, S, row, draws ? draws + (long)(i >> 1) * EPMC_MAX_DRAWS : nullptr, prev_orn ? prev_orn + (long)(i >> 1) * 4 : nullptr);
}

__global__ __launch_bounds__(PMC_WAVE) void pmc_reset_kernel(StepParams P, const int32_t* ids, int n, const int32_t* clip, const double* t0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int i = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);
  GpuLanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if (i >= n) return;
  const int env = ids ? ids[i] : i;
  int c;
  double t;
  uint32_t ep = P.ep_count[env] + 1;          // the sixteen lanes of the row read, then write, the same value
  K::sample_start(ln, P, env, ep, &c, &t);     // ML:59-63, ML:50-51
  P.ep_count[env] = ep;
  if (clip) c = clip[i];
  if (t0) t = t0[i];
  K::reset_env(ln, P, env, c, t);
  P.done[env] = 0;
  P.done_reason[env] = 0;
}

__global__ __launch_bounds__(PMC_WAVE) void pmc_probe_pd_kernel(StepParams P, const float* in, float* out, int n, int mode) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int i = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);
  GpuLanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);
  if (i >= n) return;
  K::probe_pd(ln, P, in + (long)i * 36, out + (long)i * 12, mode);
}

// ll_finish_unroll: one thread per env walks its unroll backwards (rows are [obs_dim | A 12 | neglogp | R | V | r | 1 - done])
__global__ void pmc_gae_kernel(float* block, int n_envs, int unroll, int W, int od, float gamma, float lam, const float* bootstrap) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n_envs) return;
  float* rows = block + (size_t)env * unroll * W;
  float adv = 0.0f, vnext = bootstrap[env];
  for (int t = unroll - 1; t >= 0; t--) {
    float* r = rows + (size_t)t * W + od;
    const float V = r[14], m = r[16];
    const float delta = r[15] + gamma * vnext * m - V;
    adv = delta + gamma * lam * m * adv;
    r[13] = adv + V;
    vnext = V;
  }
}

__global__ void pmc_actions_kernel(StepParams P, float* actions, float sigma) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= P.n_envs * 3) return;
  reinterpret_cast<float4*>(actions)[gid] = random_action_group(P, (uint32_t)gid, sigma);
}

#if defined(LL_KERNELS_ONLY)
// A listing of ONE step kernel (tools/issue_ledger.py, tools/isa_stats.py --one): -DLL_KERNELS_ONLY='pmc_step_kernel<1, false, true, true>(StepParams)' compiles that
// instantiation alone, in seconds instead of minutes.  No library is built this way.
template __global__ void LL_KERNELS_ONLY;
#else
struct HipBackend {
  int device;
  hipStream_t own = nullptr, stream = nullptr;
  bool timing = false;
  int simds = 1024;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  std::vector<int> ev_steps;                             // control steps of each timed launch (ll_step_random_n: more than one)
  size_t ev_used = 0;
  static constexpr size_t kMaxTimedLaunches = 16384;   // event pairs kept between two ll_kernel_time_ms() polls

  explicit HipBackend(int dev) : device(dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw PmcError(LL_ENODEV, "no HIP device available: this engine has no CPU fallback");
    if (dev < 0 || dev >= n) throw PmcError(LL_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    simds = prop.multiProcessorCount * 4;             // four SIMDs per compute unit
    simds_hw = simds;
    // LL_SHARE_SIMDS=1: always launch the 256-register builds, also when the grid would fit one 512-register wavefront per SIMD.  A
    // wavefront of the one-wave-per-SIMD builds owns its SIMD's whole register file, so any other kernel that is resident at the same
    // time -- RCCL's gather on the learner rank -- DISPLACES step-kernel waves instead of sharing SIMDs with them, and the step launch
    // ends later by the full residency of that kernel (measured with an RCCL stand-in: profiles/r03_simd_sharing.txt).
    const char* sh = getenv("LL_SHARE_SIMDS");
    if (sh && sh[0] == '1') simds = 0;
    // LL_DETERMINISTIC=1: multi-step calls run as single launches.  A multi-step launch equals k single launches bit for bit only while every one of its waves is on the chip
    // (ll_get_table_sync == 0); on a device it shares with other kernels -- a collective, other ranks -- a re-seed may have to take the newest table version there is, and which clip it
    // draws then hangs on timing.  Single launches never do.
    // LL_SEPMC_ONE_WAVE (default 1): which chase-tag build runs batches beyond one wave per SIMD.  1: the one-wave-per-SIMD build at EVERY size -- a grid of 16 x the chip's SIMDs runs as
    // waves that follow each other on a SIMD without waiting for a step's slowest wave, and none of them spills (the 256-register chase-tag build carries 868 B of scratch per lane):
    // 32768 arenas 5.18 -> 4.37 ms per step, 8192 arenas 1.45 -> 1.16, 4096 arenas 0.80 -> 0.61 (profiles/r06_sepmc_one_wave_ab.txt).  0: the 256-register build, two waves per SIMD
    // (rounds 2 - 5; still what LL_SHARE_SIMDS=1 selects).  PMC and EPMC keep their 256-register builds for larger batches: those do not spill and win there.
    const char* ow = getenv("LL_SEPMC_ONE_WAVE");
    sepmc_simds = (simds != 0 && (ow ? ow[0] == '1' : LL_SEPMC_ONE_WAVE_DEFAULT)) ? 0x7fffffff : simds;
    const char* ow2 = getenv("LL_EPMC_ONE_WAVE");                    // the same choice for the PlayGround env (default 0: its 256-register build does not spill and wins at larger batches)
    epmc_simds = (simds != 0 && ow2 && ow2[0] == '1') ? 0x7fffffff : simds;
    const char* det = getenv("LL_DETERMINISTIC");
    deterministic = det && det[0] == '1';
    // LL_SPLIT_RAYS (EPMC / SEPMC): 0 = the step kernel casts the 778 rays of a row itself (rounds 1 - 5); 1 = single-step launches leave them to epmc_percept_kernel behind the step
    // kernel; 2 = multi-step calls too run as single steps, each followed by the ray kernel.  Defaults by the A/B on one box (profiles/r06_split_rays_ab.txt): EPMC 2 (hurdles: single steps
    // 0.2983 -> 0.2905 ms, 32-step calls 0.2894 -> 0.2898; cube stairs 0.3163 -> 0.2937), SEPMC 1 (single steps 0.3341 -> 0.3308; 32-step calls would lose 4 %: 0.3153 -> 0.3286)
    const char* sr = getenv("LL_SPLIT_RAYS");
    split_rays_epmc = sr ? atoi(sr) : 2;
    split_rays_sepmc = sr ? atoi(sr) : 1;
    stream = own;
  }
  ~HipBackend() {
    (void)hipSetDevice(device);
    for (auto& p : evs) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (own) (void)hipStreamDestroy(own);
  }
  void use() { HIPCHK(hipSetDevice(device)); }
  int simds_hw = 1024;
  int split_rays_epmc = 2, split_rays_sepmc = 1;
  int sepmc_simds = 1024, epmc_simds = 1024;
  // can every workgroup of a step launch be on the chip at once?  (one 512-register wave per SIMD while the grid fits, two 256-register waves otherwise:
  // launch_step.)  A multi-step launch needs it -- its waves wait for each other's finished episodes (PmcEngine::step)
  bool deterministic = false;
  bool co_resident(const StepParams& P) const {
    const int blocks = (P.n_envs + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    if (deterministic) return false;
    return blocks <= simds || blocks <= 2 * simds_hw;
  }
  void set_stream(void* s) { stream = s ? (hipStream_t)s : own; }
  void* stream_handle() { return (void*)stream; }
  void* alloc(size_t bytes) {
    use();
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, bytes ? bytes : 4));
    return p;
  }
  void release(void* p) { (void)hipSetDevice(device); (void)hipFree(p); }
  void zero(void* p, size_t bytes) { use(); HIPCHK(hipMemsetAsync(p, 0, bytes, stream)); }
  void h2d(void* d, const void* h, size_t bytes) {
    use();
    HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
  }
  void d2h(void* h, const void* d, size_t bytes) {
    use();
    HIPCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
  }
  void sync() { use(); HIPCHK(hipStreamSynchronize(stream)); }

  static size_t lds_bytes() { return ((size_t)LC_COUNT * 4 + (size_t)CAND_TABLE_WORDS * PMC_ROW + PMC_ENVS_PER_WAVE * 12) * sizeof(float); }
  static size_t lds_bytes_epmc() { return lds_bytes() + (size_t)PMC_ENVS_PER_WAVE * PMC_ROW_SCRATCH * sizeof(float); }
  static_assert(((size_t)LC_COUNT * 4 + (size_t)CAND_TABLE_WORDS * PMC_ROW + PMC_ENVS_PER_WAVE * 12 + (size_t)PMC_ENVS_PER_WAVE * PMC_ROW_SCRATCH) * sizeof(float) <= 160 * 1024 / 8,
                "eight single-wave workgroups per CU (two per SIMD) must fit the 160 KB of LDS");
  std::pair<hipEvent_t, hipEvent_t>* timing_begin(int n_steps = 1) {
    if (!timing) return nullptr;
    if (ev_used == kMaxTimedLaunches) return nullptr;   // un-polled timing does not grow without bound: later launches go untimed
    if (ev_used == evs.size()) {
      hipEvent_t a, b;
      HIPCHK(hipEventCreate(&a));
      HIPCHK(hipEventCreate(&b));
      evs.push_back(std::make_pair(a, b));
      ev_steps.push_back(1);
    }
    ev_steps[ev_used] = n_steps;
    std::pair<hipEvent_t, hipEvent_t>* ev = &evs[ev_used++];
    HIPCHK(hipEventRecord(ev->first, stream));
    return ev;
  }
  // the rays of the step's observation by the kernel of their own?  Not when the caller plays rayTestBatch (scripted rays) -- and a multi-step call only under LL_SPLIT_RAYS=2, as single steps
  static bool rays_split(const StepParams& P, const EpmcParams& E, int mode) { return !E.scr_ray_hit && (P.n_steps == 1 ? mode >= 1 : mode >= 2); }
  void launch_percept(const StepParams& P, const EpmcParams& E) {
    hipLaunchKernelGGL(epmc_percept_kernel, dim3(P.n_envs), dim3(PERCEPT_THREADS), 0, stream, P, E);
  }
  void launch_epmc_step(const StepParams& P, const EpmcParams& E_in) {
    use();
    const int blocks = (P.n_envs + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    std::pair<hipEvent_t, hipEvent_t>* ev = timing_begin(P.n_steps);
    const bool cone = P.friction_mode == 2, split = rays_split(P, E_in, split_rays_epmc);
    EpmcParams E = E_in;
    E.split_rays = split ? 1 : 0;
#define LL_GO(KERNEL, PARAMS) hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, PARAMS, E)
    if (pmc_wants_xrows_terrain(P)) {                    // the extended contact rows (round 6): the cone builds with XROWS, every step a launch of its own
      if (!cone) throw PmcError(LL_EINVAL, "self_friction / leg_edges need friction_mode 2 (the extended contact rows exist in the cone builds)");
      StepParams Q = P;
      Q.n_steps = 1;
      for (int sl = 0; sl < P.n_steps; sl++, Q.step_count++) {
        if (blocks <= epmc_simds) LL_GO((epmc_step_kernel<1, false, true, true>), Q); else LL_GO((epmc_step_kernel<2, false, true, true>), Q);
        if (split) launch_percept(Q, E);
      }
    } else
    if (P.n_steps == 1) {
      if (blocks <= epmc_simds) { if (cone) LL_GO((epmc_step_kernel<1, false, true>), P); else LL_GO((epmc_step_kernel<1>), P); }
      else                 { if (cone) LL_GO((epmc_step_kernel<2, false, true>), P); else LL_GO((epmc_step_kernel<2>), P); }
      if (split) launch_percept(P, E);
    } else if (blocks <= epmc_simds && !split) {
      if (cone) LL_GO((epmc_step_kernel<1, true, true>), P); else LL_GO((epmc_step_kernel<1, true>), P);
    } else {
      StepParams Q = P;                                  // larger batches, and every batch with the rays split off: the steps of the call as single launches (see epmc_step_kernel)
      Q.n_steps = 1;
      for (int sl = 0; sl < P.n_steps; sl++, Q.step_count++) {
        if (blocks <= epmc_simds) { if (cone) LL_GO((epmc_step_kernel<1, false, true>), Q); else LL_GO((epmc_step_kernel<1>), Q); }
        else                 { if (cone) LL_GO((epmc_step_kernel<2, false, true>), Q); else LL_GO((epmc_step_kernel<2>), Q); }
        if (split) launch_percept(Q, E);
      }
    }
#undef LL_GO
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->second, stream));
  }
  void launch_epmc_reset(const StepParams& P, const EpmcParams& E, const int32_t* ids, int n, const float* draws, const float* prev_orn) {
    use();
    const int blocks = (n + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    hipLaunchKernelGGL(epmc_reset_kernel, dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P, E, ids, n, draws, prev_orn);
    HIPCHK(hipGetLastError());
  }
  void launch_sepmc_step(const StepParams& P, const SepmcParams& S_in) {
    use();
    const int blocks = (P.n_envs + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    std::pair<hipEvent_t, hipEvent_t>* ev = timing_begin(P.n_steps);
    // (beyond one wave per SIMD a multi-step call is better off as single steps with the rays split off: 32768 arenas 4.52 -> 4.37 ms per step, profiles/r06_sepmc_one_wave_ab.txt)
    const bool cone = P.friction_mode == 2, split = rays_split(P, S_in.e, (split_rays_sepmc == 1 && blocks > simds_hw) ? 2 : split_rays_sepmc);
    SepmcParams S = S_in;
    S.e.split_rays = split ? 1 : 0;
#define LL_GO(KERNEL, PARAMS) hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, PARAMS, S)
    if (pmc_wants_xrows_terrain(P)) {                    // the extended contact rows (round 6): see launch_epmc_step
      if (!cone) throw PmcError(LL_EINVAL, "self_friction / pair_friction / max_pair / leg_edges need friction_mode 2 (the extended contact rows exist in the cone builds)");
      StepParams Q = P;
      Q.n_steps = 1;
      for (int sl = 0; sl < P.n_steps; sl++, Q.step_count++) {
        if (blocks <= sepmc_simds) LL_GO((sepmc_step_kernel<1, false, true, true>), Q); else LL_GO((sepmc_step_kernel<2, false, true, true>), Q);
        if (split) launch_percept(Q, S.e);
      }
    } else
    if (P.n_steps == 1) {
      if (blocks <= sepmc_simds) { if (cone) LL_GO((sepmc_step_kernel<1, false, true>), P); else LL_GO((sepmc_step_kernel<1>), P); }
      else                 { if (cone) LL_GO((sepmc_step_kernel<2, false, true>), P); else LL_GO((sepmc_step_kernel<2>), P); }
      if (split) launch_percept(P, S.e);
    } else if (blocks <= sepmc_simds && !split) {
      if (cone) LL_GO((sepmc_step_kernel<1, true, true>), P); else LL_GO((sepmc_step_kernel<1, true>), P);
    } else {
      StepParams Q = P;                                  // larger batches, and every batch with the rays split off: single launches (see epmc_step_kernel)
      Q.n_steps = 1;
      for (int sl = 0; sl < P.n_steps; sl++, Q.step_count++) {
        if (blocks <= sepmc_simds) { if (cone) LL_GO((sepmc_step_kernel<1, false, true>), Q); else LL_GO((sepmc_step_kernel<1>), Q); }
        else                 { if (cone) LL_GO((sepmc_step_kernel<2, false, true>), Q); else LL_GO((sepmc_step_kernel<2>), Q); }
        if (split) launch_percept(Q, S.e);
      }
    }
#undef LL_GO
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->second, stream));
  }
  void launch_sepmc_reset(const StepParams& P, const SepmcParams& S, const int32_t* ids, int n, const float* draws, const float* prev_orn) {
    use();
    const int blocks = (n + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    hipLaunchKernelGGL(sepmc_reset_kernel, dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P, S, ids, n, draws, prev_orn);
    HIPCHK(hipGetLastError());
  }
  void launch_step(const StepParams& P) {
    use();
    const int blocks = (P.n_envs + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    std::pair<hipEvent_t, hipEvent_t>* ev = timing_begin(P.n_steps);
    const bool one = blocks <= simds, multi = P.n_steps > 1;
    if (pmc_wants_xrows(P)) {                            // the extended contact rows (round 6): flat-ground cone builds with XROWS
      if (P.friction_mode != 2 || P.set_obstacle) throw PmcError(LL_EINVAL, "self_friction needs friction_mode 2 and no jump obstacle (the extended contact rows exist in the flat-ground cone builds)");
      if (one) { if (multi) hipLaunchKernelGGL((pmc_step_kernel<1, false, true, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<1, false, false, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P); }
      else     { if (multi) hipLaunchKernelGGL((pmc_step_kernel<2, false, true, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<2, false, false, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P); }
    } else
    if (P.set_obstacle && P.friction_mode == 2) {
      if (one) { if (multi) hipLaunchKernelGGL((pmc_step_kernel<1, true, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<1, true, false, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P); }
      else     { if (multi) hipLaunchKernelGGL((pmc_step_kernel<2, true, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<2, true, false, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P); }
    } else if (P.set_obstacle) {
      if (one) { if (multi) hipLaunchKernelGGL((pmc_step_kernel<1, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<1, true, false>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P); }
      else     { if (multi) hipLaunchKernelGGL((pmc_step_kernel<2, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<2, true, false>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P); }
    } else if (P.friction_mode == 2) {
      if (one) { if (multi) hipLaunchKernelGGL((pmc_step_kernel<1, false, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<1, false, false, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P); }
      else     { if (multi) hipLaunchKernelGGL((pmc_step_kernel<2, false, true, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P);      // (row scratch: WithConeInLds)
                 else       hipLaunchKernelGGL((pmc_step_kernel<2, false, false, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes_epmc(), stream, P); }
    } else {
      if (one) { if (multi) hipLaunchKernelGGL((pmc_step_kernel<1, false, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<1, false, false>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P); }
      else     { if (multi) hipLaunchKernelGGL((pmc_step_kernel<2, false, true>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P);
                 else       hipLaunchKernelGGL((pmc_step_kernel<2, false, false>), dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P); }
    }
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->second, stream));
  }
  void launch_reset(const StepParams& P, const int32_t* ids, int n, const int32_t* clip, const double* t0) {
    use();
    const int blocks = (n + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    hipLaunchKernelGGL(pmc_reset_kernel, dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P, ids, n, clip, t0);
    HIPCHK(hipGetLastError());
  }
  void launch_probe_pd(const StepParams& P, const float* in, float* out, int n, int mode) {
    use();
    hipLaunchKernelGGL(pmc_probe_pd_kernel, dim3((n + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE), dim3(PMC_WAVE), lds_bytes(), stream, P, in, out, n, mode);
    HIPCHK(hipGetLastError());
  }
  void launch_gae(float* block, int n_envs, int unroll, int W, int od, float gamma, float lam, const float* bootstrap) {
    use();
    hipLaunchKernelGGL(pmc_gae_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, stream, block, n_envs, unroll, W, od, gamma, lam, bootstrap);
    HIPCHK(hipGetLastError());
  }
  void launch_actions(const StepParams& P, float* actions, float sigma) {
    use();
    const int threads = 256;
    hipLaunchKernelGGL(pmc_actions_kernel, dim3((P.n_envs * 3 + threads - 1) / threads), dim3(threads), 0, stream, P, actions, sigma);
    HIPCHK(hipGetLastError());
  }
  void enable_timing(bool on) { timing = on; }
  void collect_timing(double* avg_ms, int* n, long long* n_steps = nullptr) {
    sync();
    double tot = 0;
    long long steps = 0;
    for (size_t i = 0; i < ev_used; i++) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, evs[i].first, evs[i].second));
      tot += ms;
      steps += ev_steps[i];
    }
    *n = (int)ev_used;
    *avg_ms = ev_used ? tot / ev_used : 0.0;
    if (n_steps) *n_steps = steps;
    ev_used = 0;
  }
};

typedef PmcEngine<HipBackend> ENGINE;
#include "pmc_capi.inc"
typedef EpmcEngine<HipBackend> EPMC_ENGINE;
#include "epmc_capi.inc"
typedef SepmcEngine<HipBackend> SEPMC_ENGINE;
#include "sepmc_capi.inc"
#include "pmc_policy.inc"
#include "xfer_capi.inc"
#endif  // LL_KERNELS_ONLY
