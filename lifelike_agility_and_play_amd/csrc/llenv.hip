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

#include "lanes.hpp"
#include "pmc_engine.hpp"
#include "pmc_step.hpp"

#define HIPCHK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess) throw PmcError(LL_EHIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

typedef Pmc<GpuLanes> K;

__global__ __launch_bounds__(PMC_WAVE, 2) void pmc_step_kernel(StepParams P) {   // <= 256 registers: two waves per SIMD
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int env = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 4);      // one env = one 16-lane DPP row
  GpuLanes ln(lds);
  ln.stage_consts(P.legc, LC_COUNT, P.candc, CAND_TABLE_WORDS);           // all 64 lanes copy, also those without an env
  if (env >= P.n_envs) return;
  K::step_env(ln, P, env);
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

// PLE:235-240 for the batch, one 256-thread block: fold the statistics published by finished episodes into the per-clip
// table (one thread per clip), then rebuild p ~ (1 - avg_reward_sum)^factor and its inclusive CDF.
__global__ __launch_bounds__(256) void pmc_table_kernel(StepParams P, double* avg_r, double* avg_l, double* prob, double* cdf) {
  __shared__ double red[256];
  __shared__ int any_pending;
  const int tid = threadIdx.x;
  if (tid == 0) any_pending = 0;
  __syncthreads();
  for (int c = tid; c < P.n_clips; c += 256) {
    unsigned long long pr = P.pending_reward[c], pl = P.pending_len[c];
    if (pr) {
      avg_r[c] = (double)__uint_as_float((uint32_t)pr);
      avg_l[c] = (double)__uint_as_float((uint32_t)pl);
      P.pending_reward[c] = 0ull;
      P.pending_len[c] = 0ull;
      any_pending = 1;
    }
  }
  __syncthreads();
  if (!any_pending) return;
  double part = 0.0;
  for (int c = tid; c < P.n_clips; c += 256) {
    double p = pow(1.0 - avg_r[c], P.sample_factor);
    prob[c] = p;
    part += p;
  }
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double inv = 1.0 / red[0];
  for (int c = tid; c < P.n_clips; c += 256) prob[c] *= inv;
  __syncthreads();
  if (tid == 0) {
    double acc = 0.0;
    for (int c = 0; c < P.n_clips; c++) { acc += prob[c]; cdf[c] = acc; }
    cdf[P.n_clips - 1] = 1.0;
  }
}

// synthetic random policy: a ~ N(0, sigma^2), Philox4x32-10 keyed on (seed; element, step) + Box-Muller, four per thread
__global__ void pmc_actions_kernel(StepParams P, float* actions, float sigma) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= P.n_envs * 3) return;
  uint32_t r[4];
  philox4x32((uint32_t)gid, (uint32_t)P.step_count, (uint32_t)(P.step_count >> 32), 0xAC710u, (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r);
  const float k = 2.3283064365386963e-10f;   // 2^-32
  float u1 = fminf(((float)r[0] + 1.0f) * k, 1.0f), u2 = (float)r[1] * k, u3 = fminf(((float)r[2] + 1.0f) * k, 1.0f), u4 = (float)r[3] * k;
  float m1 = sqrtf(-2.0f * logf(u1)), m2 = sqrtf(-2.0f * logf(u3));
  float4 o;
  o.x = sigma * m1 * cosf(6.283185307179586f * u2); o.y = sigma * m1 * sinf(6.283185307179586f * u2);
  o.z = sigma * m2 * cosf(6.283185307179586f * u4); o.w = sigma * m2 * sinf(6.283185307179586f * u4);
  reinterpret_cast<float4*>(actions)[gid] = o;
}

struct HipBackend {
  int device;
  hipStream_t own = nullptr, stream = nullptr;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  size_t ev_used = 0;

  explicit HipBackend(int dev) : device(dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw PmcError(LL_ENODEV, "no HIP device available: this engine has no CPU fallback");
    if (dev < 0 || dev >= n) throw PmcError(LL_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    stream = own;
  }
  ~HipBackend() {
    (void)hipSetDevice(device);
    for (auto& p : evs) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (own) (void)hipStreamDestroy(own);
  }
  void use() { HIPCHK(hipSetDevice(device)); }
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

  static size_t lds_bytes() { return ((size_t)LC_COUNT * 4 + (size_t)CAND_TABLE_WORDS * PMC_ROW) * sizeof(float); }
  void launch_step(const StepParams& P) {
    use();
    const int blocks = (P.n_envs + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (timing) {
      if (ev_used == evs.size()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        evs.push_back(std::make_pair(a, b));
      }
      ev = &evs[ev_used++];
      HIPCHK(hipEventRecord(ev->first, stream));
    }
    hipLaunchKernelGGL(pmc_step_kernel, dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P);
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev->second, stream));
  }
  void launch_reset(const StepParams& P, const int32_t* ids, int n, const int32_t* clip, const double* t0) {
    use();
    const int blocks = (n + PMC_ENVS_PER_WAVE - 1) / PMC_ENVS_PER_WAVE;
    hipLaunchKernelGGL(pmc_reset_kernel, dim3(blocks), dim3(PMC_WAVE), lds_bytes(), stream, P, ids, n, clip, t0);
    HIPCHK(hipGetLastError());
  }
  void launch_prestep(const StepParams& P, double* avg_r, double* avg_l, double* prob, double* cdf, float* actions, float sigma) {
    use();
    hipLaunchKernelGGL(pmc_table_kernel, dim3(1), dim3(256), 0, stream, P, avg_r, avg_l, prob, cdf);
    HIPCHK(hipGetLastError());
    if (actions) {
      const int threads = 256;
      hipLaunchKernelGGL(pmc_actions_kernel, dim3((P.n_envs * 3 + threads - 1) / threads), dim3(threads), 0, stream, P, actions, sigma);
      HIPCHK(hipGetLastError());
    }
  }
  void enable_timing(bool on) { timing = on; }
  void collect_timing(double* avg_ms, int* n) {
    sync();
    double tot = 0;
    for (size_t i = 0; i < ev_used; i++) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, evs[i].first, evs[i].second));
      tot += ms;
    }
    *n = (int)ev_used;
    *avg_ms = ev_used ? tot / ev_used : 0.0;
    ev_used = 0;
  }
};

typedef PmcEngine<HipBackend> ENGINE;
#include "pmc_capi.inc"
