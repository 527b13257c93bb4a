// llenv.hip -- the product: HIP kernels for gfx950 + the C ABI of include/llenv.h.
//
//   pmc_step_kernel     one 50 Hz control step of every env, fully fused (pmc_step.hpp); 1 env = 4 lanes,
//                       64-thread workgroups (one wavefront, 16 envs), contact rows staged in LDS
//   pmc_reset_kernel    PLE:150-171 for a list of envs
//   pmc_prestep_kernel  folds finished-episode statistics into the sampling table (PLE:235-240) and, on request,
//                       draws the synthetic random-policy actions (Philox + Box-Muller)
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

__global__ __launch_bounds__(PMC_WAVE) void pmc_step_kernel(StepParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // envs_per_wave < 16 leaves lanes idle on purpose: at 4096 envs there are more SIMDs (1024) than 16-env waves (256),
  // and the solver's wave-uniform work is the union over the envs of a wave, so fewer envs per wave = shorter kernel.
  const int quad = threadIdx.x >> 2;
  const int env = blockIdx.x * P.envs_per_wave + quad;
  if (quad >= P.envs_per_wave || env >= P.n_envs) return;
  GpuLanes ln(lds);
  K::clear_scratch(ln);
  K::step_env(ln, P, env);
}

__global__ __launch_bounds__(PMC_WAVE) void pmc_reset_kernel(StepParams P, const int32_t* ids, int n, const int32_t* clip, const double* t0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int i = blockIdx.x * PMC_ENVS_PER_WAVE + (threadIdx.x >> 2);
  if (i >= n) return;
  GpuLanes ln(lds);
  const int env = ids ? ids[i] : i;
  int c;
  double t;
  uint32_t ep = P.ep_count[env] + 1;          // the four lanes of the quad read, then write, the same value
  K::sample_start(P, env, ep, &c, &t);         // ML:59-63, ML:50-51
  P.ep_count[env] = ep;
  if (clip) c = clip[i];
  if (t0) t = t0[i];
  K::reset_env(ln, P, env, c, t);
  P.done[env] = 0;
  P.done_reason[env] = 0;
}

__global__ void pmc_prestep_kernel(StepParams P, double* avg_r, double* avg_l, double* prob, double* cdf, float* actions, float sigma) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid == 0) pmc_finalize_table(P, avg_r, avg_l, prob, cdf);
  if (actions) {
    const int total = P.n_envs * 3;              // four normals per thread
    if (gid < total) {
      uint32_t r[4];
      philox4x32((uint32_t)gid, (uint32_t)P.step_count, (uint32_t)(P.step_count >> 32), 0xAC710u, (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r);
      const float k = 2.3283064365386963e-10f;   // 2^-32
      float u1 = ((float)r[0] + 1.0f) * k, u2 = (float)r[1] * k, u3 = ((float)r[2] + 1.0f) * k, u4 = (float)r[3] * k;
      u1 = fminf(u1, 1.0f); u3 = fminf(u3, 1.0f);
      float m1 = sqrtf(-2.0f * logf(u1)), m2 = sqrtf(-2.0f * logf(u3));
      float4 o;
      o.x = sigma * m1 * cosf(6.283185307179586f * u2); o.y = sigma * m1 * sinf(6.283185307179586f * u2);
      o.z = sigma * m2 * cosf(6.283185307179586f * u4); o.w = sigma * m2 * sinf(6.283185307179586f * u4);
      reinterpret_cast<float4*>(actions)[gid] = o;
    }
  }
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
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    n_simd = prop.multiProcessorCount * 4;
    if (const char* e = getenv("LL_ENVS_PER_WAVE")) { int v = atoi(e); if (v >= 1 && v <= PMC_ENVS_PER_WAVE) epw_override = v; }
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

  static size_t lds_bytes() { return (size_t)LW_COUNT * PMC_WAVE * sizeof(float); }
  int epw_override = 0, n_simd = 1024;
  int envs_per_wave(int n_envs) const {
    if (epw_override > 0) return epw_override;
    int e = (n_envs + n_simd - 1) / n_simd;       // aim for at least one wave per SIMD
    return e < 1 ? 1 : (e > PMC_ENVS_PER_WAVE ? PMC_ENVS_PER_WAVE : e);
  }
  void launch_step(const StepParams& Pin) {
    use();
    StepParams P = Pin;
    P.envs_per_wave = envs_per_wave(P.n_envs);
    const int blocks = (P.n_envs + P.envs_per_wave - 1) / P.envs_per_wave;
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
    const int threads = 256;
    const int blocks = actions ? (P.n_envs * 3 + threads - 1) / threads : 1;
    hipLaunchKernelGGL(pmc_prestep_kernel, dim3(blocks), dim3(actions ? threads : 64), 0, stream, P, avg_r, avg_l, prob, cdf, actions, sigma);
    HIPCHK(hipGetLastError());
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
