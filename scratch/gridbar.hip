// What does a device-wide barrier cost on MI355X next to a kernel boundary?  (the question behind "one
// persistent kernel per dense block".)  A dense layer on the 20x20 / 10x10 maps is, per convolution, "every workgroup leaves a
// statistic row -> somebody sums the rows in row order -> every workgroup of the next kernel reads the coefficients".  Two forms:
//
//   A  launches:   producer kernel (256 workgroups, each stores its row of C floats) -> coefficient kernel (C / 16 workgroups sum the
//                  rows in order, write C coefficients) -> next producer reads the coefficients ...; all in one stream (and as a hipGraph)
//   B  persistent: ONE kernel, 256 workgroups (one per CU); per iteration: store the row, release, agent-scope counter add, the
//                  workgroups poll the counter (sc1 loads), acquire, then EVERY workgroup sums the rows of its 1/256 of the channels ...
//                  no: the coefficient vector is needed by all, so form B1 lets every workgroup sum all rows of all channels
//                  (redundant reads from L2), form B2 lets workgroup w sum channels [w*C/256, ...) and publishes them behind a SECOND barrier.
//
// Reported: microseconds per iteration (= per convolution boundary).  Every poll loop is bounded by a clock (a barrier that cannot
// complete -- workgroups not co-resident -- sets an error flag and falls through, so the grid always drains).
//
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scratch/gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NT = 512;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
  // MI355X_MICROARCH.md's counter barrier: every storing wave drains its stores, the workgroup meets, ONE lane releases (L2
  // write-back, agent scope), adds to the counter, polls it with relaxed sc1 loads, acquires once (L1 invalidate), waits for the
  // invalidate; the other waves load after the workgroup barrier that lane then joins
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (wall_clock64() - t0 > 100000000ll) { *err = 1; ok = false; break; }     // 1 s at 100 MHz: never hang the GPU
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return ok;
}

// ---- form A
__global__ __launch_bounds__(NT) void producer_kernel(float* rows, const float* coef, int C, int it) {
  // reads the coefficients (dependency on the previous coefficient kernel), leaves its row
  const int w = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += NT) rows[(size_t)w * C + c] = coef[c] * 0.5f + (float)(w + it);
}
__global__ __launch_bounds__(1024) void coef_kernel(const float* rows, float* coef, int C, int R) {
  // 16 channels x 64 row lanes per workgroup, rows summed in a fixed order (as bn_coef_kernel does)
  __shared__ float part[64][17];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
  float s = 0.f;
  for (int r = rl; r < R; r += 64) s += rows[(size_t)r * C + c];
  part[rl][threadIdx.x & 15] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += part[i][threadIdx.x];
    coef[blockIdx.x * 16 + threadIdx.x] = t * (1.f / R);
  }
}

// ---- form B1: one barrier, every workgroup sums all rows of all channels
__global__ __launch_bounds__(NT) void persistent_b1(float* rows, float* out, unsigned* counter, int* err, int C, int iters) {
  const int w = blockIdx.x, G = gridDim.x;
  __shared__ float coef[2048];
  for (int c = threadIdx.x; c < C; c += NT) coef[c] = 0.f;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    float* rw = rows + (size_t)(it & 1) * G * C;                  // two row buffers: iteration it+1 may start storing while a slow
    for (int c = threadIdx.x; c < C; c += NT) rw[(size_t)w * C + c] = coef[c] * 0.5f + (float)(w + it);    // workgroup still reads it
    if (!grid_barrier(counter, (unsigned)(it + 1) * G, err)) break;
    for (int c = threadIdx.x; c < C; c += NT) {
      float s = 0.f;
      for (int r = 0; r < G; ++r) s += rw[(size_t)r * C + c];
      coef[c] = s * (1.f / G);
    }
    __syncthreads();
  }
  if (w == 0) for (int c = threadIdx.x; c < C; c += NT) out[c] = coef[c];
}

// ---- form B2: two barriers, the channels are divided over the workgroups (each sums its share in row order), then published
__global__ __launch_bounds__(NT) void persistent_b2(float* rows, float* gcoef, float* out, unsigned* counter, int* err, int C, int iters) {
  const int w = blockIdx.x, G = gridDim.x;
  __shared__ float coef[2048];
  for (int c = threadIdx.x; c < C; c += NT) coef[c] = 0.f;
  __syncthreads();
  unsigned phase = 0;
  for (int it = 0; it < iters; ++it) {
    for (int c = threadIdx.x; c < C; c += NT) rows[(size_t)w * C + c] = coef[c] * 0.5f + (float)(w + it);
    if (!grid_barrier(counter, ++phase * G, err)) break;
    // channel c is summed by workgroup c % G; 64 row lanes per channel inside the workgroup
    for (int c = w; c < C; c += G) {
      float s = 0.f;
      for (int r = threadIdx.x; r < G; r += NT) s += rows[(size_t)r * C + c];
      // fixed-order fold over the workgroup (LDS)
      __shared__ float part[NT];
      part[threadIdx.x] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (G < NT ? G : NT); ++i) t += part[i];
        gcoef[c] = t * (1.f / G);
      }
      __syncthreads();
    }
    if (!grid_barrier(counter, ++phase * G, err)) break;
    for (int c = threadIdx.x; c < C; c += NT) coef[c] = gcoef[c];
    __syncthreads();
  }
  if (w == 0) for (int c = threadIdx.x; c < C; c += NT) out[c] = coef[c];
}

// ---- barrier alone
__global__ __launch_bounds__(NT) void barrier_only(unsigned* counter, int* err, int iters) {
  for (int it = 0; it < iters; ++it)
    if (!grid_barrier(counter, (unsigned)(it + 1) * gridDim.x, err)) break;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  printf("device %s, %d CUs: grid = %d workgroups of %d threads\n", prop.name, G, G, NT);
  const int iters = 400;
  float *rows, *coef, *gcoef, *out;
  unsigned* counter;
  int* err;
  CK(hipMalloc(&rows, (size_t)2 * G * 2048 * 4));
  CK(hipMalloc(&coef, 2048 * 4));
  CK(hipMalloc(&gcoef, 2048 * 4));
  CK(hipMalloc(&out, 2048 * 4));
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&err, 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto reset = [&] { CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(coef, 0, 2048 * 4, st)); };
  auto elapsed_us = [&](int n) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / n; };
  auto check_err = [&](const char* what) { int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost)); if (h) printf("  !! %s: a barrier timed out (workgroups not co-resident?)\n", what); };

  // barrier alone
  for (int rep = 0; rep < 2; ++rep) {
    reset();
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(barrier_only, dim3(G), dim3(NT), 0, st, counter, err, iters);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    if (rep) printf("grid barrier alone (one lane: release, agent-scope add, relaxed sc1 poll, acquire; %d workgroups): %.2f us per barrier\n", G, elapsed_us(iters));
  }
  check_err("barrier_only");

  for (int C : {32, 128, 512, 1024}) {
    // form A, eager stream
    float a_eager = 0, a_graph = 0, b1 = 0, b2 = 0;
    for (int rep = 0; rep < 2; ++rep) {
      reset();
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(producer_kernel, dim3(G), dim3(NT), 0, st, rows, coef, C, it);
        hipLaunchKernelGGL(coef_kernel, dim3(C / 16), dim3(1024), 0, st, rows, coef, C, G);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      a_eager = elapsed_us(iters);
    }
    // form A as a hipGraph (what the training step replays)
    {
      hipGraph_t graph;
      hipGraphExec_t exec;
      reset();
      CK(hipStreamSynchronize(st));
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(producer_kernel, dim3(G), dim3(NT), 0, st, rows, coef, C, it);
        hipLaunchKernelGGL(coef_kernel, dim3(C / 16), dim3(1024), 0, st, rows, coef, C, G);
      }
      CK(hipStreamEndCapture(st, &graph));
      CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        a_graph = elapsed_us(iters);
      }
      CK(hipGraphExecDestroy(exec));
      CK(hipGraphDestroy(graph));
    }
    for (int rep = 0; rep < 2; ++rep) {
      reset();
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent_b1, dim3(G), dim3(NT), 0, st, rows, out, counter, err, C, iters);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      b1 = elapsed_us(iters);
    }
    check_err("persistent_b1");
    for (int rep = 0; rep < 2; ++rep) {
      reset();
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent_b2, dim3(G), dim3(NT), 0, st, rows, gcoef, out, counter, err, C, iters);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      b2 = elapsed_us(iters);
    }
    check_err("persistent_b2");
    printf("C = %4d channels, %d rows: launches (producer + coefficient kernel) eager %.2f us, hipGraph %.2f us | persistent, one barrier, "
           "every workgroup sums all rows %.2f us | persistent, two barriers, channels divided %.2f us   (per convolution boundary)\n",
           C, G, a_eager, a_graph, b1, b2);
  }
  return 0;
}
