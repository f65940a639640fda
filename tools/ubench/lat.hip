// Latency micro-benchmarks that size the mjhip design (tools only, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 lat.hip -o lat && ./lat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define N 4096
__global__ void k_clock(long long* out, int spin) {
  long long w0 = wall_clock64(); long long c0 = clock64();
  long long w = w0;
  while (w - w0 < spin) w = wall_clock64();
  out[0] = w - w0; out[1] = clock64() - c0;
}
// pointer chase: p = a[p]
__global__ void k_chase_global(const int* a, int* out, long long* t, int iters) {
  int p = threadIdx.x & 0;   // all lanes same chain
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) p = a[p];
  long long c1 = clock64();
  if (threadIdx.x == 0) { t[blockIdx.x] = c1 - c0; out[blockIdx.x] = p; }
}
__global__ void k_chase_scalar(const int* __restrict__ a, int* out, long long* t, int iters) {
  int p = 0;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) p = __builtin_amdgcn_readfirstlane(a[__builtin_amdgcn_readfirstlane(p)]);
  long long c1 = clock64();
  if (threadIdx.x == 0) { t[blockIdx.x] = c1 - c0; out[blockIdx.x] = p; }
}
extern __shared__ int lds[];
__global__ void k_chase_ds(const int* a, int* out, long long* t, int iters) {
  for (int i = threadIdx.x; i < N; i += 64) lds[i] = a[i];
  __syncthreads();
  int p = 0;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) p = lds[p];
  long long c1 = clock64();
  if (threadIdx.x == 0) { t[blockIdx.x] = c1 - c0; out[blockIdx.x] = p; }
}
__device__ __forceinline__ int* flat_lds() {
  unsigned long long base; asm volatile("s_mov_b64 %0, src_shared_base" : "=s"(base)); return (int*)base;
}
__global__ void k_chase_flat(const int* a, int* out, long long* t, int iters, int use_lds) {
  for (int i = threadIdx.x; i < N; i += 64) lds[i] = a[i];
  __syncthreads();
  const int* q = use_lds ? flat_lds() : a;
  int p = 0;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) p = q[p];
  long long c1 = clock64();
  if (threadIdx.x == 0) { t[blockIdx.x] = c1 - c0; out[blockIdx.x] = p; }
}
// store -> sync -> load by another lane, via flat LDS / ds / global
__global__ void k_pingpong(int* g, int* out, long long* t, int iters, int mode) {
  int* q = mode == 0 ? (int*)lds : (mode == 1 ? flat_lds() : g + blockIdx.x * 64);
  int lane = threadIdx.x;
  int v = lane;
  q[lane] = v;
  __syncthreads();
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) {
    v = q[(lane + 1) & 63] + 1;
    __syncthreads();
    q[lane] = v;
    __syncthreads();
  }
  long long c1 = clock64();
  if (lane == 0) { t[blockIdx.x] = c1 - c0; out[blockIdx.x] = v; }
}
// dependent fp64 add chain, readlane chain
__global__ void k_fadd(double* out, long long* t, int iters, double x) {
  double a = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) a = a * x + x;
  long long c1 = clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ void k_readlane(double* out, long long* t, int iters, double x) {
  double a = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) {
    int lo = __builtin_amdgcn_readlane(__double2loint(a), i & 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(a), i & 63);
    a = a + __hiloint2double(hi, lo) * x;
  }
  long long c1 = clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ void k_shfl(double* out, long long* t, int iters, double x) {
  double a = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < iters; i++) a = a + __shfl(a, (threadIdx.x + i) & 63, 64) * x;
  long long c1 = clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
  out[blockIdx.x * 64 + threadIdx.x] = a;
}

int main() {
  int* a; int* out; long long* t; double* dout;
  hipMalloc(&a, N * 4 * 64); hipMalloc(&out, 65536 * 4); hipMalloc(&t, 65536 * 8); hipMalloc(&dout, 65536 * 64 * 8);
  std::vector<int> h(N);
  for (int i = 0; i < N; i++) h[i] = (i * 1237 + 1) % N;
  hipMemcpy(a, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<long long> ht(65536);
  long long* clk; hipMalloc(&clk, 16);
  // clock rates
  {
    auto t0 = std::chrono::steady_clock::now();
    k_clock<<<1, 64>>>(clk, 10000000);   // spin for 1e7 wall_clock ticks
    hipDeviceSynchronize();
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("wall_clock64: %lld ticks in %.4f s host => %.2f MHz ; clock64: %lld ticks => %.2f MHz\n", hc[0], el, hc[0] / el / 1e6, hc[1], hc[1] / el / 1e6);
  }
  auto report = [&](const char* name, int blocks, int iters) {
    hipDeviceSynchronize();
    hipMemcpy(ht.data(), t, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; i++) s += ht[i];
    printf("%-34s blocks=%5d : %.1f clock64 ticks / iter\n", name, blocks, s / blocks / iters);
  };
  int iters = 2000;
  for (int blocks : {1, 2048, 4096}) {
    k_chase_global<<<blocks, 64>>>(a, out, t, iters); report("vector global chase (L2/L1 hit)", blocks, iters);
    k_chase_scalar<<<blocks, 64>>>(a, out, t, iters); report("scalar (s_load) chase", blocks, iters);
    k_chase_ds<<<blocks, 64, N * 4>>>(a, out, t, iters); report("ds_read chase", blocks, iters);
    k_chase_flat<<<blocks, 64, N * 4>>>(a, out, t, iters, 1); report("flat->LDS chase", blocks, iters);
    k_chase_flat<<<blocks, 64, N * 4>>>(a, out, t, iters, 0); report("flat->global chase", blocks, iters);
    for (int mode = 0; mode < 3; mode++) {
      k_pingpong<<<blocks, 64, N * 4>>>(a + N, out, t, iters, mode);
      report(mode == 0 ? "pingpong ds (ld,sync,st,sync)" : mode == 1 ? "pingpong flat-LDS" : "pingpong global", blocks, iters);
    }
    k_fadd<<<blocks, 64>>>(dout, t, iters, 1.0000001); report("dependent fp64 mul+add", blocks, iters);
    k_readlane<<<blocks, 64>>>(dout, t, iters, 1.0000001); report("readlane x2 + fp64 mul,add", blocks, iters);
    k_shfl<<<blocks, 64>>>(dout, t, iters, 1.0000001); report("__shfl(double) + fp64 mul,add", blocks, iters);
  }
  return 0;
}
