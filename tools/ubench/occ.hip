// Residency census: how many one-wavefront workgroups run concurrently for a given LDS / VGPR / scratch
// footprint (tools only).  hipcc --offload-arch=gfx950 -O3 occ.hip -o occ && ./occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern __shared__ char lds[];
template <int SCRATCH_WORDS, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k(long long* t, int spin_ticks, int sel, double* sink) {
  long long t0 = wall_clock64();
  double acc = 0;
  if (SCRATCH_WORDS > 0) {
    volatile int buf[SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1];
    for (int i = 0; i < SCRATCH_WORDS; i++) buf[i] = i + sel;
    acc += buf[(sel * 7 + threadIdx.x) % (SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1)];
  }
  lds[threadIdx.x] = (char)sel;
  while (wall_clock64() - t0 < spin_ticks) { acc += 1.0; }
  if (threadIdx.x == 0) { t[2*blockIdx.x] = t0; t[2*blockIdx.x+1] = wall_clock64(); }
  if (sel == 12345) sink[threadIdx.x] = acc + lds[threadIdx.x ^ 1];
}
template <int S, int W> void run(const char* name, int lds_bytes, long long* t, double* sink) {
  const int N = 4096, T = 200000;   // 2 ms at 100 MHz
  hipFuncSetAttribute((const void*)k<S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<S, W><<<N, 64, lds_bytes>>>(t, 1000, 1, sink);
  hipDeviceSynchronize();
  k<S, W><<<N, 64, lds_bytes>>>(t, T, 1, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(2*N);
  hipMemcpy(h.data(), t, 2*N*8, hipMemcpyDeviceToHost);
  long long mn = h[0], mx = h[1];
  std::vector<long long> st(N);
  for (int i = 0; i < N; i++) { mn = std::min(mn, h[2*i]); mx = std::max(mx, h[2*i+1]); st[i] = h[2*i]; }
  std::sort(st.begin(), st.end());
  printf("%-34s lds=%6d : span %.2f ms (ideal rounds x 2 ms), avg concurrency %.0f, start quantiles us: 25%%=%.0f 50%%=%.0f 75%%=%.0f 100%%=%.0f\n",
         name, lds_bytes, (mx - mn) / 1e5, (double)N * T / (mx - mn), (st[N/4]-mn)/100.0, (st[N/2]-mn)/100.0, (st[3*N/4]-mn)/100.0, (st[N-1]-mn)/100.0);
}
int main() {
  long long* t; double* sink; hipMalloc(&t, 4096*16); hipMalloc(&sink, 4096);
  for (int lds_bytes : {0, 10240, 16384, 20480, 24576}) {
    run<0, 2>("no scratch, 2 waves/SIMD budget", lds_bytes, t, sink);
    run<64, 2>("256 B/lane scratch, 2 waves/SIMD", lds_bytes, t, sink);
    run<128, 2>("512 B/lane scratch, 2 waves/SIMD", lds_bytes, t, sink);
    run<0, 4>("no scratch, 4 waves/SIMD budget", lds_bytes, t, sink);
    run<128, 4>("512 B/lane scratch, 4 waves/SIMD", lds_bytes, t, sink);
  }
  return 0;
}
