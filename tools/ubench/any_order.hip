// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950 (AQL barrier bit cleared)?
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/any_order.hip -o tools/ubench/any_order ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(long long cycles, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, 1);
}
int main() {
  int *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long long cyc = 2000;   // wall_clock64 ticks at 100 MHz: 20 us
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, d);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, cyc, d);
      else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1ll, d);   // a dependent third kernel
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.1f us\n", mode == 0 ? "two kernels, in order      " : mode == 1 ? "second with AnyOrderLaunch " : "one kernel                 ", best * 1e3f);
  }
  return 0;
}
