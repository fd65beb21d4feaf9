// How fast does the chip start workgroups?  Empty kernels (exit at once) of 64..1024 threads, with and without a static
// LDS block, over grids of 128..8192 workgroups; back-to-back launches on one stream, hipEvent time per launch.  The slope of
// time(workgroups) is the dispatch cost per workgroup; the intercept is the kernel boundary.
//   hipcc -O3 --offload-arch=gfx950 tools/dispatch_rate.hip -o tools/dispatch_rate && tools/dispatch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int T, int LDS_FLOATS>
__global__ __launch_bounds__(T) void empty_kernel(float* out, int never) {
  __shared__ float lds[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
  if (never) {  // keeps the LDS block and the output alive
    lds[threadIdx.x] = (float)blockIdx.x;
    __syncthreads();
    out[blockIdx.x * T + threadIdx.x] = lds[(threadIdx.x + 1) % T];
  }
}

template <int T, int L>
static void sweep(const char* name, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  printf("%s\n", name);
  double t_first = 0, wg_first = 0;
  for (int wgs : {128, 256, 512, 1024, 2048, 4096, 8192}) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((empty_kernel<T, L>), dim3(wgs), dim3(T), 0, 0, out, 0);
    hipDeviceSynchronize();
    const int reps = 400;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((empty_kernel<T, L>), dim3(wgs), dim3(T), 0, 0, out, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    if (wgs == 128) t_first = us, wg_first = wgs;
    printf("  %5d workgroups  %7.2f us per launch   (%.1f ns per extra workgroup over the 128-workgroup launch)\n", wgs, us,
           wgs > 128 ? (us - t_first) * 1e3 / (wgs - wg_first) : 0.0);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 8192 * 1024 * sizeof(float));
  sweep<64, 0>("64 threads, no LDS", out);
  sweep<256, 0>("256 threads, no LDS", out);
  sweep<256, 9216>("256 threads, 36 KB LDS", out);
  sweep<512, 0>("512 threads, no LDS", out);
  sweep<1024, 0>("1024 threads, no LDS", out);
  return 0;
}
