// How much dynamic LDS may a 256-thread workgroup use before only ONE of them is resident per CU?  512 workgroups that each
// wait ~20 us: co-resident pairs finish in one wait, serialized ones in two.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) spin(int* out, long long ticks) {
  extern __shared__ int sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) out[blockIdx.x] = sm[255];
}
int main() {
  int* out; hipMalloc(&out, 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int kb2 = 60 * 2; kb2 <= 84 * 2; kb2 += 1) {
    const int bytes = kb2 * 512;
    if (hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { printf("%d B: attr failed\n", bytes); continue; }
    spin<<<512, 256, bytes>>>(out, 2000); hipDeviceSynchronize();
    hipEventRecord(a); spin<<<512, 256, bytes>>>(out, 2000); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%6d B dynamic LDS: %6.1f us for 512 workgroups of 20 us (%s)\n", bytes, ms * 1e3, ms * 1e3 < 32 ? "two per CU" : "one per CU");
  }
  return 0;
}
