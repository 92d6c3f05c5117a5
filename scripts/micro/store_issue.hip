// Micro-benchmark: how fast can one CU issue 16-byte-per-lane stores, by shape of the 1 KiB a wave-instruction covers?
// 256 workgroups x 8 waves; every wave issues ITER buffer stores of distinct addresses (streaming, no re-use).
//   shape 0: 1 row  x 1024 B     shape 1: 2 rows x 512 B     shape 2: 4 rows x 256 B     shape 3: 16 rows x 64 B
// Build/run: hipcc --offload-arch=gfx950 -O3 scripts/micro/store_issue.hip -o gpurun_out/store_issue && gpurun_out/store_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int SHAPE, int NT, int WIDTH16>
__global__ void __launch_bounds__(512) store_kernel(unsigned char* out, int iters, long row_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int lanes_per_row = SHAPE == 0 ? 64 : SHAPE == 1 ? 32 : SHAPE == 2 ? 16 : 4;
  constexpr int rows = 64 / lanes_per_row;
  const int r = lane / lanes_per_row, c = lane % lanes_per_row;
  // each wave owns a private slab: iters * rows rows
  unsigned char* base = out + ((long)(blockIdx.x * 8 + wave) * iters * rows) * row_stride;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFFu, 0x00020000);
  u32x4 v = {(unsigned)lane, (unsigned)wave, (unsigned)blockIdx.x, 7u};
  const unsigned voff = (unsigned)(r * row_stride + c * 16);
  for (int i = 0; i < iters; ++i) {
    const unsigned soff = (unsigned)((long)i * rows * row_stride);
    if (WIDTH16) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, NT ? 2 : 0);
    else {
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[0], v[1]}, rs, voff / 2, soff / 2, NT ? 2 : 0);
    }
    asm volatile("s_nop 3");
    v[3] += 1;
  }
}

static int g_grid = 256;
template <int SHAPE, int NT, int W16>
static void run(const char* name, unsigned char* buf, int iters, long row_stride) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<SHAPE, NT, W16><<<g_grid, 512>>>(buf, iters, row_stride);
  hipEventRecord(e0);
  for (int k = 0; k < 5; ++k) store_kernel<SHAPE, NT, W16><<<g_grid, 512>>>(buf, iters, row_stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)g_grid * 8 * iters * (W16 ? 1024 : 512);
  const double instr = 8.0 * iters;  // wave-instructions per CU
  printf("%-44s %8.1f us  %6.2f TB/s  %6.1f B/clk/CU  %6.1f clk per wave-store (2.4 GHz)\n", name, ms * 1e3, bytes / ms / 1e9,
         bytes / g_grid / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / instr);
}

int main(int argc, char** argv) {
  if (argc > 1) g_grid = atoi(argv[1]);
  printf("grid = %d workgroups (one per CU)\n", g_grid);
  const int iters = 256;
  const long stride = 8192;
  size_t bytes = (size_t)256 * 8 * iters * 16 * stride;  // worst case rows
  if (bytes > (size_t)60 << 30) bytes = (size_t)60 << 30;
  unsigned char* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<0, 1, 1>("x4 nt, 1 row x 1024 B", buf, iters, 1024);
  run<1, 1, 1>("x4 nt, 2 rows x 512 B (stride 3072)", buf, iters, 3072);
  run<2, 1, 1>("x4 nt, 4 rows x 256 B (stride 6144)", buf, iters, 6144);
  run<2, 1, 1>("x4 nt, 4 rows x 256 B (stride 256 = dense)", buf, iters, 256);
  run<3, 1, 1>("x4 nt, 16 rows x 64 B (stride 4608)", buf, iters, 4608);
  run<0, 0, 1>("x4 plain, 1 row x 1024 B", buf, iters, 1024);
  run<2, 0, 1>("x4 plain, 4 rows x 256 B (stride 6144)", buf, iters, 6144);
  run<2, 1, 0>("x2 nt, 4 rows x 128 B (stride 3072)", buf, iters, 6144);
  return 0;
}
