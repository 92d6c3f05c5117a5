// Micro-benchmark: L2 -> LDS rate of buffer_load_dwordx4 ... lds by the shape of the 1 KiB a wave-instruction fetches.
// 256 workgroups x 8 waves stream a panel that every workgroup of an XCD shares (L2 hits), like the GEMM operand stream:
//   rows of `pitch` bytes, `piece` contiguous bytes taken from each row per instruction (64 = one K-stage of 32 bf16,
//   128 = a whole cache line, 1024 = fully contiguous).
// Build/run: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_dma_rate.hip -o scripts/micro/bin/lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned num_records) {
  const unsigned long long pa = (unsigned long long)base;
  return i32x4{(int)__builtin_amdgcn_readfirstlane((unsigned)pa), (int)(__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) & 0xFFFF),
               (int)__builtin_amdgcn_readfirstlane(num_records), 0x00020000};
}
__device__ __forceinline__ void dma16s(i32x4 rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

// panel: rows x pitch bytes (shared by all workgroups of an XCD -> L2 resident after the first touch)
template <int PIECE, int PAIR = 0>
__global__ void __launch_bounds__(512) dma_kernel(const unsigned char* panel, int rows, int pitch, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[131072];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int lanes_per_row = PIECE / 16, rows_per_instr = 64 / lanes_per_row;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const i32x4 rs = make_srd(panel, (unsigned)((long)rows * pitch));
  const int r = lane / lanes_per_row, c = lane % lanes_per_row;
  const unsigned voff = (unsigned)(r * pitch + c * 16);
  const int pieces_per_row = pitch / PIECE;
  // the workgroup sweeps the panel like a GEMM stage stream: instruction i of wave w takes row group (i*8 + w), column piece
  // advancing every full sweep of the rows
  const int groups = rows / rows_per_instr;  // row groups
  int g = wave, col = 0;
  for (int i = 0; i < iters; ++i) {
    if (PAIR) {  // instructions 2j and 2j+1 fetch the two 64-B halves of the same 16 lines back to back
      const unsigned soff = (unsigned)(g * rows_per_instr * pitch + (col * 2 + (i & 1)) * PIECE);
      dma16s(rs, lds0 + ((i & 15) * 8 + wave) * 1024, voff, soff);
      if (i & 1) {
        g += 8;
        if (g >= groups) { g -= groups; col = (col + 1) % (pieces_per_row / 2); }
      }
    } else {
      const unsigned soff = (unsigned)(g * rows_per_instr * pitch + col * PIECE);
      dma16s(rs, lds0 + ((i & 15) * 8 + wave) * 1024, voff, soff);
      g += 8;
      if (g >= groups) { g -= groups; col = (col + 1) % pieces_per_row; }
    }
    if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // 8..16 KiB per wave in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)smem;
}

template <int PIECE, int PAIR = 0>
static void run(const char* name, const unsigned char* panel, int rows, int pitch, unsigned* sink) {
  const int iters = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dma_kernel<PIECE, PAIR><<<256, 512>>>(panel, rows, pitch, iters, sink);
  hipEventRecord(e0);
  for (int k = 0; k < 5; ++k) dma_kernel<PIECE, PAIR><<<256, 512>>>(panel, rows, pitch, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = 256.0 * 8 * iters * 1024;
  printf("%-52s %8.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ms / 1e6);
}

int main() {
  unsigned char* panel; unsigned* sink;
  hipMalloc(&panel, 64 << 20); hipMemset(panel, 1, 64 << 20); hipMalloc(&sink, 4096);
  // 2 MiB panels (fit the 4 MiB L2 of every XCD): 1536-B pitch = a K=768 bf16 row
  run<64>("64 B per row  (K-stage of 32 bf16), pitch 1536, 2 MiB", panel, 1365, 1536, sink);
  run<64, 1>("64 B per row, halves of a line back to back,  2 MiB", panel, 1365, 1536, sink);
  run<128>("128 B per row (whole line),        pitch 1536, 2 MiB", panel, 1365, 1536, sink);
  run<256>("256 B per row,                     pitch 1536, 2 MiB", panel, 1364, 1536, sink);
  run<1024>("1024 B contiguous,                 pitch 1024, 2 MiB", panel, 2048, 1024, sink);
  // 32 MiB panels: beyond L2, inside the Infinity Cache
  run<64>("64 B per row,  pitch 1536, 32 MiB (Infinity Cache)", panel, 21845, 1536, sink);
  run<128>("128 B per row, pitch 1536, 32 MiB (Infinity Cache)", panel, 21845, 1536, sink);
  return 0;
}
