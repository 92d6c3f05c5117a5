// Micro-test of the store-data hazard recorded in DESIGN.md section 5: a 16-byte buffer store whose data registers are
// overwritten by the very next VALU instructions.  Everything sits in ONE inline-asm block with fixed registers, so no
// compiler pass (hazard recognizer included) touches the sequence:
//     v[100:103] = value of this (iteration, lane)        4 x v_mov_b32, then s_nop 4
//     buffer_store_dwordx4 v[100:103], voff, rsrc, SOFF offen       SOFF = an SGPR (variants S*) or the literal 0 (variants I*)
//     <GUARD: nothing | s_nop 0 | s_nop 1 | s_nop 3>
//     v[100:103] = 0xDEADBEEF                              4 x v_mov_b32
// 256 workgroups x 8 waves x ITERS stores, every dword checked on the device afterwards.  A variant that needs wait states
// shows "corrupt dwords > 0"; the production epilogue uses the SGPR form with `s_nop 3` behind every 16-byte store.
// Build / run:  hipcc --offload-arch=gfx950 -O3 scripts/micro/store_data_hazard.hip -o gpurun_out/store_data_hazard && gpurun_out/store_data_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;

#define SEQ(SOFF_OPERAND, GUARD)                                                                                              \
  asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\ts_nop 4\n\t"              \
               "buffer_store_dwordx4 v[100:103], %4, %5, " SOFF_OPERAND " offen\n\t" GUARD                                         \
               "v_mov_b32 v100, %7\n\tv_mov_b32 v101, %7\n\tv_mov_b32 v102, %7\n\tv_mov_b32 v103, %7\n\t"                          \
               ::"v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk)                                  \
               : "v100", "v101", "v102", "v103", "memory")

template <int VARIANT>
__global__ void __launch_bounds__(512) hazard_kernel(unsigned* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned wid = blockIdx.x * 8 + wave;
  unsigned* base = out + (size_t)wid * iters * 256;  // 1 KiB per store instruction
  const unsigned long long pa = (unsigned long long)base;
  const i32x4 rsrc = {(int)__builtin_amdgcn_readfirstlane((unsigned)pa), (int)(__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) & 0xFFFF),
                      (int)__builtin_amdgcn_readfirstlane((unsigned)(iters * 1024)), 0x00020000};
  const unsigned junk = 0xDEADBEEFu;
  for (int i = 0; i < iters; ++i) {
    const unsigned x0 = (wid << 16) ^ (i << 8) ^ lane, x1 = x0 + 0x1000000u, x2 = x0 + 0x2000000u, x3 = x0 + 0x3000000u;
    unsigned soff = __builtin_amdgcn_readfirstlane(i * 1024);
    unsigned voff = lane * 16;
    if (VARIANT >= 4) { voff += soff; soff = 0; }
    if (VARIANT == 0) SEQ("%6", "");
    if (VARIANT == 1) SEQ("%6", "s_nop 0\n\t");
    if (VARIANT == 2) SEQ("%6", "s_nop 1\n\t");
    if (VARIANT == 3) SEQ("%6", "s_nop 3\n\t");
    if (VARIANT == 4) SEQ("0", "");
    if (VARIANT == 5) SEQ("0", "s_nop 0\n\t");
  }
}

__global__ void check_kernel(const unsigned* out, int iters, unsigned long long* bad) {
  const size_t n = (size_t)gridDim.x * blockDim.x;
  const size_t total = (size_t)256 * 8 * iters * 256;
  unsigned long long local = 0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += n) {
    const unsigned wid = (unsigned)(k / ((size_t)iters * 256));
    const unsigned r = (unsigned)(k % ((size_t)iters * 256));
    const unsigned i = r / 256, lane = (r % 256) / 4, c = r % 4;
    const unsigned want = ((wid << 16) ^ (i << 8) ^ lane) + c * 0x1000000u;
    if (out[k] != want) ++local;
  }
  if (local) atomicAdd(bad, local);
}

template <int VARIANT>
static void run(const char* name, unsigned* buf, unsigned long long* bad, int iters) {
  unsigned long long total_bad = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(buf, 0, (size_t)256 * 8 * iters * 1024);
    hipMemset(bad, 0, 8);
    hazard_kernel<VARIANT><<<256, 512>>>(buf, iters);
    check_kernel<<<1024, 256>>>(buf, iters, bad);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    total_bad += h;
  }
  printf("%-58s corrupt dwords in 5 launches: %llu of %llu\n", name, total_bad, 5ull * 256 * 8 * iters * 256);
}

int main() {
  const int iters = 512;
  unsigned* buf; unsigned long long* bad;
  if (hipMalloc(&buf, (size_t)256 * 8 * iters * 1024) != hipSuccess || hipMalloc(&bad, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<0>("S0  soffset in an SGPR, data overwritten at once", buf, bad, iters);
  run<1>("S1  soffset in an SGPR, s_nop 0 (1 wait state)", buf, bad, iters);
  run<2>("S2  soffset in an SGPR, s_nop 1 (2 wait states)", buf, bad, iters);
  run<3>("S3  soffset in an SGPR, s_nop 3 (4 wait states: production)", buf, bad, iters);
  run<4>("I0  soffset literal 0, data overwritten at once", buf, bad, iters);
  run<5>("I1  soffset literal 0, s_nop 0", buf, bad, iters);
  return 0;
}
