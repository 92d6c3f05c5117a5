"""Which library kernels does torch.matmul pick for the step's long-K NT shapes? Run under `rocprofv3 --kernel-trace --stats`:
the Tensile kernel name spells out the macro tile, wave layout and prefetch depths of the configuration that sets the
measuring-stick numbers of profiles/r02_gemm_vs_library.log. Nothing in the package calls the library."""
import torch

dev = torch.device("cuda")
for M, N, K in [(55552, 768, 3072), (55552, 768, 2304), (55552, 3072, 768), (14080, 768, 3072), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    for _ in range(5):
        torch.matmul(A, B.t())
    torch.cuda.synchronize()
