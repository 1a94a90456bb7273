"""Which hipBLASLt kernels torch.matmul picks for the four prefill GEMM shapes (run under rocprofv3 --kernel-trace):
a yardstick only -- the names carry the macro tile, wave tiling, prefetch depths and workgroup mapping."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench_gemm
bench_gemm.yardstick(rounds=2)
