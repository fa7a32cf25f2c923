"""tools/exp/voxelize_chunk_log.hip = csrc/voxelize.hip with 100 MHz wall-clock stamps per channel chunk of the scatter / fused
voxelize kernels (lane 0 of every workgroup: chunk start, after the barrier, after B1, after B2, after the stores; points,
occupied voxels, channels, slab, cloud, workgroup).  Build and read:
  python tools/vox_chunk_log_build.py && tools/build_variant.sh voxlog voxelize=tools/exp/voxelize_chunk_log.hip
  LION_HIP_SO=$PWD/tools/exp/variants/liblion_voxlog.so python tools/vox_chunk_log.py        (needs tools/scratch/chain_clouds.npz)"""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(root, "lion_amd/csrc/voxelize.hip")).read()
s = s.replace("namespace {\n", "namespace {\n__device__ unsigned long long g_vlog[8192 * 8];\n__device__ unsigned int g_vlog_n;\n", 1)
a = s.index("    __syncthreads(); // hist/tmp (or the previous chunk's vm) are dead from here on\n")
s = s[:a] + ("    unsigned long long *vlog = nullptr;\n    if (threadIdx.x == 0) { const unsigned q = atomicAdd(&g_vlog_n, 1u); if (q < 8192) { vlog = g_vlog + q * 8; vlog[0] = wall_clock64(); "
             "vlog[5] = ((unsigned long long)my_pts << 32) | (unsigned)my_occ; vlog[6] = ((unsigned long long)ch << 32) | ((unsigned)(lo / SV) << 16) | (unsigned)b; vlog[7] = blockIdx.x; } }\n") + s[a:]
a = s.index("    __syncthreads(); // hist/tmp (or the previous chunk's vm) are dead from here on\n") + len("    __syncthreads(); // hist/tmp (or the previous chunk's vm) are dead from here on\n")
s = s[:a] + "    if (vlog) vlog[1] = wall_clock64();\n" + s[a:]
a = s.index("      __syncthreads();\n      // B2: per-voxel means from LDS")
s = s[:a] + "      __syncthreads();\n      if (vlog) vlog[2] = wall_clock64();\n" + s[a + len("      __syncthreads();\n"):]
a = s.index("    __syncthreads();\n    // C: the slab of the dense grid")
s = s[:a] + "    __syncthreads();\n    if (vlog) vlog[3] = wall_clock64();\n" + s[a + len("    __syncthreads();\n"):]
a = s.index("      if (g >= q4) { g -= q4; ++cl; }\n    }\n")
e = a + len("      if (g >= q4) { g -= q4; ++cl; }\n    }\n")
s = s[:e] + "    if (vlog) vlog[4] = wall_clock64();\n" + s[e:]
s = s.replace('int lion_avg_voxelize_backward(', '''int lion_debug_vox_log(unsigned long long *host, int reset) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_vlog_n), 4) != hipSuccess) return -1;
  if (n > 8192) n = 8192;
  if (host && n && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_vlog), (size_t)n * 64) != hipSuccess) return -1;
  if (reset) { unsigned int z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_vlog_n), &z, 4) != hipSuccess) return -1; }
  return (int)n;
}

int lion_avg_voxelize_backward(''', 1)
open(os.path.join(root, "tools/exp/voxelize_chunk_log.hip"), "w").write(s)
