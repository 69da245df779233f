// rm_math_sweep.hip -- TEST INFRASTRUCTURE: runs the product's device functions of csrc/rm_math.h over the
// sweep inputs of oracle/rm_math_ref.c (all 2^32 fp32 bit patterns; hashed pairs for atan2) and returns the
// per-block checksums, so tests/test_math_sweep.py can compare them with tests/golden/math_sweep.json.
// Built by __graft_entry__.build() into tests/_build/librm_math_sweep.so with the product's compile flags.
#include "../../ray_marching_amd/csrc/rm_math.h"

namespace {

__device__ __forceinline__ uint32_t hash_y(uint32_t i) { uint32_t h = i * 0x9E3779B1u + 0x7F4A7C15u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; return h; }
__device__ __forceinline__ uint32_t hash_x(uint32_t i) { uint32_t h = (i ^ 0x85EBCA6Bu) * 0xC2B2AE35u; h ^= h >> 13; h *= 0x297A2D39u; h ^= h >> 16; return h; }

__device__ __forceinline__ float eval(int fn, float a, float b) {
  switch (fn) {
    case 0: return rm::exp_f64path(a);
    case 1: return rm::log_f64path(a);
    case 2: return rm::sleef_powf(a, b);
    case 3: return rm::sleef_atan2f(a, b);
    default: return rm::sqrt_rn(a);
  }
}

constexpr int kGroupsPerBlock = 64;   // workgroups per 2^24-input block

__global__ void __launch_bounds__(256) k_sweep(int fn, uint32_t block_begin, float gamma, unsigned long long* sums) {
  const uint32_t blk = block_begin + blockIdx.x / kGroupsPerBlock;
  const uint32_t sub = blockIdx.x % kGroupsPerBlock;
  unsigned long long acc = 0;
  for (uint32_t j = sub * 256 + threadIdx.x; j < (1u << 24); j += kGroupsPerBlock * 256) {
    const uint32_t i = (blk << 24) + j;
    float a, b;
    if (fn == 3) { a = rm::i2f((int)hash_y(i)); b = rm::i2f((int)hash_x(i)); }
    else { a = rm::i2f((int)i); b = gamma; }
    const float r = eval(fn, a, b);
    const uint32_t bits = (r != r) ? 0x7fc00000u : (uint32_t)rm::f2i(r);
    acc += (unsigned long long)bits * (2ull * i + 1ull);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sums[blockIdx.x / kGroupsPerBlock], acc);
}

__global__ void k_eval(int fn, const float* a, const float* b, float* out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = eval(fn, a[i], b ? b[i] : 0.0f);
}

}  // namespace

extern "C" {

// sums: device uint64[n_blocks], zeroed by the caller
int rm_math_sweep(int fn, uint32_t block_begin, uint32_t n_blocks, float gamma, unsigned long long* sums, void* stream) {
  if (fn < 0 || fn > 4 || !sums || n_blocks == 0 || block_begin + n_blocks > 256) return -1;
  k_sweep<<<n_blocks * kGroupsPerBlock, 256, 0, (hipStream_t)stream>>>(fn, block_begin, gamma, sums);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

int rm_math_eval(int fn, const float* a, const float* b, float* out, long long n, void* stream) {
  if (fn < 0 || fn > 4 || !a || !out || n < 0 || ((fn == 2 || fn == 3) && !b)) return -1;
  if (n == 0) return 0;
  long long g = (n + 255) / 256;
  k_eval<<<(int)(g < 4096 ? g : 4096), 256, 0, (hipStream_t)stream>>>(fn, a, b, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // extern "C"
