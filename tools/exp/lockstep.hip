// Lock-step self-check kernel (round-3 race probe).  Every section evaluates the SAME operation twice from the same
// (opaque) inputs and counts the lanes whose two results differ — a transient hardware / scheduling fault shows up as
// a mismatch, a deterministic bug does not.  Launched next to the MFMA decoder on another stream so that both kernels
// share CUs (tools/exp/race_probe.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o liblockstep.so lockstep.hip
// Sections: 0 global dwordx4 loads (partial exec masks), 1 LDS write -> read_b128 broadcast, 2 ds_bpermute,
//           3 IEEE division + sqrt, 4 DPP row sums, 5 packed fma chains, 6 LDS float atomics
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NSEC 7
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void opaque(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void opaque(int& x) { asm volatile("" : "+v"(x)); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float row_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v;
}

// counts[sec][lane]: mismatching evaluations
__global__ __launch_bounds__(256, 2) void lockstep_kernel(const float* __restrict__ data, int n_texels, int iters,
                                                           unsigned* __restrict__ counts, unsigned sections) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 8];
  const int lane = threadIdx.x & 63;
  const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
  unsigned bad[NSEC];
#pragma unroll
  for (int s = 0; s < NSEC; ++s) bad[s] = 0;
  unsigned rng = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    rng = rng * 1664525u + 1013904223u;
    // slot-uniform pseudo-random texel (as the walk: 16 lanes read one 512-byte record), slot-dependent reload decision
    const unsigned rs = __shfl((int)rng, (lane & 48), 64);
    int texel = (int)((rs >> 8) % (unsigned)n_texels);
    const bool reload = ((rs >> 3) & 3) != 0;
    if (sections & 1u) {  // ---- 0: global loads under a slot-divergent mask, twice
      v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
      const v4f* p = reinterpret_cast<const v4f*>(data + (size_t)texel * 128 + sub * 8);
      if (reload) {
        a0 = p[0];
        a1 = p[1];
      }
      int t2 = texel;
      opaque(t2);
      const v4f* q = reinterpret_cast<const v4f*>(data + (size_t)t2 * 128 + sub * 8);
      if (reload) {
        b0 = q[0];
        b1 = q[1];
      }
      const bool ne = a0.x != b0.x || a0.y != b0.y || a0.z != b0.z || a0.w != b0.w || a1.x != b1.x || a1.y != b1.y ||
                      a1.z != b1.z || a1.w != b1.w;
      // and against the known content: data[i] = hash(i)
      const unsigned i0 = (unsigned)texel * 128u + sub * 8u;
      const bool wrong = reload && a0.x != __uint_as_float(0x3f800000u | ((i0 * 2654435761u) >> 9));
      bad[0] += (ne || wrong) ? 1u : 0u;
      keep += a0.x + b1.w;
    }
    if (sections & 2u) {  // ---- 1: LDS record written by lane `sub == it % 16` of the slot, read by all 16 (b128 broadcast), twice
      float4* rec = reinterpret_cast<float4*>(lds) + slot * 2;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (sub == (it & 15)) {
        rec[0] = make_float4(__uint_as_float(rng), __uint_as_float(rng ^ 0x5555u), (float)it, (float)slot);
        rec[1] = make_float4(__uint_as_float(rng + 1), __uint_as_float(rng + 2), __uint_as_float(rng + 3), 1.0f);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float4 r0 = rec[0], r1 = rec[1];
      const unsigned want = __shfl((int)rng, (lane & 48) + (it & 15), 64);
      bad[1] += (__float_as_uint(r0.x) != want || r0.z != (float)it || __float_as_uint(r1.z) != want + 3) ? 1u : 0u;
      keep += r0.y + r1.x;
    }
    if (sections & 4u) {  // ---- 2: ds_bpermute butterflies, twice
      float v = __uint_as_float(0x3f800000u | (rng >> 9));
      float a = v, b = v;
      opaque(b);
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64);
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) b += __shfl_xor(b, m, 64);
      bad[2] += (a != b) ? 1u : 0u;
      keep += a;
    }
    if (sections & 8u) {  // ---- 3: IEEE division + sqrt, twice
      float x = __uint_as_float(0x3f800000u | (rng >> 9)), y = __uint_as_float(0x3f800000u | ((rng * 7u) >> 9));
      float x2 = x, y2 = y;
      opaque(x2);
      opaque(y2);
      const float r1 = x / (sqrtf(y) * sqrtf(x + y));
      const float r2 = x2 / (sqrtf(y2) * sqrtf(x2 + y2));
      bad[3] += (r1 != r2) ? 1u : 0u;
      keep += r1;
    }
    if (sections & 16u) {  // ---- 4: DPP row sums, twice
      float v = __uint_as_float(0x3f800000u | ((rng * 3u) >> 9));
      float w = v;
      opaque(w);
      const float a = row_sum(v), b = row_sum(w);
      bad[4] += (a != b) ? 1u : 0u;
      keep += a;
    }
    if (sections & 32u) {  // ---- 5: packed fma chains, twice
      v2f a = {__uint_as_float(0x3f800000u | (rng >> 9)), __uint_as_float(0x3f800000u | ((rng * 5u) >> 9))};
      v2f b = a;
      asm volatile("" : "+v"(b));
      v2f acc1 = {0, 0}, acc2 = {0, 0};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc1 = __builtin_elementwise_fma(a, a + (float)k, acc1);
        acc2 = __builtin_elementwise_fma(b, b + (float)k, acc2);
      }
      bad[5] += (acc1.x != acc2.x || acc1.y != acc2.y) ? 1u : 0u;
      keep += acc1.x;
    }
    if (sections & 64u) {  // ---- 6: LDS float atomics (one owner lane per address), read back
      float* cs = lds + 1024 + slot * 16;
      cs[sub] = 0.0f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float c0 = __uint_as_float(0x3f800000u | (rng >> 9));
      __hip_atomic_fetch_add(cs + sub, c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(cs + sub, 2.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float got = cs[(sub + 1) & 15];
      const float want = __shfl(c0, (lane & 48) + ((sub + 1) & 15), 64) + 2.0f;
      bad[6] += (got != want) ? 1u : 0u;
      keep += got;
    }
  }
#pragma unroll
  for (int s = 0; s < NSEC; ++s)
    if (bad[s]) atomicAdd(counts + s * 64 + lane, bad[s]);
  if (keep == 12345.678f) counts[NSEC * 64] = 1;  // keeps every result alive
}

__global__ void lockstep_fill(float* data, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    data[i] = __uint_as_float(0x3f800000u | (((unsigned)i * 2654435761u) >> 9));
}

extern "C" int lockstep_fill_data(float* data, int64_t n, void* stream) {
  hipLaunchKernelGGL(lockstep_fill, dim3(1024), dim3(256), 0, (hipStream_t)stream, data, (size_t)n);
  return (int)hipGetLastError();
}

extern "C" int lockstep_launch(const float* data, int n_texels, int iters, unsigned* counts, unsigned sections, int grid,
                               void* stream) {
  hipLaunchKernelGGL(lockstep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, data, n_texels, iters, counts, sections);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Sentinel kernel: a wave parks known patterns in NV vector registers (and NS scalar registers), idles / waits on
// memory for a while WITHOUT touching them, and then checks them.  A register that changed was written by somebody
// else.  out_v[reg * 64 + lane] / out_s[reg] count the mismatches; out_first[0..7] keeps one (reg, lane, got, want).
__device__ __forceinline__ unsigned pat(unsigned i, unsigned t) { return (i * 0x9E3779B1u) ^ (t * 0x85EBCA6Bu) ^ 0xA5A5A5A5u; }

template <int NV, int NS>
__global__ __launch_bounds__(256, 2) void sentinel_kernel(const float* __restrict__ data, int n_floats, int iters, int idle_mode,
                                                           unsigned* __restrict__ out_v, unsigned* __restrict__ out_s,
                                                           unsigned* __restrict__ out_first) {
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned r[NV];
  unsigned s[NS];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    r[i] = pat(i, tid);
    asm volatile("" : "+v"(r[i]));
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    s[i] = __builtin_amdgcn_readfirstlane(pat(i, blockIdx.x * 4u + (threadIdx.x >> 6)));
    asm volatile("" : "+s"(s[i]));
  }
  float keep = 0.f;
  unsigned rng = tid * 747796405u + 2891336453u;
  for (int it = 0; it < iters; ++it) {
    // ---- idle phase: nothing here reads or writes r[] / s[]
    if (idle_mode == 0) {
      __builtin_amdgcn_s_sleep(8);
    } else if (idle_mode == 1) {  // parked on vmcnt
      rng = rng * 1664525u + 1013904223u;
      const float x = data[(rng >> 4) % (unsigned)n_floats];
      keep += x;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (idle_mode == 2) {  // parked on lgkmcnt (ds_bpermute)
      keep += __shfl_xor(keep + 1.0f, 1 + (it & 31), 64);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {  // issue-stalled: s_nop trains
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    }
    // ---- keep the sentinels live and opaque (no instruction is emitted for these)
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
    for (int i = 0; i < NS; ++i) asm volatile("" : "+s"(s[i]));
    if ((it & 15) == 15) {
      const unsigned tk = (tid * 0x85EBCA6Bu) ^ 0xA5A5A5A5u;
      unsigned first_bad = 0xffffffffu, got_bad = 0, n_bad = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const unsigned want = ((unsigned)i * 0x9E3779B1u) ^ tk;
        const bool b = r[i] != want;
        n_bad += b ? 1u : 0u;
        got_bad = (b && first_bad == 0xffffffffu) ? r[i] : got_bad;
        first_bad = (b && first_bad == 0xffffffffu) ? (unsigned)i : first_bad;
        r[i] = want;  // re-park
        asm volatile("" : "+v"(r[i]));
      }
      if (n_bad) {
        atomicAdd(out_v + first_bad * 64 + lane, n_bad);
        const unsigned k = atomicAdd(out_first, 1u);
        if (k < 8) {
          unsigned* o = out_first + 8 + k * 6;
          o[0] = first_bad, o[1] = lane, o[2] = got_bad, o[3] = (first_bad * 0x9E3779B1u) ^ tk, o[4] = blockIdx.x, o[5] = it;
        }
      }
      const unsigned sk = ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 0x85EBCA6Bu) ^ 0xA5A5A5A5u;
      unsigned s_bad = 0xffffffffu;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const unsigned want = __builtin_amdgcn_readfirstlane(((unsigned)i * 0x9E3779B1u) ^ sk);
        s_bad = (s[i] != want && s_bad == 0xffffffffu) ? (unsigned)i : s_bad;
        s[i] = want;
        asm volatile("" : "+s"(s[i]));
      }
      if (s_bad != 0xffffffffu && lane == 0) atomicAdd(out_s + s_bad, 1u);
    }
  }
  if (keep == 12345.678f) out_first[7] = 1;
}

extern "C" int sentinel_launch(const float* data, int n_floats, int iters, int idle_mode, unsigned* out_v, unsigned* out_s,
                               unsigned* out_first, int grid, int variant, void* stream) {
  if (variant == 0)  // registers only
    hipLaunchKernelGGL((sentinel_kernel<72, 12>), dim3(grid), dim3(256), 0, (hipStream_t)stream, data, n_floats, iters, idle_mode,
                       out_v, out_s, out_first);
  else               // more uniform values than scalar registers: the compiler parks them in vector-register lanes
    hipLaunchKernelGGL((sentinel_kernel<24, 120>), dim3(grid), dim3(256), 0, (hipStream_t)stream, data, n_floats, iters, idle_mode,
                       out_v, out_s, out_first);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// E4: is a VALU-written lane mask (v_cmp -> vcc / SGPR pair) that the scalar unit consumes right away
// (s_and_saveexec_b64) stale in its upper 16 bits while ANOTHER wave of the SIMD issues 16-bit 32x32x16 MFMAs?
// Victim: per-lane pseudo-random predicate p; `a` counts p through a real branch (exec masking), `b` through
// arithmetic.  a != b  <=>  some lane executed / skipped a block it should not have.
__global__ __launch_bounds__(256, 2) void mask_victim_kernel(int iters, unsigned* __restrict__ counts /* [3][64] */) {
  const int lane = threadIdx.x & 63;
  unsigned rng = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 777u;
  int a = 0, b = 0, c = 0, d = 0;
  float fa = 0.f, fb = 0.f;
  for (int it = 0; it < iters; ++it) {
    rng = rng * 1664525u + 1013904223u;
    const int bit = (int)((rng >> 9) & 1u);
    int bit2 = bit;
    asm volatile("" : "+v"(bit2));
    if (bit) {  // v_cmp -> s_and_saveexec -> block
      a += 1;
      asm volatile("" : "+v"(a));
    }
    b += bit2;
    // slot-uniform predicate (as the walk's tap reload: 16 lanes decide alike), v_cmp_ne on two registers
    const unsigned rs = (unsigned)__shfl((int)rng, lane & 48, 64);
    int want = (int)((rs >> 11) & 3u), have = (int)((rs >> 13) & 3u);
    asm volatile("" : "+v"(want), "+v"(have));
    if (want != have) {
      c += 1;
      asm volatile("" : "+v"(c));
    }
    d += (((rs >> 11) & 3u) != ((rs >> 13) & 3u)) ? 1 : 0;
    // mask consumed by the vector unit (v_cndmask with an SGPR-pair mask)
    const float x = __uint_as_float(0x3f800000u | (rng >> 9));
    const bool m = x > 1.5f;
    fa += m ? x : 0.25f;
    float x2 = x;
    asm volatile("" : "+v"(x2));
    fb += (x2 > 1.5f) ? x2 : 0.25f;
  }
  if (a != b) atomicAdd(counts + lane, 1u);
  if (c != d) atomicAdd(counts + 64 + lane, 1u);
  if (fa != fb) atomicAdd(counts + 128 + lane, 1u);
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// Partner: nothing but one kind of matrix instruction (or packed VALU) in a loop
template <int KIND>
__global__ __launch_bounds__(256, 2) void mfma_partner_kernel(int iters, float* __restrict__ sink) {
  const float seed = (float)(threadIdx.x & 7) * 0.125f;
  f16v acc0 = (f16v)(seed), acc1 = (f16v)(seed + 1.f);
  f4v q0 = {seed, 0, 0, 0}, q1 = {0, seed, 0, 0};
  h8 a8, b8v;
  b8 ab8, bb8;
  h4 a4, b4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a8[i] = (_Float16)(seed + i);
    b8v[i] = (_Float16)(1.0f - seed * i);
    ab8[i] = (__bf16)(seed + i);
    bb8[i] = (__bf16)(1.0f - seed * i);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a4[i] = (_Float16)(seed + i);
    b4[i] = (_Float16)(0.5f - seed * i);
  }
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {seed, seed + 1.f}, p1 = {seed + 2.f, seed + 3.f};
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8v, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8v, acc1, 0, 0, 0);
    } else if constexpr (KIND == 1) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc1, 0, 0, 0);
    } else if constexpr (KIND == 2) {
      q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8v, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8v, q1, 0, 0, 0);
    } else if constexpr (KIND == 3) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, acc1, 0, 0, 0);
    } else if constexpr (KIND == 4) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab8, bb8, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab8, bb8, acc1, 0, 0, 0);
    } else if constexpr (KIND == 5) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        p0 = __builtin_elementwise_fma(p0, p1, p0);
        p1 = __builtin_elementwise_fma(p1, p0, p1);
      }
    } else if constexpr (KIND == 6) {
      q0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, q1, 0, 0, 0);
    } else if constexpr (KIND == 7) {  // the split-fp16 operand preparation without any matrix instruction
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const v2 ab = p0 * 4.0f;
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, h2));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(p0.x), "v"(4.0f), "v"(h));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(p0.y), "v"(4.0f), "v"(h));
        p0.x = r0 + p1.x;
        p0.y = r1 + p1.y;
        p1 = p1 * 0.999f;
      }
    } else {  // KIND 8: 32x32x16 f16 with the A operands re-read from LDS every step (ds_read_b128), as the decoder does
      __shared__ __attribute__((aligned(16))) unsigned wl[64 * 4 * 8];
      if (it == 0) {
        for (int i = threadIdx.x; i < 64 * 4 * 8; i += 256) wl[i] = 0x3c003c00u;
        __syncthreads();
      }
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4* wp = reinterpret_cast<const u4*>(wl) + (threadIdx.x & 63);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const h8 af = __builtin_bit_cast(h8, wp[64 * (2 * k)]), ag = __builtin_bit_cast(h8, wp[64 * (2 * k + 1)]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b8v, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, b8v, acc1, 0, 0, 0);
      }
    }
  }
  float r = acc0[0] + acc1[5] + q0[0] + q1[1] + p0.x + p1.y;
  if (r == 12345.678f) sink[0] = r;
}

extern "C" int mask_victim_launch(int iters, unsigned* counts, int grid, void* stream) {
  hipLaunchKernelGGL(mask_victim_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, iters, counts);
  return (int)hipGetLastError();
}
extern "C" int mfma_partner_launch(int kind, int iters, float* sink, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL((mfma_partner_kernel<0>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 1: hipLaunchKernelGGL((mfma_partner_kernel<1>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 2: hipLaunchKernelGGL((mfma_partner_kernel<2>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 3: hipLaunchKernelGGL((mfma_partner_kernel<3>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 4: hipLaunchKernelGGL((mfma_partner_kernel<4>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 5: hipLaunchKernelGGL((mfma_partner_kernel<5>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 6: hipLaunchKernelGGL((mfma_partner_kernel<6>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    case 7: hipLaunchKernelGGL((mfma_partner_kernel<7>), dim3(grid), dim3(256), 0, st, iters, sink); break;
    default: hipLaunchKernelGGL((mfma_partner_kernel<8>), dim3(grid), dim3(256), 0, st, iters, sink); break;
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// E6: which packed-fp32 instruction form loses results next to 16-bit 32x32x16 MFMAs of another wave?
// Two identical chains of ONE instruction form (forced with inline asm) on separate registers; lanes whose chains
// end differently are counted.  FORM: 0 v_pk_fma_f32 (plain), 1 v_pk_fma_f32 op_sel_hi:[1,0,1] (broadcast weight, as the
// cost volume's interpolation), 2 v_pk_mul_f32, 3 v_pk_add_f32, 4 v_fma_f32 (one fp32 per lane), 5 v_pk_mul_f32 op_sel_hi:[1,0]
typedef float pk2 __attribute__((ext_vector_type(2)));
template <int FORM>
__global__ __launch_bounds__(256, 2) void pk_victim_kernel(int iters, unsigned* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const float s0 = 1.0f + (float)(threadIdx.x & 31) * 0.03125f;
  pk2 a1 = {s0, s0 + 0.5f}, a2 = a1;
  pk2 t = {1.0f - 1.0f / 1024.0f, 1.0f - 1.0f / 2048.0f}, w = {1.0f / 4096.0f, 1.0f / 8192.0f};
  asm volatile("" : "+v"(a2), "+v"(t), "+v"(w));
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (FORM == 0) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(t), "v"(w));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(t), "v"(w));
      } else if constexpr (FORM == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a1) : "v"(t), "v"(w));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a2) : "v"(t), "v"(w));
      } else if constexpr (FORM == 2) {
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(t));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(t));
      } else if constexpr (FORM == 3) {
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(w));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2) : "v"(w));
      } else if constexpr (FORM == 4) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1.x) : "v"(t.x), "v"(w.x));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2.x) : "v"(t.x), "v"(w.x));
      } else {
        asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a1) : "v"(t));
        asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a2) : "v"(t));
      }
    }
    if (a1.x != a2.x || a1.y != a2.y) {
      ++bad;
      a2 = a1;
    }
    if ((it & 63) == 63) {  // keep the values in range
      a1 = pk2{s0, s0 + 0.5f};
      a2 = a1;
      asm volatile("" : "+v"(a2));
    }
  }
  if (bad) atomicAdd(counts + lane, bad);
}

extern "C" int pk_victim_launch(int form, int iters, unsigned* counts, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (form) {
    case 0: hipLaunchKernelGGL((pk_victim_kernel<0>), dim3(grid), dim3(256), 0, st, iters, counts); break;
    case 1: hipLaunchKernelGGL((pk_victim_kernel<1>), dim3(grid), dim3(256), 0, st, iters, counts); break;
    case 2: hipLaunchKernelGGL((pk_victim_kernel<2>), dim3(grid), dim3(256), 0, st, iters, counts); break;
    case 3: hipLaunchKernelGGL((pk_victim_kernel<3>), dim3(grid), dim3(256), 0, st, iters, counts); break;
    case 4: hipLaunchKernelGGL((pk_victim_kernel<4>), dim3(grid), dim3(256), 0, st, iters, counts); break;
    default: hipLaunchKernelGGL((pk_victim_kernel<5>), dim3(grid), dim3(256), 0, st, iters, counts); break;
  }
  return (int)hipGetLastError();
}
