// Two-level fan-in for the per-channel statistics that the convolution kernels leave in their epilogues.
//
// A kernel whose every workgroup ends with one fp64 atomicAdd per value on red[n][c][..] puts 512 .. 4096 atomic requests
// on ONE cache line; device-scope atomics on a line are retired one after the other at the memory side (measured: 8 .. 50 ns
// each, i.e. 8 us of a 22 us conv launch, 12 us of a 25 us 1x1 launch, whatever the volume holds).  Here a workgroup adds its
// values into one of FAN_REP replicas of the unit's sums (a line of its own each, in a caller-provided workspace), counts
// itself in on the replica's counter; the workgroup that completes a replica counts the replica in on the unit's top counter;
// the workgroup that completes the top counter collects the replicas by atomic exchange -- which also leaves every word zero
// for the next launch -- and gets the unit totals back to add them to red[] itself.  Every access to the arena is a RETURNING
// device-scope atomic executed at the memory side: a contributor's adds have returned before it counts itself in, and the
// collector never reads through a cache (the L2s of the 8 XCDs are not coherent with each other).
//
// Unit = the set of workgroups that sum into the same values (one (sample, channel block) of a launch).  The block is the
// CALLER's (xh_conv_ptrs.fan, xh_fanin_bytes() bytes, zero on entry, left zero): the library owns no device memory.  Launches
// ordered on one stream never overlap and may share a block; launches that can run concurrently need a block each.
#pragma once
#include <hip/hip_runtime.h>

constexpr int FAN_REP = 32;                         // replicas per unit
constexpr int FAN_NV = 32;                          // values per unit (fp64), at most
constexpr int FAN_STRIDE = FAN_NV * 8 + 128;        // bytes per replica: payload, then its counter on a line of its own
constexpr int FAN_UNIT_BYTES = (FAN_REP + 1) * FAN_STRIDE;
constexpr int FAN_UNITS = 64;                       // units per launch, at most (else the kernel keeps its direct atomics)
constexpr int FAN_MIN_WGS = 256;                    // fewer workgroups per unit: direct atomics are cheaper
constexpr long long FAN_BLOCK_BYTES = (long long)FAN_UNITS * FAN_UNIT_BYTES;

// host: the caller's fan-in block for a launch with `units` units of `wgs` workgroups each, or nullptr (direct atomics)
unsigned char* xh_fan_block(void* fan, long long fan_bytes, long long units, long long wgs);

// Device side.  Call from ALL threads of the workgroup (contains barriers); s_tot[0..NV) holds this workgroup's values
// (LDS, written before the call, no barrier needed in between), s_flag is one LDS word.  Returns true in exactly one
// workgroup per unit, with the unit totals in s_tot[0..NV) for threads 0..NV-1 to add to their destination.
template <int NV>
__device__ __forceinline__ bool fan_in(unsigned char* unit, int wg, int nwg, double* s_tot, int* s_flag) {
  static_assert(NV <= FAN_NV && NV <= 64, "fan_in: too many values per unit");
  const int tid = threadIdx.x;
  const int nrep = nwg < FAN_REP ? nwg : FAN_REP;
  const int rep = wg % FAN_REP;
  unsigned char* line = unit + rep * FAN_STRIDE;
  __syncthreads();                                    // s_tot complete
  if (tid < NV) {
    const double old = atomicAdd(reinterpret_cast<double*>(line) + tid, s_tot[tid]);
    asm volatile("" ::"v"(old));                      // returned = performed
  }
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every add of this wave (NV <= 64: they all belong to wave 0)
    const unsigned mine = ((unsigned)nwg - rep + FAN_REP - 1) / FAN_REP;           // workgroups of this replica
    int last = 0;
    if (__hip_atomic_fetch_add(reinterpret_cast<unsigned*>(line + FAN_NV * 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1)
      last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(unit + FAN_REP * FAN_STRIDE), 1u, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nrep - 1;
    *s_flag = last;
  }
  __syncthreads();
  if (!*s_flag) return false;                         // workgroup-uniform
  if (tid < NV) {
    unsigned long long bits[FAN_REP];
#pragma unroll
    for (int r = 0; r < FAN_REP; ++r)                 // independent exchanges: all in flight together
      bits[r] = r < nrep ? __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(unit + r * FAN_STRIDE) + tid, 0ull,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : 0ull;
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < FAN_REP; ++r) tot += __builtin_bit_cast(double, bits[r]);
    s_tot[tid] = tot;
  }
  if (tid >= 64 && tid < 64 + nrep)
    __hip_atomic_exchange(reinterpret_cast<unsigned*>(unit + (tid - 64) * FAN_STRIDE + FAN_NV * 8), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 128) __hip_atomic_exchange(reinterpret_cast<unsigned*>(unit + FAN_REP * FAN_STRIDE), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();                                    // totals visible to threads 0..NV-1 of any wave
  return true;
}
