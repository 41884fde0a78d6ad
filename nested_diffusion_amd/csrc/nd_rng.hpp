// nd_rng.hpp -- launchers of the in-library noise generator (nd_rng.hip) for the sampler's translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// out[nm][T][mc*B][C] standard normals for ensemble members m0 .. m0+nm-1 (the member word of the counter is m0 + k).  state_dev != nullptr: {seed, batch counter | first image << 32} is read on the DEVICE at
// run time (hipGraph replays see the current values); else the three scalar arguments are used.
hipError_t nd_launch_philox_normal(float* out, const unsigned long long* state_dev, unsigned long long seed, uint32_t batch, uint32_t first,
                                   int m0, int nm, int T, int B, int mc, int C, hipStream_t st);
// batch counter += 1 (last node of a sampling graph)
hipError_t nd_launch_rng_advance(unsigned long long* state_dev, hipStream_t st);
// kernel handles for explicit hipGraph kernel nodes; argument lists:
//   philox:  (float* out, const unsigned long long* state, unsigned long long seed, uint32_t batch, uint32_t first, int m0, int nm, int T, int B, int mc, int C)
//            grid ((nm*T*B*mc*ceil(C/4) + 255) / 256), block 256
//   advance: (unsigned long long* state), grid 1, block 64
void* nd_philox_normal_kernel();
void* nd_rng_advance_kernel();
