// nd_persist.hpp -- the ONE-LAUNCH form of a whole p_sample_loop (csrc/nd_persist.hip): interface between that translation unit and
// the graph builder in csrc/nd_sampler.hip.  gfx950 only.
//
// Reference: diffusion/diffusion_utils.py:133-163 (the T-loop), diffusion/latent_model.py:173-184 (what one step evaluates).
#pragma once
#include "nd_common.hpp"
#include "nd_step.hpp"

// Barrier block in the handle's workspace: words g*32 and g*32 + 16 = arrival counter and generation of the launch's g-th member (a
// 128-byte line per member); word ND_INLINE_DESCS*32 = sticky error word (a barrier wait gave up).  Zeroed by nd_bind_workspace and by
// nd_persist_status(reset); never by a launch (the barriers clean up after themselves, csrc/nd_persist.hip).
#define ND_PERSIST_BAR_WORDS ((ND_INLINE_DESCS + 1) * 32)
#define ND_PERSIST_ERR_WORD (ND_INLINE_DESCS * 32)

struct PersistScalars {
    unsigned* bar;
    int nm, B, M, maxM, F, T;
    int skew;          // start offset between consecutive members, in ticks of the 100 MHz wall clock (0: all start together)
    int spin_ticks;    // a barrier wait longer than this sets the error word and the workgroup leaves the kernel
    int active;        // members that run (experiments: ND_PERSIST_ACTIVE; the others' workgroups leave at once, their outputs are not written)
    int fake_resident; // TIMING ABLATION ONLY (ND_PERSIST_FAKE_RESIDENT = n, results are WRONG): the weight loads of the first n of a wave's
                       // register stages per layer all go to the wave's FIRST stage (cache hits instead of fabric traffic) -- an upper bound on
                       // what holding that share of the weights on chip could buy, with the instruction stream and the MFMAs unchanged
};

struct PersistArgs {                         // by value in the kernel arguments, read through the constant address space
    SkinnyDesc l2[ND_INLINE_DESCS];          // lin2 block of each member (x = h1, out = h2, scale / shift = the [T, F] folds)
    SkinnyDesc l3[ND_INLINE_DESCS];          // lin3 + lin4 block (x = h2, pw = lin4.weight, part = the partial-sum buffer)
    MemberDev mem[ND_INLINE_DESCS];
    StepIO io;
    PersistScalars s;
};

struct PersistPlan {
    bool ok;             // the shape runs as one launch (else: the per-step kernels)
    void* fn;
    dim3 grid, block;
    unsigned lds;
    hipError_t err;      // preparing the launch (dynamic-LDS attribute of the kernel on this device)
    const char* why;     // when !ok
};

// M rows per member, nm members, C classes, F = feature width (K = N = F); half: fp16 operands.
PersistPlan nd_persist_plan(int F, int M, int nm, int C, int half);
