// ph_sample_hooks.h -- experiment hooks of the phased sampling kernel; compiled only into variant libraries
// (make -C pnr_amd/csrc variant NAME=reread DEFS="-DPNR_EXPERIMENT_HOOKS -DPNR_EXP_REREAD=1"), never into the product.
//
// PNR_EXP_REREAD (round 4): what would it cost the sampling kernel to take the ordered mean (ZNCC pass 1) over?
// The only form that fits its registers and LDS (EXPERIMENTS.md, round 4) is: the wave that has stored an item's 125 samples per
// lane reads them back and adds them in order behind the running sum of the item in front.  This hook does the reading and the
// adding WITHOUT the hand-over (no waiting for another wave: the lower bound of the cost) and leaves the results untouched -- ph_sums
// still runs both passes, the graph is the product's.
//   1: the stash stores stay non-temporal (as in the product); 2: plain stores (the re-read may then hit L2)
#pragma once
#if PNR_EXP_REREAD == 2 // (included in front of smc_device.h, which keeps a STASH_ST it finds defined)
#define STASH_ST(ptr, v) (*(ptr) = (v))
#endif
__device__ float g_reread_sink[64];
__device__ __forceinline__ void pnr_hook_reread(const float *seg_lane, int cnt /* values of this lane's segment, wave-uniform */)
{
    float acc = 0.f;
    for (int k0 = 0; k0 < cnt; k0 += 25) { // a row of 25 at a time: 25 loads in flight, then the ordered adds
        float v[25];
#pragma unroll
        for (int j = 0; j < 25; j++) v[j] = (k0 + j < cnt) ? seg_lane[(size_t)(k0 + j) * 64] : 0.f;
#pragma unroll
        for (int j = 0; j < 25; j++)
            if (k0 + j < cnt) acc += v[j];
    }
    if (acc == 12345.678f) g_reread_sink[threadIdx.x & 63] = acc; // (keeps the sum alive)
}
#define PNR_HOOK_AFTER_FULL_ITEM(seg_lane, cnt) pnr_hook_reread(seg_lane, cnt)
